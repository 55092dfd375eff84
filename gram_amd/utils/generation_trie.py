"""Prefix tree over candidate item-token sequences, plus its flat (CSR) form for the device.

Host-side mirror of the reference's src/utils/generation_trie.py: same class name, constructor,
attributes (``trie_dict``, ``len``) and methods (``add``, ``get``, ``append``, ``load_from_dict``,
iteration), same ``prefix_allowed_tokens_fn(trie)`` closure shape (generation_trie.py:89-95), so the
runner code that builds ``gt.Trie(encoded_candidates)`` (single_runner_gram.py:617-619) works
unchanged.  The addition is :class:`FlatTrie`: the nested dict laid out as three int32 arrays in
HBM, which is what ``gram_beam_step`` walks instead of calling back into Python per beam.
"""
from __future__ import annotations

from typing import Dict, Iterable, Iterator, List, Optional, Sequence

import numpy as np


class Trie:
    def __init__(self, sequences: Optional[Iterable[Sequence[int]]] = None):
        self.trie_dict: Dict[int, dict] = {}
        self.len = 0
        self.append_trie = None
        self.bos_token_id = None
        for seq in sequences or ():
            self.add(seq)

    def append(self, trie: "Trie", bos_token_id: int) -> None:
        self.append_trie = trie
        self.bos_token_id = bos_token_id

    def add(self, sequence: Sequence[int]) -> None:
        node = self.trie_dict
        for tok in sequence:
            nxt = node.get(tok)
            if nxt is None:
                nxt = node[tok] = {}
            node = nxt
        self.len += 1

    def get(self, prefix_sequence: Sequence[int]) -> List[int]:
        """Tokens that may follow ``prefix_sequence`` ([] if the prefix is not in the tree)."""
        node = self.trie_dict
        for tok in prefix_sequence:
            nxt = node.get(tok)
            if nxt is None:
                return self.append_trie.get(prefix_sequence) if self.append_trie else []
            node = nxt
        out = list(node.keys())
        if self.append_trie and self.bos_token_id in out:
            out.remove(self.bos_token_id)
            out += list(self.append_trie.trie_dict.keys())
        return out

    @staticmethod
    def load_from_dict(trie_dict: dict) -> "Trie":
        t = Trie()
        t.trie_dict = trie_dict
        t.len = sum(1 for _ in t)
        return t

    def __iter__(self) -> Iterator[List[int]]:
        stack = [([], self.trie_dict)]
        while stack:
            prefix, node = stack.pop()
            if not node:
                if prefix or node is not self.trie_dict:
                    yield prefix
                continue
            for tok in reversed(list(node.keys())):
                stack.append((prefix + [tok], node[tok]))

    def __len__(self) -> int:
        return self.len

    def __getitem__(self, value: Sequence[int]) -> List[int]:
        return self.get(value)


def prefix_allowed_tokens_fn(candidate_trie: Trie):
    """The closure HF's PrefixConstrainedLogitsProcessor calls (generation_trie.py:89-95).
    ``GRAM.generate`` recognises it (via ``__closure__``) and uses the flat Trie on the device."""

    def prefix_allowed_tokens(batch_id, sentence):
        sentence = sentence.tolist() if hasattr(sentence, "tolist") else list(sentence)
        return candidate_trie.get(sentence)

    return prefix_allowed_tokens


def exact_match(predictions, targets, k):
    """generation_trie.py:98-108: number of targets found among their k predictions."""
    correct = 0
    for b, t in enumerate(targets):
        if t in predictions[b * k:(b + 1) * k]:
            correct += 1
    return correct


class FlatTrie:
    """CSR layout of a :class:`Trie` (include/gram_hip.h ``gram_trie_t``).

    Node 0 is the root; children of node n are ``child_tok/child_node[child_off[n]:child_off[n+1]]``
    sorted by token id (the device binary-searches them).  Breadth-first numbering keeps each
    level contiguous: the upper levels, which every beam of every user touches, stay L2-resident.
    """

    def __init__(self, trie: Trie):
        if trie.append_trie is not None:
            raise ValueError("FlatTrie: append_trie is not supported (never set by the GRAM runners)")
        off, toks, nodes = [0], [], []
        queue = [trie.trie_dict]
        depth = [0]
        head = 0
        min_leaf = 0  # tokens on the path to the shallowest leaf = the shortest candidate sequence
        while head < len(queue):
            node = queue[head]
            if not node and head > 0 and (min_leaf == 0 or depth[head] < min_leaf):
                min_leaf = depth[head]
            for tok in sorted(node.keys()):
                toks.append(int(tok))
                nodes.append(len(queue))
                queue.append(node[tok])
                depth.append(depth[head] + 1)
            head += 1
            off.append(len(toks))
        self.min_seq_len = int(min_leaf)
        self.child_off = np.asarray(off, dtype=np.int32)
        self.child_tok = np.asarray(toks, dtype=np.int32)
        self.child_node = np.asarray(nodes, dtype=np.int32)
        self.n_nodes = len(queue)
        self.n_edges = len(toks)
        fan = np.diff(self.child_off)
        self.max_fanout = int(fan.max()) if len(fan) else 0
        self.n_sequences = len(trie)
        self._device = {}
        self._node_item = None  # (candidate list id, np.int32 [n_nodes]), see node_items()

    def leaf_of(self, sequence: Sequence[int]) -> int:
        """Node the whole sequence ends on (host walk of the CSR), or -1 if it leaves the tree."""
        node = 0
        for tok in sequence:
            lo, hi = int(self.child_off[node]), int(self.child_off[node + 1])
            i = lo + int(np.searchsorted(self.child_tok[lo:hi], tok))
            if i >= hi or self.child_tok[i] != tok:
                return -1
            node = int(self.child_node[i])
        return node

    def node_items(self, candidates: Sequence[Sequence[int]]) -> np.ndarray:
        """int32 [n_nodes]: index into `candidates` (the token sequences the Trie was built from, in the runner's order) of the
        candidate a LEAF node completes -- the first one when several candidates share a sequence -- and -1 elsewhere.  What
        ``gram_trie_item_index`` reads to turn returned sequences into item indices.  Level-wise walk: all candidates advance one
        token per round through the sorted child arrays (vectorised; 12 101 Beauty candidates take milliseconds)."""
        n = len(candidates)
        out = np.full(self.n_nodes, -1, dtype=np.int32)
        if n == 0:
            return out
        lens = np.fromiter((len(c) for c in candidates), dtype=np.int64, count=n)
        width = int(lens.max())
        toks = np.full((n, width), -1, dtype=np.int64)
        for i, c in enumerate(candidates):
            toks[i, : len(c)] = c
        node = np.zeros(n, dtype=np.int64)
        # global edge key (parent node, token): edges are stored parent-major with tokens ascending, so the keys are sorted
        parent = np.repeat(np.arange(self.n_nodes, dtype=np.int64), np.diff(self.child_off))
        edge_key = parent * (1 << 32) + self.child_tok.astype(np.int64)
        for p in range(width):
            live = (lens > p) & (node >= 0)
            key = node[live] * (1 << 32) + toks[live, p]
            e = np.searchsorted(edge_key, key)
            ok = (e < len(edge_key)) & (edge_key[np.minimum(e, len(edge_key) - 1)] == key)
            nxt = np.where(ok, self.child_node[np.minimum(e, len(edge_key) - 1)], -1)
            node[live] = nxt
        if (node < 0).any():
            raise ValueError("node_items: a candidate is not in the Trie it is being indexed against")
        fan = np.diff(self.child_off)
        leaf = fan[node] == 0
        idx = np.nonzero(leaf)[0]
        # first candidate wins: assign in reverse so that earlier indices overwrite later ones
        out[node[idx[::-1]]] = idx[::-1].astype(np.int32)
        return out

    # host-side walk with the same arrays the device uses (for tests)
    def get(self, prefix: Sequence[int]) -> List[int]:
        node = 0
        for tok in prefix:
            lo, hi = int(self.child_off[node]), int(self.child_off[node + 1])
            i = lo + int(np.searchsorted(self.child_tok[lo:hi], tok))
            if i >= hi or self.child_tok[i] != tok:
                return []
            node = int(self.child_node[i])
        return self.child_tok[self.child_off[node]:self.child_off[node + 1]].tolist()

    def to_device(self, device):
        """Upload once per device; returns (ctypes gram_trie_t, keep-alive tensors)."""
        import torch
        from .. import _lib

        key = str(device)
        if key not in self._device:
            t_off = torch.from_numpy(self.child_off).to(device)
            t_tok = torch.from_numpy(self.child_tok).to(device)
            t_node = torch.from_numpy(self.child_node).to(device)
            c = _lib.Trie(t_off.data_ptr(), t_tok.data_ptr(), t_node.data_ptr(), self.n_nodes, self.n_edges, self.max_fanout,
                          self.min_seq_len)
            self._device[key] = (c, (t_off, t_tok, t_node))
        return self._device[key]

    def node_items_on(self, device, candidates: Sequence[Sequence[int]]):
        """Device copy of :meth:`node_items` for this candidate list, built once per (list object, device)."""
        import torch

        key = ("items", str(device), id(candidates), len(candidates))
        hit = self._device.get(key)
        if hit is None or hit[0] is not candidates:
            hit = (candidates, torch.from_numpy(self.node_items(candidates)).to(device))
            self._device = {k: v for k, v in self._device.items() if not (isinstance(k, tuple) and k[0] == "items")}
            self._device[key] = hit
        return hit[1]
