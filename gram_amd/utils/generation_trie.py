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
    """The nested dict is built LAZILY: sequences are remembered as they are added and inserted into ``trie_dict`` on its first
    access.  ``GRAM.generate`` never needs the dict -- :class:`FlatTrie` is built straight from the sequences -- so an evaluation
    over 12 101 candidates does not pay for 100 k Python dict inserts; anything that reads ``trie_dict`` / ``get`` / iterates (the
    reference's own uses) sees exactly the dict the reference builds."""

    def __init__(self, sequences: Optional[Iterable[Sequence[int]]] = None):
        self._dict: Dict[int, dict] = {}
        self._pending: List[List[int]] = []   # added, not yet in the dict
        self._sequences: Optional[List[List[int]]] = []  # every sequence added so far; None once the dict may have other sources
        self.len = 0
        self.append_trie = None
        self.bos_token_id = None
        for seq in sequences or ():
            self.add(seq)

    @property
    def trie_dict(self) -> Dict[int, dict]:
        if self._pending:
            pending, self._pending = self._pending, []
            for seq in pending:
                node = self._dict
                for tok in seq:
                    nxt = node.get(tok)
                    if nxt is None:
                        nxt = node[tok] = {}
                    node = nxt
        self._sequences = None  # the caller holds the dict now and may change it: the sequence list is no longer authoritative
        return self._dict

    @trie_dict.setter
    def trie_dict(self, value: Dict[int, dict]) -> None:
        self._dict, self._pending, self._sequences = value, [], None

    def sequences(self) -> Optional[List[List[int]]]:
        """Every sequence added, in order -- or None when the tree may hold anything else (dict handed out or assigned)."""
        return self._sequences

    def append(self, trie: "Trie", bos_token_id: int) -> None:
        self.append_trie = trie
        self.bos_token_id = bos_token_id

    def add(self, sequence: Sequence[int]) -> None:
        seq = [int(t) for t in sequence]
        self._pending.append(seq)
        if self._sequences is not None:
            self._sequences.append(seq)
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
        self._device = {}
        self._node_item = None
        seqs = trie.sequences() if isinstance(trie, Trie) else None
        if seqs:
            self._from_sequences(seqs)
            self.n_sequences = len(trie)
            return
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

    def _from_sequences(self, seqs: Sequence[Sequence[int]]) -> None:
        """The same CSR (same breadth-first numbering, children sorted by token) without the nested dict: sort the sequences
        lexicographically; the nodes of depth p are the distinct length-p prefixes in that order."""
        n = len(seqs)
        lens = np.fromiter((len(q) for q in seqs), dtype=np.int64, count=n)
        width = int(lens.max()) if n else 0
        toks = np.full((n, max(width, 1)), -1, dtype=np.int64)
        for i, q in enumerate(seqs):
            toks[i, : len(q)] = q
        if n and int(toks[toks >= 0].min(initial=0)) < 0:
            raise ValueError("token ids must be >= 0")
        order = np.lexsort(toks.T[::-1]) if n else np.zeros(0, dtype=np.int64)
        toks, lens = toks[order], lens[order]
        node = np.zeros(n, dtype=np.int64)     # node of each sequence's prefix at the current depth
        n_nodes, tok_parts, node_parts, parent_parts, depth_parts = 1, [], [], [], [np.zeros(1, dtype=np.int64)]
        same = np.ones(n, dtype=bool)          # same[i]: row i has the same prefix (at this depth) as the previous LIVE row
        for p in range(width):
            live = np.nonzero(lens > p)[0]
            if live.size == 0:
                break
            sub = toks[live, : p + 1]
            new = np.ones(live.size, dtype=bool)
            new[1:] = (sub[1:] != sub[:-1]).any(axis=1)
            ids = n_nodes + np.cumsum(new) - 1
            first = live[new]
            tok_parts.append(toks[first, p])
            parent_parts.append(node[first])
            node_parts.append(ids[new])
            depth_parts.append(np.full(int(new.sum()), p + 1, dtype=np.int64))
            node[live] = ids
            n_nodes += int(new.sum())
        parent = np.concatenate(parent_parts) if parent_parts else np.zeros(0, dtype=np.int64)
        self.child_tok = (np.concatenate(tok_parts) if tok_parts else np.zeros(0, dtype=np.int64)).astype(np.int32)
        self.child_node = (np.concatenate(node_parts) if node_parts else np.zeros(0, dtype=np.int64)).astype(np.int32)
        self.child_off = np.concatenate([[0], np.cumsum(np.bincount(parent, minlength=n_nodes))]).astype(np.int32)
        self.n_nodes, self.n_edges = n_nodes, int(self.child_tok.size)
        fan = np.diff(self.child_off)
        self.max_fanout = int(fan.max()) if len(fan) else 0
        depth = np.concatenate(depth_parts)
        leaves = (fan == 0) & (np.arange(n_nodes) > 0)
        self.min_seq_len = int(depth[leaves].min()) if leaves.any() else 0
        # the candidate a leaf completes: the FIRST sequence (original order) that ends there
        end_leaf = fan[node] == 0
        item = np.full(n_nodes, -1, dtype=np.int64)
        idx = np.nonzero(end_leaf)[0]
        by_orig = idx[np.argsort(order[idx], kind="stable")][::-1]  # descending original index: the smallest is written last
        item[node[by_orig]] = order[by_orig]
        self._node_item = (seqs, item.astype(np.int32))

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
        if self._node_item is not None and (self._node_item[0] is candidates or (
                len(self._node_item[0]) == len(candidates) and list(map(list, candidates)) == self._node_item[0])):
            return self._node_item[1]  # (built with the tree, from the very sequences it is being asked about)
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
