"""Input contract of the scoring path: mirror of the reference's ``CollatorGRAM``
(src/processor/Collator.py:152-450).

What ``GRAM.generate`` receives is defined here: per user a stack of granularity passages -- passage 0 the
coarse user prompt, passages 1..h the item prompts, most recent first -- as ``item_text_ids`` int64
(B, N, L) / ``item_text_masks`` bool (B, N, L) with N = min(max passages in the batch, max_his) + 1,
all-zero ids and all-False masks for missing passages, and L trimmed to the longest valid passage of the
batch; plus ``target_ids`` (B, T) with -100 at padded positions.

For ``item_id_type == "split"`` the two piece-separator ids (1820 = '|', 9175 = '▁|') are removed after
tokenisation, each row is cut to the length limit, an EOS (1) is forced into the last kept position when the
cut removed it, and rows are zero-padded back to the limit (Collator.py:281-340, 342-450).

Host cost.  The reference tokenises every passage of every user in every batch, one tokenizer call per
user and a Python loop per row.  A passage's token row depends on its text alone (each row of a ``batch_encode_plus`` call is
tokenised on its own, and the strip / cut / forced-EOS rule is per row), and a user's passages 1..h are ``item2input[item]``
strings shared by every user that has the item in its history (test_dataset_gram.py:112-123).  So: the rows of item prompts
and of target ids are tokenised ONCE and kept in a table (`_RowTable`: text -> row), a batch tokenises only the texts it has
not seen (one tokenizer call for all of them: the per-user prompts, mostly), `_strip_rows` is vectorised, and the (B, N, L)
tensors are assembled by one gather.  Same tensors bit for bit (tests/test_host_logic.py: the reference class's own outputs).

Parity: tests/golden/collator_cases.json holds outputs of the reference class itself, driven by
tests/stub_tokenizer.py (oracle/make_collator_fixtures.py); tests/test_host_logic.py compares bit for bit.
"""
from __future__ import annotations

from typing import Dict, List, Sequence

import numpy as np
import torch

SPLIT_IDS = (1820, 9175)  # '|', '▁|' (hard-coded in the reference too)


def _strip_rows(ids: torch.Tensor, mask: torch.Tensor, limit: int):
    """Drop the separator ids from every row, keep the first `limit` survivors, force an EOS into the last
    kept slot if none survived, zero-pad to `limit`.  ids/mask: (R, W) int64 -> two (R, limit) tensors.
    Vectorised: a survivor's destination column is the number of survivors before it."""
    ids, mask = ids.to(torch.long), mask.to(torch.long)
    R = ids.size(0)
    keep = (ids != SPLIT_IDS[0]) & (ids != SPLIT_IDS[1])
    dest = keep.cumsum(1) - 1
    kept = keep.sum(1)
    if R and int(kept.min()) == 0:
        raise ValueError("a passage consists of separator tokens only")  # the reference fails on this input too
    sel = keep & (dest < limit)
    rows = torch.arange(R).unsqueeze(1).expand_as(ids)[sel]
    cols = dest[sel]
    out_ids = torch.zeros(R, limit, dtype=torch.long)
    out_mask = torch.zeros(R, limit, dtype=torch.long)
    out_ids[rows, cols] = ids[sel]
    out_mask[rows, cols] = mask[sel]
    no_eos = ~(out_ids == 1).any(dim=1)
    if bool(no_eos.any()):
        last = kept.clamp(max=limit) - 1  # the last slot of the cut row (Collator.py: tmp_input_ids[-1] = 1)
        r = no_eos.nonzero().squeeze(1)
        out_ids[r, last[r]] = 1
    return out_ids, out_mask


class _RowTable:
    """text -> (ids row, mask row) of a fixed width, filled on first sight.  Row 0 is the all-zero / all-False row of a
    missing passage (Collator.py:410-436)."""

    def __init__(self, width: int):
        self.width = width
        self.index: Dict[str, int] = {}
        self.ids = np.zeros((1024, width), dtype=np.int64)
        self.mask = np.zeros((1024, width), dtype=np.bool_)
        self.n = 1

    def add(self, texts: Sequence[str], ids: torch.Tensor, mask: torch.Tensor) -> None:
        need = self.n + len(texts)
        if need > self.ids.shape[0]:
            cap = max(need, 2 * self.ids.shape[0])
            self.ids = np.concatenate([self.ids, np.zeros((cap - self.ids.shape[0], self.width), dtype=np.int64)])
            self.mask = np.concatenate([self.mask, np.zeros((cap - self.mask.shape[0], self.width), dtype=np.bool_)])
        self.ids[self.n:need] = ids.numpy()
        self.mask[self.n:need] = mask.numpy() != 0
        for i, t in enumerate(texts):
            self.index[t] = self.n + i
        self.n = need


class CollatorGRAM:
    def __init__(self, tokenizer, args=None, mode: str = "train"):
        self.tokenizer, self.args, self.mode = tokenizer, args, mode
        self.item_prompt_max_len = args.item_prompt_max_len
        self.target_max_len = args.target_max_len
        self.max_item_num = args.max_his
        self.item_id_type = args.item_id_type
        self.hierarchical_id_type = getattr(args, "hierarchical_id_type", None)
        self._passage_rows = _RowTable(self.item_prompt_max_len)   # item prompts (user-independent, §8f N2)
        self._target_rows = _RowTable(self.target_max_len) if self.item_id_type == "split" else None

    # ------------------------------------------------------------------ one tokenizer call for a list of texts
    def _tokenise_passages(self, texts: Sequence[str]):
        """(P, L) ids / masks of stand-alone passages: every row exactly as Collator.py:342-450 produces it inside a user's stack
        (rows of one call are tokenised independently; padding to the longest row of THIS call instead of to 999 only changes
        how many zero columns the strip sees)."""
        L = self.item_prompt_max_len
        if self.item_id_type == "split":
            enc = self.tokenizer.batch_encode_plus(list(texts), max_length=999, padding="longest", return_tensors="pt", truncation=True)
            return _strip_rows(enc["input_ids"], enc["attention_mask"], L)
        enc = self.tokenizer.batch_encode_plus(list(texts), max_length=L, pad_to_max_length=True, return_tensors="pt", truncation=True)
        return enc["input_ids"], enc["attention_mask"]

    def _rows_of(self, table: _RowTable, texts: Sequence[str], tokenise, remember) -> np.ndarray:
        """Row numbers of `texts` in `table`; unseen texts are tokenised in ONE call.  remember[i] False: text i is used for this
        batch only (a per-user prompt would never be looked up again) -- its row is appended past the table's end and dropped
        with the next call."""
        rows = np.empty(len(texts), dtype=np.int64)
        new, keep, pending = [], [], {}
        for i, t in enumerate(texts):
            r = table.index.get(t)
            if r is None:
                j = pending.get(t)
                if j is None:
                    j = pending[t] = len(new)
                    new.append(t)
                    keep.append(remember is None or bool(remember[i]))
                r = -1 - j
            rows[i] = r
        if new:
            perm = sorted(range(len(new)), key=lambda j: not keep[j])  # the texts to remember first (stable)
            ordered, n_keep = [new[j] for j in perm], sum(keep)
            ids, mask = tokenise(ordered)
            base = table.n
            table.add(ordered, ids, mask)
            for t in ordered[n_keep:]:
                del table.index[t]
            table.n = base + n_keep  # the one-off rows stay readable until the next add()
            new_row = np.empty(len(new), dtype=np.int64)
            new_row[np.asarray(perm)] = base + np.arange(len(new))
            fresh = rows < 0
            rows[fresh] = new_row[-1 - rows[fresh]]
        return rows

    # ------------------------------------------------------------------ targets (Collator.py:170-198, 281-340)
    def _tokenise_targets(self, texts: Sequence[str]):
        enc = self.tokenizer.batch_encode_plus(list(texts), max_length=99, padding="longest", return_tensors="pt", truncation=True)
        return _strip_rows(enc["input_ids"], enc["attention_mask"], self.target_max_len)

    def _targets(self, texts: Sequence[str]):
        if self.item_id_type == "t5_token":
            rows = [self.tokenizer.convert_tokens_to_ids(t.split(" ")) + [1] for t in texts]
            width = max(len(r) for r in rows)
            ids = torch.tensor([r + [0] * (width - len(r)) for r in rows])
            mask = torch.tensor([[1] * len(r) + [0] * (width - len(r)) for r in rows])
        elif self.item_id_type == "split":
            t = self._target_rows
            rows = self._rows_of(t, texts, self._tokenise_targets, None)
            ids, mask = torch.from_numpy(t.ids[rows]), torch.from_numpy(t.mask[rows])
            width = int(mask.sum(-1).max())
            ids, mask = ids[:, :width], mask[:, :width]
        else:
            capped = self.target_max_len > 0
            enc = self.tokenizer.batch_encode_plus(list(texts), max_length=self.target_max_len if capped else None,
                                                   padding="longest", return_tensors="pt", truncation=capped)
            ids, mask = enc["input_ids"], enc["attention_mask"]
        mask = mask.bool()
        return ids.masked_fill(~mask, -100), mask

    # ------------------------------------------------------------------ passages (Collator.py:224-279, 342-450)
    def _passages(self, users: Sequence[Sequence[str]]):
        n_slots = min(max(len(u) for u in users), self.max_item_num) + 1  # + the coarse-grained user prompt
        flat, where, remember = [], [], []
        for b, passages in enumerate(users):
            if len(passages) > n_slots:
                raise ValueError(f"user {b} has {len(passages)} passages for {n_slots} slots (max_his = {self.max_item_num})")
            flat.extend(passages)
            where.extend(b * n_slots + s for s in range(len(passages)))
            remember.extend(s > 0 for s in range(len(passages)))  # passage 0 is the user's own prompt, 1.. are item prompts
        t = self._passage_rows
        rows = np.zeros(len(users) * n_slots, dtype=np.int64)  # row 0 = a missing passage
        rows[np.asarray(where, dtype=np.int64)] = self._rows_of(t, flat, self._tokenise_passages, remember)
        ids = torch.from_numpy(t.ids[rows]).view(len(users), n_slots, t.width)
        mask = torch.from_numpy(t.mask[rows]).view(len(users), n_slots, t.width)
        width = int(mask.sum(-1).max())  # trim to the longest valid passage of the batch
        return ids[:, :, :width], mask[:, :, :width]

    def encode_passages(self, texts: Sequence[str]):
        """Token ids / masks (P, L) of stand-alone passages, tokenised exactly as `_passages` tokenises them inside a
        user's stack (each row on its own).  Used to register the dataset's item prompts (`item2input`,
        test_dataset_gram.py:115-123) with `GRAM.cache_passages`; the rows stay in the table for the batches to come."""
        t = self._passage_rows
        rows = self._rows_of(t, list(texts), self._tokenise_passages, None)
        return torch.from_numpy(t.ids[rows]), torch.from_numpy(t.mask[rows])

    def __call__(self, batch: List[Dict]):
        target_ids, target_masks = self._targets([x["output"] for x in batch])
        item_text_ids, item_text_masks = self._passages([x["input"] for x in batch])
        return {
            "target_ids": target_ids, "target_masks": target_masks,          # B x T
            "item_text_ids": item_text_ids, "item_text_masks": item_text_masks,  # B x N x L
            "neg_item_ids": None, "neg_item_masks": None,
            "user_ids": [x["user_id"] for x in batch],
        }
