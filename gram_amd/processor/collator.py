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

Parity: tests/golden/collator_cases.json holds outputs of the reference class itself, driven by
tests/stub_tokenizer.py (oracle/make_collator_fixtures.py); tests/test_host_logic.py compares bit for bit.
"""
from __future__ import annotations

from typing import Dict, List, Sequence

import torch

SPLIT_IDS = (1820, 9175)  # '|', '▁|' (hard-coded in the reference too)


def _strip_rows(ids: torch.Tensor, mask: torch.Tensor, limit: int):
    """Drop the separator ids from every row, keep the first `limit` survivors, force an EOS into the last
    kept slot if none survived, zero-pad to `limit`.  ids/mask: (R, W) int64 -> two (R, limit) tensors."""
    keep = (ids != SPLIT_IDS[0]) & (ids != SPLIT_IDS[1])
    out_ids = torch.zeros(ids.size(0), limit, dtype=torch.long)
    out_mask = torch.zeros(ids.size(0), limit, dtype=torch.long)
    for r in range(ids.size(0)):
        row, m = ids[r][keep[r]][:limit].clone(), mask[r][keep[r]][:limit]
        if row.numel() == 0:
            raise ValueError("a passage consists of separator tokens only")  # the reference fails on this input too
        if not bool((row == 1).any()):
            row[-1] = 1
        out_ids[r, : row.numel()] = row
        out_mask[r, : m.numel()] = m
    return out_ids, out_mask


class CollatorGRAM:
    def __init__(self, tokenizer, args=None, mode: str = "train"):
        self.tokenizer, self.args, self.mode = tokenizer, args, mode
        self.item_prompt_max_len = args.item_prompt_max_len
        self.target_max_len = args.target_max_len
        self.max_item_num = args.max_his
        self.item_id_type = args.item_id_type
        self.hierarchical_id_type = getattr(args, "hierarchical_id_type", None)

    # ------------------------------------------------------------------ targets (Collator.py:170-198, 281-340)
    def _targets(self, texts: Sequence[str]):
        if self.item_id_type == "t5_token":
            rows = [self.tokenizer.convert_tokens_to_ids(t.split(" ")) + [1] for t in texts]
            width = max(len(r) for r in rows)
            ids = torch.tensor([r + [0] * (width - len(r)) for r in rows])
            mask = torch.tensor([[1] * len(r) + [0] * (width - len(r)) for r in rows])
        elif self.item_id_type == "split":
            enc = self.tokenizer.batch_encode_plus(list(texts), max_length=99, padding="longest", return_tensors="pt", truncation=True)
            ids, mask = _strip_rows(enc["input_ids"], enc["attention_mask"], self.target_max_len)
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
        L = self.item_prompt_max_len
        n_slots = min(max(len(u) for u in users), self.max_item_num) + 1  # + the coarse-grained user prompt
        ids = torch.zeros(len(users), n_slots, L, dtype=torch.long)
        mask = torch.zeros(len(users), n_slots, L, dtype=torch.long)
        split = self.item_id_type == "split"
        for b, passages in enumerate(users):
            if len(passages) > n_slots:
                raise ValueError(f"user {b} has {len(passages)} passages for {n_slots} slots (max_his = {self.max_item_num})")
            enc = self.tokenizer.batch_encode_plus(list(passages), max_length=999 if split else L, pad_to_max_length=True,
                                                   return_tensors="pt", truncation=True)
            p_ids, p_mask = enc["input_ids"], enc["attention_mask"]
            if split:
                p_ids, p_mask = _strip_rows(p_ids, p_mask, L)
            ids[b, : len(passages)], mask[b, : len(passages)] = p_ids, p_mask
        width = int(mask.sum(-1).max())  # trim to the longest valid passage of the batch
        return ids[:, :, :width], mask[:, :, :width].bool()

    def encode_passages(self, texts: Sequence[str]):
        """Token ids / masks (P, L) of stand-alone passages, tokenised exactly as `_passages` tokenises them inside a
        user's stack (each row on its own).  Used to register the dataset's item prompts (`item2input`,
        test_dataset_gram.py:115-123) with `GRAM.cache_passages`."""
        L = self.item_prompt_max_len
        split = self.item_id_type == "split"
        enc = self.tokenizer.batch_encode_plus(list(texts), max_length=999 if split else L, pad_to_max_length=True,
                                               return_tensors="pt", truncation=True)
        ids, mask = enc["input_ids"], enc["attention_mask"]
        if split:
            ids, mask = _strip_rows(ids, mask, L)
        return ids, mask.bool()

    def __call__(self, batch: List[Dict]):
        target_ids, target_masks = self._targets([x["output"] for x in batch])
        item_text_ids, item_text_masks = self._passages([x["input"] for x in batch])
        return {
            "target_ids": target_ids, "target_masks": target_masks,          # B x T
            "item_text_ids": item_text_ids, "item_text_masks": item_text_masks,  # B x N x L
            "neg_item_ids": None, "neg_item_masks": None,
            "user_ids": [x["user_id"] for x in batch],
        }
