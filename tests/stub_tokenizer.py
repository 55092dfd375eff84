"""A deterministic stand-in for the T5 SentencePiece tokenizer (none exists offline): whitespace pieces -> ids
by CRC, '|' -> 1820 and '▁|' -> 9175 (the two ids the reference collator strips), EOS 1 appended, pad 0.
It implements just the calls the reference collator makes (`batch_encode_plus`, `convert_tokens_to_ids`) with
HF's conventions (truncation keeps room for EOS; `pad_to_max_length` / `padding="longest"`; `return_tensors="pt"`),
and is used both by oracle/make_collator_fixtures.py (driving the reference's own CollatorGRAM) and by the tests
(driving gram_amd.processor.CollatorGRAM), so the comparison pins everything the collator itself does."""
import zlib

import torch

SPECIAL = {"|": 1820, "▁|": 9175}


class StubTokenizer:
    pad_token_id, eos_token_id = 0, 1

    def convert_tokens_to_ids(self, tokens):
        return [self._id(t) for t in tokens]

    @staticmethod
    def _id(tok):
        if tok in SPECIAL:
            return SPECIAL[tok]
        v = 2 + zlib.crc32(tok.encode("utf-8")) % 32000
        return v + 1 if v in (1820, 9175) else v

    def encode(self, text):
        return self.convert_tokens_to_ids(text.split()) + [1]

    def batch_decode(self, rows, skip_special_tokens=True):
        """ids are CRCs, not invertible: decode to the id string, which keeps the equality relation the metric uses"""
        return [" ".join(str(t) for t in r if not (skip_special_tokens and t in (0, 1))) for r in rows]

    def batch_encode_plus(self, texts, max_length=None, padding=False, pad_to_max_length=False, truncation=False,
                          return_tensors=None):
        rows = []
        for t in texts:
            ids = self.convert_tokens_to_ids(t.split())
            if truncation and max_length is not None:
                ids = ids[: max_length - 1]
            rows.append(ids + [1])
        if pad_to_max_length:
            width = max_length
        elif padding in ("longest", True):
            width = max(len(r) for r in rows)
        else:
            width = None
        masks = [[1] * len(r) for r in rows]
        if width is not None:
            masks = [m + [0] * (width - len(m)) for m in masks]
            rows = [r + [0] * (width - len(r)) for r in rows]
        if return_tensors == "pt":
            return {"input_ids": torch.tensor(rows, dtype=torch.long), "attention_mask": torch.tensor(masks, dtype=torch.long)}
        return {"input_ids": rows, "attention_mask": masks}
