"""Golden vectors for the input contract: run the REFERENCE's own CollatorGRAM
(/root/reference/src/processor/Collator.py:152-450, loaded by path, nothing copied) with tests/stub_tokenizer.py
on synthetic batches and store inputs + outputs in tests/golden/collator_cases.json.
TEST INFRASTRUCTURE: run once in the build container (the reference does not exist on the GPU box).
    PYTHONDONTWRITEBYTECODE=1 python oracle/make_collator_fixtures.py"""
import importlib.util
import json
import os
import random
import sys
from types import SimpleNamespace

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.stub_tokenizer import StubTokenizer  # noqa: E402

spec = importlib.util.spec_from_file_location("ref_collator", "/root/reference/src/processor/Collator.py")
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)

WORDS = ["item:", "similar", "items:", "what", "would", "user", "purchase", "after", ";", "?", "rene", "furterer", "complexe",
         "shampoo", "oil", "|", "▁|", "organic", "5", "kernel", "mango", "butter", "lend", "said", "obtained", "generation"]


def text(rng, lo, hi):
    return " ".join(rng.choice(WORDS) for _ in range(rng.randint(lo, hi)))


def batch(rng, B, max_pass, lo, hi):
    out = []
    for b in range(B):
        n = rng.randint(1, max_pass)
        out.append({"input": [text(rng, lo, hi) for _ in range(n)], "output": text(rng, 2, 9), "user_id": f"U{rng.randint(0, 10 ** 6)}"})
    return out


def main():
    rng = random.Random(2023)
    cases = []
    for item_id_type, max_his, ipl, tml, B, max_pass, lo, hi in [
        ("split", 20, 128, 32, 4, 9, 3, 40), ("split", 3, 16, 8, 5, 4, 1, 30), ("split", 20, 32, 6, 1, 1, 40, 60),
        ("t5_token", 5, 24, 32, 3, 4, 2, 20), ("other", 4, 20, 10, 4, 5, 2, 30), ("other", 4, 20, -1, 2, 3, 2, 12),
    ]:
        args = SimpleNamespace(item_prompt_max_len=ipl, target_max_len=tml, max_his=max_his, item_id_type=item_id_type,
                               hierarchical_id_type="none")
        col = ref.CollatorGRAM(StubTokenizer(), args, mode="test")
        b = batch(rng, B, max_pass, lo, hi)
        o = col(b)
        cases.append({"args": vars(args), "batch": b,
                      "target_ids": o["target_ids"].tolist(), "target_masks": o["target_masks"].long().tolist(),
                      "item_text_ids": o["item_text_ids"].tolist(), "item_text_masks": o["item_text_masks"].long().tolist(),
                      "user_ids": o["user_ids"]})
    path = os.path.join(ROOT, "tests", "golden", "collator_cases.json")
    json.dump(cases, open(path, "w"))
    print("wrote", path, [(c["args"]["item_id_type"], len(c["item_text_ids"]), len(c["item_text_ids"][0]), len(c["item_text_ids"][0][0])) for c in cases])


if __name__ == "__main__":
    main()
