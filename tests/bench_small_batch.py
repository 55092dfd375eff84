"""Latency of one generate() call at small batch sizes (the reference evaluates one user per call: --eval_batch_size 1), configs[1]
shapes (T5-base, 3 x 128 passages, Beauty Trie, beam 20).  Not collected by pytest.
    python tests/bench_small_batch.py [--precision f16x3] [--batches 1,2,4,8,16,32,64] [--iters 20]
A/B hooks are process-wide environment variables (GRAM_GEMM_STREAM_MAXM, ...): run it once per setting."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--precision", default="f16x3")
    ap.add_argument("--batches", default="1,2,4,8,16,32,64")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--tag", default="")
    args = ap.parse_args()
    import gram_amd
    from gram_amd.utils import generation_trie as gt
    dev = torch.device("cuda:0")
    torch.manual_seed(2023)
    model = gram_amd.create_model("gram", gram_amd.T5Config.named("t5-base")).to(dev).eval()
    model.set_precision(args.precision)
    z = np.load(os.path.join(ROOT, "tests", "golden", "tries.npz"))
    cands = [[int(x) for x in row if x >= 0] for row in z["Beauty_cands"]]
    fn = gt.prefix_allowed_tokens_fn(gt.Trie(cands))
    max_length = max(len(c) for c in cands)
    res = {}
    for B in [int(x) for x in args.batches.split(",")]:
        g = torch.Generator().manual_seed(1000)
        ids = torch.randint(2, 32100, (B, 3, 128), generator=g)
        ids[:, :, -1] = 1
        ids_d, mask_d = ids.to(dev), torch.ones(B, 3, 128, dtype=torch.bool, device=dev)

        def step():
            return model.generate(input_ids=ids_d, attention_mask=mask_d, max_length=max_length, prefix_allowed_tokens_fn=fn, num_beams=20,
                                  num_return_sequences=20, output_scores=True, return_dict_in_generate=True, length_penalty=1.0)
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.iters):
            step()
        torch.cuda.synchronize()
        res[B] = round((time.perf_counter() - t0) / args.iters * 1e3, 3)
    print(json.dumps({"tag": args.tag, "precision": args.precision, "stream_max_m": os.environ.get("GRAM_GEMM_STREAM_MAXM", "default"),
                      "ms_per_generate": res, "users_per_s": {b: round(b / v * 1e3, 1) for b, v in res.items()}}))


if __name__ == "__main__":
    main()
