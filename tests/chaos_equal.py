"""Whole-path race screen (not a pytest; run on an MI355X): generate() at the bench's shape -- T5-base, the Beauty Trie, 3 x 128-token passages,
beam 20 -- once per library in a process of its own, results compared BIT FOR BIT: the product library against the chaos build
(make -C gram_amd/csrc CHAOS=1: a random sleep of up to ~3.5 us behind every workgroup barrier, common.h).  Nothing in the path depends on the
order in which workgroups or waves run (no floating-point atomics, no split reductions), so any difference is a race.
    python tests/chaos_equal.py [users] [ragged 0/1]        (parent)      GRAM_LIB=... python tests/chaos_equal.py --child out.pt users ragged"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(out_path, users, ragged):
    import numpy as np
    import torch

    import gram_amd
    from gram_amd.utils import generation_trie as gt
    dev = torch.device("cuda:0")
    cfg = gram_amd.T5Config.named("t5-base")
    torch.manual_seed(2023)
    model = gram_amd.create_model("gram", cfg)
    with torch.no_grad():
        for name, p_ in model.named_parameters():
            if name.endswith(".q.weight"):
                p_.mul_(4.0)
    model = model.to(dev).eval()
    z = np.load(os.path.join(ROOT, "tests", "golden", "tries.npz"))
    cands = [[int(x) for x in row if x >= 0] for row in z["Beauty_cands"]]
    fn = gt.prefix_allowed_tokens_fn(gt.Trie(cands))
    max_length = max(len(c) for c in cands)
    B, N, L, K = users, 3, 128, 20
    g = torch.Generator().manual_seed(99)
    ids = torch.randint(2, 32100, (B, N, L), generator=g)
    ids[:, :, -1] = 1
    mask = torch.ones(B, N, L, dtype=torch.bool)
    if ragged:
        lens = torch.randint(32, L + 1, (B, N), generator=g)
        n_user = torch.randint(1, N + 1, (B,), generator=g)
        mask = (torch.arange(L)[None, None, :] < lens[:, :, None]) & (torch.arange(N)[None, :, None] < n_user[:, None, None])
        ids[~mask] = 0
    outs = []
    for rep in range(2):
        out = model.generate(input_ids=ids.to(dev), attention_mask=mask.to(dev), max_length=max_length, prefix_allowed_tokens_fn=fn,
                             num_beams=K, num_return_sequences=K, output_scores=True, return_dict_in_generate=True, length_penalty=1.0)
        outs.append((out["sequences"].cpu(), out["sequences_scores"].cpu()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]), "two runs of one library differ"
    torch.save({"sequences": outs[0][0], "scores": outs[0][1]}, out_path)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        return child(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]))
    import tempfile

    import torch
    users = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    ragged = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    libs = {"product": os.path.join(ROOT, "gram_amd", "csrc", "libgram_hip.so"), "chaos": os.path.join(ROOT, "gram_amd", "csrc", "libgram_hip_chaos.so")}
    res = {}
    with tempfile.TemporaryDirectory() as tmp:
        for name, lib in libs.items():
            out = os.path.join(tmp, name + ".pt")
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", out, str(users), str(ragged)], env=dict(os.environ, GRAM_LIB=lib),
                               capture_output=True, text=True, timeout=900)
            if p.returncode != 0:
                print(f"[chaos equal] {name} failed:\n{p.stdout[-800:]}{p.stderr[-1500:]}")
                return 1
            res[name] = torch.load(out, weights_only=True)
    same_seq = torch.equal(res["product"]["sequences"], res["chaos"]["sequences"])
    same_sc = torch.equal(res["product"]["scores"], res["chaos"]["scores"])
    d = (res["product"]["scores"] - res["chaos"]["scores"]).abs()
    print(f"[chaos equal] {users} users, ragged={ragged}: sequences {'identical' if same_seq else 'DIFFER'}, scores {'identical' if same_sc else 'DIFFER'} "
          f"(max |diff| {float(d.max()):.3g}, {int((d > 0).sum())} of {d.numel()} scores)")
    return 0 if same_seq and same_sc else 1


if __name__ == "__main__":
    sys.exit(main())
