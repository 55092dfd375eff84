"""GPU micro-benchmark (not a test): the encoder self-attention at the bench shape (P = 3 * batch passages of 128 tokens, 12 heads,
two-piece operands), in ms per launch and TB/s of its algorithmic bytes (q|k|v pieces in, interleaved pieces out).
    python tests/bench_enc_attn.py [--batch 4096]
    GRAM_LIB=gram_amd/csrc/libgram_hip_eabl<n>.so python tests/bench_enc_attn.py      # ablation builds (make EABL=n; results wrong)"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gram_amd import _lib  # noqa: E402
from tests import gpu_util as G  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--iters", type=int, default=5)
    a = ap.parse_args()
    lib = _lib.load()
    P, L, H = a.batch * 3, 128, 12
    inner = H * 64
    g = torch.Generator(device=G.DEV).manual_seed(0)
    qkv = torch.empty(2, P * L, 3 * inner, dtype=G.DT, device=G.DEV)
    for pc in range(2):
        for lo in range(0, P * L, 1 << 18):
            qkv[pc, lo:lo + (1 << 18)] = (torch.randn(min(1 << 18, P * L - lo), 3 * inner, generator=g, device=G.DEV) * (1.0 if pc == 0 else 2.0 ** -11)).to(G.DT)
    bias = (torch.randn(H, 255, generator=g, device=G.DEV) * 0.5).contiguous()
    mask = torch.ones(P, L, dtype=torch.uint8, device=G.DEV)
    out = torch.empty(P * L, 2 * inner, dtype=G.DT, device=G.DEV)
    run = lambda: _lib.check(lib.gram_enc_self_attn_split(G.p(qkv), G.p(bias), G.p(mask), G.p(out), P, L, H, 2, qkv[0].numel(), G.stream()), "enc_attn")
    for _ in range(2):
        run()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(a.iters):
        run()
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / a.iters
    nbytes = qkv.numel() * 2 + out.numel() * 2
    print(f"enc_attn P={P} L={L} H={H} two-piece: {ms:7.3f} ms per launch, {nbytes / ms / 1e9:6.2f} TB/s of {nbytes / 1e9:.1f} GB algorithmic", flush=True)


if __name__ == "__main__":
    main()
