"""GPU micro-benchmark (not a test): gram_gemm_bf16 variants on the shapes of the scoring path.
    python tests/bench_gemm.py [--batch 128]
Checks each variant against torch on a sampled block, then times it with HIP events."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gram_amd import _lib  # noqa: E402
from tests import gpu_util as G  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--variants", default="1,3,22", help="0 reg-staged 128x128, 1 LDS-DMA 128x128, 3 256x128, 22 ping-pong 256x256, 30 skinny, 31 64x128")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--only", default="")
    ap.add_argument("--stagger", type=int, default=-1)
    ap.add_argument("--fullcheck", action="store_true", help="compare every output row against torch (default: first 256)")
    ap.add_argument("--fuse", action="store_true", help="time the folded-layernorm epilogues (gram_norm_fusion_t)")
    a = ap.parse_args()
    lib = _lib.load()
    if a.stagger >= 0:
        lib.gram_debug_set_gemm_variant(1000 + a.stagger)
    Me, Md = a.batch * 384, a.batch * 20
    shapes = [("enc qkv", Me, 2304, 768, _lib.EPI_BF16), ("enc o", Me, 768, 768, _lib.EPI_F32_ADD),
              ("enc wi", Me, 3072, 768, _lib.EPI_BF16_RELU), ("enc wo", Me, 768, 3072, _lib.EPI_F32_ADD),
              ("dec qkv", Md, 2304, 768, _lib.EPI_BF16), ("dec o/q", Md, 768, 768, _lib.EPI_F32_ADD),
              ("dec wi", Md, 3072, 768, _lib.EPI_BF16_RELU), ("dec wo", Md, 768, 3072, _lib.EPI_F32_ADD),
              ("lm_head", Md, 32128, 768, _lib.EPI_F32)]
    g = torch.Generator().manual_seed(0)
    for name, M, N, K, epi in shapes:
        if a.only and a.only != name:
            continue
        A = (torch.randn(M, K, generator=g)).to(G.DEV).to(G.DT)
        W = (torch.randn(N, K, generator=g) * K ** -0.5).to(G.DEV).to(G.DT)
        f32 = epi in (_lib.EPI_F32_ADD, _lib.EPI_F32)
        C = torch.zeros(M, N, dtype=torch.float32 if f32 else G.DT, device=G.DEV)
        ref = (A[:256].float() @ W.float().T)
        if epi == _lib.EPI_BF16_RELU:
            ref = ref.clamp(min=0)
        line = f"{name:8s} M={M:6d} N={N:5d} K={K:4d}"
        for v in [int(x) for x in a.variants.split(",")]:
            lib.gram_debug_set_gemm_variant(v)
            C.zero_()
            try:
                G.gemm(A, W, epi, C)
            except Exception:
                line += f" | v{v}: n/a"
                continue
            torch.cuda.synchronize()
            err = (C[:256].float() - ref).abs().max().item()
            if a.fullcheck:
                for r0 in range(0, M, 8192):
                    rr = A[r0:r0 + 8192].float() @ W.float().T
                    if epi == _lib.EPI_BF16_RELU:
                        rr = rr.clamp(min=0)
                    err = max(err, (C[r0:r0 + 8192].float() - rr).abs().max().item())
            ok = err < (2e-3 if f32 else 3e-2)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            run = lambda: G.gemm(A, W, epi, C)
            if a.fuse and epi != _lib.EPI_F32:
                import ctypes as ct
                if epi == _lib.EPI_F32_ADD:
                    xb = torch.empty(M, N, dtype=G.DT, device=G.DEV)
                    ss = torch.empty(M, N // 64, dtype=torch.float32, device=G.DEV)
                    nf = _lib.NormFusion(xb.data_ptr(), ss.data_ptr(), None, 0, 0, 0.0)
                else:
                    ss = torch.rand(M, K // 64, dtype=torch.float32, device=G.DEV) + 1
                    nf = _lib.NormFusion(None, None, ss.data_ptr(), K // 64, K, 1e-6)
                run = lambda: _lib.check(lib.gram_gemm_bf16_ex(G.p(A), G.p(W), G.p(C), M, N, K, K, N, epi, None, ct.byref(nf), G.stream()), "ex")
            for _ in range(3):
                run()
            s.record()
            for _ in range(a.iters):
                run()
            e.record()
            torch.cuda.synchronize()
            us = s.elapsed_time(e) * 1e3 / a.iters
            line += f" | v{v}: {us:8.1f} us {2.0 * M * N * K / us / 1e6:7.1f} TF {'ok' if ok else 'BAD err=%.3g' % err}"
        print(line, flush=True)
        del A, W, C
    lib.gram_debug_set_gemm_variant(-1)


if __name__ == "__main__":
    main()
