"""GPU micro-benchmark (not a test): the two-piece ("x3") GEMMs of the scoring path at the bench shapes, with the epilogues the path
uses, in EXECUTED MFMA TFLOP/s (3 products per logical product).
    python tests/bench_gemm_x3.py [--batch 4096] [--only "enc qkv"]
    GRAM_LIB=gram_amd/csrc/libgram_hip_abl1.so python tests/bench_gemm_x3.py     # ablation builds (make ABL=n; results not checked)
ABL bits: 1 no tile-end epilogue, 2 no operand DMA after the prologue, 4 no LDS fragment reads, 16 the epilogue without its stores.
Every line carries the time-weighted in-kernel clock of its launches (gram_prof_pp_clock)."""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gram_amd import _lib  # noqa: E402
from tests import gpu_util as G  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--iters", type=int, default=6)
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    lib = _lib.load()
    Me, Md = a.batch * 384, a.batch * 20
    d, F, V = 768, 3072, 32128
    shapes = [("enc qkv", Me, 3 * d, d, _lib.EPI_BF16), ("enc o", Me, d, d, _lib.EPI_F32_ADD), ("enc wi", Me, F, d, _lib.EPI_BF16_RELU),
              ("enc wo", Me, d, F, _lib.EPI_F32_ADD), ("dec qkv", Md, 3 * d, d, _lib.EPI_BF16), ("dec wi", Md, F, d, _lib.EPI_BF16_RELU),
              ("dec wo", Md, d, F, _lib.EPI_F32_ADD), ("lm_head", Md, V, d, "lse"), ("kv bank", Me, 12 * 2 * d, d, "kv")]
    g = torch.Generator(device=G.DEV).manual_seed(0)
    for name, M, N, K, epi in shapes:
        if a.only and a.only != name:
            continue
        # interleaved two-piece operands [rows][2K]: random halves are as good as real pieces for timing (and for the chip's clock)
        A = (torch.randn(M, 2 * K, generator=g, device=G.DEV)).to(G.DT)
        W = (torch.randn(N, 2 * K, generator=g, device=G.DEV) * K ** -0.5).to(G.DT)
        keep = []
        if epi == "lse":
            part = torch.empty(M, N // 64 + 1, 2, dtype=torch.float32, device=G.DEV)
            sp = _lib.Split(2, 0, 0, 0, 0.0)
            run = lambda: _lib.check(lib.gram_gemm_bf16_lse_split(G.p(A), G.p(W), None, G.p(part), M, N, K, 2 * K, 0, C.byref(sp), G.stream()), "lse")
        elif epi == "kv":
            Bu, S, H, layers = a.batch, 384, 12, 12
            k = torch.empty(2, layers, Bu, H, S, 64, dtype=G.DT, device=G.DEV)
            vt = torch.empty(2, layers, Bu, H, S // 32, 64, 32, dtype=G.DT, device=G.DEV)
            bank = _lib.KVBank(k.data_ptr(), vt.data_ptr(), layers, Bu, H, S)
            sp = _lib.Split(2, 0, 0, k[0].numel(), 0.0)
            keep = [k, vt]
            run = lambda: _lib.check(lib.gram_gemm_bf16_split(G.p(A), G.p(W), None, M, N, K, 2 * K, 0, _lib.EPI_KV_BANK, C.byref(bank), None, C.byref(sp), G.stream()), "kv")
        elif epi == _lib.EPI_F32_ADD:
            x = torch.zeros(M, N, dtype=torch.float32, device=G.DEV)
            xb = torch.empty(M, 2 * N, dtype=G.DT, device=G.DEV)
            ss = torch.empty(M, N // 64, dtype=torch.float32, device=G.DEV)
            nf = _lib.NormFusion(xb.data_ptr(), ss.data_ptr(), None, 0, 0, 0.0, 0)
            sp = _lib.Split(2, 0, 0, 0, 0.0)
            keep = [x, xb, ss]
            run = lambda: _lib.check(lib.gram_gemm_bf16_split(G.p(A), G.p(W), G.p(x), M, N, K, 2 * K, N, epi, None, C.byref(nf), C.byref(sp), G.stream()), "f32add")
        else:
            inter = epi == _lib.EPI_BF16_RELU  # FFN-in feeds a GEMM (interleaved), QKV feeds the attention (planar pieces)
            y = torch.empty(2, M, N, dtype=G.DT, device=G.DEV)
            rs = torch.rand(M, dtype=torch.float32, device=G.DEV) + 0.5
            nf = _lib.NormFusion(None, None, rs.data_ptr(), 0, K, 1e-6, 0)
            sp = _lib.Split(2, int(inter), 0 if inter else M * N, 0, 0.0)
            keep = [y, rs]
            run = lambda: _lib.check(lib.gram_gemm_bf16_split(G.p(A), G.p(W), G.p(y), M, N, K, 2 * K, 2 * N if inter else N, epi, None, C.byref(nf), C.byref(sp), G.stream()), "bf16")
        for _ in range(2):
            run()
        torch.cuda.synchronize()
        lib.gram_prof_pp_clock_enable(1)
        _lib.check(lib.gram_prof_pp_clock(None, 1), "clock reset")
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(a.iters):
            run()
        e.record()
        torch.cuda.synchronize()
        us = s.elapsed_time(e) * 1e3 / a.iters
        line = f"{name:8s} M={M:8d} N={N:5d} K={K:4d}  {us:9.1f} us  {6.0 * M * N * K / us / 1e6:7.1f} TF executed"
        ghz = C.c_double(0.0)
        _lib.check(lib.gram_prof_pp_clock(C.byref(ghz), 1), "clock")
        line += f"  in-kernel clock {ghz.value:.3f} GHz"
        print(line, flush=True)
        del A, W, keep


if __name__ == "__main__":
    main()
