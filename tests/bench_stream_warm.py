"""GPU micro-benchmark (not a test): what a one-user decoder GEMM (gemm_stream_kernel, two-piece operands) costs with its weights
cold (cycling through more weight matrices than the 256-MiB Infinity Cache holds, as a decode step does: ~400 MB of decoder weights)
against warm (the same matrix again and again: L2 / Infinity Cache resident) -- the upper bound of what prefetching the NEXT launch's
weights from otherwise idle CUs could buy.
    python tests/bench_stream_warm.py"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gram_amd import _lib  # noqa: E402
from tests import gpu_util as G  # noqa: E402


def main():
    lib = _lib.load()
    g = torch.Generator(device=G.DEV).manual_seed(0)
    for name, M, N, K, epi in (("o/q", 20, 768, 768, _lib.EPI_F32_ADD), ("qkv", 20, 2304, 768, _lib.EPI_BF16), ("wi", 20, 3072, 768, _lib.EPI_BF16_RELU),
                               ("wo", 20, 768, 3072, _lib.EPI_F32_ADD), ("o/q B=4", 80, 768, 768, _lib.EPI_F32_ADD)):
        nW = max(2, int(600e6 // (N * 2 * K * 2)))  # > 256 MiB of weights in the cold cycle
        Ws = [(torch.randn(N, 2 * K, generator=g, device=G.DEV) * K ** -0.5).to(G.DT) for _ in range(nW)]
        A = torch.randn(M, 2 * K, generator=g, device=G.DEV).to(G.DT)
        sp = _lib.Split(2, 1, 0, 0, 0.0)
        if epi == _lib.EPI_F32_ADD:
            Cc = torch.zeros(M, N, dtype=torch.float32, device=G.DEV)
            xb = torch.empty(M, 2 * N, dtype=G.DT, device=G.DEV)
            ss = torch.empty(M, N // 16, dtype=torch.float32, device=G.DEV)
            nf = _lib.NormFusion(xb.data_ptr(), ss.data_ptr(), None, 0, 0, 0.0, 1)
            sp = _lib.Split(2, 0, 0, 0, 0.0)
            ldc = N
        else:
            Cc = torch.empty(M, 2 * N, dtype=G.DT, device=G.DEV)
            rs = torch.rand(M, dtype=torch.float32, device=G.DEV) + 0.5
            nf = _lib.NormFusion(None, None, rs.data_ptr(), 0, K, 1e-6, 0)
            ldc = 2 * N

        def run(W):
            _lib.check(lib.gram_gemm_bf16_split(G.p(A), G.p(W), G.p(Cc), M, N, K, 2 * K, ldc, epi, None, C.byref(nf), C.byref(sp), G.stream()), "gemm")

        res = {}
        for mode in ("cold", "warm"):
            for _ in range(3):
                run(Ws[0])
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 3 * nW
            s.record()
            for i in range(reps):
                run(Ws[i % nW] if mode == "cold" else Ws[0])
            e.record()
            torch.cuda.synchronize()
            res[mode] = s.elapsed_time(e) * 1e3 / reps
        print(f"{name:8s} M={M:3d} N={N:5d} K={K:4d}  {nW:3d} weight matrices  cold {res['cold']:6.2f} us  warm {res['warm']:6.2f} us per launch (back to back, launch floor included)", flush=True)
        del Ws


if __name__ == "__main__":
    main()
