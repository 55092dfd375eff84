"""Race screen for the TWO-PIECE ping-pong GEMM (not a pytest; companion of stress_gemm_pp.py): its 16-bit outputs leave in the load
slots around the tile boundary with stores counted into the DMA waits (gemm.hip, "in-load-slot epilogue") -- a wrong count reads a
half-tile before its DMA has landed only when the timing allows it, so the schedule is screened over many shapes and runs.  Random
shapes / epilogues / layouts; every output of variant 22 must equal the 256x128-tile kernel's (variant 3) bit for bit, repeatedly.
    python tests/stress_gemm_pp_x3.py [seconds]"""
import ctypes as ct
import os
import random
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gram_amd import _lib  # noqa: E402
from tests import gpu_util as G  # noqa: E402


def main(budget=None, stamps=None, entry_delay=None):
    """-> 0 (all bit-identical) / 1 (a mismatch, printed).  Callable in-process (tests/test_gpu_kernels.py) or as a script."""
    budget = float(budget if budget is not None else sys.argv[1] if len(sys.argv) > 1 else 120.0)
    stamps = int(stamps if stamps is not None else os.environ.get("STAMPS", "0"))
    entry_delay = int(entry_delay if entry_delay is not None else os.environ.get("ENTRY_DELAY", "0"))
    lib = _lib.load()
    try:
        return _screen(lib, budget, stamps, entry_delay)
    finally:  # leave the library as the product path runs it
        lib.gram_debug_set_gemm_variant(-1)
        lib.gram_debug_set_gemm_variant(2000)
        lib.gram_prof_pp_clock_enable(0)


def _screen(lib, budget, stamps, entry_delay):
    lib.gram_prof_pp_clock_enable(stamps)  # the diagnostic clock stamps change the kernel's timing: screen both
    # ENTRY_DELAY=n: wave group 1 enters the prologue n x 512 cycles late (gram_debug_set_gemm_variant(2000 + n); honoured by the chaos build,
    # GRAM_LIB=.../libgram_hip_chaos.so: the timing that exposed the prologue's missing barrier in round 4, made deterministic)
    lib.gram_debug_set_gemm_variant(2000 + entry_delay)
    rng = random.Random(4321)
    t0, n_cases, n_runs = time.time(), 0, 0
    while time.time() - t0 < budget:
        K = rng.choice([128, 256, 384, 768, 1024, 3072])  # logical; physical 2K (>= 4 k-tiles of 64)
        N = 256 * rng.randint(1, 12)
        M = rng.choice([256 * rng.randint(1, 300) + rng.choice([0, 0, 32, 100, 129, 255]), rng.randint(20000, 160000), 256 * rng.randint(600, 1200)])
        epi = rng.choice([_lib.EPI_BF16, _lib.EPI_BF16_RELU, _lib.EPI_BF16, _lib.EPI_BF16_RELU, _lib.EPI_F32_ADD])
        inter = rng.random() < 0.5
        fused = rng.random() < 0.7
        g = torch.Generator(device=G.DEV).manual_seed(rng.randint(0, 1 << 30))
        A = torch.randn(M, 2 * K, generator=g, device=G.DEV).to(G.DT)
        W = (torch.randn(N, 2 * K, generator=g, device=G.DEV) * K ** -0.5).to(G.DT)
        rs = torch.rand(M, generator=g, device=G.DEV) + 0.5
        base = torch.randn(M, N, generator=g, device=G.DEV) if epi == _lib.EPI_F32_ADD else None
        # the row factors of the 16-bit copy (gram_norm_fusion_t.xs_in: a power of two per row), as every generate() passes them
        xs = torch.pow(2.0, torch.randint(-6, 7, (M,), generator=g, device=G.DEV).float()) if rng.random() < 0.7 else None

        def run(v):
            lib.gram_debug_set_gemm_variant(v)
            extra = []
            if epi == _lib.EPI_F32_ADD:
                C = base.clone()
                xb = torch.zeros(M, 2 * N, dtype=G.DT, device=G.DEV)
                ss = torch.zeros(M, N // 64, dtype=torch.float32, device=G.DEV)
                nf = _lib.NormFusion(xb.data_ptr(), ss.data_ptr(), None, 0, 0, 0.0, 0, xs.data_ptr() if xs is not None else None, None) if fused else None
                extra = [xb, ss]
                sp = _lib.Split(2, 0, 0, 0, 0.0)
                ldc = N
            else:
                C = torch.zeros(M, 2 * N, dtype=G.DT, device=G.DEV)
                nf = _lib.NormFusion(None, None, rs.data_ptr(), 0, K, 1e-6, 0) if fused else None
                sp = _lib.Split(2, int(inter), 0 if inter else M * N, 0, 0.0)
                ldc = 2 * N if inter else N
            _lib.check(lib.gram_gemm_bf16_split(G.p(A), G.p(W), G.p(C), M, N, K, 2 * K, ldc, epi, None, ct.byref(nf) if nf else None,
                                                ct.byref(sp), G.stream()), "gemm")
            torch.cuda.synchronize()
            return [C] + extra

        ref = run(3)
        for rep in range(3):
            got = run(22)
            n_runs += 1
            for a, b in zip(ref, got):
                if not torch.equal(a, b):
                    bad = (a.float() - b.float()).abs()
                    idx = torch.nonzero(bad > 0)
                    print(f"MISMATCH M={M} N={N} K={K} epi={epi} inter={inter} fused={fused} rep={rep}: {idx.shape[0]} elements, max {bad.max().item():.4g}, "
                          f"first at {idx[0].tolist()}", flush=True)
                    return 1
        n_cases += 1
        del A, W, ref, got
        if n_cases % 20 == 0:
            print(f"[stress x3] {n_cases} cases, {n_runs} runs, {time.time() - t0:.0f} s", flush=True)
    print(f"[stress x3] OK: {n_cases} cases x 3 runs bit-identical to the 256x128-tile kernel in {time.time() - t0:.0f} s")
    return 0


if __name__ == "__main__":
    sys.exit(main())
