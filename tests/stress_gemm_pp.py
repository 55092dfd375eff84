"""Race screen for the ping-pong GEMM (not a pytest): a new barrier/vmcnt schedule has to be screened over many runs
at several sizes -- an early read of a staged buffer passes whenever the DMA happens to land first.  Random shapes and
epilogues; every output of variant 22 must equal the 256x128-tile kernel's (variant 3) bit for bit, repeatedly.
    python tests/stress_gemm_pp.py [seconds]"""
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
    rng = random.Random(1234)
    t0, n_cases, n_runs = time.time(), 0, 0
    while time.time() - t0 < budget:
        K = rng.choice([256, 512, 768, 1024, 2048, 3072])
        N = 256 * rng.randint(1, 12)
        M = rng.choice([rng.randint(1, 600), 256 * rng.randint(1, 300) + rng.choice([0, 0, 32, 100, 255]), rng.randint(20000, 120000)])
        epi = rng.choice([_lib.EPI_BF16, _lib.EPI_BF16_RELU, _lib.EPI_F32_ADD, _lib.EPI_F32])
        fused = rng.random() < 0.6
        g = torch.Generator().manual_seed(rng.randint(0, 1 << 30))
        A = torch.randn(M, K, generator=g).to(G.DEV).to(G.DT)
        W = (torch.randn(N, K, generator=g) * K ** -0.5).to(G.DEV).to(G.DT)
        rs = (torch.rand(M, generator=g) + 0.5).to(G.DEV)
        xs = torch.pow(2.0, torch.randint(-6, 7, (M,), generator=g).float()).to(G.DEV) if rng.random() < 0.7 else None  # gram_norm_fusion_t.xs_in
        base = torch.randn(M, N, generator=g).to(G.DEV) if epi == _lib.EPI_F32_ADD else None
        f32 = epi in (_lib.EPI_F32_ADD, _lib.EPI_F32)

        def run(v):
            lib.gram_debug_set_gemm_variant(v)
            C = base.clone() if base is not None else torch.zeros(M, N, dtype=torch.float32 if f32 else G.DT, device=G.DEV)
            extra = []
            if fused and epi == _lib.EPI_F32_ADD:
                xb = torch.zeros(M, N, dtype=G.DT, device=G.DEV)
                ss = torch.zeros(M, N // 64, dtype=torch.float32, device=G.DEV)
                nf = _lib.NormFusion(xb.data_ptr(), ss.data_ptr(), None, 0, 0, 0.0, 0, xs.data_ptr() if xs is not None else None, None)
                extra = [xb, ss]
            elif fused and not f32:
                nf = _lib.NormFusion(None, None, rs.data_ptr(), 0, K, 1e-6)
            else:
                nf = None
            _lib.check(lib.gram_gemm_bf16_ex(G.p(A), G.p(W), G.p(C), M, N, K, K, N, epi, None, ct.byref(nf) if nf else None,
                                             G.stream()), "gemm")
            torch.cuda.synchronize()
            return [C] + extra

        ref = run(3)
        for rep in range(3):
            got = run(22)
            n_runs += 1
            for a, b in zip(ref, got):
                if not torch.equal(a, b):
                    bad = (a.float() - b.float()).abs()
                    print(f"MISMATCH M={M} N={N} K={K} epi={epi} fused={fused} rep={rep}: {int((bad > 0).sum())} elements, max {float(bad.max())}")
                    return 1
        n_cases += 1
        del A, W, ref, got
    print(f"ok: {n_cases} random cases, {n_runs} ping-pong runs, all bit-identical to the reference kernel")
    return 0


if __name__ == "__main__":
    sys.exit(main())
