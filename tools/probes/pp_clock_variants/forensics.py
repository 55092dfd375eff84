"""What exactly is wrong in a mismatching tile of the variant builds (tools/probes/pp_clock_variants/build.sh; GRAM_LIB selects one)?
Runs the two-piece race screen's case sequence (tests/stress_gemm_pp_x3.py, fp32-residual cases only); on the first mismatch it maps the
wrong elements onto the ping-pong kernel's geometry (256 x 256 tiles, wave blocks of 128 rows x 64 columns, the workgroup that owned the
tile) and fits the error against "one k-tile's contribution computed from a stale LDS half-tile" hypotheses.
    GRAM_LIB=.../libgram_hip_bad.so python tools/probes/pp_clock_variants/forensics.py [seconds]"""
import ctypes as ct
import os
import random
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from gram_amd import _lib  # noqa: E402
from tests import gpu_util as G  # noqa: E402


def pieces(X, K):
    """[rows][2K] interleaved -> (p0, p1) fp64 [rows][K]"""
    x = X.double().view(X.shape[0], K // 32, 2, 32)
    return x[:, :, 0, :].reshape(X.shape[0], K), x[:, :, 1, :].reshape(X.shape[0], K)


def contrib(a0, a1, w0, w1, ks):
    return a0[:, ks] @ w1[:, ks].T + a1[:, ks] @ w0[:, ks].T + a0[:, ks] @ w0[:, ks].T


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    lib = _lib.load()
    rng = random.Random(4321)
    t0 = time.time()
    n = 0
    while time.time() - t0 < budget:
        K = rng.choice([128, 256, 384, 768, 1024, 3072])
        N = 256 * rng.randint(1, 12)
        M = rng.choice([256 * rng.randint(1, 300) + rng.choice([0, 0, 32, 100, 129, 255]), rng.randint(20000, 160000), 256 * rng.randint(600, 1200)])
        epi = rng.choice([_lib.EPI_BF16, _lib.EPI_BF16_RELU, _lib.EPI_BF16, _lib.EPI_BF16_RELU, _lib.EPI_F32_ADD])
        inter = rng.random() < 0.5
        fused = rng.random() < 0.7
        seed = rng.randint(0, 1 << 30)
        if epi != _lib.EPI_F32_ADD:
            rng.random()  # (the screen draws the xs coin for every case)
            continue
        g = torch.Generator(device=G.DEV).manual_seed(seed)
        A = torch.randn(M, 2 * K, generator=g, device=G.DEV).to(G.DT)
        W = (torch.randn(N, 2 * K, generator=g, device=G.DEV) * K ** -0.5).to(G.DT)
        torch.rand(M, generator=g, device=G.DEV)  # (the screen's row scales: keeps the generator in step)
        base = torch.randn(M, N, generator=g, device=G.DEV)
        use_xs = rng.random() < 0.7

        xs = torch.pow(2.0, torch.randint(-6, 7, (M,), generator=g, device=G.DEV).float()) if use_xs else None

        def run(v):
            lib.gram_debug_set_gemm_variant(v)
            C = base.clone()
            xb = torch.zeros(M, 2 * N, dtype=G.DT, device=G.DEV)
            ss = torch.zeros(M, N // 64, dtype=torch.float32, device=G.DEV)
            nf = _lib.NormFusion(xb.data_ptr(), ss.data_ptr(), None, 0, 0, 0.0, 0, xs.data_ptr() if xs is not None else None, None) if fused else None
            sp = _lib.Split(2, 0, 0, 0, 0.0)
            _lib.check(lib.gram_gemm_bf16_split(G.p(A), G.p(W), G.p(C), M, N, K, 2 * K, N, epi, None, ct.byref(nf) if nf else None,
                                                ct.byref(sp), G.stream()), "gemm")
            torch.cuda.synchronize()
            return C

        ref = run(3)
        for rep in range(6):
            got = run(22)
            n += 1
            if torch.equal(ref, got):
                continue
            lib.gram_debug_set_gemm_variant(-1)
            d = (got.double() - ref.double())
            bad = d != 0
            print(f"MISMATCH M={M} N={N} K={K} (physical {2 * K}, {2 * K // 64} k-tiles) rep={rep}: {int(bad.sum())} elements, max {float(d.abs().max()):.4g}")
            mt_n, nt_n = (M + 255) // 256, N // 256
            props = torch.cuda.get_device_properties(0)
            Gd = min(props.multi_processor_count, mt_n * nt_n)
            print(f"tiles {mt_n} x {nt_n} = {mt_n * nt_n}, workgroups <= {Gd} (persistent: tile, tile + G, ...)")
            # wave blocks: 128 rows x 64 cols
            rb, cb = (M + 127) // 128, N // 64
            pad = torch.zeros(rb * 128, N, dtype=torch.bool, device=G.DEV)
            pad[:M] = bad
            blk = pad.view(rb, 128, cb, 64).permute(0, 2, 1, 3).reshape(rb, cb, -1).sum(-1)
            wrong = torch.nonzero(blk)
            print(f"{wrong.shape[0]} wave blocks (128 x 64) hold wrong elements; wrong elements per block: min {int(blk[blk > 0].min())} max {int(blk.max())} of 8192")
            a0, a1 = pieces(A, K)
            w0, w1 = pieces(W, K)
            nk = K // 32  # physical k-tiles of 64 columns = 32 logical columns
            seen_tiles = {}
            for r, c in wrong.tolist()[:12]:
                mt, nt, wr, wc = r // 2, c // 4, r % 2, c % 4
                rows = slice(r * 128, min(r * 128 + 128, M))
                cols = slice(c * 64, c * 64 + 64)
                dd = d[rows, cols]
                nz = dd != 0
                rws = torch.nonzero(nz.any(1)).flatten()
                cls = torch.nonzero(nz.any(0)).flatten()
                line = (f"  tile (m {mt}, n {nt}) wave block wr={wr} wc={wc}: rows {int(rws.min())}..{int(rws.max())} cols {int(cls.min())}..{int(cls.max())} of the block, "
                        f"|d| mean {float(dd.abs().mean()):.3g}")
                # hypotheses: one k-tile b computed with a stale A half (the buffer's previous content: k-tile b-2 of the same rows) or a
                # stale W half, or missing, or doubled
                best = []
                Ar0, Ar1 = a0[rows], a1[rows]
                Wc0, Wc1 = w0[cols], w1[cols]
                for b in range(nk):
                    ks = slice(b * 32, b * 32 + 32)
                    P = contrib(Ar0, Ar1, Wc0, Wc1, ks)
                    cands = {"missing": -P, "doubled": P}
                    for back in (1, 2):
                        if b - back >= 0:
                            ko = slice((b - back) * 32, (b - back) * 32 + 32)
                            cands[f"A of k-tile {b - back}"] = (Ar0[:, ko] @ Wc1[:, ks].T + Ar1[:, ko] @ Wc0[:, ks].T + Ar0[:, ko] @ Wc0[:, ks].T) - P
                            cands[f"W of k-tile {b - back}"] = (Ar0[:, ks] @ Wc1[:, ko].T + Ar1[:, ks] @ Wc0[:, ko].T + Ar0[:, ks] @ Wc0[:, ko].T) - P
                    for name, cand in cands.items():
                        res = float((dd - cand).norm() / dd.norm())
                        best.append((res, b, name))
                best.sort()
                line += "; best fits: " + ", ".join(f"k-tile {b} {name} (residual {res:.3f})" for res, b, name in best[:3])
                print(line)
                seen_tiles.setdefault((mt, nt), []).append((wr, wc))
            tiles = sorted({(r // 2, c // 4) for r, c in wrong.tolist()})
            print("wrong tiles (m, n) -> linear index n-fastest:", [(t, t[0] * nt_n + t[1]) for t in tiles][:24])
            groups = sorted({r % 2 for r, c in wrong.tolist()})
            print("wave groups (wr) with wrong blocks:", groups)
            return 1
        del A, W, base, ref
    lib.gram_debug_set_gemm_variant(-1)
    print(f"no mismatch in {n} runs ({time.time() - t0:.0f} s)")
    return 0


if __name__ == "__main__":
    sys.exit(main())
