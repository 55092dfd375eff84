"""GPU micro-benchmark (not a test): the cross-attention kernel at the bench shape per (waves per workgroup, ring depth)
variant.  Needs a library built with -DGRAM_XA_AB=1; each variant runs in its own process (the selector is read once).
    python tests/bench_xattn.py            # all variants, both piece counts
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def one(pieces, B=4096, H=12, S=384, K=20, reps=20, use_bits=True, layers=1):
    import torch
    from gram_amd import _lib
    DT = _lib.piece_dtype()
    lib = _lib.load()
    dev = "cuda:0"
    inner = H * 64
    q = torch.randn(pieces, B * K, inner, device=dev).to(DT)
    # layers > 1: every launch reads another layer's slice of a [pieces][layers] bank, as a decode step does (the whole bank is
    # 12 x 9.7 GB at the bench shape: no launch finds its pages where the previous one left the TLBs)
    kbs = torch.randn(pieces, layers, B, H, S, 64, device=dev, dtype=DT)
    vts = torch.randn(pieces, layers, B, H, S // 32, 64, 32, device=dev, dtype=DT)
    kb, vt = kbs[:, 0], vts[:, 0]
    pstride = kbs[0].numel()
    lstride = kbs[0, 0].numel() * 2  # bytes
    it = [0]
    mask = torch.ones(B, S, dtype=torch.uint8, device=dev)
    out = torch.empty(B * K, pieces * inner, dtype=DT, device=dev)  # (two pieces: interleaved rows)
    st = torch.cuda.current_stream().cuda_stream
    bits = torch.zeros(B, 128, dtype=torch.int32, device=dev)
    _lib.check(lib.gram_mask_key_bits(mask.data_ptr(), bits.data_ptr(), B, S, st), "bits")

    def run():
        ly = it[0] % layers
        it[0] += 1
        _lib.check(lib.gram_cross_attn_decode_split(q.data_ptr(), kbs.data_ptr() + ly * lstride, vts.data_ptr() + ly * lstride, mask.data_ptr(), out.data_ptr(), B, K, H, S,
                                                    None, None, pieces, q[0].numel(), pstride, bits.data_ptr() if use_bits else None, st), "xattn")
    for _ in range(3):
        run()
    if os.environ.get("XA_HEAT"):
        # every timed launch right behind a large GEMM, as in a decode step (does the kernel see the clocks / power state the GEMMs
        # leave behind?): per-launch events, the GEMMs are not in the sum
        M, N_, K_ = 81920, 2304, 768
        ga = torch.randn(M, K_, device=dev).to(DT)
        gw = torch.randn(N_, K_, device=dev).to(DT)
        gc = torch.empty(M, N_, dtype=DT, device=dev)
        ev = []
        for _ in range(reps):
            for _ in range(int(os.environ["XA_HEAT"])):
                _lib.check(lib.gram_gemm_bf16(ga.data_ptr(), gw.data_ptr(), gc.data_ptr(), M, N_, K_, K_, N_, _lib.EPI_BF16, None, st), "gemm")
            a, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            run()
            b_.record()
            ev.append((a, b_))
        torch.cuda.synchronize()
        us = sum(a.elapsed_time(b_) for a, b_ in ev) * 1e3 / reps
        return us, 4.0 * B * H * S * 64 * pieces / (us * 1e-6) / 1e9
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    return us, 4.0 * B * H * S * 64 * pieces / (us * 1e-6) / 1e9


if __name__ == "__main__":
    if len(sys.argv) > 2:  # pieces B H S K: one shape, e.g. config 5's  `2 512 16 2688 50`
        pieces, B, H, S, K = (int(x) for x in sys.argv[1:6])
        layers = int(sys.argv[6]) if len(sys.argv) > 6 else 1
        us, gbs = one(pieces, B=B, H=H, S=S, K=K, layers=layers, reps=max(20, 2 * layers))
        print(json.dumps({"pieces": pieces, "B": B, "H": H, "S": S, "K": K, "layers": layers, "us": round(us, 1), "GBps": round(gbs, 1)}))
    elif len(sys.argv) > 1:
        for ub in (True, True):
            us, gbs = one(int(sys.argv[1]), use_bits=ub)
            print(json.dumps({"variant": os.environ.get("GRAM_XA_VARIANT", "default"), "pieces": int(sys.argv[1]), "key_bits": ub, "us": round(us, 1), "GBps": round(gbs, 1)}))
    else:
        for pieces, variants in ((1, ["", "21", "22", "12", "13", "41", "42"]), (2, ["", "21", "22", "11", "12", "41"])):
            for v in variants:
                env = dict(os.environ)
                if v:
                    env["GRAM_XA_VARIANT"] = v
                else:
                    env.pop("GRAM_XA_VARIANT", None)
                p = subprocess.run([sys.executable, os.path.abspath(__file__), str(pieces)], env=env, capture_output=True, text=True)
                print(p.stdout.strip() or p.stderr[-300:], flush=True)
