"""-m gpu: the two-piece precision mode ("f16x3": gram_split_t in include/gram_hip.h) -- every kernel through the C ABI against an
fp64 restatement of the same op on the UNROUNDED fp32 inputs, and the whole path against the fp32 oracle.  Tolerances are the
mode's design error: a product of two values carried as two IEEE-half pieces is exact to ~2^-22 relative (two bf16 pieces, the
PIECE=bf16 build: ~2^-18), accumulation is fp32 as in the reference.

Layouts under test (gram_hip.h): GEMM operands and everything a GEMM reads are INTERLEAVED ([rows][cols/32][2][32]); Q/K/V rows,
the KV bank and the self-attention cache are planar pieces."""
import ctypes as C

import pytest
import torch

from oracle import gram_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from tests import gpu_util
    return gpu_util


def tol(G):
    """max |error| / rms(reference) allowed for a two-piece GEMM-like result; observed values are printed"""
    return 2e-5 if G.F16 else 1.5e-4


def _r(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def relerr(a, ref):
    return float((a.double() - ref.double()).abs().max() / ref.double().pow(2).mean().sqrt())


def split_gemm(G, A, W, C_out, M, N, K, epi, inter_c=False, c_ps=0, nf=None, bank=None, bank_ps=0, out_scale=0.0):
    """gram_gemm_bf16_split on interleaved two-piece operands A [M][2K], W [N][2K]"""
    from gram_amd import _lib
    sp = _lib.Split(2, int(inter_c), c_ps, bank_ps, out_scale)
    ldc = 2 * N if inter_c else N
    return G.lib().gram_gemm_bf16_split(G.p(A), G.p(W), G.p(C_out), M, N, K, 2 * K, ldc, epi, None if bank is None else C.byref(bank),
                                        None if nf is None else C.byref(nf), C.byref(sp), G.stream())


@pytest.mark.parametrize("M,N,K", [(20, 256, 128), (300, 384, 768), (1500, 256, 3072)])
def test_split_gemm_epilogues(G, M, N, K):
    """16-bit out planar and interleaved, + ReLU (+ folded-norm row scale) / fp32 / fp32 residual add (+ interleaved 16-bit copy + sum
    of squares), with a power-of-two weight scale undone by out_scale: skinny, 64- and 128-row tiles."""
    from gram_amd import _lib
    a32, w32 = _r(M, K, seed=1).to(G.DEV), _r(N, K, seed=2, scale=K ** -0.5).to(G.DEV)
    wscale = 256.0
    A, W = G.inter(a32), G.inter(w32 * wscale)
    ref = a32.double() @ w32.double().T
    rs = (torch.rand(M, generator=torch.Generator().manual_seed(13)) + 0.5).to(G.DEV)
    for epi, act in ((_lib.EPI_BF16, lambda t: t), (_lib.EPI_BF16_RELU, lambda t: t.clamp(min=0))):
        cons = _lib.NormFusion(None, None, rs.data_ptr(), 0, K, 1e-6)
        y = torch.zeros(2, M, N, dtype=G.DT, device=G.DEV)
        _lib.check(split_gemm(G, A, W, y, M, N, K, epi, False, M * N, cons, out_scale=1 / wscale), "planar")
        e = relerr(G.join(y), act(ref * rs.double()[:, None]))
        assert e < tol(G), (epi, e)
        yi = torch.zeros(M, 2 * N, dtype=G.DT, device=G.DEV)
        _lib.check(split_gemm(G, A, W, yi, M, N, K, epi, True, 0, cons, out_scale=1 / wscale), "interleaved")
        assert torch.equal(_lib.deinterleave(yi), y), epi  # the same pieces, the other layout
    f = torch.zeros(M, N, dtype=torch.float32, device=G.DEV)
    _lib.check(split_gemm(G, A, W, f, M, N, K, _lib.EPI_F32, out_scale=1 / wscale), "f32")
    e32 = relerr(f, ref)
    base = _r(M, N, seed=14).to(G.DEV)
    x = base.clone()
    xb = torch.zeros(M, 2 * N, dtype=G.DT, device=G.DEV)
    ss = torch.full((M, N // 64), float("nan"), dtype=torch.float32, device=G.DEV)
    prod = _lib.NormFusion(xb.data_ptr(), ss.data_ptr(), None, 0, 0, 0.0)
    _lib.check(split_gemm(G, A, W, x, M, N, K, _lib.EPI_F32_ADD, nf=prod, out_scale=1 / wscale), "add")
    torch.cuda.synchronize()
    print(f"\n[split gemm] {M}x{N}x{K}: fp32-out max err / rms = {e32:.2e}")
    assert e32 < tol(G)
    assert relerr(x, base.double() + ref) < tol(G)
    assert relerr(G.join_inter(xb), x) < (2e-6 if G.F16 else 5e-5)  # the pieces reproduce the stored fp32 value
    assert torch.allclose(ss.sum(-1).double(), x.double().pow(2).sum(-1), rtol=1e-5)
    # a scale that is not a power of two is refused (it would not be exact wherever a kernel applies it)
    assert split_gemm(G, A, W, f, M, N, K, _lib.EPI_F32, out_scale=0.3) == _lib.E_ARG


def _host_row_xscale(ss):
    """gram_norm_fusion_t.xs_out restated on the host from the 64-column partials [M][nblk] (fp32, the device's own floats)."""
    s = torch.zeros(ss.shape[0], dtype=torch.float32, device=ss.device)
    for i in range(ss.shape[1] // 2):
        s = s + (ss[:, 2 * i] + ss[:, 2 * i + 1])
    want = 0.25 * torch.rsqrt(ss.min(dim=1).values / 64.0)
    cap = 1024.0 * torch.rsqrt(s)
    e = torch.floor(torch.log2(torch.minimum(want, cap).double())).clamp(-40, 20)
    return torch.pow(2.0, e).float()


@pytest.mark.parametrize("pieces", [2, 1])
@pytest.mark.parametrize("M", [20, 300, 1500, 33000])  # skinny / streaming, 64- and 128-row tiles, the ping-pong kernel
def test_row_factors_of_the_16_bit_residual_copy(G, M, pieces):
    """gram_norm_fusion_t.xs_in / xs_out: residual rows of magnitude 1e-2 .. 3e5 with outlier features (what a trained T5 stream looks
    like; IEEE half ends at 65 504) go producer -> consumer with a power-of-two factor per row on the 16-bit copy.  The consumer's
    output matches the fp64 norm + Linear at the two-piece precision for EVERY row, the published factors are the documented function
    of the partials, and without the factors the large rows are lost (f16 build)."""
    from gram_amd import _lib
    d, Kp, F = 768, 128, 256
    g = torch.Generator().manual_seed(M + pieces)
    mag = torch.pow(10.0, torch.rand(M, 1, generator=g) * 7.5 - 2.0)          # 1e-2 .. 3e5 per row
    base = torch.randn(M, d, generator=g) * mag
    base[:, [5, 300, 701]] *= 300.0                                          # outlier features (dominate the rms)
    base = base.clamp(-3.0e8, 3.0e8).to(G.DEV)
    a32 = (torch.randn(M, Kp, generator=g) * mag.clamp(max=1.0e3)).to(G.DEV)   # the sublayer's output (itself a 16-bit operand)
    wo32, w232 = _r(d, Kp, seed=3, scale=Kp ** -0.5).to(G.DEV), _r(F, d, seed=4, scale=d ** -0.5).to(G.DEV)
    enc = G.inter if pieces == 2 else G.bf
    A, Wo, W2 = enc(a32), enc(wo32), enc(w232)
    sp = _lib.Split(pieces, int(pieces == 2), 0, 0, 1.0)
    lda = pieces * Kp
    # xs0: the factor the PREVIOUS norm point would have published -- from the rows as they stand before the add
    xs0 = _host_row_xscale((base * base).view(M, d // 64, 64).sum(-1))
    outs = {}
    for use_xs in (True, False):
        x = base.clone()
        xb = torch.zeros(M, pieces * d, dtype=G.DT, device=G.DEV)
        ss = torch.full((M, d // 64), float("nan"), dtype=torch.float32, device=G.DEV)
        xs1 = torch.zeros(M, device=G.DEV)
        prod = _lib.NormFusion(xb.data_ptr(), ss.data_ptr(), None, 0, 0, 0.0, 0, xs0.data_ptr() if use_xs else None, None)
        _lib.check(G.lib().gram_gemm_bf16_split(G.p(A), G.p(Wo), G.p(x), M, d, Kp, lda, d, _lib.EPI_F32_ADD, None, C.byref(prod), C.byref(sp),
                                                G.stream()), "producer")
        y = torch.zeros(M, pieces * F, dtype=G.DT, device=G.DEV)
        if M >= 32768:  # the ping-pong kernels take 1/rms (already divided by the factor) per row
            rs = torch.zeros(M, device=G.DEV)
            _lib.check(G.lib().gram_row_rscale_xs(G.p(ss), G.p(rs), G.p(xs0) if use_xs else None, G.p(xs1), M, d // 64, d, 1e-6, G.stream()),
                       "row_rscale_xs")
            cons = _lib.NormFusion(None, None, rs.data_ptr(), 0, d, 1e-6)
        else:
            cons = _lib.NormFusion(None, None, ss.data_ptr(), d // 64, d, 1e-6, 0, xs0.data_ptr() if use_xs else None, xs1.data_ptr())
        _lib.check(G.lib().gram_gemm_bf16_split(G.p(xb), G.p(W2), G.p(y), M, F, d, pieces * d, pieces * F, _lib.EPI_BF16, None, C.byref(cons),
                                                C.byref(sp), G.stream()), "consumer")
        torch.cuda.synchronize()
        outs[use_xs] = (x, ss, xs1, G.join_inter(y) if pieces == 2 else y.double())
    x, ss, xs1, y = outs[True]
    xr = base.double() + a32.double() @ wo32.double().T
    ref = (xr * torch.rsqrt(xr.pow(2).mean(-1, keepdim=True) + 1e-6)) @ w232.double().T
    row_err = (y - ref).norm(dim=1) / ref.norm(dim=1)
    bound = (tol(G) if pieces == 2 else 2e-3 if G.F16 else 1.5e-2)
    print(f"\n[row factors] M={M} pieces={pieces}: worst row error {float(row_err.max()):.2e} (bound {bound:.1e})")
    assert bool(torch.isfinite(y).all()) and float(row_err.max()) < bound
    # the published factors: the documented function of the partials (v_rsq_f32 vs torch.rsqrt may differ in the last bit: at most a
    # handful of rows sit on a power-of-two boundary)
    host = _host_row_xscale(ss)
    ratio = (xs1 / host).cpu()
    assert bool(((ratio == 1) | (ratio == 2) | (ratio == 0.5)).all()) and float((ratio == 1).float().mean()) > 0.99
    assert bool((torch.log2(xs1) == torch.log2(xs1).round()).all())
    if G.F16:  # without the factors the rows beyond the half range are lost; the ones inside it are the same values
        y0 = outs[False][3]
        big = (xr.abs().max(dim=1).values > 7.0e4)
        assert big.any() and not bool(torch.isfinite(y0[big]).all())


def test_split_gemm_kv_bank(G):
    from gram_amd import _lib
    B, S, H, layers, d = 3, 96, 2, 2, 256
    inner = H * 64
    M = B * S
    a32, w32 = _r(M, d, seed=41).to(G.DEV), _r(layers * 2 * inner, d, seed=42, scale=d ** -0.5).to(G.DEV)
    A, W = G.inter(a32), G.inter(w32)
    ref = (a32.double() @ w32.double().T).view(B, S, layers, 2, H, 64)
    k = torch.zeros(2, layers, B, H, S, 64, dtype=G.DT, device=G.DEV)
    vt = torch.zeros(2, layers, B, H, S // 32, 64, 32, dtype=G.DT, device=G.DEV)  # V^T blocked by 32 keys
    bank = _lib.KVBank(k.data_ptr(), vt.data_ptr(), layers, B, H, S)
    _lib.check(split_gemm(G, A, W, None, M, layers * 2 * inner, d, _lib.EPI_KV_BANK, bank=bank, bank_ps=k[0].numel()), "kv")
    torch.cuda.synchronize()
    kref = ref[:, :, :, 0].permute(2, 0, 3, 1, 4)   # (layers,B,H,S,64)
    vref = ref[:, :, :, 1].permute(2, 0, 3, 4, 1)   # (layers,B,H,64,S)
    assert relerr(G.join(k), kref) < tol(G)
    assert relerr(G.join(G.vt_unblocked(vt)), vref) < tol(G)


@pytest.mark.parametrize("L", [32, 128])
def test_split_enc_self_attn(G, L):
    from gram_amd import _lib
    from gram_amd.model.gram import relative_position_bucket
    P, H = 4, 2
    inner = H * 64
    g = torch.Generator().manual_seed(L)
    qkv32 = torch.randn(P * L, 3 * inner, generator=g).to(G.DEV)
    qkv = G.pieces_of(qkv32)
    table = torch.randn(32, H, generator=g) * 0.5
    bias = table[relative_position_bucket(torch.arange(-127, 128), True, 32, 128)].t().contiguous().to(G.DEV)
    mask = torch.zeros(P, L, dtype=torch.bool)
    for p_ in range(P):
        mask[p_, : int(torch.randint(1, L + 1, (1,), generator=g))] = True
    mask[0] = True
    out = torch.empty(P * L, 2 * inner, dtype=G.DT, device=G.DEV)  # interleaved: the O GEMM's A operand
    m8 = mask.to(G.DEV).view(torch.uint8).contiguous()
    _lib.check(G.lib().gram_enc_self_attn_split(G.p(qkv), G.p(bias), G.p(m8), G.p(out), P, L, H, 2, qkv[0].numel(), G.stream()), "enc_attn")
    x = qkv32.double().cpu().view(P, L, 3, H, 64)
    q, k, v = (x[:, :, i].permute(0, 2, 1, 3) for i in range(3))
    cfg = O.OracleConfig(num_heads=H)
    b = (O.position_bias(table, L, L, True, cfg) + ((1.0 - mask.float())[:, None, None, :] * O.FMIN)).double()
    sc = torch.matmul(q, k.transpose(3, 2)) + b
    ref = torch.matmul(torch.softmax(sc, -1), v).transpose(1, 2).reshape(P * L, inner)
    e = relerr(G.join_inter(out).cpu(), ref)
    print(f"\n[split enc attn] L={L}: {e:.2e}")
    assert e < 10 * tol(G)  # scores of magnitude ~8 go through exp: absolute score error times |softmax'|


@pytest.mark.parametrize("K,S", [(20, 384), (1, 32), (16, 96), (50, 640), (8, 160)])
def test_split_cross_attn(G, K, S):
    from gram_amd import _lib
    B, H = 3, 2
    inner = H * 64
    g = torch.Generator().manual_seed(K * 1000 + S)
    q32 = (torch.randn(B * K, inner, generator=g) * 0.3).to(G.DEV)
    k32 = torch.randn(B, H, S, 64, generator=g).to(G.DEV)
    v32 = torch.randn(B, H, S, 64, generator=g).to(G.DEV)
    q, kb, vt = G.pieces_of(q32), G.pieces_of(k32), G.pieces_of(G.vt_blocked(v32.transpose(2, 3).contiguous()))
    mask = torch.rand(B, S, generator=g) > 0.3
    mask[1, : S // 2] = False
    if S >= 64:
        mask[2, 32:] = False
    out = torch.empty(B * K, 2 * inner, dtype=G.DT, device=G.DEV)  # interleaved
    m8 = mask.to(G.DEV).view(torch.uint8).contiguous()
    _lib.check(G.lib().gram_cross_attn_decode_split(G.p(q), G.p(kb), G.p(vt), G.p(m8), G.p(out), B, K, H, S, None, None, 2,
                                                    q[0].numel(), kb[0].numel(), None, G.stream()), "xattn")
    # the same with the mask's bit view precomputed once (what gram_generate does): identical output
    bits = torch.full((B, 128), -1, dtype=torch.int32, device=G.DEV)
    _lib.check(G.lib().gram_mask_key_bits(G.p(m8), G.p(bits), B, S, G.stream()), "bits")
    out2 = torch.empty_like(out)
    _lib.check(G.lib().gram_cross_attn_decode_split(G.p(q), G.p(kb), G.p(vt), G.p(m8), G.p(out2), B, K, H, S, None, None, 2,
                                                    q[0].numel(), kb[0].numel(), G.p(bits), G.stream()), "xattn")
    assert torch.equal(out, out2)
    qh = q32.double().cpu().view(B, K, H, 64).permute(0, 2, 1, 3)
    ext = ((1.0 - mask.float()) * O.FMIN)[:, None, None, :].double()
    sc = torch.matmul(qh, k32.double().cpu().transpose(3, 2)) + ext
    ref = torch.matmul(torch.softmax(sc, -1), v32.double().cpu()).transpose(1, 2).reshape(B * K, inner)
    e = relerr(G.join_inter(out).cpu(), ref)
    print(f"\n[split cross attn] K={K} S={S}: {e:.2e}")
    assert e < 10 * tol(G)


@pytest.mark.parametrize("Tmax,steps", [(10, 5), (64, 64)])  # (GRAM_MAX_DEC_LEN positions: distances up to 63 through T5's log buckets)
def test_split_dec_self_attn(G, Tmax, steps):
    from gram_amd import _lib
    from gram_amd.model.gram import relative_position_bucket
    R, H = 12, 3
    inner = H * 64
    g = torch.Generator().manual_seed(21)
    table = torch.randn(32, H, generator=g) * 0.5
    bias = table[relative_position_bucket(-torch.arange(0, _lib.GRAM_MAX_DEC_LEN), False, 32, 128)].t().contiguous().to(G.DEV)
    kc = torch.zeros(2, Tmax, R, inner, dtype=G.DT, device=G.DEV)
    vc = torch.zeros_like(kc)
    anc = torch.arange(R, dtype=torch.int32).repeat(Tmax, 1).to(G.DEV)
    cfg = O.OracleConfig(num_heads=H)
    ks, vs = None, None
    for t in range(steps):
        qkv32 = (torch.randn(R, 3 * inner, generator=g) * 0.5).to(G.DEV)
        qkv = G.pieces_of(qkv32)
        out = torch.empty(R, 2 * inner, dtype=G.DT, device=G.DEV)  # interleaved
        _lib.check(G.lib().gram_dec_self_attn_split(G.p(qkv), G.p(kc), G.p(vc), G.p(anc), G.p(bias), G.p(out), R, R, None, H, t, Tmax,
                                                    2, qkv[0].numel(), kc[0].numel(), G.stream()), "dec_attn")
        x = G.join(qkv).cpu().view(R, 1, 3, H, 64)
        q, k, v = (x[:, :, i].permute(0, 2, 1, 3) for i in range(3))
        ks = k if ks is None else torch.cat([ks, k], 2)
        vs = v if vs is None else torch.cat([vs, v], 2)
        b = O.position_bias(table, t + 1, t + 1, False, cfg)[:, :, -1:, :].double()
        sc = torch.matmul(q, ks.transpose(3, 2)) + b
        ref = torch.matmul(torch.softmax(sc, -1), vs).transpose(1, 2).reshape(R, inner)
        assert relerr(G.join_inter(out).cpu(), ref) < 10 * tol(G), t
        parent = torch.randint(0, R, (R,), generator=g)
        ks, vs = ks.index_select(0, parent), vs.index_select(0, parent)
        a = anc.cpu()
        new = a.clone()
        new[: t, :] = a[: t, parent]
        new[t, :] = parent.int()
        anc.copy_(new)


def test_split_rowops_write_interleaved_pieces(G):
    """gram_embed_ex_split / gram_rmsnorm_bf16_split: the interleaved two-piece copy of an fp32 row reproduces it to ~2^-22."""
    from gram_amd import _lib
    rows, d, V = 37, 256, 50
    g = torch.Generator().manual_seed(5)
    table = torch.randn(V, d, generator=g).to(G.DEV)
    ids = torch.randint(0, V, (rows,), generator=g).to(G.DEV)
    x = torch.empty(rows, d, device=G.DEV)
    xb = torch.zeros(rows, 2 * d, dtype=G.DT, device=G.DEV)
    ss = torch.zeros(rows, d // 64, device=G.DEV)
    _lib.check(G.lib().gram_embed_ex_split(G.p(table), G.p(ids), 1, G.p(x), G.p(xb), G.p(ss), d // 64, rows, d, 2, G.stream()), "embed")
    assert torch.equal(x, table[ids])
    assert relerr(G.join_inter(xb), x) < (2e-6 if G.F16 else 5e-5)
    assert torch.equal(_lib.deinterleave(xb)[0], x.to(G.DT))  # piece 0 = the value rounded to 16 bits
    w = (torch.rand(d, generator=g) + 0.5).to(G.DEV)
    out = torch.zeros(rows, 2 * d, dtype=G.DT, device=G.DEV)
    _lib.check(G.lib().gram_rmsnorm_bf16_split(G.p(x), G.p(w), G.p(out), rows, d, 1e-6, 0.5, None, 1, 1, None, 2, G.stream()), "rmsnorm")
    ref = x.double() * torch.rsqrt(x.double().pow(2).mean(-1, keepdim=True) + 1e-6) * w.double() * 0.5
    assert relerr(G.join_inter(out), ref) < 2e-6


def _cfgs(name):
    if name == "tiny":
        oc = O.OracleConfig(vocab_size=256, d_model=128, d_kv=64, d_ff=256, num_layers=2, num_decoder_layers=2, num_heads=2,
                            max_item_num=5)
    elif name == "small":
        oc = O.OracleConfig.named("t5-small", max_item_num=4)
    else:
        oc = O.OracleConfig.named(name)
    from gram_amd import T5Config
    gc = T5Config(vocab_size=oc.vocab_size, d_model=oc.d_model, d_ff=oc.d_ff, num_layers=oc.num_layers,
                  num_decoder_layers=oc.num_decoder_layers, num_heads=oc.num_heads, max_item_num=oc.max_item_num)
    return oc, gc


@pytest.mark.parametrize("name,B,N,L,K", [("tiny", 6, 3, 32, 6), ("small", 3, 2, 64, 5)])
def test_split_generate_vs_oracle(G, name, B, N, L, K):
    """Whole path in the two-piece mode against the fp32 CPU oracle: identical sequences (unless two of the oracle's own scores
    are closer than the tolerance) and scores within 2e-5 (observed 1.9e-6: fp32's own rounding) -- one piece needs 1e-2 here."""
    import gram_amd
    from gram_amd.utils import generation_trie as gt
    tol_ = 2e-5 if G.F16 else 3e-4  # observed 1.9e-6
    oc, gc = _cfgs(name)
    sd = O.init_state_dict(oc, 11)
    m = gram_amd.create_model("gram", gc)
    m.load_state_dict(sd)
    m = m.to(G.DEV).eval()
    m.set_precision(m.default_precision())
    assert m._PIECES[m.precision] == 2
    g = torch.Generator().manual_seed(3)
    V = min(oc.vocab_size, 32100)
    ids = torch.randint(2, V, (B, N, L), generator=g)
    mask = torch.ones(B, N, L, dtype=torch.bool)
    for b in range(B):
        for n in range(N):
            ln = int(torch.randint(L // 3, L + 1, (1,), generator=g))
            mask[b, n, ln:] = False
            ids[b, n, ln - 1] = 1
            ids[b, n, ln:] = 0
    mask[B - 1, N - 1] = False
    ids[B - 1, N - 1] = 0
    cands = sorted({tuple([0] + torch.randint(2, min(V, 300), (int(torch.randint(2, 5, (1,), generator=g)),), generator=g).tolist() + [1])
                    for _ in range(200)})
    cands = [list(c) for c in cands]
    max_length = max(len(c) for c in cands)
    ref = O.generate(sd, oc, ids, mask, max_length, O.prefix_allowed_tokens_fn(O.Trie(cands)), K, K, 1.0)
    out = m.generate(input_ids=ids.to(G.DEV), attention_mask=mask.to(G.DEV), max_length=max_length,
                     prefix_allowed_tokens_fn=gt.prefix_allowed_tokens_fn(gt.Trie(cands)), num_beams=K, num_return_sequences=K)
    rs, ds = ref["sequences_scores"], out["sequences_scores"].cpu()
    rq, dq = ref["sequences"], out["sequences"].cpu()
    dev = float((rs - ds).abs().max())
    print(f"\n[split generate] {m.precision} {name}: max |score diff| = {dev:.2e}; sequences equal: {bool(rq.shape == dq.shape and torch.equal(rq, dq))}")
    if rq.shape == dq.shape and torch.equal(rq, dq):
        assert dev < tol_
    else:  # a reordering is only acceptable between oracle scores closer than the tolerance
        gaps = (rs.view(B, K)[:, :-1] - rs.view(B, K)[:, 1:]).abs()
        assert float(gaps.min()) < tol_, "sequences differ although no two oracle scores are within the tolerance"
        assert float((rs.sort().values - ds.sort().values).abs().max()) < tol_


@pytest.mark.parametrize("M,N,K,variants", [(256 * 40 + 100, 768, 256, (3, 22)), (520, 512, 512, (3, 22)), (256 * 9, 2304, 768, (3, 22)),
                                            (1280, 768, 768, (3, 33, 34, 1, 31)), (333, 512, 1024, (3, 33, 34, 1, 31))])
def test_split_gemm_persistent_matches_tiles(G, M, N, K, variants):
    """The persistent ping-pong kernel (22), the deep-ring tiles (33, 34) and the 128- / 64-row tiles (1, 31) on two-piece operands
    against the 256x128-tile kernel (3): the same MFMA sequence per accumulator (a0*w1, a1*w0, a0*w0 per 32-column block, blocks in
    order), so every output -- 16-bit pieces planar and interleaved, fp32, residual + its interleaved copy + sums of squares, LSE
    partials -- is compared bit for bit; several tiles per workgroup and an M tail."""
    from gram_amd import _lib
    L_ = G.lib()
    a32, w32 = _r(M, K, seed=11).to(G.DEV), _r(N, K, seed=12, scale=K ** -0.5).to(G.DEV)
    A, W = G.inter(a32), G.inter(w32)
    rs = (torch.rand(M, generator=torch.Generator().manual_seed(13)) + 0.5).to(G.DEV)
    base = _r(M, N, seed=14).to(G.DEV)
    outs = {}
    try:
        for v in variants:
            L_.gram_debug_set_gemm_variant(v)
            o = {}
            for epi in (_lib.EPI_BF16, _lib.EPI_BF16_RELU):
                cons = _lib.NormFusion(None, None, rs.data_ptr(), 0, K, 1e-6)
                y = torch.zeros(2, M, N, dtype=G.DT, device=G.DEV)
                _lib.check(split_gemm(G, A, W, y, M, N, K, epi, False, M * N, cons), "planar")
                o[("planar", epi)] = y
                yi = torch.zeros(M, 2 * N, dtype=G.DT, device=G.DEV)
                _lib.check(split_gemm(G, A, W, yi, M, N, K, epi, True, 0, cons), "interleaved")
                o[("inter", epi)] = yi
            f = torch.zeros(M, N, dtype=torch.float32, device=G.DEV)
            _lib.check(split_gemm(G, A, W, f, M, N, K, _lib.EPI_F32), "f32")
            o["f32"] = f
            x = base.clone()
            xb = torch.zeros(M, 2 * N, dtype=G.DT, device=G.DEV)
            ss = torch.full((M, N // 64), float("nan"), dtype=torch.float32, device=G.DEV)
            prod = _lib.NormFusion(xb.data_ptr(), ss.data_ptr(), None, 0, 0, 0.0)
            _lib.check(split_gemm(G, A, W, x, M, N, K, _lib.EPI_F32_ADD, nf=prod), "add")
            o["add"], o["xb"], o["ss"] = x, xb, ss
            part = torch.full((M, N // 64, 2), float("nan"), dtype=torch.float32, device=G.DEV)
            sp = _lib.Split(2, 0, 0, 0, 0.0)
            _lib.check(L_.gram_gemm_bf16_lse_split(G.p(A), G.p(W), None, G.p(part), M, N, K, 2 * K, N, C.byref(sp), G.stream()), "lse")
            o["lse"] = part
            torch.cuda.synchronize()
            outs[v] = o
    finally:
        L_.gram_debug_set_gemm_variant(-1)
    ref = a32.double() @ w32.double().T
    assert relerr(outs[variants[0]]["f32"], ref) < tol(G)
    assert torch.equal(_lib.deinterleave(outs[3][("inter", _lib.EPI_BF16)]), outs[3][("planar", _lib.EPI_BF16)])
    for v in variants[1:]:
        for key in outs[3]:
            assert torch.equal(outs[3][key], outs[v][key]), (v, key)


@pytest.mark.parametrize("compact", [False, True])
def test_split_gemm_kv_bank_persistent(G, compact):
    from gram_amd import _lib
    L_ = G.lib()
    layers, H, d, pN, pL = 2, 4, 256, 3, 64
    inner, S = H * 64, pN * pL
    if compact:
        B = 30
        keep = torch.rand(B * pN, generator=torch.Generator().manual_seed(5)) < 0.85
        keep[0] = True
        pmap = torch.nonzero(keep).flatten().to(torch.int32)
        P = int(pmap.numel())
        if (P * pL) % 256 == 0:
            pmap, P = pmap[:-1], P - 1
        M = P * pL
    else:
        B = 33
        M, pmap = B * S, None
    a32, w32 = _r(M, d, seed=41).to(G.DEV), _r(layers * 2 * inner, d, seed=42, scale=d ** -0.5).to(G.DEV)
    A, W = G.inter(a32), G.inter(w32 * 64.0)
    res = {}
    try:
        for v in (3, 22):
            L_.gram_debug_set_gemm_variant(v)
            k = torch.zeros(2, layers, B, H, S, 64, dtype=G.DT, device=G.DEV)
            vt = torch.zeros(2, layers, B, H, S // 32, 64, 32, dtype=G.DT, device=G.DEV)
            pm = pmap.to(G.DEV) if compact else None
            bank = _lib.KVBank(k.data_ptr(), vt.data_ptr(), layers, B, H, S, pm.data_ptr() if compact else None, pN, pL)
            _lib.check(split_gemm(G, A, W, None, M, layers * 2 * inner, d, _lib.EPI_KV_BANK, bank=bank, bank_ps=k[0].numel(), out_scale=1 / 64.0), "kv")
            torch.cuda.synchronize()
            res[v] = (k, vt)
    finally:
        L_.gram_debug_set_gemm_variant(-1)
    assert res[22][0].abs().sum() > 0
    assert torch.equal(res[3][0], res[22][0]) and torch.equal(res[3][1], res[22][1])
    if not compact:
        ref = (a32.double() @ w32.double().T).view(B, S, layers, 2, H, 64)
        assert relerr(G.join(res[22][0]), ref[:, :, :, 0].permute(2, 0, 3, 1, 4)) < tol(G)


@pytest.mark.parametrize("pieces", [1, 2])
@pytest.mark.parametrize("M", [1, 16, 20, 33, 64, 100, 384])
@pytest.mark.parametrize("N,K", [(768, 768), (2304, 768), (768, 3072), (1408, 512), (256, 128)])
def test_gemm_stream_matches_tiled(G, pieces, M, N, K):
    """The small-M streaming kernel (variant 32: one 16-column n-tile per workgroup, W and A through an LDS ring filled by
    LDS-DMA) and the skinny kernel (30, M <= 64) against the tiled 128 x 128 kernel (variant 1), bit for bit, one- and two-piece
    operands: 16-bit / +ReLU with the folded-norm row scale from 64-column AND from 16-column ("quarter") partials, the fp32
    residual add, and the producer outputs -- the residual, its 16-bit copy (interleaved pieces), and quarter partials whose
    (q0 + q1) + (q2 + q3) are the tiled kernel's 64-column partials."""
    from gram_amd import _lib
    L_ = G.lib()
    a32, w32 = _r(M, K, seed=31).to(G.DEV), _r(N, K, seed=32, scale=K ** -0.5).to(G.DEV)
    if pieces == 2:
        A, W = G.inter(a32), G.inter(w32)
    else:
        A, W = G.bf(a32), G.bf(w32)
    sp = _lib.Split(pieces, 0, M * N, 0, 0.0)
    lda = pieces * K
    q_in = (torch.rand(M, K // 16, generator=torch.Generator().manual_seed(33)) * 16 + 0.25).to(G.DEV)
    q4 = q_in.view(M, K // 64, 4)
    ss_in = ((q4[..., 0] + q4[..., 1]) + (q4[..., 2] + q4[..., 3])).contiguous()  # fp32, the order the epilogues use
    base = _r(M, N, seed=34).to(G.DEV)
    outs = {}

    def gemm(C_out, epi, nf=None):
        return L_.gram_gemm_bf16_split(G.p(A), G.p(W), G.p(C_out), M, N, K, lda, N, epi, None, None if nf is None else C.byref(nf),
                                       C.byref(sp), G.stream())
    variants = (1, 32, 30) if M <= 64 else (1, 32)
    try:
        for v in variants:
            L_.gram_debug_set_gemm_variant(v)
            o = {}
            for epi in (_lib.EPI_BF16, _lib.EPI_BF16_RELU):
                y = torch.zeros(pieces, M, N, dtype=G.DT, device=G.DEV)
                _lib.check(gemm(y, epi), "plain")
                o[("plain", epi)] = y
                y2 = torch.zeros(pieces, M, N, dtype=G.DT, device=G.DEV)
                _lib.check(gemm(y2, epi, _lib.NormFusion(None, None, ss_in.data_ptr(), K // 64, K, 1e-6)), "consumer")
                o[("scaled", epi)] = y2
                if v == 32:
                    y3 = torch.zeros(pieces, M, N, dtype=G.DT, device=G.DEV)
                    _lib.check(gemm(y3, epi, _lib.NormFusion(None, None, q_in.data_ptr(), K // 64, K, 1e-6, 1)), "quarter consumer")
                    o[("quarter", epi)] = y3
            x0 = base.clone()
            _lib.check(gemm(x0, _lib.EPI_F32_ADD), "add")
            o["add_plain"] = x0
            x = base.clone()
            xb = torch.zeros(M, pieces * N, dtype=G.DT, device=G.DEV)
            quarter = v == 32
            ss = torch.full((M, N // (16 if quarter else 64)), float("nan"), dtype=torch.float32, device=G.DEV)
            _lib.check(gemm(x, _lib.EPI_F32_ADD, _lib.NormFusion(xb.data_ptr(), ss.data_ptr(), None, 0, 0, 0.0, int(quarter))), "producer")
            if quarter:
                s4 = ss.view(M, N // 64, 4)
                ss = (s4[..., 0] + s4[..., 1]) + (s4[..., 2] + s4[..., 3])
            o["add"], o["xb"], o["ss"] = x, xb, ss
            torch.cuda.synchronize()
            outs[v] = o
    finally:
        L_.gram_debug_set_gemm_variant(-1)
    ref = a32.double() @ w32.double().T
    assert relerr(outs[32]["add_plain"], base.double() + ref) < (tol(G) if pieces == 2 else 2e-2)
    for v in variants[1:]:
        for key in outs[1]:
            assert torch.equal(outs[1][key], outs[v][key]), (v, key)
    for epi in (_lib.EPI_BF16, _lib.EPI_BF16_RELU):
        assert torch.equal(outs[32][("quarter", epi)], outs[1][("scaled", epi)]), ("quarter", epi)
    # not on the streaming kernel: a quarter layout is refused, not misread
    L_.gram_debug_set_gemm_variant(1)
    try:
        y = torch.zeros(pieces, M, N, dtype=G.DT, device=G.DEV)
        assert gemm(y, _lib.EPI_BF16, _lib.NormFusion(None, None, q_in.data_ptr(), K // 64, K, 1e-6, 1)) == _lib.E_ARG
    finally:
        L_.gram_debug_set_gemm_variant(-1)


def test_pp_clock_counters(G):
    """gram_prof_pp_clock: the ping-pong GEMM's workgroups add their s_memtime / s_memrealtime differences to two device counters; the
    quotient is the shader clock in GHz (bench.py prices the MFMA peak at it).  A diagnostic: nothing is stamped unless it is switched on
    (gram_prof_pp_clock_enable); zero after a reset, a plausible clock after a launch, the same output bits either way."""
    from gram_amd import _lib
    L_ = G.lib()
    ghz = C.c_double(-1.0)
    _lib.check(L_.gram_prof_pp_clock(C.byref(ghz), 1), "clock")
    _lib.check(L_.gram_prof_pp_clock(C.byref(ghz), 0), "clock")
    assert ghz.value == 0.0
    M, N, K = 256 * 160, 512, 256
    a32, w32 = _r(M, K, seed=1).to(G.DEV), _r(N, K, seed=2, scale=K ** -0.5).to(G.DEV)
    outs = []
    try:
        L_.gram_debug_set_gemm_variant(22)
        for on in (0, 1):
            L_.gram_prof_pp_clock_enable(on)
            y = torch.zeros(M, 2 * N, dtype=G.DT, device=G.DEV)
            _lib.check(split_gemm(G, G.inter(a32), G.inter(w32), y, M, N, K, _lib.EPI_BF16, True, 0), "pp gemm")
            _lib.check(L_.gram_prof_pp_clock(C.byref(ghz), 1), "clock")
            outs.append((y, ghz.value))
    finally:
        L_.gram_debug_set_gemm_variant(-1)
        L_.gram_prof_pp_clock_enable(0)
    print(f"\n[pp clock] {outs[1][1]:.3f} GHz")
    assert outs[0][1] == 0.0 and 0.4 < outs[1][1] < 2.6, outs
    assert torch.equal(outs[0][0], outs[1][0])
