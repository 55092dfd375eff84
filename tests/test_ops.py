"""The ``gram::`` PyTorch custom ops (gram_amd/ops.py): registered with schemas and meta implementations on any host; their
kernels exist for the ROCm device only -- a CPU tensor must fail loudly (no CPU path).  -m gpu: values through torch.ops."""
import pytest
import torch

import gram_amd  # noqa: F401
from gram_amd import _lib, ops  # noqa: F401

DT = _lib.piece_dtype()  # the loaded library's 16-bit operand type (float16; bfloat16 for the PIECE=bf16 build)


def test_ops_are_registered_with_fake_impls():
    for name in ("generate", "linear", "enc_self_attn", "cross_attn_decode", "trie_step"):
        assert hasattr(torch.ops.gram, name), name
    from torch._subclasses.fake_tensor import FakeTensorMode
    with FakeTensorMode():
        a = torch.empty(20, 768, dtype=DT, device="cuda")
        w = torch.empty(2304, 768, dtype=DT, device="cuda")
        assert torch.ops.gram.linear(a, w).shape == (20, 2304)
        q = torch.empty(40, 768, dtype=DT, device="cuda")
        kb = torch.empty(2, 12, 384, 64, dtype=DT, device="cuda")
        vt = torch.empty(2, 12, 384 // 32, 64, 32, dtype=DT, device="cuda")
        mk = torch.empty(2, 384, dtype=torch.uint8, device="cuda")
        assert torch.ops.gram.cross_attn_decode(q, kb, vt, mk, 20).shape == (40, 768)
        ids = torch.empty(3, 2, 32, dtype=torch.int64, device="cuda")
        ws = torch.empty(1024, dtype=torch.uint8, device="cuda")
        t = torch.empty(8, dtype=torch.int32, device="cuda")
        s, sc, wd = torch.ops.gram.generate(ids, ids.to(torch.uint8), 0, ws, t, t, t, 4, 3, 5, 5, 7, 1.0, None, None, None, None, None, 0, 0)
        assert s.shape == (15, 7) and sc.shape == (15,) and wd.shape == (1,)


def test_no_cpu_kernels():
    a = torch.zeros(16, 64, dtype=DT)
    w = torch.zeros(128, 64, dtype=DT)
    with pytest.raises((NotImplementedError, RuntimeError)):
        torch.ops.gram.linear(a, w)


@pytest.mark.gpu
def test_ops_values_on_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from oracle import gram_oracle as O
    g = torch.Generator().manual_seed(4)
    a = torch.randn(77, 256, generator=g).to(DT).cuda()
    w = (torch.randn(384, 256, generator=g) / 16).to(DT).cuda()
    ref = a.float() @ w.float().T
    assert torch.allclose(torch.ops.gram.linear(a, w).float(), ref, atol=2e-2, rtol=1e-2)
    assert torch.allclose(torch.ops.gram.linear(a, w, True).float(), ref.clamp(min=0), atol=2e-2, rtol=1e-2)
    B, H, K, S = 2, 2, 5, 96
    q = (torch.randn(B * K, H * 64, generator=g) * 0.3).to(DT).cuda()
    kb = torch.randn(B, H, S, 64, generator=g).to(DT).cuda()
    v_t = torch.randn(B, H, 64, S, generator=g).to(DT)
    vt = v_t.unflatten(-1, (S // 32, 32)).transpose(-3, -2).contiguous().cuda()  # the bank's V^T, blocked by 32 keys
    mask = torch.rand(B, S, generator=g) > 0.3
    out = torch.ops.gram.cross_attn_decode(q, kb, vt, mask.cuda().view(torch.uint8).contiguous(), K)
    qh = q.float().cpu().view(B, K, H, 64).permute(0, 2, 1, 3)
    ext = ((1.0 - mask.float()) * O.FMIN)[:, None, None, :]
    ref = O._attend(qh, kb.float().cpu(), v_t.float().transpose(2, 3), ext).reshape(B * K, H * 64)
    assert torch.allclose(out.float().cpu(), ref, atol=1e-2, rtol=1e-2)


@pytest.mark.gpu
def test_ops_refuse_malformed_tensors():
    """The ops take raw data_ptr()s into the C ABI: a wrong dtype, a transposed or sliced view, a layout that is not the documented
    one or inconsistent shapes must raise instead of reading garbage or past the allocation."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    a = torch.zeros(32, 256, dtype=DT, device="cuda")
    w = torch.zeros(384, 256, dtype=DT, device="cuda")
    assert torch.ops.gram.linear(a, w).shape == (32, 384)
    assert torch.ops.gram.linear(torch.zeros(32, 512, dtype=DT, device="cuda")[:, :256], w).shape == (32, 384)  # row-strided A is fine
    for bad_a, bad_w in ((a.float(), w), (a, w.float()), (a.t().contiguous().t(), w), (a, torch.zeros(256, 384, dtype=DT, device="cuda").t()),
                         (a[:, ::2], w[:, ::2]), (a, w[:, :128]), (a[:, 1:65], w[:, :64]), (a, torch.zeros(100, 256, dtype=DT, device="cuda"))):
        with pytest.raises((ValueError, RuntimeError)):
            torch.ops.gram.linear(bad_a, bad_w)
    B, H, K, S = 2, 2, 5, 96
    q = torch.zeros(B * K, H * 64, dtype=DT, device="cuda")
    kb = torch.zeros(B, H, S, 64, dtype=DT, device="cuda")
    vt = torch.zeros(B, H, S // 32, 64, 32, dtype=DT, device="cuda")
    mk = torch.ones(B, S, dtype=torch.uint8, device="cuda")
    assert torch.ops.gram.cross_attn_decode(q, kb, vt, mk, K).shape == q.shape
    for args in ((q, kb, torch.zeros(B, H, 64, S, dtype=DT, device="cuda"), mk, K),      # plain [64][S] V^T, not blocked by 32 keys
                 (q[:-1], kb, vt, mk, K), (q, kb, vt, mk[:, :64].contiguous(), K), (q.float(), kb, vt, mk, K),
                 (q, kb.transpose(1, 2), vt, mk, K), (q, kb, vt, mk.bool(), K), (q, kb, vt, mk, 65)):
        with pytest.raises((ValueError, RuntimeError)):
            torch.ops.gram.cross_attn_decode(*args)
    qkv = torch.zeros(2 * 32, 3 * 128, dtype=DT, device="cuda")
    bias = torch.zeros(2, 255, device="cuda")
    m2 = torch.ones(2, 32, dtype=torch.uint8, device="cuda")
    assert torch.ops.gram.enc_self_attn(qkv, bias, m2, 2).shape == (64, 128)
    for args in ((qkv[:, :256], bias, m2, 2), (qkv, bias[:, :200].contiguous(), m2, 2), (qkv, bias, torch.ones(2, 48, dtype=torch.uint8, device="cuda"), 2)):
        with pytest.raises((ValueError, RuntimeError)):
            torch.ops.gram.enc_self_attn(*args)
