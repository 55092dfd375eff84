"""Helpers for the -m gpu parity tests: call libgram_hip.so through its C ABI on torch tensors."""
import ctypes as C

import numpy as np
import torch

from gram_amd import _lib

DEV = "cuda:0"
DT = _lib.piece_dtype()  # the library's 16-bit operand type: float16 (bfloat16 for the PIECE=bf16 build)
F16 = DT == torch.float16


def lib():
    return _lib.load()


def stream():
    return torch.cuda.current_stream().cuda_stream


def p(t):
    return None if t is None else t.data_ptr()


def bf(t):
    """fp32 -> the library's 16-bit type, on the device"""
    return t.to(DEV, torch.float32).to(DT).contiguous()


def pieces_of(x, s=2):
    """[s][...] 16-bit pieces of an fp32 tensor (device): p_i = r16(x - p_0 - .. - p_{i-1})."""
    r, out = x.float().clone(), []
    for _ in range(s):
        out.append(r.to(DT))
        r = r - out[-1].float()
    return torch.stack(out).contiguous()


def inter(x):
    """fp32 [rows][cols] -> the interleaved two-piece operand [rows][2 * cols] a GEMM reads (gram_hip.h)"""
    return _lib.interleave(pieces_of(x, 2))


def join(p):
    """[s][...] planar pieces -> fp64 value"""
    return p.double().sum(0)


def join_inter(x):
    """interleaved [rows][2 * cols] -> fp64 [rows][cols]"""
    return _lib.deinterleave(x).double().sum(0)


def gemm(A, W, epi, C_out=None, bank=None):
    M, K = A.shape
    N = W.shape[0]
    rc = lib().gram_gemm_bf16(p(A), p(W), p(C_out), M, N, K, A.stride(0), 0 if C_out is None else C_out.stride(0), epi,
                              None if bank is None else C.byref(bank), stream())
    _lib.check(rc, "gram_gemm_bf16")
    return C_out


def make_beam_state(B, K, Tmax, lp=1.0, device=DEV, cand_scratch=False):
    """cand_scratch: also give the state the scratch of the small-batch sparse-logits kernel (gram_beam_state_t.cand_logits)"""
    R = B * K
    t = dict(
        tokens=torch.zeros(R, dtype=torch.int32, device=device), node=torch.zeros(R, dtype=torch.int32, device=device),
        beam_scores=torch.zeros(R, dtype=torch.float32, device=device), seq=torch.zeros(R, Tmax, dtype=torch.int32, device=device),
        anc=torch.zeros(Tmax, R, dtype=torch.int32, device=device), done=torch.zeros(B, dtype=torch.int32, device=device),
        n_hyps=torch.zeros(B, dtype=torch.int32, device=device), hyp_score=torch.zeros(B, K + 1, dtype=torch.float64, device=device),
        worst=torch.zeros(B, dtype=torch.float64, device=device), hyp_len=torch.zeros(B, K + 1, dtype=torch.int32, device=device),
        hyp_tok=torch.zeros(B, K + 1, Tmax, dtype=torch.int32, device=device), error=torch.zeros(4, dtype=torch.int32, device=device),
    )
    extra = {}
    if cand_scratch:
        t["cand_logits"] = torch.full((B, 16384), float("nan"), dtype=torch.float32, device=device)
        extra = dict(cand_logits_users=B, cand_logits_stride=16384)
    st = _lib.BeamState(B=B, K=K, Tmax=Tmax, length_penalty=lp, eos=1, pad=0, **{k: v.data_ptr() for k, v in t.items()}, **extra)
    return st, t


def device_beam_search(logits_per_step, flat_trie, B, K, max_length, lp=1.0, nret=None):
    """Run gram_beam_init/step/finalize on pre-computed logits [T][R][V] (fp32, device)."""
    nret = nret or K
    V = logits_per_step[0].shape[-1]
    st, keep = make_beam_state(B, K, max_length, lp)
    ctrie, keep2 = flat_trie.to_device(torch.device(DEV))
    L = lib()
    _lib.check(L.gram_beam_init(C.byref(st), C.byref(ctrie), 0, stream()), "beam_init")
    lse = torch.empty(B * K, dtype=torch.float32, device=DEV)
    trace = []
    for t in range(max_length - 1):
        lg = logits_per_step[t].contiguous()
        _lib.check(L.gram_row_lse(p(lg), p(lse), B * K, V, stream()), "row_lse")
        _lib.check(L.gram_beam_step(C.byref(st), C.byref(ctrie), p(lg), p(lse), V, t + 1, K, stream()), "beam_step")
        trace.append(dict(tokens=keep["tokens"].clone(), scores=keep["beam_scores"].clone(), seq=keep["seq"].clone(),
                          anc=keep["anc"].clone(), done=keep["done"].clone()))
    seqs = torch.empty(B * nret, max_length, dtype=torch.int64, device=DEV)
    scores = torch.empty(B * nret, dtype=torch.float32, device=DEV)
    width = torch.zeros(4, dtype=torch.int32, device=DEV)
    _lib.check(L.gram_beam_finalize(C.byref(st), nret, max_length, p(seqs), p(scores), p(width), stream()), "beam_finalize")
    torch.cuda.synchronize()
    w = int(width[0])
    return seqs[:, :w].cpu(), scores.cpu(), int(keep["error"][0]), trace


def vt_blocked(vt):
    """[..., 64, S] (V transposed) -> the bank's layout [..., S/32, 64, 32]: V^T blocked by 32 keys, so that the 64 x 32 tile of a
    32-key step is 4 KiB contiguous (include/gram_hip.h, gram_kv_bank_t)."""
    S = vt.shape[-1]
    return vt.unflatten(-1, (S // 32, 32)).transpose(-3, -2).contiguous()


def vt_unblocked(vtb):
    """inverse of vt_blocked: [..., S/32, 64, 32] -> [..., 64, S]"""
    return vtb.transpose(-3, -2).flatten(-2)
