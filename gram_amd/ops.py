"""PyTorch-ROCm custom ops of the scoring path, namespace ``gram``.

Thin ``torch.library`` wrappers over the C ABI (include/gram_hip.h): tensors in, tensors out, stream-ordered on
PyTorch's current HIP stream, with meta ("fake") implementations so the ops trace under ``torch.compile`` /
``FakeTensorMode``.  They are registered for the ``cuda`` device type ONLY: calling one on CPU tensors raises
(there is no CPU path in gram_amd; the CPU restatement lives in oracle/ and is test infrastructure).

    torch.ops.gram.generate          GRAM.generate's whole path (gram_generate_ex) -- what ``GRAM.generate`` calls
    torch.ops.gram.linear            nn.Linear without bias: A @ W^T on the 16-bit MFMA (gram_gemm_bf16, 16-bit epilogue)
    torch.ops.gram.enc_self_attn     T5Attention self branch on (P*L, 3*inner) q|k|v rows (gram_enc_self_attn)
    torch.ops.gram.cross_attn_decode the fusion read of one decoder layer and step over the beam-shared bank
    torch.ops.gram.trie_step         one Trie-constrained beam-search step on dense logits (gram_row_lse + gram_beam_step)
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Tuple

import torch

from . import _lib

Tensor = torch.Tensor


def _stream(t: Tensor) -> int:
    return torch.cuda.current_stream(t.device).cuda_stream


def _p(t: Optional[Tensor]):
    return None if t is None else t.data_ptr()


def _check(name: str, t: Tensor, dtype, shape=None, device=None, contiguous: bool = True) -> None:
    """Everything the C ABI assumes about a tensor before its data_ptr() is taken: a wrong dtype, a transposed or sliced view or
    a shape that does not match the kernel's indexing would otherwise be silent garbage or an out-of-bounds device read."""
    if t.dtype != dtype:
        raise ValueError(f"{name}: dtype {t.dtype}, expected {dtype}")
    if device is not None and t.device != device:
        raise ValueError(f"{name}: on {t.device}, expected {device}")
    if contiguous and not t.is_contiguous():
        raise ValueError(f"{name}: must be contiguous (got strides {t.stride()})")
    if t.numel() and t.data_ptr() % 16:
        raise ValueError(f"{name}: data pointer must be 16-byte aligned")
    if shape is not None:
        if t.dim() != len(shape) or any(e is not None and int(s_) != int(e) for s_, e in zip(t.shape, shape)):
            raise ValueError(f"{name}: shape {tuple(t.shape)}, expected {tuple('*' if e is None else e for e in shape)}")


# ---------------------------------------------------------------------------------------------- generate
@torch.library.custom_op("gram::generate", mutates_args=("workspace",), device_types="cuda")
def generate(input_ids: Tensor, attention_mask: Tensor, handle: int, workspace: Tensor, trie_child_off: Tensor,
             trie_child_tok: Tensor, trie_child_node: Tensor, trie_max_fanout: int, trie_min_seq_len: int, num_beams: int,
             num_return_sequences: int, max_length: int, length_penalty: float, comp_map: Optional[Tensor],
             comp_ids: Optional[Tensor], comp_mask: Optional[Tensor], cache_slot: Optional[Tensor], cache_x: Optional[Tensor],
             n_cached: int, cache_L: int) -> Tuple[Tensor, Tensor, Tensor]:
    """input_ids i64 (B,N,L), attention_mask u8 (B,N,L), ``handle`` a gram_model_t* as int, ``workspace`` a u8 scratch tensor of
    gram_workspace_bytes; the Trie as its CSR arrays (gram_trie_t); comp_* / cache_* the gram_compaction_t fields (None = off).
    Returns (sequences i64 (B*nret, max_length), sequences_scores f32 (B*nret) -- empty for greedy search --, width i32 (1,))."""
    lib = _lib.load()
    B, N, L = input_ids.shape
    dev = input_ids.device
    K, nret = num_beams, num_return_sequences
    trie = _lib.Trie(trie_child_off.data_ptr(), trie_child_tok.data_ptr(), trie_child_node.data_ptr(), trie_child_off.numel() - 1,
                     trie_child_tok.numel(), trie_max_fanout, trie_min_seq_len)
    comp = None
    if comp_map is not None:
        comp = _lib.Compaction(comp_map.numel(), comp_map.data_ptr(), _p(comp_ids), _p(comp_mask), n_cached, cache_L, _p(cache_x),
                               _p(cache_slot))
    seqs = torch.empty(B * nret, max_length, dtype=torch.int64, device=dev)
    scores = torch.empty(B * nret if K > 1 else 0, dtype=torch.float32, device=dev)
    width = C.c_int32(0)
    with torch.cuda.device(dev):
        rc = lib.gram_generate_ex(handle, input_ids.data_ptr(), attention_mask.data_ptr(), B, N, L, K, nret, max_length,
                                  float(length_penalty), C.byref(trie), C.byref(comp) if comp is not None else None,
                                  workspace.data_ptr(), workspace.numel(), seqs.data_ptr(), scores.data_ptr() if K > 1 else None,
                                  C.byref(width), _stream(input_ids))
    _lib.check(rc, "gram_generate")
    return seqs, scores, torch.tensor([width.value], dtype=torch.int32)


@generate.register_fake
def _(input_ids, attention_mask, handle, workspace, trie_child_off, trie_child_tok, trie_child_node, trie_max_fanout,
      trie_min_seq_len, num_beams, num_return_sequences, max_length, length_penalty, comp_map, comp_ids, comp_mask, cache_slot,
      cache_x, n_cached, cache_L):
    B = input_ids.shape[0]
    return (input_ids.new_empty(B * num_return_sequences, max_length),
            input_ids.new_empty(B * num_return_sequences if num_beams > 1 else 0, dtype=torch.float32),
            torch.empty(1, dtype=torch.int32))


# ---------------------------------------------------------------------------------------------- linear
@torch.library.custom_op("gram::linear", mutates_args=(), device_types="cuda")
def linear(a: Tensor, w: Tensor, relu: bool = False) -> Tensor:
    """a (M, K) row-major (rows may be strided: stride(1) == 1, stride(0) % 8 == 0), w (N, K) contiguous ([out][in], as nn.Linear
    stores it), both of the library's 16-bit type (_lib.piece_dtype()): (M, N) = a @ w^T, fp32 accumulate.  N % 128 == 0, K % 64 == 0."""
    dt = _lib.piece_dtype()
    if a.dim() != 2 or w.dim() != 2:
        raise ValueError("linear: a must be (M, K) and w (N, K)")
    M, K = a.shape
    N = w.shape[0]
    _check("a", a, dt, contiguous=False)
    _check("w", w, dt, (None, K), a.device)
    if M < 1 or a.stride(1) != 1 or a.stride(0) < K or a.stride(0) % 8 or N % 128 or K % 64:
        raise ValueError(f"linear: need M >= 1, a.stride(1) == 1, a.stride(0) >= K and % 8 == 0, N % 128 == 0, K % 64 == 0 "
                         f"(got a {tuple(a.shape)} strides {a.stride()}, w {tuple(w.shape)})")
    out = torch.empty(M, N, dtype=dt, device=a.device)
    with torch.cuda.device(a.device):
        rc = _lib.load().gram_gemm_bf16(a.data_ptr(), w.data_ptr(), out.data_ptr(), M, N, K, a.stride(0), N,
                                        _lib.EPI_BF16_RELU if relu else _lib.EPI_BF16, None, _stream(a))
    _lib.check(rc, "gram_gemm_bf16")
    return out


@linear.register_fake
def _(a, w, relu=False):
    return a.new_empty(a.shape[0], w.shape[0])


# ---------------------------------------------------------------------------------------------- encoder self-attention
@torch.library.custom_op("gram::enc_self_attn", mutates_args=(), device_types="cuda")
def enc_self_attn(qkv: Tensor, bias: Tensor, mask: Tensor, num_heads: int) -> Tensor:
    """qkv 16-bit (P*L, 3*inner) rows q|k|v; bias f32 (H, 255) by (key - query + 127); mask u8 (P, L) -> 16-bit (P*L, inner)."""
    dt = _lib.piece_dtype()
    if mask.dim() != 2:
        raise ValueError("enc_self_attn: mask must be (P, L)")
    P, L = mask.shape
    inner = num_heads * 64
    _check("qkv", qkv, dt, (P * L, 3 * inner))
    _check("bias", bias, torch.float32, (num_heads, 255), qkv.device)
    _check("mask", mask, torch.uint8, None, qkv.device)
    if L < 32 or L > _lib.GRAM_MAX_PASSAGE_LEN or L % 32 or num_heads < 1:
        raise ValueError(f"enc_self_attn: L must be a multiple of 32 in [32, {_lib.GRAM_MAX_PASSAGE_LEN}] (got {L})")
    out = torch.empty(P * L, inner, dtype=dt, device=qkv.device)
    with torch.cuda.device(qkv.device):
        rc = _lib.load().gram_enc_self_attn(qkv.data_ptr(), bias.data_ptr(), mask.data_ptr(), out.data_ptr(), P, L, num_heads,
                                            _stream(qkv))
    _lib.check(rc, "gram_enc_self_attn")
    return out


@enc_self_attn.register_fake
def _(qkv, bias, mask, num_heads):
    return qkv.new_empty(qkv.shape[0], num_heads * 64)


# ---------------------------------------------------------------------------------------------- cross-attention decode
@torch.library.custom_op("gram::cross_attn_decode", mutates_args=(), device_types="cuda")
def cross_attn_decode(q: Tensor, k_bank: Tensor, vt_bank: Tensor, mask: Tensor, num_beams: int) -> Tensor:
    """q bf16 (B*K, inner); k_bank bf16 (B, H, S, 64); vt_bank bf16 (B, H, S/32, 64, 32) = V transposed, blocked by 32 keys (one
    copy per user, shared by its K beams); mask u8 (B, S) -> bf16 (B*K, inner)."""
    dt = _lib.piece_dtype()
    if k_bank.dim() != 4 or k_bank.shape[-1] != 64:
        raise ValueError("cross_attn_decode: k_bank must be (B, H, S, 64)")
    B, H, S, _ = k_bank.shape
    if S < 32 or S % 32 or S > 4096 or not 1 <= num_beams <= _lib.GRAM_MAX_BEAMS:
        raise ValueError(f"cross_attn_decode: S must be a multiple of 32 in [32, 4096] and 1 <= num_beams <= {_lib.GRAM_MAX_BEAMS}")
    _check("k_bank", k_bank, dt)
    _check("vt_bank", vt_bank, dt, (B, H, S // 32, 64, 32), k_bank.device)  # V^T blocked by 32 keys
    _check("q", q, dt, (B * num_beams, H * 64), k_bank.device)
    _check("mask", mask, torch.uint8, (B, S), k_bank.device)
    out = torch.empty_like(q)
    with torch.cuda.device(q.device):
        rc = _lib.load().gram_cross_attn_decode(q.data_ptr(), k_bank.data_ptr(), vt_bank.data_ptr(), mask.data_ptr(), out.data_ptr(),
                                                B, num_beams, H, S, _stream(q))
    _lib.check(rc, "gram_cross_attn_decode")
    return out


@cross_attn_decode.register_fake
def _(q, k_bank, vt_bank, mask, num_beams):
    return torch.empty_like(q)


# ---------------------------------------------------------------------------------------------- Trie-constrained search step
@torch.library.custom_op("gram::trie_step", mutates_args=("tokens", "node", "beam_scores", "seq", "anc", "done", "n_hyps", "hyp_score",
                                                          "worst", "hyp_len", "hyp_tok", "error"), device_types="cuda")
def trie_step(logits: Tensor, tokens: Tensor, node: Tensor, beam_scores: Tensor, seq: Tensor, anc: Tensor, done: Tensor,
              n_hyps: Tensor, hyp_score: Tensor, worst: Tensor, hyp_len: Tensor, hyp_tok: Tensor, error: Tensor,
              trie_child_off: Tensor, trie_child_tok: Tensor, trie_child_node: Tensor, trie_max_fanout: int, num_beams: int,
              cur_len: int, length_penalty: float) -> Tensor:
    """One HF-4.26 beam-search step on dense logits f32 (B*K, V): log-softmax normaliser, Trie mask, top-2K, BeamSearchScorer.process,
    the beam state (gram_beam_state_t fields) advanced in place.  Returns the row LSE f32 (B*K)."""
    lib = _lib.load()
    if logits.dim() != 2 or seq.dim() != 2:
        raise ValueError("trie_step: logits must be (B*K, V) and seq (B*K, Tmax)")
    R, V = logits.shape
    K = num_beams
    if K < 1 or K > _lib.GRAM_MAX_BEAMS or R % K or not 1 <= cur_len < seq.shape[1] or seq.shape[1] > _lib.GRAM_MAX_DEC_LEN:
        raise ValueError("trie_step: rows must be B * num_beams, 1 <= cur_len < Tmax <= %d" % _lib.GRAM_MAX_DEC_LEN)
    Bq, Tq, dev = R // K, seq.shape[1], logits.device
    _check("logits", logits, torch.float32)
    for name, t, dtype, shape in (("tokens", tokens, torch.int32, (R,)), ("node", node, torch.int32, (R,)),
                                  ("beam_scores", beam_scores, torch.float32, (R,)), ("seq", seq, torch.int32, (R, Tq)),
                                  ("anc", anc, torch.int32, (Tq, R)), ("done", done, torch.int32, (Bq,)),
                                  ("n_hyps", n_hyps, torch.int32, (Bq,)), ("hyp_score", hyp_score, torch.float64, (Bq, K + 1)),
                                  ("worst", worst, torch.float64, (Bq,)), ("hyp_len", hyp_len, torch.int32, (Bq, K + 1)),
                                  ("hyp_tok", hyp_tok, torch.int32, (Bq, K + 1, Tq)), ("error", error, torch.int32, None),
                                  ("trie_child_off", trie_child_off, torch.int32, None),
                                  ("trie_child_tok", trie_child_tok, torch.int32, None),
                                  ("trie_child_node", trie_child_node, torch.int32, (trie_child_tok.numel(),))):
        _check(name, t, dtype, shape, dev)
    if error.numel() < 1 or trie_child_off.numel() < 2:
        raise ValueError("trie_step: error needs >= 1 element, trie_child_off >= 2")
    st = _lib.BeamState(B=R // K, K=K, Tmax=seq.shape[1], length_penalty=length_penalty, eos=1, pad=0, tokens=tokens.data_ptr(),
                        node=node.data_ptr(), beam_scores=beam_scores.data_ptr(), seq=seq.data_ptr(), anc=anc.data_ptr(),
                        done=done.data_ptr(), n_hyps=n_hyps.data_ptr(), hyp_score=hyp_score.data_ptr(), worst=worst.data_ptr(),
                        hyp_len=hyp_len.data_ptr(), hyp_tok=hyp_tok.data_ptr(), error=error.data_ptr())
    trie = _lib.Trie(trie_child_off.data_ptr(), trie_child_tok.data_ptr(), trie_child_node.data_ptr(), trie_child_off.numel() - 1,
                     trie_child_tok.numel(), trie_max_fanout, 0)
    lse = torch.empty(R, dtype=torch.float32, device=logits.device)
    with torch.cuda.device(logits.device):
        _lib.check(lib.gram_row_lse(logits.data_ptr(), lse.data_ptr(), R, V, _stream(logits)), "gram_row_lse")
        _lib.check(lib.gram_beam_step(C.byref(st), C.byref(trie), logits.data_ptr(), lse.data_ptr(), V, cur_len, K, _stream(logits)),
                   "gram_beam_step")
    return lse


@trie_step.register_fake
def _(logits, tokens, node, beam_scores, seq, anc, done, n_hyps, hyp_score, worst, hyp_len, hyp_tok, error, trie_child_off,
      trie_child_tok, trie_child_node, trie_max_fanout, num_beams, cur_len, length_penalty):
    return logits.new_empty(logits.shape[0])


__all__ = ["generate", "linear", "enc_self_attn", "cross_attn_decode", "trie_step"]
