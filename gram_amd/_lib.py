"""ctypes binding of libgram_hip.so (include/gram_hip.h).

The product path has NO fallback: if the shared library is missing or a symbol is absent, the
import fails loudly.  Build it with ``python -c 'import __graft_entry__ as g; g.build()'`` or
``make -C gram_amd/csrc``.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GRAM_LIB") or os.path.join(_HERE, "csrc", "libgram_hip.so")  # (GRAM_LIB: an A/B build of the same ABI)

GRAM_MAX_BEAMS = 64
GRAM_MAX_DEC_LEN = 64
GRAM_MAX_PASSAGE_LEN = 128
EPI_BF16, EPI_BF16_RELU, EPI_F32_ADD, EPI_F32, EPI_KV_BANK = range(5)
K_GEMM, K_ENC_ATTN, K_CROSS_ATTN, K_DEC_SELF_ATTN, K_ROWOPS, K_LSE, K_BEAM = range(7)
E_ARG, E_WORKSPACE, E_BEAM, E_NONFINITE = -1, -2, -3, -4
ABI_VERSION = 7
# two-piece mode (gram_hip.h, gram_split_t): 16-bit pieces per value -> MFMA products per product
MAX_PIECES = 2
SPLIT_NPROD = (0, 1, 3)


def interleave(pieces):
    """[2][rows][cols] pieces -> the interleaved [rows][2 * cols] matrix a GEMM reads (gram_hip.h): per 32-column block, piece 0's
    32 values then piece 1's."""
    p, rows, cols = pieces.shape
    assert p == 2 and cols % 32 == 0
    return pieces.view(2, rows, cols // 32, 32).permute(1, 2, 0, 3).reshape(rows, 2 * cols).contiguous()


def deinterleave(x):
    """inverse of ``interleave``: [rows][2 * cols] -> [2][rows][cols]"""
    rows, c2 = x.shape
    return x.view(rows, c2 // 64, 2, 32).permute(2, 0, 1, 3).reshape(2, rows, c2 // 2).contiguous()


# gram_debug_set_stage_pieces: the stages of a sensitivity sweep (gram_hip.h: enum gram_stage)
STAGES = ("enc_attn", "enc_ffn", "bank_k", "bank_v", "dec_self", "dec_cross", "dec_ffn", "lm_head")

vp = C.c_void_p
i32 = C.c_int32
i64 = C.c_int64
f32 = C.c_float


class KVBank(C.Structure):
    _fields_ = [("k", vp), ("vt", vp), ("n_layers", i32), ("B", i32), ("H", i32), ("S", i32), ("passage_map", vp), ("N", i32),
                ("L", i32)]


class Trie(C.Structure):
    _fields_ = [("child_off", vp), ("child_tok", vp), ("child_node", vp), ("n_nodes", i32), ("n_edges", i32),
                ("max_fanout", i32), ("min_seq_len", i32)]


class LiveRows(C.Structure):
    _fields_ = [("rows", vp), ("rowpos", vp), ("users", vp), ("tokens", vp), ("counts", vp)]


class BeamState(C.Structure):
    _fields_ = [("B", i32), ("K", i32), ("Tmax", i32), ("length_penalty", f32), ("eos", i32), ("pad", i32),
                ("tokens", vp), ("node", vp), ("beam_scores", vp), ("seq", vp), ("anc", vp), ("done", vp),
                ("n_hyps", vp), ("hyp_score", vp), ("worst", vp), ("hyp_len", vp), ("hyp_tok", vp), ("error", vp),
                ("cand_logits", vp), ("cand_logits_users", i32), ("cand_logits_stride", i64)]


class Split(C.Structure):
    _fields_ = [("pieces", i32), ("c_interleaved", i32), ("c_pstride", i64), ("bank_pstride", i64), ("out_scale", f32)]


class NormFusion(C.Structure):
    _fields_ = [("xb_out", vp), ("ss_out", vp), ("ss_in", vp), ("nblk_in", i32), ("d", i32), ("eps", f32), ("quarter", i32),
                ("xs_in", vp), ("xs_out", vp)]


class Compaction(C.Structure):
    _fields_ = [("n_active", i32), ("passage_map", vp), ("ids", vp), ("mask", vp),
                ("n_cached", i32), ("cache_L", i32), ("cache_x", vp), ("cache_slot", vp)]


class ModelDesc(C.Structure):
    _fields_ = [
        ("vocab", i32), ("d_model", i32), ("d_ff", i32), ("n_heads", i32), ("n_enc_layers", i32),
        ("n_dec_layers", i32), ("max_passages", i32), ("tie_word_embeddings", i32),
        ("use_position_embedding", i32), ("fold_norm", i32), ("eps", f32),
        ("embed_f32", vp), ("lm_head_bf16", vp), ("pos_emb_f32", vp), ("enc_bias_f32", vp), ("dec_bias_f32", vp),
        ("enc_final_ln", vp), ("dec_final_ln", vp),
        ("enc_ln1", C.POINTER(vp)), ("enc_wqkv", C.POINTER(vp)), ("enc_wo", C.POINTER(vp)),
        ("enc_ln2", C.POINTER(vp)), ("enc_wi", C.POINTER(vp)), ("enc_wo2", C.POINTER(vp)),
        ("dec_ln1", C.POINTER(vp)), ("dec_wqkv", C.POINTER(vp)), ("dec_wo", C.POINTER(vp)),
        ("dec_ln2", C.POINTER(vp)), ("dec_wq_x", C.POINTER(vp)), ("dec_wo_x", C.POINTER(vp)),
        ("dec_ln3", C.POINTER(vp)), ("dec_wi", C.POINTER(vp)), ("dec_wo2", C.POINTER(vp)),
        ("dec_wkv_x_all", vp),
        ("pieces", i32), ("lm_head_f32", vp), ("w_scales", C.POINTER(f32)),
    ]


# name -> (restype, argtypes); every symbol include/gram_hip.h declares
SIGNATURES = {
    "gram_abi_version": (C.c_int, []),
    "gram_piece_format": (C.c_int, []),
    "gram_gemm_bf16": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(KVBank), vp]),
    "gram_gemm_bf16_ex": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(KVBank), C.POINTER(NormFusion), vp]),
    "gram_row_rscale": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, f32, vp]),
    "gram_row_rscale_xs": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, f32, vp]),
    "gram_embed_ex_xs": (C.c_int, [vp, vp, C.c_int, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "gram_embed_ex": (C.c_int, [vp, vp, C.c_int, vp, vp, vp, C.c_int, C.c_int, C.c_int, vp]),
    "gram_gemm_bf16_lse": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "gram_lse_combine": (C.c_int, [vp, vp, C.c_int, C.c_int, vp]),
    "gram_embed_i64": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, vp]),
    "gram_embed_i32": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, vp]),
    "gram_rmsnorm_bf16": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, f32, f32, vp, C.c_int, C.c_int, vp]),
    "gram_enc_self_attn": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, vp]),
    "gram_cross_attn_decode": (C.c_int, [vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "gram_dec_self_attn": (C.c_int, [vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "gram_row_lse": (C.c_int, [vp, vp, C.c_int, C.c_int, vp]),
    "gram_beam_init": (C.c_int, [C.POINTER(BeamState), C.POINTER(Trie), C.c_int, vp]),
    "gram_beam_step": (C.c_int, [C.POINTER(BeamState), C.POINTER(Trie), vp, vp, C.c_int, C.c_int, C.c_int, vp]),
    "gram_beam_step_sparse": (C.c_int, [C.POINTER(BeamState), C.POINTER(Trie), vp, vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, vp]),
    "gram_greedy_step": (C.c_int, [C.POINTER(BeamState), C.POINTER(Trie), vp, C.c_int, C.c_int, vp]),
    "gram_greedy_finalize": (C.c_int, [C.POINTER(BeamState), C.c_int, vp, vp, vp]),
    "gram_beam_finalize": (C.c_int, [C.POINTER(BeamState), C.c_int, C.c_int, vp, vp, vp, vp]),
    "gram_trie_item_index": (C.c_int, [C.POINTER(Trie), vp, vp, C.c_int, C.c_int, vp, vp]),
    "gram_model_create": (vp, [C.POINTER(ModelDesc)]),
    "gram_model_destroy": (None, [vp]),
    "gram_workspace_bytes": (i64, [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "gram_workspace_encoder_x_offset": (i64, [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "gram_encode_fused": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_int, vp, i64, C.c_int, C.c_int, vp, vp]),
    "gram_decode_step": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, i64, vp, vp]),
    "gram_prof_pp_clock": (C.c_int, [C.POINTER(C.c_double), C.c_int]),
    "gram_prof_pp_clock_enable": (C.c_int, [C.c_int]),
    "gram_debug_set_gemm_variant": (C.c_int, [C.c_int]),
    "gram_gemm_stream_max_m": (C.c_int, []),
    "gram_debug_set_live_rows": (C.c_int, [C.c_int]),
    "gram_debug_set_stage_pieces": (C.c_int, [C.POINTER(i32), C.c_int]),
    "gram_debug_stream_read": (C.c_int, [vp, C.c_size_t, vp, vp]),
    "gram_debug_stream_read_variant": (C.c_int, [vp, C.c_size_t, vp, C.c_int, C.c_int, vp]),
    "gram_prof_enable": (C.c_int, [C.c_uint32, C.c_int]),
    "gram_prof_reset": (C.c_int, []),
    "gram_prof_collect": (C.c_int, [C.c_int, C.POINTER(C.c_double), C.POINTER(i64), C.POINTER(C.c_double), C.POINTER(i64)]),
    "gram_rmsnorm_bf16_map": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, f32, f32, vp, C.c_int, C.c_int, vp, vp]),
    "gram_generate_ex": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, f32, C.POINTER(Trie),
                                   C.POINTER(Compaction), vp, i64, vp, vp, C.POINTER(i32), vp]),
    "gram_encode_passages": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, vp, i64, vp, vp]),
    "gram_gather_passage_x": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "gram_live_rows": (C.c_int, [C.POINTER(BeamState), C.POINTER(Trie), C.POINTER(LiveRows), vp]),
    "gram_dec_self_attn_live": (C.c_int, [vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, vp, C.c_int, C.c_int, C.c_int, vp]),
    "gram_cross_attn_decode_live": (C.c_int, [vp, vp, vp, vp, vp, C.c_int, vp, vp, C.c_int, C.c_int, C.c_int, vp]),
    "gram_beam_step_sparse_live": (C.c_int, [C.POINTER(BeamState), C.POINTER(Trie), vp, vp, C.c_int, vp, C.c_int, C.c_int, vp, vp]),
    "gram_gemm_bf16_split": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(KVBank),
                                       C.POINTER(NormFusion), C.POINTER(Split), vp]),
    "gram_gemm_bf16_lse_split": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(Split), vp]),
    "gram_embed_ex_split": (C.c_int, [vp, vp, C.c_int, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "gram_rmsnorm_bf16_split": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, f32, f32, vp, C.c_int, C.c_int, vp, C.c_int, vp]),
    "gram_enc_self_attn_split": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, i64, vp]),
    "gram_cross_attn_decode_split": (C.c_int, [vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, C.c_int, i64, i64, vp, vp]),
    "gram_mask_key_bits": (C.c_int, [vp, vp, C.c_int, C.c_int, vp]),
    "gram_dec_self_attn_split": (C.c_int, [vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int, i64, i64, vp]),
    "gram_beam_step_sparse_split": (C.c_int, [C.POINTER(BeamState), C.POINTER(Trie), vp, vp, C.c_int, vp, C.c_int, C.c_int, C.c_int,
                                              vp, C.c_int, vp]),
    "gram_generate": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, f32, C.POINTER(Trie),
                                vp, i64, vp, vp, C.POINTER(i32), vp]),
}

# the `_f16` aliases (gram_hip.h): same signatures as the `_bf16` names
for _old, _new in (("gram_gemm_bf16", "gram_gemm_f16"), ("gram_gemm_bf16_ex", "gram_gemm_f16_ex"), ("gram_gemm_bf16_split", "gram_gemm_f16_split"),
                   ("gram_gemm_bf16_lse", "gram_gemm_f16_lse"), ("gram_gemm_bf16_lse_split", "gram_gemm_f16_lse_split"),
                   ("gram_rmsnorm_bf16", "gram_rmsnorm_f16"), ("gram_rmsnorm_bf16_map", "gram_rmsnorm_f16_map"),
                   ("gram_rmsnorm_bf16_split", "gram_rmsnorm_f16_split")):
    SIGNATURES[_new] = SIGNATURES[_old]

_lib = None


def load() -> C.CDLL:
    """Load libgram_hip.so and bind every declared symbol; raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch ships its own HIP runtime (torch/lib/libamdhip64.so); the library's device pointers and streams come from PyTorch,
    # so both must talk to ONE runtime instance: load torch's first, then libgram_hip.so's libamdhip64.so.7 dependency resolves
    # to the copy already in the process.  (Loaded the other way round, generate() fails with hipErrorNoDevice.)
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the HIP extension is not built (run __graft_entry__.build() or "
            "`make -C gram_amd/csrc`).  gram_amd has no CPU fallback."
        )
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.gram_abi_version() != ABI_VERSION:
        raise ImportError("libgram_hip.so ABI version mismatch")
    _lib = lib
    return lib


class GramHipError(RuntimeError):
    pass


def piece_dtype():
    """torch dtype of the loaded library's 16-bit operands (gram_piece_format(): float16 for the default build, bfloat16 for
    ``make PIECE=bf16``)."""
    import torch
    return torch.float16 if load().gram_piece_format() == 1 else torch.bfloat16


def check(code: int, what: str) -> None:
    if code == 0:
        return
    names = {E_ARG: "GRAM_E_ARG (bad shape / unsupported size)", E_WORKSPACE: "GRAM_E_WORKSPACE",
             E_BEAM: "GRAM_E_BEAM (impossible beam state; HF would raise here)",
             E_NONFINITE: "GRAM_E_NONFINITE (a returned score is NaN / +inf: an activation left the range of the 16-bit pieces -- IEEE half, "
                          "|x| <= 65504; build and load the bfloat16 library: make -C gram_amd/csrc PIECE=bf16, GRAM_LIB=.../libgram_hip_bf16.so)"}
    raise GramHipError(f"{what} failed: {names.get(code, f'hipError {code}')}")
