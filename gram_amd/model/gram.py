"""GRAM model wrapper: the reference's ``src/model/gram.py`` interface over libgram_hip.so.

Same constructor (``GRAM(config)``), same state-dict key layout, same
``load_t5`` / ``load_state_dict`` / ``generate`` signatures as the reference class
(gram.py:14-107,162-165), so ``main_generative_gram.py`` and the runners drive it unchanged.
``generate`` runs entirely on the MI355X through ONE C-ABI call (``gram_generate``): encoder,
late fusion, the beam-shared KV bank, and the Trie-constrained beam search all stay on the
device; there is no PyTorch-op or CPU fallback.

Out of scope here (training half): ``forward`` with labels.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from typing import Callable, Dict, Optional

import torch
from torch import nn

from .. import _lib
from ..utils.generation_trie import FlatTrie, Trie


class GenerateOutput(dict):
    """Mapping with attribute access, standing in for HF's BeamSearchEncoderDecoderOutput:
    the runner reads ``prediction["sequences"]`` / ``prediction["sequences_scores"]``
    (single_runner_gram.py:654-655)."""

    __getattr__ = dict.get


class _Node(nn.Module):
    """Anonymous container used to reproduce the reference's dotted parameter names."""


def _register(root: nn.Module, dotted: str, param: nn.Parameter) -> None:
    parts = dotted.split(".")
    mod = root
    for p in parts[:-1]:
        child = mod._modules.get(p)
        if child is None:
            child = _Node()
            mod.add_module(p, child)
        mod = child
    mod.register_parameter(parts[-1], param)


def relative_position_bucket(rel: torch.Tensor, bidirectional: bool, num_buckets: int, max_distance: int) -> torch.Tensor:
    """Bucket index of a relative position (key - query).  Same integer results as
    T5Attention._relative_position_bucket (gram_t5_modeling.py:398-450), evaluated once on the
    host to build the per-head bias tables the kernels index."""
    rel = rel.to(torch.long)
    out = torch.zeros_like(rel)
    n = num_buckets
    if bidirectional:
        n //= 2
        out += (rel > 0).long() * n
        dist = rel.abs()
    else:
        dist = (-rel).clamp(min=0)
    exact = n // 2
    log_part = exact + (torch.log(dist.float() / exact) / math.log(max_distance / exact) * (n - exact)).long()
    log_part = log_part.clamp(max=n - 1)
    return out + torch.where(dist < exact, dist, log_part)


class GRAM(nn.Module):
    main_input_name = "input_ids"

    def __init__(self, config):
        super().__init__()
        self.config = config
        c = config
        if c.d_kv != 64:
            raise ValueError("gram_amd kernels are built for d_kv == 64 (every T5 checkpoint the reference uses)")
        if getattr(c, "feed_forward_proj", "relu") not in ("relu",):
            raise ValueError("only the ReLU feed-forward of t5-{small,base,large} is on the path")
        self.max_seq_len = getattr(c, "max_seq_len", 128)
        self.max_item_num = getattr(c, "max_item_num", 20)
        self.use_position_embedding = bool(getattr(c, "use_position_embedding", True))
        d, inner, F, V = c.d_model, c.num_heads * c.d_kv, c.d_ff, c.vocab_size
        self.model_dim = d
        dec_layers = c.num_decoder_layers if getattr(c, "num_decoder_layers", None) is not None else c.num_layers
        self._n_enc, self._n_dec = c.num_layers, dec_layers

        def P(*shape):
            return nn.Parameter(torch.empty(*shape, dtype=torch.float32))

        shared = P(V, d)
        _register(self, "shared.weight", shared)
        _register(self, "encoder.encoder.embed_tokens.weight", shared)
        _register(self, "decoder.embed_tokens.weight", shared)

        def attn(prefix, rel):
            _register(self, prefix + ".q.weight", P(inner, d))
            _register(self, prefix + ".k.weight", P(inner, d))
            _register(self, prefix + ".v.weight", P(inner, d))
            _register(self, prefix + ".o.weight", P(d, inner))
            if rel:
                _register(self, prefix + ".relative_attention_bias.weight", P(c.relative_attention_num_buckets, c.num_heads))

        def ff(prefix):
            _register(self, prefix + ".DenseReluDense.wi.weight", P(F, d))
            _register(self, prefix + ".DenseReluDense.wo.weight", P(d, F))

        for i in range(self._n_enc):
            p = f"encoder.encoder.block.{i}.module.layer"
            attn(p + ".0.SelfAttention", i == 0)
            _register(self, p + ".0.layer_norm.weight", P(d))
            ff(p + ".1")
            _register(self, p + ".1.layer_norm.weight", P(d))
        _register(self, "encoder.encoder.final_layer_norm.weight", P(d))
        for i in range(self._n_dec):
            p = f"decoder.block.{i}.layer"
            attn(p + ".0.SelfAttention", i == 0)
            _register(self, p + ".0.layer_norm.weight", P(d))
            attn(p + ".1.EncDecAttention", False)
            _register(self, p + ".1.layer_norm.weight", P(d))
            ff(p + ".2")
            _register(self, p + ".2.layer_norm.weight", P(d))
        _register(self, "decoder.final_layer_norm.weight", P(d))
        if self.use_position_embedding:
            pos = P(self.max_item_num + 1, d)  # one extra for the coarse user prompt (gram.py:23-26)
            _register(self, "position_embedding.weight", pos)
            _register(self, "encoder.position_embedding.weight", pos)
        _register(self, "lm_head.weight", shared if getattr(c, "tie_word_embeddings", True) else P(V, d))
        self._init_weights()
        self._packed = None  # (device, version, handle, keepalive)
        self._version = 0
        self._workspace = None
        self._tries: Dict[int, tuple] = {}  # id(trie) -> (trie, FlatTrie)
        # The reference computes in fp32; "f16x3" (two IEEE-half pieces per value, three MFMA products per product) is the cheapest
        # arithmetic that keeps Recall@5 / NDCG@5 within 1e-4 of it, so it is what a drop-in model starts in.
        # GRAM_PRECISION / set_precision() choose another.
        self._precision = os.environ.get("GRAM_PRECISION") or None  # None: the loaded library's default, resolved at first use
        if self._precision is not None and self._precision not in self.PRECISIONS:
            raise ValueError(f"GRAM_PRECISION={self._precision!r}: choose from {self.PRECISIONS}")
        self._pcache = None  # passage cache: dict(x, canon, keys, perm) on the device
        self._stage_caps: Dict[str, int] = {}  # sensitivity sweeps only (set_stage_pieces)

    # ------------------------------------------------------------------ weights
    def _init_weights(self) -> None:
        """Distributions of the reference initialiser (gram_t5_modeling.py:865-929, gram.py:32-33)."""
        c = self.config
        d, dk, H, F = c.d_model, c.d_kv, c.num_heads, c.d_ff
        with torch.no_grad():
            for name, p in self.named_parameters():
                if name.endswith("layer_norm.weight"):
                    p.fill_(1.0)
                elif name.endswith(".q.weight"):
                    p.normal_(0.0, (d * dk) ** -0.5)
                elif name.endswith((".k.weight", ".v.weight", "wi.weight", "relative_attention_bias.weight")):
                    p.normal_(0.0, d ** -0.5)
                elif name.endswith(".o.weight"):
                    p.normal_(0.0, (H * dk) ** -0.5)
                elif name.endswith("wo.weight"):
                    p.normal_(0.0, F ** -0.5)
                elif name.endswith("position_embedding.weight"):
                    p.normal_(0.0, 0.02)
                else:
                    p.normal_(0.0, 1.0)

    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        out = super().load_state_dict(state_dict, strict=strict, **kw)
        self._invalidate()
        return out

    def load_t5(self, state_dict):
        """gram.py:162-165: load a plain T5 checkpoint (encoder blocks are not wrapped there:
        ``encoder.block.{i}.layer...``), non-strict, leaving ``position_embedding`` as initialised."""
        remapped = {}
        for k, v in state_dict.items():
            if k.startswith("encoder.block."):
                parts = k.split(".")
                k = "encoder.encoder.block." + parts[2] + ".module." + ".".join(parts[3:])
            elif k.startswith("encoder.") and not k.startswith("encoder.encoder."):
                k = "encoder." + k
            remapped[k] = v
        return self.load_state_dict(remapped, strict=False)

    # "f16" / "bf16": one 16-bit piece per value (11 / 8 significant bits).  "f16x3" / "bf16x3": every value travels as two pieces and
    # every product is three MFMA products (gram_hip.h, gram_split_t): ~2^-22 / ~2^-18 relative error at 3x the MFMA work.
    # The 16-bit type is a property of the library build (gram_piece_format(): libgram_hip.so, the default build, computes on IEEE
    # half; `make PIECE=bf16` builds libgram_hip_bf16.so on bfloat16); the mode's prefix must name the loaded library's type.
    PRECISIONS = ("bf16", "bf16x3", "f16", "f16x3")
    _PIECES = {"bf16": 1, "bf16x3": 2, "f16": 1, "f16x3": 2}

    @staticmethod
    def default_precision() -> str:
        return "f16x3" if _lib.load().gram_piece_format() == 1 else "bf16x3"

    @property
    def precision(self) -> str:
        if self._precision is None:
            self._precision = self.default_precision()
        return self._precision

    def set_precision(self, mode: str) -> None:
        """Arithmetic of the GEMM / attention operands (accumulation, residual stream, softmax and scores are fp32 in
        every mode).  The packed weights depend on it, so changing it re-packs on the next ``generate``."""
        if mode not in self.PRECISIONS:
            raise ValueError(f"unknown precision {mode!r}; choose from {self.PRECISIONS}")
        if mode != self._precision:
            self._precision = mode
            self._invalidate()

    def set_stage_pieces(self, caps: Optional[Dict[str, int]] = None) -> None:
        """Sensitivity sweeps (tests/precision_population.py): stage -> number of pieces its operands keep, for the stages of
        ``_lib.STAGES``; the other stages keep the mode's own count.  The kernels and their cost do not change: the upper pieces of
        a capped stage's weights are zeroed here and those of its activations in generate.hip (gram_debug_set_stage_pieces), which
        is arithmetically the smaller piece count.  The caps are process-wide in the library; ``None`` / ``{}`` lifts them."""
        caps = dict(caps or {})
        for k in caps:
            if k not in _lib.STAGES:
                raise ValueError(f"unknown stage {k!r}; choose from {_lib.STAGES}")
        arr = (C.c_int32 * len(_lib.STAGES))(*[int(caps.get(k, 99)) for k in _lib.STAGES])
        _lib.check(_lib.load().gram_debug_set_stage_pieces(arr if caps else None, len(_lib.STAGES)), "gram_debug_set_stage_pieces")
        if caps != self._stage_caps:
            self._stage_caps = caps
            self._invalidate()

    def _invalidate(self) -> None:
        self._version += 1
        self._pcache = None  # cached encoder states belong to the old weights

    def _apply(self, fn, *a, **kw):  # .to()/.cuda()/.float() all funnel through here
        out = super()._apply(fn, *a, **kw)
        self._invalidate()
        return out

    def train(self, mode: bool = True):
        # parameters may have been updated by an optimizer while training: repack on leaving train mode
        if self.training != bool(mode):
            self._invalidate()
        return super().train(mode)

    def wrap_encoder(self, use_checkpoint=False):  # API no-ops kept for drop-in compatibility
        pass

    def unwrap_encoder(self):
        pass

    def set_checkpoint(self, use_checkpoint):
        pass

    def forward(self, input_ids=None, attention_mask=None, **kwargs):
        raise NotImplementedError(
            "gram_amd implements the generative *scoring* path (GRAM.generate). The teacher-forced training "
            "forward/backward (SURVEY.md §8f N4) is out of scope of this build."
        )

    # ------------------------------------------------------------------ device packing
    def _device(self) -> torch.device:
        return self.get_parameter("shared.weight").device

    def _pack(self):
        dev = self._device()
        if dev.type != "cuda":
            raise RuntimeError(
                "gram_amd.GRAM.generate needs the model on a ROCm device (model.to('cuda')); there is no CPU path"
            )
        if self._packed is not None and self._packed[0] == dev and self._packed[1] == self._version:
            return self._packed[2]
        lib = _lib.load()
        if self._packed is not None:
            lib.gram_model_destroy(self._packed[2])
            self._packed = None
        c = self.config
        H = c.num_heads
        sd = {k: v.detach() for k, v in self.state_dict().items()}
        keep = []

        def f32(t):
            t = t.to(dev, torch.float32).contiguous()
            keep.append(t)
            return t.data_ptr()

        pieces = self._PIECES[self.precision]
        f16 = lib.gram_piece_format() == 1
        if self.precision.startswith("f16") != f16:
            raise _lib.GramHipError(f"precision {self.precision!r} needs the {'f16' if not f16 else 'bf16'} build of libgram_hip "
                                    f"(loaded: {_lib.LIB_PATH}; GRAM_LIB selects another build)")
        tdt = torch.float16 if f16 else torch.bfloat16
        w_scales = []

        caps = self._stage_caps

        def b16(t, stage=None, row_caps=None, no_scale=False):
            """[out][in] fp32 -> the MFMA weight operand: one 16-bit matrix, or (two-piece mode) the INTERLEAVED [out][in/32][2][32]
            matrix of W = p0 + p1, p0 = r16(W), p1 = r16(W - p0): a 64-column k-tile holds both pieces of a 32-column block.
            stage / row_caps: sensitivity sweeps (set_stage_pieces) -- the upper piece is zeroed (per row: row_caps)."""
            t = t.to(dev, torch.float32)
            # f16 build: the matrix is scaled by a power of two so that its largest entry sits in [2^13, 2^14) -- the low pieces of
            # small weights then stay normal numbers (an f16 subnormal keeps fewer bits); the GEMM multiplies its result by the
            # inverse (gram_model_desc_t.w_scales).  The 16-bit lm_head of the one-piece mode is read unscaled by the beam kernel.
            scale = 1.0
            if f16 and not (no_scale and pieces == 1):
                amax = float(t.abs().max())
                if amax > 0.0 and math.isfinite(amax):
                    scale = 2.0 ** (13 - math.floor(math.log2(amax)))
                t = t * scale
            w_scales.append(scale)
            if pieces == 1:
                t = t.to(tdt).contiguous()
            else:
                ps, r = [], t.clone()
                for _ in range(pieces):
                    ps.append(r.to(tdt))
                    r -= ps[-1].float()
                cap = caps.get(stage, 99) if stage else 99
                for j in range(pieces):
                    if j >= cap:
                        ps[j].zero_()
                    elif row_caps is not None:
                        ps[j][row_caps.to(dev) <= j] = 0
                t = _lib.interleave(torch.stack(ps))
            keep.append(t)
            return t.data_ptr()

        def ptr_array(vals):
            arr = (C.c_void_p * len(vals))(*vals)
            keep.append(arr)
            return C.cast(arr, C.POINTER(C.c_void_p))

        # T5LayerNorm folding (gram_norm_fusion_t): the gain g of the norm in front of a Linear is folded into
        # that Linear's columns (W[n][k] * g[k], in fp32, then one bf16 rounding); the 1/rms factor is applied
        # per row in the GEMM epilogue.  GRAM_FOLD_NORM=0 keeps the separate norm kernels (A/B, debugging).
        fold = os.environ.get("GRAM_FOLD_NORM", "1") != "0" or pieces > 1

        def lin(wname, gname, stage):
            w = sd[wname].to(dev, torch.float32)
            return b16(w * sd[gname].to(dev, torch.float32)[None, :], stage) if fold else b16(w, stage)

        def qkv(prefix, gname, stage):
            w = torch.cat([sd[prefix + ".q.weight"], sd[prefix + ".k.weight"], sd[prefix + ".v.weight"]], 0).to(dev, torch.float32)
            return b16(w * sd[gname].to(dev, torch.float32)[None, :], stage) if fold else b16(w, stage)

        # relative-bias tables: encoder [H][255] by (key - query + 127), decoder [H][GRAM_MAX_DEC_LEN] by distance
        nb, md = c.relative_attention_num_buckets, c.relative_attention_max_distance
        rel = torch.arange(-127, 128)
        enc_tab = sd["encoder.encoder.block.0.module.layer.0.SelfAttention.relative_attention_bias.weight"].cpu().float()
        enc_bias = enc_tab[relative_position_bucket(rel, True, nb, md)].t().contiguous()  # (H,255)
        dist = torch.arange(0, _lib.GRAM_MAX_DEC_LEN)
        dec_tab = sd["decoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight"].cpu().float()
        dec_bias = dec_tab[relative_position_bucket(-dist, False, nb, md)].t().contiguous()  # (H, GRAM_MAX_DEC_LEN)

        e = "encoder.encoder.block.{}.module.layer"
        dd = "decoder.block.{}.layer"
        ne, nd = self._n_enc, self._n_dec
        wkv_all = torch.cat(
            [torch.cat([sd[dd.format(i) + ".1.EncDecAttention.k.weight"], sd[dd.format(i) + ".1.EncDecAttention.v.weight"]], 0)
             for i in range(nd)], 0)
        inner_ = H * c.d_kv  # rows of wkv_all: per layer the K rows, then the V rows
        kv_row_caps = None
        if "bank_k" in caps or "bank_v" in caps:
            kv_row_caps = torch.tensor([caps.get("bank_k", 99), caps.get("bank_v", 99)]).repeat_interleave(inner_).repeat(nd)
        desc = _lib.ModelDesc(
            vocab=c.vocab_size, d_model=c.d_model, d_ff=c.d_ff, n_heads=H, n_enc_layers=ne, n_dec_layers=nd,
            max_passages=self.max_item_num + 1, tie_word_embeddings=int(bool(getattr(c, "tie_word_embeddings", True))),
            use_position_embedding=int(self.use_position_embedding), fold_norm=int(fold), eps=float(c.layer_norm_epsilon),
            embed_f32=f32(sd["shared.weight"]), lm_head_bf16=b16(sd["lm_head.weight"], "lm_head", no_scale=True),
            pos_emb_f32=f32(sd["position_embedding.weight"]) if self.use_position_embedding else None,
            enc_bias_f32=f32(enc_bias), dec_bias_f32=f32(dec_bias),
            enc_final_ln=f32(sd["encoder.encoder.final_layer_norm.weight"]),
            dec_final_ln=f32(sd["decoder.final_layer_norm.weight"]),
            enc_ln1=ptr_array([f32(sd[e.format(i) + ".0.layer_norm.weight"]) for i in range(ne)]),
            enc_wqkv=ptr_array([qkv(e.format(i) + ".0.SelfAttention", e.format(i) + ".0.layer_norm.weight", "enc_attn") for i in range(ne)]),
            enc_wo=ptr_array([b16(sd[e.format(i) + ".0.SelfAttention.o.weight"], "enc_attn") for i in range(ne)]),
            enc_ln2=ptr_array([f32(sd[e.format(i) + ".1.layer_norm.weight"]) for i in range(ne)]),
            enc_wi=ptr_array([lin(e.format(i) + ".1.DenseReluDense.wi.weight", e.format(i) + ".1.layer_norm.weight", "enc_ffn") for i in range(ne)]),
            enc_wo2=ptr_array([b16(sd[e.format(i) + ".1.DenseReluDense.wo.weight"], "enc_ffn") for i in range(ne)]),
            dec_ln1=ptr_array([f32(sd[dd.format(i) + ".0.layer_norm.weight"]) for i in range(nd)]),
            dec_wqkv=ptr_array([qkv(dd.format(i) + ".0.SelfAttention", dd.format(i) + ".0.layer_norm.weight", "dec_self") for i in range(nd)]),
            dec_wo=ptr_array([b16(sd[dd.format(i) + ".0.SelfAttention.o.weight"], "dec_self") for i in range(nd)]),
            dec_ln2=ptr_array([f32(sd[dd.format(i) + ".1.layer_norm.weight"]) for i in range(nd)]),
            dec_wq_x=ptr_array([lin(dd.format(i) + ".1.EncDecAttention.q.weight", dd.format(i) + ".1.layer_norm.weight", "dec_cross") for i in range(nd)]),
            dec_wo_x=ptr_array([b16(sd[dd.format(i) + ".1.EncDecAttention.o.weight"], "dec_cross") for i in range(nd)]),
            dec_ln3=ptr_array([f32(sd[dd.format(i) + ".2.layer_norm.weight"]) for i in range(nd)]),
            dec_wi=ptr_array([lin(dd.format(i) + ".2.DenseReluDense.wi.weight", dd.format(i) + ".2.layer_norm.weight", "dec_ffn") for i in range(nd)]),
            dec_wo2=ptr_array([b16(sd[dd.format(i) + ".2.DenseReluDense.wo.weight"], "dec_ffn") for i in range(nd)]),
            dec_wkv_x_all=b16(wkv_all, None, kv_row_caps),
            pieces=pieces, lm_head_f32=f32(sd["lm_head.weight"]) if pieces > 1 else None,
        )
        # b16() ran in the field order of the constructor call above: lm_head first, then the per-layer families, then wkv_all.
        # gram_model_desc_t.w_scales wants [enc_wqkv][enc_wo][enc_wi][enc_wo2][dec_wqkv][dec_wo][dec_wq_x][dec_wo_x][dec_wi][dec_wo2][wkv][lm_head]
        assert len(w_scales) == 4 * ne + 6 * nd + 2
        order = w_scales[1:] + w_scales[:1]
        arr = (C.c_float * len(order))(*order)
        keep.append(arr)
        desc.w_scales = C.cast(arr, C.POINTER(C.c_float))
        handle = lib.gram_model_create(C.byref(desc))
        if not handle:
            raise _lib.GramHipError("gram_model_create rejected the configuration (dims must be multiples of 128, heads <= 16)")
        self._packed = (dev, self._version, handle, keep)
        return handle

    def _get_workspace(self, handle, B, N, L, K, max_length) -> torch.Tensor:
        lib = _lib.load()
        need = lib.gram_workspace_bytes(handle, B, N, L, K, max_length)
        if need < 0:
            raise _lib.GramHipError(
                f"unsupported problem size B={B} N={N} L={L} K={K} max_length={max_length} "
                f"(N <= max_item_num+1, L <= 128, K <= 64, max_length <= {_lib.GRAM_MAX_DEC_LEN})"
            )
        dev = self._device()
        if self._workspace is None or self._workspace.device != dev or self._workspace.numel() < need:
            self._workspace = None
            torch.cuda.empty_cache()  # hand the old workspace (and whatever else the caching allocator holds unused) back first:
            #                           the new one is most of the HBM, and the allocator does not always manage that by itself
            self._workspace = torch.empty(int(need), dtype=torch.uint8, device=dev)
        return self._workspace

    def max_users_per_call(self, N: int, L: int, K: int, max_length: int, limit: int = 4096, headroom: float = 0.9) -> int:
        """Largest batch B <= limit whose ``generate`` workspace (gram_workspace_bytes) fits the device's free HBM: what the runners
        size their GPU batches with, whatever ``--eval_batch_size`` the loader was built with (results do not depend on the batch a
        user is scored in).  Memory held by this model's current workspace counts as free (it is re-used or replaced)."""
        handle = self._pack()
        lib = _lib.load()
        dev = self._device()
        Lp = (int(L) + 31) // 32 * 32
        torch.cuda.empty_cache()  # unused blocks of PyTorch's caching allocator go back to the driver first (once per evaluation)
        free, _total = torch.cuda.mem_get_info(dev)
        if self._workspace is not None and self._workspace.device == dev:
            free += self._workspace.numel()
        budget = int(free * headroom)
        lo, hi = 1, max(1, int(limit))
        if lib.gram_workspace_bytes(handle, hi, N, Lp, K, max_length) <= budget:
            return hi
        while lo < hi:  # workspace bytes grow monotonically with B
            mid = (lo + hi + 1) // 2
            need = lib.gram_workspace_bytes(handle, mid, N, Lp, K, max_length)
            if 0 <= need <= budget:
                lo = mid
            else:
                hi = mid - 1
        return lo

    @staticmethod
    def _closure_trie(fn: Callable):
        for cell in getattr(fn, "__closure__", None) or ():
            obj = cell.cell_contents
            if isinstance(obj, Trie):  # (checked first: reading `trie_dict` of a gram_amd Trie builds its nested dict)
                return obj
            if hasattr(obj, "trie_dict") and hasattr(obj, "get"):  # the reference's own utils.generation_trie.Trie
                return obj
        return None

    def _flat_trie(self, fn: Callable) -> FlatTrie:
        """Device CSR of the closure's Trie, cached per Trie OBJECT.  The entry holds a strong reference to its
        source, so the id cannot be recycled by a later Trie while the entry is alive (the runner builds a fresh
        Trie per evaluation), and the item count guards against in-place ``add`` calls."""
        trie = self._closure_trie(fn)
        cached = self._tries.get(id(trie))
        if cached is None or cached[0] is not trie or cached[1].n_sequences != len(trie):
            cached = (trie, FlatTrie(trie))
            self._tries = {id(trie): cached}  # one live Trie per eval; drop stale ones
        return cached[1]

    # ------------------------------------------------------------------ which passages the encoder runs on
    _CACHE_L = _lib.GRAM_MAX_PASSAGE_LEN
    _KEY_MULT: Dict[torch.device, torch.Tensor] = {}

    @staticmethod
    def _canonical(rows_ids: torch.Tensor, rows_mask: torch.Tensor) -> torch.Tensor:
        """(P, 128) int32: token id where valid, -1 elsewhere -- the identity of a passage (position-wise)."""
        canon = torch.where(rows_mask.bool(), rows_ids, rows_ids.new_full((), -1)).to(torch.int32)
        return torch.nn.functional.pad(canon, (0, GRAM._CACHE_L - canon.shape[1]), value=-1)

    @staticmethod
    def _passage_keys(canon: torch.Tensor) -> torch.Tensor:
        # 64-bit key, overflow-free ((id+1) < 2^15, multiplier < 2^31, 128 terms); a key match is always
        # confirmed against the stored tokens, so a collision costs a miss, never a wrong hit
        mult = GRAM._KEY_MULT.get(canon.device)
        if mult is None:
            g = torch.Generator().manual_seed(0x6772616D)
            mult = torch.randint(1, 2 ** 31 - 1, (GRAM._CACHE_L,), generator=g, dtype=torch.int64).to(canon.device)
            GRAM._KEY_MULT[canon.device] = mult
        return ((canon.to(torch.int64) + 1) * mult).sum(dim=1)

    # The passage cache: x f32 [capacity][128][d] (the encoder's residual stream before the final norm), canon i32 [capacity][128]
    # (the tokens: the identity of a passage), n rows in use, keys (sorted) / perm over the rows in use.
    def _pcache_rows(self, n_new: int):
        """Make room for n_new more passages; returns (cache dict, first new row).  The buffers grow geometrically (a dataset's item
        prompts arrive over several batches; `reserve_passage_cache` sizes them once)."""
        dev, d = self._device(), self.config.d_model
        pc = self._pcache
        used = 0 if pc is None else pc["n"]
        cap = 0 if pc is None else pc["x"].shape[0]
        if used + n_new > cap:
            new_cap = max(used + n_new, 2 * cap, int(getattr(self, "_pcache_reserve", 0)))
            x = torch.empty(new_cap, self._CACHE_L, d, dtype=torch.float32, device=dev)
            canon = torch.full((new_cap, self._CACHE_L), -1, dtype=torch.int32, device=dev)
            if used:
                x[:used].copy_(pc["x"][:used])
                canon[:used].copy_(pc["canon"][:used])
            pc = dict(x=x, canon=canon, n=used, keys=None if pc is None else pc["keys"], perm=None if pc is None else pc["perm"])
            self._pcache = pc
        return pc, used

    def _pcache_commit(self, pc, n_total: int) -> None:
        keys = self._passage_keys(pc["canon"][:n_total])
        pc["keys"], pc["perm"] = torch.sort(keys, stable=True)
        pc["n"] = n_total

    def reserve_passage_cache(self, n_passages: int) -> None:
        """Capacity hint: the number of distinct passages that will be cached (a dataset's item count)."""
        self._pcache_reserve = int(n_passages)

    def set_passage_harvest(self, on: bool = True, first_slot: int = 1) -> None:
        """With harvesting on, every passage a ``generate`` call had to encode in a slot >= first_slot joins the passage cache when
        the call returns -- its residual-stream rows are still in the workspace (gram_workspace_encoder_x_offset), nothing is
        encoded twice -- so an evaluation needs no separate cache-fill pass: a batch pays for the item prompts it is the first to
        see and the later ones find them.  Slot 0 is the user's own prompt (test_dataset_gram.py:203-210), never seen again.
        Results do not change (the cache is result-neutral bit for bit)."""
        self._harvest = bool(on)
        self._harvest_first_slot = int(first_slot)

    @torch.no_grad()
    def cache_passages(self, input_ids, attention_mask, chunk: int = 1024) -> int:
        """Pre-encode user-independent passages.

        A GRAM input is passage 0 = the user prompt plus up to ``max_item_num`` ITEM prompts
        (test_dataset_gram.py:115-123,203-210); EncoderWrapper.forward (gram.py:200-256) encodes every passage on
        its own and adds the slot's position embedding afterwards, so an item prompt's encoder states are the same
        for every user and every slot.  Passages registered here (``(P, L)`` or ``(B, N, L)`` ids + mask, e.g. all
        item prompts of the dataset) are encoded once; ``generate`` then recognises them by their tokens and runs the
        encoder only on the rest.  Results are bit-identical with and without the cache.  The cache holds the fp32
        residual stream (128 x d_model x 4 B per passage) on the device and is dropped whenever the weights change.
        Returns the number of passages cached so far."""
        handle = self._pack()
        lib = _lib.load()
        dev = self._device()
        ids = input_ids.to(dev, torch.int64).reshape(-1, input_ids.shape[-1])
        mask = attention_mask.to(dev).reshape(-1, attention_mask.shape[-1]).ne(0)
        if ids.shape != mask.shape or ids.shape[1] > self._CACHE_L:
            raise ValueError(f"passages must be (P, L <= {self._CACHE_L}) ids with a mask of the same shape")
        keep = mask.any(dim=1)
        canon = self._canonical(ids[keep], mask[keep])
        canon = canon[self._first_unseen(canon)]
        n_new = int(canon.shape[0])
        if n_new == 0:
            return 0 if self._pcache is None else int(self._pcache["n"])
        full_ids = canon.clamp(min=0).to(torch.int64).contiguous()
        full_mask = canon.ge(0).view(torch.uint8).contiguous()
        pc, base = self._pcache_rows(n_new)
        pc["canon"][base:base + n_new] = canon
        stream = torch.cuda.current_stream(dev).cuda_stream
        with torch.cuda.device(dev):
            for lo in range(0, n_new, chunk):
                n = min(chunk, n_new - lo)
                ws = self._get_workspace(handle, n, 1, self._CACHE_L, 1, 2)
                _lib.check(lib.gram_encode_passages(handle, full_ids[lo:lo + n].data_ptr(), full_mask[lo:lo + n].data_ptr(), n,
                                                    self._CACHE_L, ws.data_ptr(), ws.numel(), pc["x"][base + lo:base + lo + n].data_ptr(),
                                                    stream), "gram_encode_passages")
        self._pcache_commit(pc, base + n_new)
        return base + n_new

    def clear_passage_cache(self) -> None:
        self._pcache = None

    def _first_unseen(self, canon: torch.Tensor) -> torch.Tensor:
        """bool [P]: the first occurrence of every passage of `canon` that the cache does not hold yet."""
        keys = self._passage_keys(canon)
        order = torch.argsort(keys, stable=True)
        first = torch.ones_like(keys, dtype=torch.bool)
        first[order[1:]] = keys[order[1:]] != keys[order[:-1]]
        if self._pcache is not None and self._pcache["n"] > 0:
            first &= self._lookup(canon, keys)[0].logical_not()
        return first

    def _lookup(self, canon: torch.Tensor, keys: torch.Tensor):
        pc = self._pcache
        pos = torch.searchsorted(pc["keys"], keys).clamp(max=pc["keys"].numel() - 1)
        slot = pc["perm"][pos]
        hit = (pc["keys"][pos] == keys) & (pc["canon"].index_select(0, slot) == canon).all(dim=1)
        return hit, slot

    def _plan_encoder(self, ids, mask, B, N, Lp):
        """gram_compaction_t for this batch as (struct, keep-alive list, tensors), or None when the encoder simply runs on all B*N passages."""
        return self._plan(ids, mask, B, N, Lp)[0]

    def _plan(self, ids, mask, B, N, Lp):
        """(gram_compaction_t for this batch -- or None when the encoder simply runs on all B*N passages --, harvest plan or None).

        Ragged batches: the Collator pads every user to the batch's largest passage count with fully masked passages
        (Collator.py:410-436); the encoder skips them.  Passages found in the passage cache skip it too.  One small
        D2H sync for the counts; the gathers are input plumbing.  GRAM_COMPACT=0 disables both.
        Harvest plan (set_passage_harvest): (compact rows of the passages to add to the cache after the call, their tokens)."""
        harvest = bool(getattr(self, "_harvest", False)) and N > int(getattr(self, "_harvest_first_slot", 1))
        have = self._pcache is not None and self._pcache["n"] > 0
        if os.environ.get("GRAM_COMPACT", "1") == "0" or (N == 1 and not have):
            return None, None
        rows_ids, rows_mask = ids.view(B * N, Lp), mask.view(B * N, Lp)
        active = rows_mask.ne(0).any(dim=1)
        canon = self._canonical(rows_ids, rows_mask) if (have or harvest) else None
        if have:
            hit, slot = self._lookup(canon, self._passage_keys(canon))
            hit &= active
        else:
            hit = torch.zeros_like(active)
        miss = active & ~hit
        n_miss, n_hit, users = torch.stack([miss.sum(), hit.sum(), active.view(B, N).any(dim=1).sum()]).tolist()
        if users < B:
            raise ValueError("every user needs at least one passage with a valid token")
        plan = None
        if harvest and n_miss:
            # misses in slots >= first_slot, one per distinct passage; compact row of flat passage f = its rank among the misses
            slot_ok = (torch.arange(B * N, device=ids.device) % N) >= self._harvest_first_slot
            cand = (miss & slot_ok).nonzero().squeeze(1)
            if cand.numel():
                c_canon = canon.index_select(0, cand)
                cand = cand[self._first_unseen(c_canon)]
                rank = torch.cumsum(miss.to(torch.int64), 0) - 1
                plan = (rank.index_select(0, cand), canon.index_select(0, cand))
        if n_hit == 0 and n_miss == B * N:
            return None, plan
        midx, hidx = miss.nonzero().squeeze(1), hit.nonzero().squeeze(1)
        c_ids = rows_ids.index_select(0, midx).contiguous()
        c_mask = rows_mask.index_select(0, midx).contiguous()
        c_map = torch.cat([midx, hidx]).to(torch.int32).contiguous()
        keep = [c_ids, c_mask, c_map]
        comp = _lib.Compaction(n_miss + n_hit, c_map.data_ptr(), c_ids.data_ptr(), c_mask.data_ptr(), 0, 0, None, None)
        tens = dict(comp_map=c_map, comp_ids=c_ids, comp_mask=c_mask, cache_slot=None, cache_x=None, n_cached=0, cache_L=0)
        if n_hit:
            slots = slot.index_select(0, hidx).to(torch.int32).contiguous()
            keep += [slots, self._pcache["x"]]
            comp.n_cached, comp.cache_L = n_hit, self._CACHE_L
            comp.cache_x, comp.cache_slot = self._pcache["x"].data_ptr(), slots.data_ptr()
            tens.update(cache_slot=slots, cache_x=self._pcache["x"], n_cached=n_hit, cache_L=self._CACHE_L)
        return (comp, keep, tens), plan

    def _harvest_from_workspace(self, plan, ws, handle, B, N, Lp, K, max_length) -> None:
        """Add the planned passages to the cache from the residual stream the generate() call left in the workspace."""
        rows, canon = plan
        n_new = int(rows.numel())
        if n_new == 0:
            return
        d = self.config.d_model
        off = _lib.load().gram_workspace_encoder_x_offset(handle, B, N, Lp, K, max_length)
        if off < 0:
            raise _lib.GramHipError("gram_workspace_encoder_x_offset rejected the shape")
        wx = ws[off: off + B * N * Lp * d * 4].view(torch.float32).view(B * N, Lp, d)
        pc, base = self._pcache_rows(n_new)
        dst = pc["x"][base:base + n_new]
        torch.index_select(wx, 0, rows, out=dst[:, :Lp])
        if Lp < self._CACHE_L:
            dst[:, Lp:].zero_()  # positions past the batch's L are padding for these passages (gram_compaction_t.cache_L semantics)
        pc["canon"][base:base + n_new] = canon
        self._pcache_commit(pc, base + n_new)

    # ------------------------------------------------------------------ the hot path
    @torch.no_grad()
    def generate(self, input_ids, attention_mask, max_length, prefix_allowed_tokens_fn=None, num_beams=1,
                 num_return_sequences=None, output_scores=True, return_dict_in_generate=True, length_penalty=1.0,
                 **unused):
        """GRAM.generate (gram.py:74-107) with the kwargs single_runner_gram.py:641-651 passes.

        input_ids (B,N,L) int64, attention_mask (B,N,L) bool on the model's device.  Returns a
        mapping with ``sequences`` (B*num_return_sequences, T) int64 -- user-major, best first,
        0-padded, starting with the decoder start token -- and ``sequences_scores`` (fp32)."""
        if prefix_allowed_tokens_fn is None:
            raise NotImplementedError("unconstrained generation is not on GRAM's scoring path (a Trie is always passed)")
        # HF kwargs that would change the search: refuse the ones this path does not implement instead of ignoring them
        neutral = {"do_sample": False, "early_stopping": False, "num_beam_groups": 1, "repetition_penalty": 1.0,
                   "no_repeat_ngram_size": 0, "diversity_penalty": 0.0, "use_cache": True, "temperature": 1.0, "top_k": 50,
                   "top_p": 1.0, "min_length": 0, "pad_token_id": 0, "eos_token_id": 1, "decoder_start_token_id": 0}
        for k, v in unused.items():
            if k in neutral:
                if v is not None and v != neutral[k] and k not in ("temperature", "top_k", "top_p"):
                    raise NotImplementedError(f"generate({k}={v!r}) is not supported by the HIP scoring path "
                                              f"(only {k}={neutral[k]!r}, the value the GRAM runners use)")
            elif v is not None:
                import warnings
                warnings.warn(f"gram_amd.GRAM.generate ignores the keyword argument {k!r}")
        if input_ids.dim() != 3:
            raise ValueError("input_ids must be (B, N, L)")
        handle = self._pack()
        lib = _lib.load()
        dev = self._device()
        K = int(num_beams)
        nret = int(num_return_sequences or K)
        B, N, L = input_ids.shape
        ids = input_ids.to(dev, torch.int64)
        mask = attention_mask.to(dev).ne(0).view(torch.uint8) if attention_mask.dtype != torch.bool else attention_mask.to(dev).view(torch.uint8)
        Lp = (L + 31) // 32 * 32  # masked padding is invisible to attention: pad L to the kernel tile
        if Lp != L:
            ids = torch.nn.functional.pad(ids, (0, Lp - L))
            mask = torch.nn.functional.pad(mask, (0, Lp - L))
        ids, mask = ids.contiguous(), mask.contiguous()
        if self._closure_trie(prefix_allowed_tokens_fn) is None:
            if K == 1:
                raise NotImplementedError("greedy search needs the Trie closure form of prefix_allowed_tokens_fn")
            return self._generate_with_callback(ids, mask, B, N, Lp, K, nret, int(max_length), float(length_penalty),
                                                prefix_allowed_tokens_fn, return_dict_in_generate)
        comp, harvest_plan = self._plan(ids, mask, B, N, Lp)
        flat = self._flat_trie(prefix_allowed_tokens_fn)
        _ctrie, (t_off, t_tok, t_node) = flat.to_device(dev)
        ws = self._get_workspace(handle, B, N, Lp, K, int(max_length))
        if K == 1 and nret != 1:
            raise ValueError("num_return_sequences must be 1 for greedy search (num_beams == 1), as in HF generate")
        ct = comp[2] if comp else dict(comp_map=None, comp_ids=None, comp_mask=None, cache_slot=None, cache_x=None, n_cached=0, cache_L=0)
        # the whole path is ONE PyTorch-ROCm custom op over the C ABI (gram_amd/ops.py -> gram_generate_ex)
        from .. import ops as _ops  # noqa: F401  (registers torch.ops.gram.*)
        seqs, scores, width = torch.ops.gram.generate(
            ids, mask, int(handle), ws, t_off, t_tok, t_node, int(flat.max_fanout), int(flat.min_seq_len), K, nret, int(max_length),
            float(length_penalty), ct["comp_map"], ct["comp_ids"], ct["comp_mask"], ct["cache_slot"], ct["cache_x"],
            int(ct["n_cached"]), int(ct["cache_L"]))
        if harvest_plan is not None:
            self._harvest_from_workspace(harvest_plan, ws, handle, B, N, Lp, K, int(max_length))
        seqs = seqs[:, : int(width[0])]
        scores = scores if K > 1 else None  # num_beams == 1 is HF's greedy_search: it has no sequences_scores
        if not return_dict_in_generate:
            return seqs
        return GenerateOutput(sequences=seqs, sequences_scores=scores)

    @torch.no_grad()
    def sequence_items(self, sequences, prefix_allowed_tokens_fn, candidates):
        """Which candidate each row of ``generate(...)["sequences"]`` spells: int32 (rows,) indices into ``candidates`` -- the
        token-id sequences the closure's Trie was built from, in the caller's order -- or -1 for a row that is not a candidate
        (HF's -inf filler beams).  Every hypothesis of a Trie-constrained search is a leaf of the Trie, so the runner needs one
        ``batch_decode`` of the candidate list per evaluation instead of one of B*K generated rows per batch
        (single_runner_gram.py:657-662); ``gram_trie_item_index`` walks the flat Trie on the device."""
        flat = self._flat_trie(prefix_allowed_tokens_fn)
        dev = self._device()
        ctrie, _keep = flat.to_device(dev)
        node_item = flat.node_items_on(dev, candidates)
        seqs = sequences.to(dev, torch.int64).contiguous()
        out = torch.empty(seqs.shape[0], dtype=torch.int32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(_lib.load().gram_trie_item_index(C.byref(ctrie), node_item.data_ptr(), seqs.data_ptr(), seqs.shape[0], seqs.shape[1],
                                                        out.data_ptr(), torch.cuda.current_stream(dev).cuda_stream), "gram_trie_item_index")
        return out

    def _generate_with_callback(self, ids, mask, B, N, Lp, K, nret, max_length, length_penalty, fn, return_dict):
        """Slow path for an ARBITRARY ``prefix_allowed_tokens_fn(batch_id, sent) -> List[int]`` (HF's
        PrefixConstrainedLogitsProcessor contract): semantics preserved, one host round trip per step like the
        reference (generation_trie.py:89-95).  The model still runs on the device through the same C-ABI entry
        points (encode, decode step, LSE, beam step, finalize); only the allowed-token lists come from Python,
        uploaded each step as a one-level CSR in which row r owns node r + 1."""
        lib = _lib.load()
        dev = self._device()
        handle = self._pack()
        ws = self._get_workspace(handle, B, N, Lp, K, max_length)
        stream = torch.cuda.current_stream(dev).cuda_stream
        R, V = B * K, self.config.vocab_size
        i32 = dict(dtype=torch.int32, device=dev)
        t = dict(tokens=torch.zeros(R, **i32), node=torch.zeros(R, **i32), beam_scores=torch.zeros(R, dtype=torch.float32, device=dev),
                 seq=torch.zeros(R, max_length, **i32), anc=torch.zeros(max_length, R, **i32), done=torch.zeros(B, **i32),
                 n_hyps=torch.zeros(B, **i32), hyp_score=torch.zeros(B, K + 1, dtype=torch.float64, device=dev),
                 worst=torch.zeros(B, dtype=torch.float64, device=dev), hyp_len=torch.zeros(B, K + 1, **i32),
                 hyp_tok=torch.zeros(B, K + 1, max_length, **i32), error=torch.zeros(4, **i32))
        st = _lib.BeamState(B=B, K=K, Tmax=max_length, length_penalty=length_penalty, eos=1, pad=0,
                            **{k: v.data_ptr() for k, v in t.items()})
        logits = torch.empty(R, V, dtype=torch.float32, device=dev)
        lse = torch.empty(R, dtype=torch.float32, device=dev)

        def csr(lists):
            off = [0, 0]
            toks = []
            for l in lists:
                toks += sorted(set(int(x) for x in l))
                off.append(len(toks))
            fan = max((off[i + 1] - off[i] for i in range(1, len(off) - 1)), default=0)
            if K * max(fan, 1) > 16384:
                raise _lib.GramHipError("prefix_allowed_tokens_fn returned too many tokens for the on-chip candidate sort")
            a = torch.tensor(off, **i32)
            b = torch.tensor(toks or [0], **i32)
            c = torch.full((max(len(toks), 1),), -1, **i32)
            return _lib.Trie(a.data_ptr(), b.data_ptr(), c.data_ptr(), len(off) - 1, len(toks), max(fan, 1)), (a, b, c)

        with torch.cuda.device(dev):
            _lib.check(lib.gram_encode_fused(handle, ids.data_ptr(), mask.data_ptr(), B, N, Lp, ws.data_ptr(), ws.numel(), K,
                                             max_length, None, stream), "gram_encode_fused")
            start_trie, keep = csr([[0]])  # root -> start token, so that gram_beam_init finds node 1 for every row
            start_trie.n_nodes = 2
            _lib.check(lib.gram_beam_init(C.byref(st), C.byref(start_trie), 0, stream), "gram_beam_init")
            for step in range(max_length - 1):
                sent = t["seq"][:, : step + 1].cpu()
                lists = [fn(r // K, sent[r]) for r in range(R)]  # the reference's per-beam callback
                step_trie, keep = csr(lists)
                t["node"].copy_(torch.arange(1, R + 1, **i32))
                _lib.check(lib.gram_decode_step(handle, t["tokens"].data_ptr(), t["anc"].data_ptr(), mask.data_ptr(), B, N, Lp, K,
                                                max_length, step, ws.data_ptr(), ws.numel(), logits.data_ptr(), stream),
                           "gram_decode_step")
                _lib.check(lib.gram_row_lse(logits.data_ptr(), lse.data_ptr(), R, V, stream), "gram_row_lse")
                _lib.check(lib.gram_beam_step(C.byref(st), C.byref(step_trie), logits.data_ptr(), lse.data_ptr(), V, step + 1, K,
                                              stream), "gram_beam_step")
                torch.cuda.current_stream(dev).synchronize()  # the CSR tensors must outlive the launch
            seqs = torch.empty(B * nret, max_length, dtype=torch.int64, device=dev)
            scores = torch.empty(B * nret, dtype=torch.float32, device=dev)
            width = torch.zeros(4, **i32)
            _lib.check(lib.gram_beam_finalize(C.byref(st), nret, max_length, seqs.data_ptr(), scores.data_ptr(), width.data_ptr(),
                                              stream), "gram_beam_finalize")
            torch.cuda.current_stream(dev).synchronize()
        if int(t["error"][0]) != 0:
            _lib.check(_lib.E_BEAM, "beam search")
        seqs = seqs[:, : int(width[0])]
        return GenerateOutput(sequences=seqs, sequences_scores=scores) if return_dict else seqs

    def __del__(self):
        try:
            if self._packed is not None:
                _lib.load().gram_model_destroy(self._packed[2])
        except Exception:
            pass
