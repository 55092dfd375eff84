"""CPU oracle for GRAM's multi-granular late-fusion generative scoring path.

TEST INFRASTRUCTURE ONLY.  This file is a CPU restatement (torch-CPU fp32 for the
floating-point stack, plain Python for the integer/bookkeeping parts) of the reference
algorithm.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and only as the checker / reported baseline.  The product path
(``gram_amd``) never imports anything from ``oracle/``.

Pinning status (see DESIGN.md "Oracle"):
  * encoder / late fusion / decoder step / Trie / metrics: pinned against the reference itself,
    imported in the build container by ``oracle/make_golden.py`` -> ``tests/golden/*.npz``.
  * beam search (HF ``transformers==4.26.0`` semantics, third-party, source absent from
    /root/reference): restated from the published 4.26.0 algorithm; pinned against the
    installed transformers-5.15 search driving the reference model on Tries where both
    searches provably coincide, plus hand-worked variable-length cases.  The 4.26-only
    early-stop heuristic itself is "parity unpinned".

Every function cites the reference file:line (under /root/reference/) it follows.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

Tensor = torch.Tensor
FMIN = torch.finfo(torch.float32).min


# --------------------------------------------------------------------------------------
# configuration + synthetic weights
# --------------------------------------------------------------------------------------
@dataclass
class OracleConfig:
    """Hyper-parameter contract of src/model/gram_t5_config.py:85-143 plus the four GRAM
    attributes main_generative_gram.py:67-70 adds."""

    vocab_size: int = 32128
    d_model: int = 512
    d_kv: int = 64
    d_ff: int = 2048
    num_layers: int = 6
    num_decoder_layers: int = 6
    num_heads: int = 8
    relative_attention_num_buckets: int = 32
    relative_attention_max_distance: int = 128
    layer_norm_epsilon: float = 1e-6
    tie_word_embeddings: bool = True
    pad_token_id: int = 0
    eos_token_id: int = 1
    decoder_start_token_id: int = 0
    max_item_num: int = 20
    use_position_embedding: bool = True

    @staticmethod
    def named(name: str, **kw) -> "OracleConfig":
        table = {
            "t5-small": dict(d_model=512, d_ff=2048, num_layers=6, num_decoder_layers=6, num_heads=8),
            "t5-base": dict(d_model=768, d_ff=3072, num_layers=12, num_decoder_layers=12, num_heads=12),
            "t5-large": dict(d_model=1024, d_ff=4096, num_layers=24, num_decoder_layers=24, num_heads=16),
        }
        base = dict(table[name])
        base.update(kw)
        return OracleConfig(**base)


def init_state_dict(cfg: OracleConfig, seed: int = 2023) -> Dict[str, Tensor]:
    """Synthetic weights with the key layout SURVEY.md §3.4 records for the reference
    checkpoint and the distributions of gram_t5_modeling.py:865-929 / gram.py:32-33.
    (Own generator order: this is NOT bit-identical to the reference's init; parity tests
    always load the same dict into both sides.)"""
    g = torch.Generator().manual_seed(seed)
    d, dk, H, F, V = cfg.d_model, cfg.d_kv, cfg.num_heads, cfg.d_ff, cfg.vocab_size
    inner = H * dk

    def nrm(shape, std):
        return torch.randn(shape, generator=g, dtype=torch.float32) * std

    sd: Dict[str, Tensor] = {}
    sd["shared.weight"] = nrm((V, d), 1.0)

    def attn(prefix, rel_bias):
        sd[prefix + ".q.weight"] = nrm((inner, d), (d * dk) ** -0.5)
        sd[prefix + ".k.weight"] = nrm((inner, d), d ** -0.5)
        sd[prefix + ".v.weight"] = nrm((inner, d), d ** -0.5)
        sd[prefix + ".o.weight"] = nrm((d, inner), inner ** -0.5)
        if rel_bias:
            sd[prefix + ".relative_attention_bias.weight"] = nrm(
                (cfg.relative_attention_num_buckets, H), d ** -0.5
            )

    def ff(prefix):
        sd[prefix + ".DenseReluDense.wi.weight"] = nrm((F, d), d ** -0.5)
        sd[prefix + ".DenseReluDense.wo.weight"] = nrm((d, F), F ** -0.5)

    # random (not all-ones) norm gains so that a dropped/misplaced gain is caught
    def gain():
        return 1.0 + 0.1 * torch.randn(d, generator=g, dtype=torch.float32)

    for i in range(cfg.num_layers):
        p = f"encoder.encoder.block.{i}.module.layer"
        attn(p + ".0.SelfAttention", i == 0)
        sd[p + ".0.layer_norm.weight"] = gain()
        ff(p + ".1")
        sd[p + ".1.layer_norm.weight"] = gain()
    sd["encoder.encoder.final_layer_norm.weight"] = gain()
    for i in range(cfg.num_decoder_layers):
        p = f"decoder.block.{i}.layer"
        attn(p + ".0.SelfAttention", i == 0)
        sd[p + ".0.layer_norm.weight"] = gain()
        attn(p + ".1.EncDecAttention", False)
        sd[p + ".1.layer_norm.weight"] = gain()
        ff(p + ".2")
        sd[p + ".2.layer_norm.weight"] = gain()
    sd["decoder.final_layer_norm.weight"] = gain()
    sd["position_embedding.weight"] = nrm((cfg.max_item_num + 1, d), 0.02)
    # aliases the reference state_dict also carries (same tensors, several names)
    sd["encoder.encoder.embed_tokens.weight"] = sd["shared.weight"]
    sd["decoder.embed_tokens.weight"] = sd["shared.weight"]
    sd["encoder.position_embedding.weight"] = sd["position_embedding.weight"]
    sd["lm_head.weight"] = sd["shared.weight"] if cfg.tie_word_embeddings else nrm((V, d), 1.0)
    return sd


# --------------------------------------------------------------------------------------
# T5 building blocks
# --------------------------------------------------------------------------------------
def rms_norm(x: Tensor, weight: Tensor, eps: float) -> Tensor:
    """T5LayerNorm.forward, gram_t5_modeling.py:262-276 (fp32 variance, no mean, no bias)."""
    var = x.to(torch.float32).pow(2).mean(-1, keepdim=True)
    return weight * (x * torch.rsqrt(var + eps))


def relative_position_bucket(rel: Tensor, bidirectional: bool, num_buckets: int, max_distance: int) -> Tensor:
    """T5Attention._relative_position_bucket, gram_t5_modeling.py:398-450.
    ``rel`` = memory_position - query_position (int64)."""
    buckets = torch.zeros_like(rel)
    if bidirectional:
        num_buckets //= 2
        buckets = buckets + (rel > 0).to(torch.long) * num_buckets
        rel = rel.abs()
    else:
        rel = -torch.minimum(rel, torch.zeros_like(rel))
    max_exact = num_buckets // 2
    small = rel < max_exact
    large = max_exact + (
        torch.log(rel.float() / max_exact) / math.log(max_distance / max_exact) * (num_buckets - max_exact)
    ).to(torch.long)
    large = torch.minimum(large, torch.full_like(large, num_buckets - 1))
    return buckets + torch.where(small, rel, large)


def position_bias(table: Tensor, q_len: int, k_len: int, bidirectional: bool, cfg: OracleConfig) -> Tensor:
    """T5Attention.compute_bias, gram_t5_modeling.py:452-477 -> (1, H, q_len, k_len)."""
    ctx = torch.arange(q_len, dtype=torch.long, device=table.device)[:, None]
    mem = torch.arange(k_len, dtype=torch.long, device=table.device)[None, :]
    bucket = relative_position_bucket(
        mem - ctx, bidirectional, cfg.relative_attention_num_buckets, cfg.relative_attention_max_distance
    )
    return table[bucket].permute(2, 0, 1).unsqueeze(0)


def _heads(x: Tensor, H: int, dk: int) -> Tensor:
    b, s, _ = x.shape
    return x.view(b, s, H, dk).transpose(1, 2)  # (b,H,s,dk)  gram_t5_modeling.py:513-517


def _attend(q: Tensor, k: Tensor, v: Tensor, bias: Tensor) -> Tensor:
    """scores = q k^T (UNSCALED) + bias ; fp32 softmax ; @ v ; merge heads.
    gram_t5_modeling.py:572-621."""
    scores = torch.matmul(q, k.transpose(3, 2)) + bias
    w = torch.softmax(scores.float(), dim=-1)
    o = torch.matmul(w, v)  # (b,H,q,dk)
    b, H, ql, dk = o.shape
    return o.transpose(1, 2).contiguous().view(b, ql, H * dk)


def encoder_forward(sd: Dict[str, Tensor], cfg: OracleConfig, ids: Tensor, mask: Tensor) -> Tensor:
    """T5Stack.forward in the encoder role, gram_t5_modeling.py:1037-1296, on a flat
    (P, L) batch of passages.  mask: (P, L) bool.  Returns (P, L, d) after final_layer_norm."""
    H, dk = cfg.num_heads, cfg.d_kv
    P, L = ids.shape
    x = sd["shared.weight"][ids]  # :1091
    ext = (1.0 - mask.to(torch.float32))[:, None, None, :] * FMIN  # :1130-1132
    bias = (
        position_bias(
            sd["encoder.encoder.block.0.module.layer.0.SelfAttention.relative_attention_bias.weight"],
            L, L, True, cfg,
        )
        + ext
    )  # layer-0 table shared by all layers :1246-1249 ; mask folded in :595-598
    for i in range(cfg.num_layers):
        p = f"encoder.encoder.block.{i}.module.layer"
        h = rms_norm(x, sd[p + ".0.layer_norm.weight"], cfg.layer_norm_epsilon)  # :653
        a = p + ".0.SelfAttention"
        q = _heads(h @ sd[a + ".q.weight"].T, H, dk)
        k = _heads(h @ sd[a + ".k.weight"].T, H, dk)
        v = _heads(h @ sd[a + ".v.weight"].T, H, dk)
        x = x + _attend(q, k, v, bias) @ sd[a + ".o.weight"].T  # :622, :663
        h = rms_norm(x, sd[p + ".1.layer_norm.weight"], cfg.layer_norm_epsilon)  # :349
        f = torch.relu(h @ sd[p + ".1.DenseReluDense.wi.weight"].T) @ sd[p + ".1.DenseReluDense.wo.weight"].T
        x = x + f  # :351
    return rms_norm(x, sd["encoder.encoder.final_layer_norm.weight"], cfg.layer_norm_epsilon)  # :1271


def late_fusion(sd: Dict[str, Tensor], cfg: OracleConfig, hidden: Tensor, B: int, N: int) -> Tensor:
    """EncoderWrapper.forward tail, gram.py:238-255: add the per-passage position embedding
    (broadcast over L) and view the (B*N, L, d) states as the fused (B, N*L, d) bank."""
    P, L, d = hidden.shape
    assert P == B * N
    if cfg.use_position_embedding:
        pos = sd["position_embedding.weight"][torch.arange(N, device=hidden.device)]  # (N,d)
        hidden = hidden + pos.repeat(B, 1).view(B * N, 1, d)
    return hidden.view(B, N * L, d)


def encode_fused(sd, cfg: OracleConfig, input_ids: Tensor, attention_mask: Tensor) -> Tensor:
    """GRAM.generate's encoder call, gram.py:74-91: (B,N,L) -> (B, N*L, d)."""
    B, N, L = input_ids.shape
    h = encoder_forward(sd, cfg, input_ids.reshape(B * N, L), attention_mask.reshape(B * N, L).bool())
    return late_fusion(sd, cfg, h, B, N)


def cross_kv(sd, cfg: OracleConfig, enc: Tensor) -> List[Tuple[Tensor, Tensor]]:
    """Cross-attention K/V projection of the fused bank, gram_t5_modeling.py:531-534, once
    per layer.  enc: (X, S, d) where X is B (shared-bank mode) or B*K (reference-faithful)."""
    out = []
    for i in range(cfg.num_decoder_layers):
        a = f"decoder.block.{i}.layer.1.EncDecAttention"
        out.append(
            (_heads(enc @ sd[a + ".k.weight"].T, cfg.num_heads, cfg.d_kv),
             _heads(enc @ sd[a + ".v.weight"].T, cfg.num_heads, cfg.d_kv))
        )
    return out


@dataclass
class DecodeState:
    """Decoder caches for R = B*K rows.  self_k/self_v: per layer (R,H,t,dk) tensors (the
    reference's tuple cache, gram_t5_modeling.py:536-540); cross: per layer (X,H,S,dk)."""

    cross: List[Tuple[Tensor, Tensor]]
    enc_mask_ext: Tensor  # (R,1,1,S) additive, gram_t5_modeling.py:1145-1147
    rows_per_bank: int  # K when the bank is shared by a user's beams, 1 when replicated
    self_k: List[Optional[Tensor]] = field(default_factory=list)
    self_v: List[Optional[Tensor]] = field(default_factory=list)

    def reorder(self, beam_idx: Tensor) -> None:
        """_reorder_cache, gram_t5.py:320-348.  In reference-faithful mode the cross K/V
        are index_select-ed too (they are beam-replicated there); in shared mode they are
        beam-invariant by construction and left alone."""
        self.self_k = [k.index_select(0, beam_idx) for k in self.self_k]
        self.self_v = [v.index_select(0, beam_idx) for v in self.self_v]
        if self.rows_per_bank == 1:
            self.cross = [(k.index_select(0, beam_idx), v.index_select(0, beam_idx)) for k, v in self.cross]


def decoder_step(sd, cfg: OracleConfig, tokens: Tensor, st: DecodeState) -> Tensor:
    """One cached decode step for R rows: T5ForConditionalGeneration_GRAM.forward with
    encoder_outputs given (gram_t5.py:181-254) -> decoder T5Stack (gram_t5_modeling.py:1037-1296,
    blocks :723-837).  tokens: (R,) last token of each row.  Returns logits (R, V)."""
    H, dk, d = cfg.num_heads, cfg.d_kv, cfg.d_model
    R = tokens.shape[0]
    x = sd["shared.weight"][tokens][:, None, :]  # (R,1,d)
    t = 0 if not st.self_k else st.self_k[0].shape[2]
    # unidirectional bias, last query row only (:586-593); layer-0 table shared (:1246-1249)
    bias = position_bias(
        sd["decoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight"], t + 1, t + 1, False, cfg
    )[:, :, -1:, :]
    first = not st.self_k
    for i in range(cfg.num_decoder_layers):
        p = f"decoder.block.{i}.layer"
        a = p + ".0.SelfAttention"
        h = rms_norm(x, sd[p + ".0.layer_norm.weight"], cfg.layer_norm_epsilon)
        q = _heads(h @ sd[a + ".q.weight"].T, H, dk)
        k = _heads(h @ sd[a + ".k.weight"].T, H, dk)
        v = _heads(h @ sd[a + ".v.weight"].T, H, dk)
        if first:
            st.self_k.append(k)
            st.self_v.append(v)
        else:
            st.self_k[i] = torch.cat([st.self_k[i], k], dim=2)  # :540
            st.self_v[i] = torch.cat([st.self_v[i], v], dim=2)
        x = x + _attend(q, st.self_k[i], st.self_v[i], bias) @ sd[a + ".o.weight"].T
        # cross-attention over the fused bank: zero position bias (:577-582) + mask
        c = p + ".1.EncDecAttention"
        h = rms_norm(x, sd[p + ".1.layer_norm.weight"], cfg.layer_norm_epsilon)
        q = _heads(h @ sd[c + ".q.weight"].T, H, dk)  # (R,H,1,dk)
        ck, cv = st.cross[i]
        if st.rows_per_bank > 1:  # bank shared by a user's K beams: (B,H,S,dk)
            K = st.rows_per_bank
            B = R // K
            qb = q.view(B, K, H, 1, dk).permute(0, 2, 1, 3, 4).reshape(B, H, K, dk)
            mb = st.enc_mask_ext.view(B, K, 1, 1, -1)[:, 0]  # (B,1,1,S)
            o = _attend(qb, ck, cv, mb)  # (B,K,H*dk)
            o = o.reshape(R, 1, H * dk)
        else:
            o = _attend(q, ck, cv, st.enc_mask_ext)
        x = x + o @ sd[c + ".o.weight"].T
        h = rms_norm(x, sd[p + ".2.layer_norm.weight"], cfg.layer_norm_epsilon)
        x = x + torch.relu(h @ sd[p + ".2.DenseReluDense.wi.weight"].T) @ sd[p + ".2.DenseReluDense.wo.weight"].T
    x = rms_norm(x, sd["decoder.final_layer_norm.weight"], cfg.layer_norm_epsilon)
    if cfg.tie_word_embeddings:
        x = x * (d ** -0.5)  # gram_t5.py:249-252
    return (x @ sd["lm_head.weight"].T)[:, 0, :]  # gram_t5.py:254


# --------------------------------------------------------------------------------------
# Trie (src/utils/generation_trie.py)
# --------------------------------------------------------------------------------------
class Trie:
    """Nested-dict prefix tree, generation_trie.py:5-86 (append_trie/bos_token_id are never
    set by the runners and are not restated)."""

    def __init__(self, sequences: Sequence[Sequence[int]] = ()):
        self.trie_dict: dict = {}
        self.len = 0
        for s in sequences:
            self.add(s)

    def add(self, sequence: Sequence[int]) -> None:  # :38-42 (iterative form)
        node = self.trie_dict
        for tok in sequence:
            node = node.setdefault(int(tok), {})
        self.len += 1

    def get(self, prefix: Sequence[int]) -> List[int]:  # :45-68
        node = self.trie_dict
        for tok in prefix:
            if tok not in node:
                return []
            node = node[tok]
        return list(node.keys())

    def __len__(self):
        return self.len


def prefix_allowed_tokens_fn(trie: Trie) -> Callable[[int, Tensor], List[int]]:
    """generation_trie.py:89-95."""

    def fn(batch_id: int, sentence) -> List[int]:
        return trie.get(sentence.tolist() if hasattr(sentence, "tolist") else list(sentence))

    return fn


# --------------------------------------------------------------------------------------
# Beam search: transformers==4.26.0 GenerationMixin.beam_search + BeamSearchScorer +
# BeamHypotheses + PrefixConstrainedLogitsProcessor (third-party; restated from the
# published 4.26.0 algorithm, see module docstring).  Call site gram.py:93-99, arguments
# single_runner_gram.py:641-651.
# --------------------------------------------------------------------------------------
class BeamHypotheses:
    def __init__(self, num_beams: int, length_penalty: float):
        self.num_beams = num_beams
        self.length_penalty = length_penalty
        self.beams: List[Tuple[float, List[int]]] = []
        self.worst_score = 1e9

    def __len__(self):
        return len(self.beams)

    def add(self, hyp: List[int], sum_logprobs: float) -> None:
        score = sum_logprobs / (len(hyp) ** self.length_penalty)
        if len(self) < self.num_beams or score > self.worst_score:
            self.beams.append((score, hyp))
            if len(self) > self.num_beams:
                order = sorted((s, idx) for idx, (s, _) in enumerate(self.beams))
                del self.beams[order[0][1]]
                self.worst_score = order[1][0]
            else:
                self.worst_score = min(score, self.worst_score)

    def is_done(self, best_sum_logprobs: float, cur_len: int) -> bool:
        if len(self) < self.num_beams:
            return False
        return self.worst_score >= best_sum_logprobs / cur_len ** self.length_penalty


def topk_candidates(scores: Tensor, k: int) -> Tuple[Tensor, Tensor]:
    """torch.topk(sorted=True) with the tie rule made explicit: equal scores are ordered by
    ascending flat index (the order HF's CPU path produces for distinct floats; for ties
    torch.topk is implementation-defined and this oracle DEFINES it)."""
    s, idx = torch.sort(scores, dim=-1, descending=True, stable=True)
    return s[..., :k], idx[..., :k]


def beam_search(
    step_fn: Callable[[Tensor], Tensor],
    reorder_fn: Callable[[Tensor], None],
    batch_size: int,
    num_beams: int,
    max_length: int,
    prefix_fn: Optional[Callable[[int, Tensor], List[int]]],
    length_penalty: float = 1.0,
    num_return_sequences: Optional[int] = None,
    pad_token_id: int = 0,
    eos_token_id: int = 1,
    decoder_start_token_id: int = 0,
    early_exit: bool = True,
    trace: Optional[list] = None,
) -> Tuple[Tensor, Tensor]:
    """HF 4.26 beam_search with early_stopping=False, do_sample=False, one beam group.

    step_fn(last_tokens (R,)) -> logits (R,V) advances the cached decoder by one token;
    reorder_fn(beam_idx (R,)) reorders its caches.  Returns (sequences (B*nret, T) int64,
    sequences_scores (B*nret,) fp32) exactly as ``generate(..., return_dict_in_generate=True,
    output_scores=True)`` exposes them to single_runner_gram.py:654-655.

    ``early_exit=False`` keeps stepping to max_length even when every user is done (done users
    are padded exactly as BeamSearchScorer.process pads them); results are identical, which is
    the property the device implementation relies on to avoid a per-step host sync.
    """
    K = num_beams
    nret = num_return_sequences or K
    B = batch_size
    R = B * K
    input_ids = torch.full((R, 1), decoder_start_token_id, dtype=torch.long)
    beam_scores = torch.zeros(B, K, dtype=torch.float32)
    beam_scores[:, 1:] = -1e9
    beam_scores = beam_scores.view(R)
    hyps = [BeamHypotheses(K, length_penalty) for _ in range(B)]
    done = [False] * B
    next_scores = next_tokens = next_indices = None

    while True:
        cur_len = input_ids.shape[1]
        logits = step_fn(input_ids[:, -1])
        dev = logits.device  # the floating-point stack may sit on an accelerator (tests at population scale); the
        #                      bookkeeping below always runs on the host, on the top-2K candidates only
        logp = torch.log_softmax(logits.float(), dim=-1)
        V = logp.shape[-1]
        if prefix_fn is not None:  # PrefixConstrainedLogitsProcessor
            mask = torch.full_like(logp, -math.inf)
            rows: List[int] = []
            cols: List[int] = []
            for b in range(B):
                for k in range(K):
                    allowed = prefix_fn(b, input_ids[b * K + k])  # empty list -> whole row -inf (4.26: no raise)
                    rows.extend([b * K + k] * len(allowed))
                    cols.extend(int(a) for a in allowed)
            if rows:
                mask[torch.tensor(rows, dtype=torch.long, device=dev), torch.tensor(cols, dtype=torch.long, device=dev)] = 0
            logp = logp + mask
        scores = (logp + beam_scores.to(dev)[:, None]).view(B, K * V)
        next_scores, flat = topk_candidates(scores, 2 * K)
        next_scores, flat = next_scores.cpu(), flat.cpu()
        next_indices = torch.div(flat, V, rounding_mode="floor")
        next_tokens = flat % V
        if trace is not None:
            trace.append(dict(cur_len=cur_len, next_scores=next_scores.clone(), next_tokens=next_tokens.clone(),
                              next_indices=next_indices.clone(), input_ids=input_ids.clone(),
                              beam_scores=beam_scores.clone()))

        # BeamSearchScorer.process
        nb_scores = torch.zeros(B, K, dtype=torch.float32)
        nb_tokens = torch.zeros(B, K, dtype=torch.long)
        nb_idx = torch.zeros(B, K, dtype=torch.long)
        for b in range(B):
            if done[b]:
                nb_scores[b] = 0
                nb_tokens[b] = pad_token_id
                nb_idx[b] = 0
                continue
            j = 0
            for rank in range(2 * K):
                tok = int(next_tokens[b, rank])
                sc = next_scores[b, rank]
                row = b * K + int(next_indices[b, rank])
                if tok == eos_token_id:
                    if rank >= K:
                        continue
                    hyps[b].add(input_ids[row].tolist(), float(sc))
                else:
                    nb_scores[b, j] = sc
                    nb_tokens[b, j] = tok
                    nb_idx[b, j] = row
                    j += 1
                if j == K:
                    break
            if j < K:
                raise ValueError("fewer than num_beams non-EOS candidates in the top 2*num_beams")
            done[b] = done[b] or hyps[b].is_done(float(next_scores[b].max()), cur_len)

        beam_scores = nb_scores.view(R)
        beam_idx = nb_idx.view(R)
        input_ids = torch.cat([input_ids[beam_idx], nb_tokens.view(R, 1)], dim=-1)
        reorder_fn(beam_idx)
        if (early_exit and all(done)) or input_ids.shape[1] >= max_length:
            break

    # BeamSearchScorer.finalize
    for b in range(B):
        if done[b]:
            continue
        for k in range(K):
            row = b * K + k
            hyps[b].add(input_ids[row].tolist(), float(beam_scores[row]))
    best: List[List[int]] = []
    best_scores = torch.zeros(B * nret, dtype=torch.float32)
    for b in range(B):
        ordered = sorted(hyps[b].beams, key=lambda x: x[0])
        for j in range(nret):
            s, h = ordered.pop()
            best.append(h)
            best_scores[b * nret + j] = s
    lens = [len(h) for h in best]
    sent_max_len = min(max(lens) + 1, max_length)
    decoded = torch.full((B * nret, sent_max_len), pad_token_id, dtype=torch.long)
    for i, h in enumerate(best):
        decoded[i, : len(h)] = torch.tensor(h, dtype=torch.long)
        if len(h) < sent_max_len:
            decoded[i, len(h)] = eos_token_id
    return decoded, best_scores


def greedy_search(
    step_fn: Callable[[Tensor], Tensor],
    batch_size: int,
    max_length: int,
    prefix_fn: Optional[Callable[[int, Tensor], List[int]]],
    pad_token_id: int = 0,
    eos_token_id: int = 1,
    decoder_start_token_id: int = 0,
) -> Tensor:
    """HF 4.26 ``greedy_search`` (what ``generate`` runs for num_beams=1, do_sample=False; BASELINE
    configs[0]).  The prefix constraint is applied to the raw logits (-inf outside the allowed list; an empty
    list leaves the whole row -inf and argmax returns index 0), finished rows emit pad, the loop ends when every
    row has produced EOS or at max_length.  Returns sequences (B, T) int64; there are no sequences_scores."""
    B = batch_size
    input_ids = torch.full((B, 1), decoder_start_token_id, dtype=torch.long)
    unfinished = torch.ones(B, dtype=torch.long)
    while True:
        logits = step_fn(input_ids[:, -1]).float()
        if prefix_fn is not None:
            mask = torch.full_like(logits, -math.inf)
            for b in range(B):
                mask[b, prefix_fn(b, input_ids[b])] = 0
            logits = logits + mask
        nxt = torch.argmax(logits, dim=-1)
        nxt = nxt * unfinished + pad_token_id * (1 - unfinished)
        input_ids = torch.cat([input_ids, nxt[:, None]], dim=-1)
        unfinished = unfinished * (nxt != eos_token_id).long()
        if int(unfinished.max()) == 0 or input_ids.shape[1] >= max_length:
            break
    return input_ids


def generate(
    sd: Dict[str, Tensor],
    cfg: OracleConfig,
    input_ids: Tensor,
    attention_mask: Tensor,
    max_length: int,
    prefix_allowed_tokens_fn: Optional[Callable] = None,
    num_beams: int = 1,
    num_return_sequences: Optional[int] = None,
    length_penalty: float = 1.0,
    reference_faithful: bool = False,
    early_exit: bool = True,
    trace: Optional[list] = None,
) -> Dict[str, Tensor]:
    """GRAM.generate, gram.py:74-107, with the kwargs single_runner_gram.py:641-651 passes.

    reference_faithful=True reproduces the reference's cost profile (encoder states and
    cross K/V replicated K times, every cached tensor index_select-ed each step); the default
    shares the bank across a user's beams.  Both give the same results up to fp32 rounding."""
    B, N, L = input_ids.shape
    K = num_beams
    dev = input_ids.device  # cpu everywhere except the population-scale tests, which run this same code on the GPU in fp32
    with torch.no_grad():
        enc = encode_fused(sd, cfg, input_ids, attention_mask)  # (B,S,d)
        mask2 = attention_mask.reshape(B, N * L).to(torch.float32)
        mask_rows = mask2.repeat_interleave(K, dim=0)  # HF _expand_inputs_for_generation
        ext = ((1.0 - mask_rows) * FMIN)[:, None, None, :]
        if reference_faithful:
            st = DecodeState(cross=cross_kv(sd, cfg, enc.repeat_interleave(K, dim=0)), enc_mask_ext=ext, rows_per_bank=1)
        else:
            st = DecodeState(cross=cross_kv(sd, cfg, enc), enc_mask_ext=ext, rows_per_bank=K)
        if K == 1:  # HF dispatches num_beams == 1 to greedy_search
            seqs = greedy_search(lambda tok: decoder_step(sd, cfg, tok.to(dev), st).cpu(), B, max_length, prefix_allowed_tokens_fn,
                                 cfg.pad_token_id, cfg.eos_token_id, cfg.decoder_start_token_id)
            return {"sequences": seqs, "sequences_scores": None, "encoder_last_hidden_state": enc}
        seqs, scores = beam_search(
            lambda tok: decoder_step(sd, cfg, tok.to(dev), st),
            lambda beam_idx: st.reorder(beam_idx.to(dev)),
            B, K, max_length, prefix_allowed_tokens_fn, length_penalty, num_return_sequences,
            cfg.pad_token_id, cfg.eos_token_id, cfg.decoder_start_token_id, early_exit, trace,
        )
    return {"sequences": seqs, "sequences_scores": scores, "encoder_last_hidden_state": enc}


def sequence_logprob(sd, cfg: OracleConfig, input_ids: Tensor, attention_mask: Tensor, seq: Sequence[int]) -> float:
    """Full-vocabulary-normalised sum of log-probs of ``seq[1:]`` given ``seq[0]`` (the start
    token) for ONE user -- what a beam's running score is (HF adds the mask after
    log_softmax, so scores stay full-vocab normalised).  Used by tolerance-aware parity tests."""
    with torch.no_grad():
        enc = encode_fused(sd, cfg, input_ids, attention_mask)
        B = enc.shape[0]
        assert B == 1
        ext = ((1.0 - attention_mask.reshape(1, -1).to(torch.float32)) * FMIN)[:, None, None, :]
        st = DecodeState(cross=cross_kv(sd, cfg, enc), enc_mask_ext=ext, rows_per_bank=1)
        total = 0.0
        for t in range(len(seq) - 1):
            logits = decoder_step(sd, cfg, torch.tensor([seq[t]]), st)
            total += float(torch.log_softmax(logits.float(), -1)[0, seq[t + 1]])
    return total


# --------------------------------------------------------------------------------------
# metrics (src/utils/evaluate.py)
# --------------------------------------------------------------------------------------
def rel_results(predictions: Sequence, targets: Sequence, scores: Sequence[float], k: int) -> List[List[int]]:
    """evaluate.py:5-22: per user stable-sort the k (prediction, score) pairs by score
    descending and mark equality with the gold."""
    out = []
    for b in range(len(targets)):
        pairs = list(zip(predictions[b * k:(b + 1) * k], scores[b * k:(b + 1) * k]))
        pairs = sorted(pairs, key=lambda x: x[1], reverse=True)
        out.append([1 if p == targets[b] else 0 for p, _ in pairs])
    return out


def hit_at_k(rel: Sequence[Sequence[int]], k: int) -> float:
    """evaluate.py:52-58 (sum over users, caller divides)."""
    return float(sum(1 for row in rel if sum(row[:k]) > 0))


def ndcg_at_k(rel: Sequence[Sequence[int]], k: int) -> float:
    """evaluate.py:38-49 (leave-one-out: IDCG == 1)."""
    total = 0.0
    for row in rel:
        one = 0.0
        for i, r in enumerate(row[:k]):
            one += r / math.log(i + 2, 2)
        total += one
    return total


def get_metrics_results(rel: Sequence[Sequence[int]], metrics: Sequence[str]) -> np.ndarray:
    """evaluate.py:25-35."""
    res = []
    for m in metrics:
        k = int(m.split("@")[1])
        if m.lower().startswith("hit"):
            res.append(hit_at_k(rel, k))
        elif m.lower().startswith("ndcg"):
            res.append(ndcg_at_k(rel, k))
    return np.array(res)


# --------------------------------------------------------------------------------------
# runner-level restatement (single_runner_gram.py:570-719, distributed_runner_gram.py:300-359,685-874)
# --------------------------------------------------------------------------------------
def evaluate_users(
    sd, cfg, batches, candidates: Sequence[Sequence[int]], metrics: Sequence[str], num_beams: int,
    length_penalty: float = 1.0, reference_faithful: bool = False,
):
    """test_dataset_task, single_runner_gram.py:570-719, on token ids instead of text:
    ``batches`` yields (input_ids (B,N,L), attention_mask (B,N,L), gold sequences [B][..]).
    Predictions and gold are compared as id tuples with specials (0,1) stripped, which is what
    ``batch_decode(skip_special_tokens=True)`` string equality reduces to when decode is
    injective on the candidate set."""
    trie = Trie(candidates)
    fn = prefix_allowed_tokens_fn(trie)
    max_length = max(len(c) for c in candidates)  # :633-636
    sums = np.zeros(len(metrics))
    total = 0
    for ids, mask, gold in batches:
        out = generate(sd, cfg, ids, mask, max_length, fn, num_beams, num_beams, length_penalty, reference_faithful)
        strip = lambda s: tuple(int(t) for t in s if int(t) not in (0, 1))
        preds = [strip(s) for s in out["sequences"]]
        rel = rel_results(preds, [strip(g) for g in gold], out["sequences_scores"].tolist(), num_beams)
        total += len(rel)
        sums += get_metrics_results(rel, metrics)
    return sums, total


def distributed_sampler_indices(n: int, world: int, rank: int, seed: int = 0, epoch: int = 0) -> List[int]:
    """torch DistributedSampler(shuffle=True, drop_last=False) as distributed_runner_gram.py:351
    constructs it: permutation seeded seed+epoch, padded by repetition to ceil(n/W)*W, then
    strided rank::W."""
    g = torch.Generator()
    g.manual_seed(seed + epoch)
    idx = torch.randperm(n, generator=g).tolist()
    total = math.ceil(n / world) * world
    pad = total - len(idx)
    if pad > 0:
        idx += (idx * math.ceil(pad / len(idx)))[:pad]
    return idx[rank:total:world]
