// generate.hip -- host-side orchestration of the scoring path behind the C ABI: the model
// handle, the workspace carve, and the encode -> fuse -> bank -> decode-loop launch sequences.
// Nothing here allocates device memory or synchronises (except the optional 4-byte width read
// at the end of gram_generate); every launch goes to the caller's stream, so the whole
// generate() is one stream-ordered launch train.
//
// Reference call graph being replaced:
//   GRAM.generate gram.py:74-107 -> EncoderWrapper.forward gram.py:200-256 -> T5Stack (encoder)
//   gram_t5_modeling.py:1037-1296 -> HF beam_search -> per step
//   T5ForConditionalGeneration_GRAM.forward gram_t5.py:118-287 -> T5Stack (decoder) -> lm_head.
#include <stdlib.h>
#include <string.h>
#include <vector>

#include "common.h"
#include "prof.h"

struct gram_model {
  gram_model_desc_t d;
  std::vector<const float*> enc_ln1, enc_ln2, dec_ln1, dec_ln2, dec_ln3;
  std::vector<const void*> enc_wqkv, enc_wo, enc_wi, enc_wo2, dec_wqkv, dec_wo, dec_wq_x, dec_wo_x, dec_wi, dec_wo2;
  // 1 / (power-of-two factor a weight matrix was scaled by) = the out_scale of its GEMM (gram_model_desc_t.w_scales)
  std::vector<float> s_enc_wqkv, s_enc_wo, s_enc_wi, s_enc_wo2, s_dec_wqkv, s_dec_wo, s_dec_wq_x, s_dec_wo_x, s_dec_wi, s_dec_wo2;
  float s_wkv = 1.f, s_lm = 1.f;
};

namespace {

struct Carve {
  char* base;
  int64_t off;
  explicit Carve(void* p) : base((char*)p), off(0) {}
  template <typename T>
  T* take(int64_t n) {
    off = (off + 255) & ~(int64_t)255;
    T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
    off += n * (int64_t)sizeof(T);
    return p;
  }
};

// The folded-norm partial sums of squares are [rows][d / 64] floats, or -- "quarter" layout of the streaming small-M GEMM, chosen per
// CALL from the rows that call works on (all of them, the compacted encoder rows, the live rows of a late decode step) --
// [rows_of_the_call][d / 16] with rows_of_the_call <= gram_gemm_stream_max_m() <= kStreamRowsLimit.  The buffer holds the larger of
// the two for every count the full row number allows, whatever GRAM_GEMM_STREAM_MAXM says.
constexpr int64_t kStreamRowsLimit = GRAM_STREAM_MAX_M_LIMIT;  // (common.h: gram_gemm_stream_max_m() never exceeds it)
inline int64_t ss_floats(int64_t rows, int64_t d) {
  const int64_t q = (rows < kStreamRowsLimit ? rows : kStreamRowsLimit) * (d / 16), full = rows * (d / 64);
  return q > full ? q : full;
}
struct Workspace {
  // encoder
  float* x;         // [Me][d]   residual stream (fp32)
  p16* h;          // [Me][d]   normed activations (GEMM A operand)
  p16* qkv;        // [Me][3*inner]
  p16* attn;       // [Me][inner]
  p16* u;          // [Me][d_ff]
  float* ss;        // [Me][d/64]  folded-norm partial sums of squares
  float* rs;        // [Me]        1/rms per row
  float* xs[2];     // [Me] x 2    per-row power-of-two factors of the 16-bit copy of x (gram_norm_fusion_t.xs_in / xs_out), ping-pong
  // fused bank
  p16* bank_k;     // [layers][B][H][S][64]
  p16* bank_vt;    // [layers][B][H][S/32][64][32]
  // decoder
  float* xd;        // [R][d]
  p16* hd;         // [R][d]
  p16* qkvd;       // [R][3*inner]
  p16* attnd;      // [R][inner]
  p16* qx;         // [R][inner]
  p16* ud;         // [R][d_ff]
  float* ssd;       // [R][d/64]
  float* rsd;       // [R]
  float* xsd[2];    // [R] x 2
  p16* kcache;     // [layers][Tmax][R][inner]
  p16* vcache;
  float* logits;    // [R][V]
  float* lse;       // [R]
  float* lse_part;  // [R][V/64][2]
  // beam state
  gram_beam_state_t beam;
  gram_live_rows_t live;
  int32_t* width;
  uint32_t* key_bits;  // [B][128]  the cross-attention's bit view of the mask (gram_mask_key_bits), once per generate
  int64_t bytes;
  // two-piece mode (gram_split_t): what a GEMM reads (h, attn, u, hd, attnd, ud) is ONE interleaved buffer of twice the row length;
  // what only attention kernels read (qkv, the bank, qkvd, qx, the cache) is `pieces` planar copies, these many elements apart
  int pieces;
  int64_t ps_qkv, ps_bank, ps_qkvd, ps_qx, ps_cache;
};

Workspace carve(const gram_model* m, void* ws, int B, int N, int L, int K, int Tmax) {
  const gram_model_desc_t& c = m->d;
  const int64_t d = c.d_model, inner = (int64_t)c.n_heads * 64, F = c.d_ff, V = c.vocab;
  const int64_t Me = (int64_t)B * N * L, S = (int64_t)N * L, R = (int64_t)B * K, nl = c.n_dec_layers;
  Carve cv(ws);
  Workspace w{};
  const int64_t P = c.pieces > 1 ? c.pieces : 1;
  w.pieces = (int)P;
  w.ps_qkv = Me * 3 * inner;
  w.ps_bank = nl * B * c.n_heads * S * 64;
  w.ps_qkvd = R * 3 * inner;
  w.ps_qx = R * inner;
  w.ps_cache = nl * Tmax * R * inner;
  w.x = cv.take<float>(Me * d);
  w.h = cv.take<p16>(P * Me * d);
  w.qkv = cv.take<p16>(P * w.ps_qkv);
  w.attn = cv.take<p16>(P * Me * inner);
  w.u = cv.take<p16>(P * Me * F);
  w.ss = cv.take<float>(ss_floats(Me, d));
  w.rs = cv.take<float>(Me);
  w.xs[0] = cv.take<float>(Me);
  w.xs[1] = cv.take<float>(Me);
  w.bank_k = cv.take<p16>(P * w.ps_bank);
  w.bank_vt = cv.take<p16>(P * w.ps_bank);
  w.xd = cv.take<float>(R * d);
  w.hd = cv.take<p16>(P * R * d);
  w.qkvd = cv.take<p16>(P * w.ps_qkvd);
  w.attnd = cv.take<p16>(P * R * inner);
  w.qx = cv.take<p16>(P * w.ps_qx);
  w.ud = cv.take<p16>(P * R * F);
  w.ssd = cv.take<float>(ss_floats(R, d));  // (16-column partials for a small-M step: gram_norm_fusion_t.quarter)
  w.rsd = cv.take<float>(R);
  w.xsd[0] = cv.take<float>(R);
  w.xsd[1] = cv.take<float>(R);
  w.kcache = cv.take<p16>(P * w.ps_cache);
  w.vcache = cv.take<p16>(P * w.ps_cache);
  w.logits = cv.take<float>(R * V);
  w.lse = cv.take<float>(R);
  w.lse_part = cv.take<float>(R * (V / 64) * 2);
  gram_beam_state_t& s = w.beam;
  s.B = B;
  s.K = K;
  s.Tmax = Tmax;
  s.length_penalty = 1.f;
  s.eos = 1;
  s.pad = 0;
  s.tokens = cv.take<int32_t>(R);
  s.node = cv.take<int32_t>(R);
  s.beam_scores = cv.take<float>(R);
  s.seq = cv.take<int32_t>(R * Tmax);
  s.anc = cv.take<int32_t>((int64_t)Tmax * R);
  s.done = cv.take<int32_t>(B);
  s.n_hyps = cv.take<int32_t>(B);
  s.hyp_score = cv.take<double>((int64_t)B * (K + 1));
  s.worst = cv.take<double>(B);
  s.hyp_len = cv.take<int32_t>((int64_t)B * (K + 1));
  s.hyp_tok = cv.take<int32_t>((int64_t)B * (K + 1) * Tmax);
  s.error = cv.take<int32_t>(4);
  // scratch of the small-batch sparse-logits kernel (gram_beam_state_t.cand_logits): 16 384 >= any K * max_fanout the search accepts
  s.cand_logits_users = B < 16 ? B : 16;
  s.cand_logits_stride = 16384;
  s.cand_logits = cv.take<float>((int64_t)s.cand_logits_users * s.cand_logits_stride);
  w.live.rows = cv.take<int32_t>(R);
  w.live.rowpos = cv.take<int32_t>(R);
  w.live.users = cv.take<int32_t>(B);
  w.live.tokens = cv.take<int32_t>(R);
  w.live.counts = cv.take<int32_t>(4);
  w.width = cv.take<int32_t>(4);
  w.key_bits = cv.take<uint32_t>((int64_t)B * 128);
  w.bytes = (cv.off + 255) & ~(int64_t)255;
  return w;
}

int check_shapes(const gram_model* m, int B, int N, int L, int K, int Tmax) {
  if (!m || B < 1 || N < 1 || N > m->d.max_passages || L < 32 || L > GRAM_MAX_PASSAGE_LEN || (L & 31) || K < 1 ||
      K > GRAM_MAX_BEAMS || Tmax < 2 || Tmax > GRAM_MAX_DEC_LEN)
    return GRAM_E_ARG;
  return 0;
}

constexpr int kPrecomputedRsRows = 32768;  // = the row count from which gemm.hip dispatches to the ping-pong kernel

#define TRY(x)            \
  do {                    \
    int e__ = (x);        \
    if (e__) return e__;  \
  } while (0)

// Sensitivity sweeps (gram_debug_set_stage_pieces): stage s computes on its first g_stage_cap[s] pieces only.  Implemented by ZEROING
// the upper piece of the stage's activation operands before they are consumed (a zero piece contributes exact zeros to every product
// it enters, so the arithmetic is that of one piece; the kernels and their cost are unchanged) -- the weights' upper piece is
// zeroed by the caller when it expands them.  Debug only: every cap costs a memset per use.
int g_stage_cap[GRAM_STAGE_COUNT] = {99, 99, 99, 99, 99, 99, 99, 99};
// planar buffer: `pieces` copies, ps elements apart
int cap_planar(const Workspace& w, void* buf, int64_t ps, int stage, void* st) {
  if (w.pieces < 2 || g_stage_cap[stage] >= 2) return 0;
  hipError_t e = hipMemsetAsync((p16*)buf + (size_t)ps, 0, (size_t)ps * sizeof(p16), (hipStream_t)st);
  return e == hipSuccess ? 0 : (int)e;
}
// interleaved buffer [rows][cols / 32][2][32]: piece 1 = the second 64 B of every 128 B
int cap_inter(const Workspace& w, void* buf, int64_t rows, int64_t cols, int stage, void* st) {
  if (w.pieces < 2 || g_stage_cap[stage] >= 2) return 0;
  hipError_t e = hipMemset2DAsync((char*)buf + 64, 128, 0, 64, (size_t)(rows * cols / 32), (hipStream_t)st);
  return e == hipSuccess ? 0 : (int)e;
}

// One Linear of the path.  Two-piece mode: A is the interleaved [M][2 kc] buffer, W the interleaved weight; a 16-bit C is written
// planar (c_kind 1: c_ps elements apart -- an attention kernel reads it) or interleaved (c_kind 2: [M][2 N], the next GEMM's A);
// the xb copy of a residual epilogue (nf) is always interleaved, the bank planar.
enum { C_NONE = 0, C_PLANAR = 1, C_INTER = 2 };
int linear(const Workspace& w, const void* A, const void* W, float out_scale, void* C, int c_kind, int64_t c_ps, int M, int N, int kc, int epi,
           const gram_kv_bank_t* bank, const gram_norm_fusion_t* nf, void* st) {
  const int P = w.pieces;
  const gram_split_t sp{P, c_kind == C_INTER, c_ps, w.ps_bank, out_scale};
  return gram_gemm_bf16_split(A, W, C, M, N, kc, P * kc, c_kind == C_INTER ? P * N : N, epi, bank, nf, &sp, st);
}

// The encoder layers on P passages (ids/mask [P][L]); leaves the residual stream in w.x rows [0, P*L).
int encoder_layers(const gram_model* m, const Workspace& w, const int64_t* ids, const uint8_t* mask, int L, int P, void* st) {
  const gram_model_desc_t& c = m->d;
  const int d = c.d_model, inner = c.n_heads * 64, F = c.d_ff, H = c.n_heads;
  const int Me = P * L;
  if (c.fold_norm) {
    // T5LayerNorm folded into the GEMMs around it (gram_norm_fusion_t): w.h holds xb = the 16-bit copy of x, w.ss the
    // per-row sum-of-squares partials; both are refreshed by every residual GEMM's epilogue
    // few rows (one short user): the streaming GEMM and its 16-column partials (gram_norm_fusion_t.quarter); the embedding writes 64-column ones
    const int quarter = Me <= gram_gemm_stream_max_m() && d % 128 == 0 && inner % 128 == 0 && F % 128 == 0;
    // big problems (the ping-pong GEMMs, M >= kPrecomputedRsRows) take 1/rms precomputed per row by one tiny kernel per
    // norm; below that the consumer GEMM adds the partials itself (same order, same bits) and the launch is saved --
    // a small batch is a chain of ~1 500 dependent launches and nothing else
    const bool pre_rs = Me >= kPrecomputedRsRows;
    // The 16-bit copy of the residual stream carries a power-of-two factor per row (gram_norm_fusion_t.xs_in / xs_out): T5's
    // stream leaves the IEEE-half range in trained checkpoints.  Norm point p: the producer before it scaled the copy by
    // xs[p & 1]; the consumer divides its row scale by that and publishes the factor of the NEXT producer in xs[(p + 1) & 1].
    int np = 0;  // norm point
    auto produce = [&]() { return gram_norm_fusion_t{w.h, w.ss, nullptr, 0, 0, 0.f, quarter, w.xs[np & 1], nullptr}; };
    auto consume = [&](bool from_embed) {
      const gram_norm_fusion_t nf = pre_rs ? gram_norm_fusion_t{nullptr, nullptr, w.rs, 0, d, c.eps, 0, nullptr, nullptr}
                                           : gram_norm_fusion_t{nullptr, nullptr, w.ss, d / 64, d, c.eps, from_embed ? 0 : quarter,
                                                                w.xs[np & 1], w.xs[(np + 1) & 1]};
      return nf;
    };
    auto norm_point = [&]() -> int {  // before the consumer GEMM of norm point np (big path: 1/rms / xs and the next factor)
      return pre_rs ? gram_row_rscale_xs(w.ss, w.rs, w.xs[np & 1], w.xs[(np + 1) & 1], Me, d / 64, d, c.eps, st) : 0;
    };
    TRY(gram_embed_ex_xs(c.embed_f32, ids, 1, w.x, w.h, w.ss, w.xs[0], d / 64, Me, d, w.pieces, st));
    for (int i = 0; i < c.n_enc_layers; ++i) {
      TRY(norm_point());
      TRY(cap_inter(w, w.h, Me, d, GRAM_STAGE_ENC_ATTN, st));
      {
        const gram_norm_fusion_t nf = consume(i == 0);
        TRY(linear(w, w.h, m->enc_wqkv[i], m->s_enc_wqkv[i], w.qkv, C_PLANAR, w.ps_qkv, Me, 3 * inner, d, GRAM_EPI_BF16, nullptr, &nf, st));
      }
      ++np;
      TRY(cap_planar(w, w.qkv, w.ps_qkv, GRAM_STAGE_ENC_ATTN, st));
      TRY(gram_enc_self_attn_split(w.qkv, c.enc_bias_f32, mask, w.attn, P, L, H, w.pieces, w.ps_qkv, st));
      TRY(cap_inter(w, w.attn, Me, inner, GRAM_STAGE_ENC_ATTN, st));
      {
        const gram_norm_fusion_t nf = produce();
        TRY(linear(w, w.attn, m->enc_wo[i], m->s_enc_wo[i], w.x, C_NONE, 0, Me, d, inner, GRAM_EPI_F32_ADD, nullptr, &nf, st));
      }
      TRY(norm_point());
      TRY(cap_inter(w, w.h, Me, d, GRAM_STAGE_ENC_FFN, st));
      {
        const gram_norm_fusion_t nf = consume(false);
        TRY(linear(w, w.h, m->enc_wi[i], m->s_enc_wi[i], w.u, C_INTER, 0, Me, F, d, GRAM_EPI_BF16_RELU, nullptr, &nf, st));
      }
      ++np;
      TRY(cap_inter(w, w.u, Me, F, GRAM_STAGE_ENC_FFN, st));
      {
        const gram_norm_fusion_t nf = produce();
        TRY(linear(w, w.u, m->enc_wo2[i], m->s_enc_wo2[i], w.x, C_NONE, 0, Me, d, F, GRAM_EPI_F32_ADD, nullptr, &nf, st));
      }
    }
  } else {  // (one piece only: gram_model_create insists on fold_norm in the two-piece mode)
    TRY(gram_embed_i64(c.embed_f32, ids, w.x, Me, d, st));
    for (int i = 0; i < c.n_enc_layers; ++i) {
      TRY(gram_rmsnorm_bf16_split(w.x, m->enc_ln1[i], w.h, Me, d, c.eps, 1.f, nullptr, 1, 1, nullptr, 1, st));
      TRY(linear(w, w.h, m->enc_wqkv[i], m->s_enc_wqkv[i], w.qkv, C_PLANAR, 0, Me, 3 * inner, d, GRAM_EPI_BF16, nullptr, nullptr, st));
      TRY(gram_enc_self_attn_split(w.qkv, c.enc_bias_f32, mask, w.attn, P, L, H, 1, 0, st));
      TRY(linear(w, w.attn, m->enc_wo[i], m->s_enc_wo[i], w.x, C_NONE, 0, Me, d, inner, GRAM_EPI_F32_ADD, nullptr, nullptr, st));
      TRY(gram_rmsnorm_bf16_split(w.x, m->enc_ln2[i], w.h, Me, d, c.eps, 1.f, nullptr, 1, 1, nullptr, 1, st));
      TRY(linear(w, w.h, m->enc_wi[i], m->s_enc_wi[i], w.u, C_PLANAR, 0, Me, F, d, GRAM_EPI_BF16_RELU, nullptr, nullptr, st));
      TRY(linear(w, w.u, m->enc_wo2[i], m->s_enc_wo2[i], w.x, C_NONE, 0, Me, d, F, GRAM_EPI_F32_ADD, nullptr, nullptr, st));
    }
  }
  return 0;
}

struct CachedPassages {  // gram_compaction_t's cache fields
  int n;
  int cache_L;
  const float* x;
  const int32_t* slot;
};

// P passages (all B*N, or the active ones with their flat indices in pmap): the first P - cached.n go through the
// encoder (ids/mask [P - cached.n][L]), the rest take their residual-stream rows from the passage cache.
int encode(const gram_model* m, const Workspace& w, const int64_t* ids, const uint8_t* mask, const uint8_t* full_mask, int B, int N,
           int L, int P, const int32_t* pmap, const CachedPassages& cached, void* st) {
  const gram_model_desc_t& c = m->d;
  const int d = c.d_model, inner = c.n_heads * 64, H = c.n_heads;
  const int Pe = P - cached.n, Me = P * L;
  TRY(gram_mask_key_bits(full_mask, w.key_bits, B, N * L, st));
  if (Pe > 0) TRY(encoder_layers(m, w, ids, mask, L, Pe, st));
  if (cached.n > 0) TRY(gram_gather_passage_x(cached.x, cached.slot, w.x + (size_t)Pe * L * d, cached.n, L, cached.cache_L, d, st));
  // final norm + per-passage position embedding = the late fusion (gram.py:238-255); the
  // (B*N, L, d) -> (B, N*L, d) view is free: rows are already user-major.
  TRY(gram_rmsnorm_bf16_split(w.x, c.enc_final_ln, w.h, Me, d, c.eps, 1.f, c.use_position_embedding ? c.pos_emb_f32 : nullptr, N, L,
                              pmap, w.pieces, st));
  // every decoder layer's cross K/V in ONE GEMM, scattered into the beam-shared bank
  gram_kv_bank_t bank{w.bank_k, w.bank_vt, c.n_dec_layers, B, H, N * L, pmap, N, L};
  TRY(linear(w, w.h, c.dec_wkv_x_all, m->s_wkv, nullptr, C_NONE, 0, Me, c.n_dec_layers * 2 * inner, d, GRAM_EPI_KV_BANK, &bank, nullptr, st));
  TRY(cap_planar(w, w.bank_k, w.ps_bank, GRAM_STAGE_BANK_K, st));
  TRY(cap_planar(w, w.bank_vt, w.ps_bank, GRAM_STAGE_BANK_V, st));
  return 0;
}

struct LiveStep {  // host view of gram_live_rows_t after the counts came back
  int n_rows, n_users;
  const int32_t *rows, *rowpos, *users;
};

// K = beams per user in THIS step's rows (1 for the compact step 0), R_cache = rows of the cache slots.
// live != NULL: the step runs on live->n_rows compact rows (tokens = their tokens), see gram_live_rows_t.
int decode_step(const gram_model* m, const Workspace& w, const int32_t* tokens, const int32_t* anc, const uint8_t* mask, int B,
                int N, int L, int K, int R_cache, int Tmax, int t, float* logits, float* lse_part, const LiveStep* live,
                void* st) {
  const gram_model_desc_t& c = m->d;
  const int d = c.d_model, inner = c.n_heads * 64, F = c.d_ff, H = c.n_heads, V = c.vocab;
  const int R = live ? live->n_rows : B * K, S = N * L;
  auto self_attn = [&](int i, size_t cache_layer) {
    return gram_dec_self_attn_split(w.qkvd, w.kcache + i * cache_layer, w.vcache + i * cache_layer, anc, c.dec_bias_f32, w.attnd,
                                    live ? R_cache : R, R, live ? live->rows : nullptr, H, t, Tmax, w.pieces, w.ps_qkvd, w.ps_cache, st);
  };
  auto cross_attn = [&](int i, size_t bank_layer) {
    return gram_cross_attn_decode_split(w.qx, w.bank_k + i * bank_layer, w.bank_vt + i * bank_layer, mask, w.attnd,
                                        live ? live->n_users : B, K, H, S, live ? live->users : nullptr, live ? live->rowpos : nullptr,
                                        w.pieces, w.ps_qx, w.ps_bank, w.key_bits, st);
  };
  const size_t bank_layer = (size_t)B * H * S * 64;
  const size_t cache_layer = (size_t)Tmax * R_cache * inner;
  if (c.fold_norm) {
    // a few rows (one user, or a handful): the streaming GEMM and its 16-column partials; the embedding writes 64-column ones
    const int quarter = R <= gram_gemm_stream_max_m() && d % 128 == 0 && inner % 128 == 0 && F % 128 == 0;
    const bool pre_rs = R >= kPrecomputedRsRows;  // see encoder_layers (also for the row factors xsd of the 16-bit copy)
    int np = 0;
    auto produce = [&]() { return gram_norm_fusion_t{w.hd, w.ssd, nullptr, 0, 0, 0.f, quarter, w.xsd[np & 1], nullptr}; };
    auto consume = [&](bool from_embed) {
      const gram_norm_fusion_t nf = pre_rs ? gram_norm_fusion_t{nullptr, nullptr, w.rsd, 0, d, c.eps, 0, nullptr, nullptr}
                                           : gram_norm_fusion_t{nullptr, nullptr, w.ssd, d / 64, d, c.eps, from_embed ? 0 : quarter,
                                                                w.xsd[np & 1], w.xsd[(np + 1) & 1]};
      return nf;
    };
    auto norm_point = [&]() -> int {
      return pre_rs ? gram_row_rscale_xs(w.ssd, w.rsd, w.xsd[np & 1], w.xsd[(np + 1) & 1], R, d / 64, d, c.eps, st) : 0;
    };
    TRY(gram_embed_ex_xs(c.embed_f32, tokens, 0, w.xd, w.hd, w.ssd, w.xsd[0], d / 64, R, d, w.pieces, st));
    for (int i = 0; i < c.n_dec_layers; ++i) {
      TRY(norm_point());
      TRY(cap_inter(w, w.hd, R, d, GRAM_STAGE_DEC_SELF, st));
      {
        const gram_norm_fusion_t nf = consume(i == 0);
        TRY(linear(w, w.hd, m->dec_wqkv[i], m->s_dec_wqkv[i], w.qkvd, C_PLANAR, w.ps_qkvd, R, 3 * inner, d, GRAM_EPI_BF16, nullptr, &nf, st));
      }
      ++np;
      TRY(cap_planar(w, w.qkvd, w.ps_qkvd, GRAM_STAGE_DEC_SELF, st));
      TRY(self_attn(i, cache_layer));
      TRY(cap_inter(w, w.attnd, R, inner, GRAM_STAGE_DEC_SELF, st));
      {
        const gram_norm_fusion_t nf = produce();
        TRY(linear(w, w.attnd, m->dec_wo[i], m->s_dec_wo[i], w.xd, C_NONE, 0, R, d, inner, GRAM_EPI_F32_ADD, nullptr, &nf, st));
      }
      TRY(norm_point());
      TRY(cap_inter(w, w.hd, R, d, GRAM_STAGE_DEC_CROSS, st));
      {
        const gram_norm_fusion_t nf = consume(false);
        TRY(linear(w, w.hd, m->dec_wq_x[i], m->s_dec_wq_x[i], w.qx, C_PLANAR, w.ps_qx, R, inner, d, GRAM_EPI_BF16, nullptr, &nf, st));
      }
      ++np;
      TRY(cap_planar(w, w.qx, w.ps_qx, GRAM_STAGE_DEC_CROSS, st));
      TRY(cross_attn(i, bank_layer));
      TRY(cap_inter(w, w.attnd, R, inner, GRAM_STAGE_DEC_CROSS, st));
      {
        const gram_norm_fusion_t nf = produce();
        TRY(linear(w, w.attnd, m->dec_wo_x[i], m->s_dec_wo_x[i], w.xd, C_NONE, 0, R, d, inner, GRAM_EPI_F32_ADD, nullptr, &nf, st));
      }
      TRY(norm_point());
      TRY(cap_inter(w, w.hd, R, d, GRAM_STAGE_DEC_FFN, st));
      {
        const gram_norm_fusion_t nf = consume(false);
        TRY(linear(w, w.hd, m->dec_wi[i], m->s_dec_wi[i], w.ud, C_INTER, 0, R, F, d, GRAM_EPI_BF16_RELU, nullptr, &nf, st));
      }
      ++np;
      TRY(cap_inter(w, w.ud, R, F, GRAM_STAGE_DEC_FFN, st));
      {
        const gram_norm_fusion_t nf = produce();
        TRY(linear(w, w.ud, m->dec_wo2[i], m->s_dec_wo2[i], w.xd, C_NONE, 0, R, d, F, GRAM_EPI_F32_ADD, nullptr, &nf, st));
      }
    }
  } else {  // (one piece only)
    TRY(gram_embed_i32(c.embed_f32, tokens, w.xd, R, d, st));
    for (int i = 0; i < c.n_dec_layers; ++i) {
      TRY(gram_rmsnorm_bf16_split(w.xd, m->dec_ln1[i], w.hd, R, d, c.eps, 1.f, nullptr, 1, 1, nullptr, 1, st));
      TRY(linear(w, w.hd, m->dec_wqkv[i], m->s_dec_wqkv[i], w.qkvd, C_PLANAR, 0, R, 3 * inner, d, GRAM_EPI_BF16, nullptr, nullptr, st));
      TRY(self_attn(i, cache_layer));
      TRY(linear(w, w.attnd, m->dec_wo[i], m->s_dec_wo[i], w.xd, C_NONE, 0, R, d, inner, GRAM_EPI_F32_ADD, nullptr, nullptr, st));
      TRY(gram_rmsnorm_bf16_split(w.xd, m->dec_ln2[i], w.hd, R, d, c.eps, 1.f, nullptr, 1, 1, nullptr, 1, st));
      TRY(linear(w, w.hd, m->dec_wq_x[i], m->s_dec_wq_x[i], w.qx, C_PLANAR, 0, R, inner, d, GRAM_EPI_BF16, nullptr, nullptr, st));
      TRY(cross_attn(i, bank_layer));
      TRY(linear(w, w.attnd, m->dec_wo_x[i], m->s_dec_wo_x[i], w.xd, C_NONE, 0, R, d, inner, GRAM_EPI_F32_ADD, nullptr, nullptr, st));
      TRY(gram_rmsnorm_bf16_split(w.xd, m->dec_ln3[i], w.hd, R, d, c.eps, 1.f, nullptr, 1, 1, nullptr, 1, st));
      TRY(linear(w, w.hd, m->dec_wi[i], m->s_dec_wi[i], w.ud, C_PLANAR, 0, R, F, d, GRAM_EPI_BF16_RELU, nullptr, nullptr, st));
      TRY(linear(w, w.ud, m->dec_wo2[i], m->s_dec_wo2[i], w.xd, C_NONE, 0, R, d, F, GRAM_EPI_F32_ADD, nullptr, nullptr, st));
    }
  }
  const float scale = c.tie_word_embeddings ? 1.0f / sqrtf((float)d) : 1.f;  // gram_t5.py:249-252
  TRY(gram_rmsnorm_bf16_split(w.xd, c.dec_final_ln, w.hd, R, d, c.eps, scale, nullptr, 1, 1, nullptr, w.pieces, st));
  TRY(cap_inter(w, w.hd, R, d, GRAM_STAGE_LM_HEAD, st));
  const gram_split_t sp{w.pieces, 0, 0, 0, m->s_lm};
  if (lse_part)  // log-softmax normaliser partials straight from the accumulators; logits may be NULL (not stored)
    TRY(gram_gemm_bf16_lse_split(w.hd, c.lm_head_bf16, logits, lse_part, R, V, d, w.pieces * d, V, &sp, st));
  else
    TRY(gram_gemm_bf16_split(w.hd, c.lm_head_bf16, logits, R, V, d, w.pieces * d, V, GRAM_EPI_F32, nullptr, nullptr, &sp, st));
  return 0;
}

// the search step on the hidden states decode_step left in w.hd (rowpos: live-row step, else NULL)
int search_step(const gram_model* m, const Workspace& w, const gram_trie_t* trie, int cur_len, int rows_per_user, const int32_t* rowpos,
                void* st) {
  const gram_model_desc_t& c = m->d;
  if (w.pieces > 1)
    return gram_beam_step_sparse_split(&w.beam, trie, w.hd, c.lm_head_f32, c.d_model, w.lse, c.vocab, cur_len, rows_per_user, rowpos,
                                       w.pieces, st);
  if (rowpos) return gram_beam_step_sparse_live(&w.beam, trie, w.hd, c.lm_head_bf16, c.d_model, w.lse, c.vocab, cur_len, rowpos, st);
  return gram_beam_step_sparse(&w.beam, trie, w.hd, c.lm_head_bf16, c.d_model, w.lse, c.vocab, cur_len, rows_per_user, st);
}

}  // namespace

extern "C" int gram_abi_version(void) { return GRAM_ABI_VERSION; }
extern "C" int gram_piece_format(void) { return GRAM_PIECE_FORMAT; }

extern "C" int gram_debug_set_stage_pieces(const int32_t* caps, int n) {
  if (caps && n != GRAM_STAGE_COUNT) return GRAM_E_ARG;
  for (int i = 0; i < GRAM_STAGE_COUNT; ++i) g_stage_cap[i] = caps ? (caps[i] < 1 ? 1 : caps[i]) : 99;
  return 0;
}

static int g_live_rows = -1;  // -1: the GRAM_LIVE_ROWS environment variable decides (default on)
extern "C" int gram_debug_set_live_rows(int on) {
  g_live_rows = on;
  return 0;
}

extern "C" gram_model_t* gram_model_create(const gram_model_desc_t* d) {
  if (!d || d->vocab % 128 || d->d_model % 128 || d->d_ff % 128 || d->n_heads < 1 || d->n_heads > 16 ||
      (d->n_heads * 64) % 128 || d->d_model > 1024 || d->n_enc_layers < 1 || d->n_dec_layers < 1)
    return nullptr;
  if (d->pieces < 0 || d->pieces > GRAM_MAX_PIECES || (d->pieces > 1 && (!d->lm_head_f32 || !d->fold_norm))) return nullptr;
  gram_model* m = new gram_model();
  m->d = *d;
  auto cpf = [](std::vector<const float*>& v, const float* const* src, int n) { v.assign(src, src + n); };
  auto cpv = [](std::vector<const void*>& v, const void* const* src, int n) { v.assign(src, src + n); };
  cpf(m->enc_ln1, d->enc_ln1, d->n_enc_layers);
  cpf(m->enc_ln2, d->enc_ln2, d->n_enc_layers);
  cpv(m->enc_wqkv, d->enc_wqkv, d->n_enc_layers);
  cpv(m->enc_wo, d->enc_wo, d->n_enc_layers);
  cpv(m->enc_wi, d->enc_wi, d->n_enc_layers);
  cpv(m->enc_wo2, d->enc_wo2, d->n_enc_layers);
  cpf(m->dec_ln1, d->dec_ln1, d->n_dec_layers);
  cpf(m->dec_ln2, d->dec_ln2, d->n_dec_layers);
  cpf(m->dec_ln3, d->dec_ln3, d->n_dec_layers);
  cpv(m->dec_wqkv, d->dec_wqkv, d->n_dec_layers);
  cpv(m->dec_wo, d->dec_wo, d->n_dec_layers);
  cpv(m->dec_wq_x, d->dec_wq_x, d->n_dec_layers);
  cpv(m->dec_wo_x, d->dec_wo_x, d->n_dec_layers);
  cpv(m->dec_wi, d->dec_wi, d->n_dec_layers);
  cpv(m->dec_wo2, d->dec_wo2, d->n_dec_layers);
  {
    const float* ws = d->w_scales;
    auto take = [&](std::vector<float>& v, int n) {
      v.assign(n, 1.f);
      if (ws) {
        for (int i = 0; i < n; ++i) v[i] = 1.f / ws[i];
        ws += n;
      }
    };
    take(m->s_enc_wqkv, d->n_enc_layers);
    take(m->s_enc_wo, d->n_enc_layers);
    take(m->s_enc_wi, d->n_enc_layers);
    take(m->s_enc_wo2, d->n_enc_layers);
    take(m->s_dec_wqkv, d->n_dec_layers);
    take(m->s_dec_wo, d->n_dec_layers);
    take(m->s_dec_wq_x, d->n_dec_layers);
    take(m->s_dec_wo_x, d->n_dec_layers);
    take(m->s_dec_wi, d->n_dec_layers);
    take(m->s_dec_wo2, d->n_dec_layers);
    if (ws) {
      m->s_wkv = 1.f / ws[0];
      m->s_lm = 1.f / ws[1];
    }
    m->d.w_scales = nullptr;  // (consumed)
  }
  // the descriptor's per-layer arrays now point at storage the handle owns
  m->d.enc_ln1 = m->enc_ln1.data();
  m->d.enc_ln2 = m->enc_ln2.data();
  m->d.enc_wqkv = m->enc_wqkv.data();
  m->d.enc_wo = m->enc_wo.data();
  m->d.enc_wi = m->enc_wi.data();
  m->d.enc_wo2 = m->enc_wo2.data();
  m->d.dec_ln1 = m->dec_ln1.data();
  m->d.dec_ln2 = m->dec_ln2.data();
  m->d.dec_ln3 = m->dec_ln3.data();
  m->d.dec_wqkv = m->dec_wqkv.data();
  m->d.dec_wo = m->dec_wo.data();
  m->d.dec_wq_x = m->dec_wq_x.data();
  m->d.dec_wo_x = m->dec_wo_x.data();
  m->d.dec_wi = m->dec_wi.data();
  m->d.dec_wo2 = m->dec_wo2.data();
  return m;
}

extern "C" void gram_model_destroy(gram_model_t* m) {
  if (!m) return;
  delete m;
}

extern "C" int64_t gram_workspace_bytes(const gram_model_t* m, int B, int N, int L, int K, int max_length) {
  if (check_shapes(m, B, N, L, K, max_length)) return GRAM_E_ARG;
  return carve(m, nullptr, B, N, L, K, max_length).bytes;
}

extern "C" int64_t gram_workspace_encoder_x_offset(const gram_model_t* m, int B, int N, int L, int K, int max_length) {
  if (check_shapes(m, B, N, L, K, max_length)) return GRAM_E_ARG;
  const Workspace w = carve(m, (void*)256, B, N, L, K, max_length);  // (any non-null base: only the offset is wanted)
  return (int64_t)((const char*)w.x - (const char*)256);
}

extern "C" int gram_encode_fused(const gram_model_t* m, const int64_t* input_ids, const uint8_t* mask, int B, int N, int L,
                                 void* workspace, int64_t workspace_bytes, int K, int max_length, void* enc_out_bf16,
                                 void* stream) {
  TRY(check_shapes(m, B, N, L, K, max_length));
  Workspace w = carve(m, workspace, B, N, L, K, max_length);
  if (!workspace || workspace_bytes < w.bytes) return GRAM_E_WORKSPACE;
  TRY(encode(m, w, input_ids, mask, mask, B, N, L, B * N, nullptr, CachedPassages{0, 0, nullptr, nullptr}, stream));
  if (enc_out_bf16) {  // (split modes: all the pieces, [pieces][B*N*L][d])
    hipError_t e = hipMemcpyAsync(enc_out_bf16, w.h, (size_t)w.pieces * B * N * L * m->d.d_model * sizeof(p16), hipMemcpyDeviceToDevice,  // (interleaved rows)
                                  (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
  }
  return 0;
}

extern "C" int gram_encode_passages(const gram_model_t* m, const int64_t* ids, const uint8_t* mask, int P, int L, void* workspace,
                                    int64_t workspace_bytes, float* x_out, void* stream) {
  TRY(check_shapes(m, P, 1, L, 1, 2));
  if (!ids || !mask || !x_out) return GRAM_E_ARG;
  Workspace w = carve(m, workspace, P, 1, L, 1, 2);
  if (!workspace || workspace_bytes < w.bytes) return GRAM_E_WORKSPACE;
  TRY(encoder_layers(m, w, ids, mask, L, P, stream));
  hipError_t e = hipMemcpyAsync(x_out, w.x, (size_t)P * L * m->d.d_model * sizeof(float), hipMemcpyDeviceToDevice,
                                (hipStream_t)stream);
  return e == hipSuccess ? 0 : (int)e;
}

extern "C" int gram_decode_step(const gram_model_t* m, const int32_t* tokens, const int32_t* anc, const uint8_t* mask, int B, int N,
                                int L, int K, int max_length, int t, void* workspace, int64_t workspace_bytes, float* logits,
                                void* stream) {
  TRY(check_shapes(m, B, N, L, K, max_length));
  if (t < 0 || t >= max_length - 1 || !logits) return GRAM_E_ARG;
  Workspace w = carve(m, workspace, B, N, L, K, max_length);
  if (!workspace || workspace_bytes < w.bytes) return GRAM_E_WORKSPACE;
  return decode_step(m, w, tokens, anc, mask, B, N, L, K, B * K, max_length, t, logits, nullptr, nullptr, stream);
}

extern "C" int gram_generate(const gram_model_t* m, const int64_t* input_ids, const uint8_t* mask, int B, int N, int L, int K,
                             int nret, int max_length, float length_penalty, const gram_trie_t* trie, void* workspace,
                             int64_t workspace_bytes, int64_t* sequences, float* scores, int32_t* width_host, void* stream) {
  return gram_generate_ex(m, input_ids, mask, B, N, L, K, nret, max_length, length_penalty, trie, nullptr, workspace, workspace_bytes,
                          sequences, scores, width_host, stream);
}

namespace {
// encode -> search -> finalize of one generate() on `stream`.
// (Replaying a small batch's launch train from a HIP graph was built in round 2 and measured in three rounds -- B = 1: 20.1 vs 20.2 ms,
// 14.1 vs 12.8 ms, 11.5 vs 10.65 ms eager: the chain is bound by the GPU-side dependency between ~950 tiny kernels, not by the host's
// launch cost, and a captured train cannot take the live-row step, whose row counts travel through the host -- and removed in round 4.)
int generate_body(const gram_model* m, Workspace& w, const int64_t* input_ids, const uint8_t* mask, int B, int N, int L, int K, int nret,
                  int max_length, const gram_trie_t* trie, const gram_compaction_t* comp, int64_t* sequences, float* scores,
                  void* stream) {
  if (comp)  // the encoder runs on the active passages only; padded ones leave their bank positions untouched (never read)
    TRY(encode(m, w, comp->ids, comp->mask, mask, B, N, L, comp->n_active, comp->passage_map,
               CachedPassages{comp->n_cached, comp->cache_L, comp->cache_x, comp->cache_slot}, stream));
  else
    TRY(encode(m, w, input_ids, mask, mask, B, N, L, B * N, nullptr, CachedPassages{0, 0, nullptr, nullptr}, stream));
  TRY(gram_beam_init(&w.beam, trie, /*decoder_start_token_id=*/0, stream));
  if (K == 1) {  // HF: num_beams == 1 -> greedy_search (raw logits, no hypotheses, no scores)
    for (int t = 0; t + 1 < max_length; ++t) {
      TRY(decode_step(m, w, w.beam.tokens, w.beam.anc, mask, B, N, L, 1, B, max_length, t, w.logits, nullptr, nullptr, stream));
      TRY(gram_greedy_step(&w.beam, trie, w.logits, m->d.vocab, t + 1, stream));
    }
    TRY(gram_greedy_finalize(&w.beam, max_length, sequences, w.width, stream));
  } else {
  static const bool live_rows_env = [] {
    const char* e = getenv("GRAM_LIVE_ROWS");
    return !(e && e[0] == '0');
  }();
  const bool live_rows = g_live_rows < 0 ? live_rows_env : g_live_rows != 0;  // (the live-row step needs a host round trip)
  // fixed max_length-1 steps: finished users are padded exactly as BeamSearchScorer.process
  // pads them, so skipping HF's all-done early exit changes nothing and needs no host sync
  for (int t = 0; t + 1 < max_length; ++t) {
    // step 0: every beam of a user holds the same start token and an empty cache, so the decoder,
    // the cross-attention and the lm_head run on ONE row per user (HF runs K identical rows);
    // gram_beam_step reads that shared row and points every beam's slot-0 ancestor at it
    const int Kt = t == 0 ? 1 : K;
    // From the step after the shortest candidate's EOS on, beams have left the Trie (gram_live_rows_t): the decoder
    // runs on the live rows only.  This costs the loop's one host round trip per such step (8 bytes), which is why
    // it is tried only where the Trie says rows can be dead -- with ids of l or l+1 pieces, the last step.
    if (live_rows && t >= 1 && trie->min_seq_len >= 2 && t >= trie->min_seq_len - 1) {
      int32_t counts[2] = {0, 0};
      TRY(gram_live_rows(&w.beam, trie, &w.live, stream));
      hipError_t e = hipMemcpyAsync(counts, w.live.counts, sizeof(counts), hipMemcpyDeviceToHost, (hipStream_t)stream);
      if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
      if (e != hipSuccess) return (int)e;
      if (counts[0] < 0 || counts[0] > B * K || counts[1] < 0 || counts[1] > B) return GRAM_E_BEAM;
      if (counts[0] < B * K) {
        if (counts[0] > 0) {
          const LiveStep live{counts[0], counts[1], w.live.rows, w.live.rowpos, w.live.users};
          TRY(decode_step(m, w, w.live.tokens, w.beam.anc, mask, B, N, L, K, B * K, max_length, t, nullptr, w.lse_part, &live,
                          stream));
          TRY(gram_lse_combine(w.lse_part, w.lse, counts[0], m->d.vocab / 64, stream));
        }  // else: no beam can be extended; the search step below reads no decoder row
        TRY(search_step(m, w, trie, t + 1, K, w.live.rowpos, stream));
        continue;
      }
    }
    // the [rows][V] logits are never written: LSE partials from the lm_head epilogue + sparse logits in the beam kernel
    TRY(decode_step(m, w, w.beam.tokens, w.beam.anc, mask, B, N, L, Kt, B * K, max_length, t, nullptr, w.lse_part, nullptr,
                    stream));
    TRY(gram_lse_combine(w.lse_part, w.lse, B * Kt, m->d.vocab / 64, stream));
    TRY(search_step(m, w, trie, t + 1, Kt, nullptr, stream));
  }
  TRY(gram_beam_finalize(&w.beam, nret, max_length, sequences, scores, w.width, stream));
  }
  return 0;
}

}  // namespace

extern "C" int gram_generate_ex(const gram_model_t* m, const int64_t* input_ids, const uint8_t* mask, int B, int N, int L, int K,
                                int nret, int max_length, float length_penalty, const gram_trie_t* trie,
                                const gram_compaction_t* comp, void* workspace, int64_t workspace_bytes, int64_t* sequences,
                                float* scores, int32_t* width_host, void* stream) {
  TRY(check_shapes(m, B, N, L, K, max_length));
  if (comp) {
    const int n_enc = comp->n_active - comp->n_cached;
    if (comp->n_active < B || comp->n_active > B * N || !comp->passage_map || comp->n_cached < 0 || n_enc < 0) return GRAM_E_ARG;
    if (n_enc > 0 && (!comp->ids || !comp->mask)) return GRAM_E_ARG;
    if (comp->n_cached > 0 && (!comp->cache_x || !comp->cache_slot || comp->cache_L < 1)) return GRAM_E_ARG;
  }
  if (!trie || nret < 1 || nret > K || !sequences || (!scores && K != 1)) return GRAM_E_ARG;
  if ((long long)K * trie->max_fanout > 16384) return GRAM_E_ARG;
  Workspace w = carve(m, workspace, B, N, L, K, max_length);
  if (!workspace || workspace_bytes < w.bytes) return GRAM_E_WORKSPACE;
  w.beam.length_penalty = length_penalty;
  TRY(generate_body(m, w, input_ids, mask, B, N, L, K, nret, max_length, trie, comp, sequences, scores, stream));
  if (width_host) {
    int32_t host[2] = {0, 0};
    hipError_t e = hipMemcpyAsync(&host[0], w.width, sizeof(int32_t), hipMemcpyDeviceToHost, (hipStream_t)stream);
    if (e == hipSuccess) e = hipMemcpyAsync(&host[1], w.beam.error, sizeof(int32_t), hipMemcpyDeviceToHost, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
    *width_host = host[0];
    if (host[1] == 4) return GRAM_E_NONFINITE;
    if (host[1] != 0) return GRAM_E_BEAM;
  }
  return 0;
}
