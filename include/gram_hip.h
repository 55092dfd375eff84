/*
 * gram_hip.h -- C ABI of libgram_hip.so: GRAM's multi-granular late-fusion generative scoring
 * path on MI355X (gfx950), hand-written HIP kernels.
 *
 * The reference (zhaodong-liu/GRAM) is pure Python and has no FFI; its seam for this path is
 *   src/model/gram.py:74-107        GRAM.generate(input_ids, attention_mask, max_length, **hf_kwargs)
 *   src/runner/single_runner_gram.py:641-651   the call site (beam=K, top-K, Trie closure)
 * Each entry point below names the reference function(s) it replaces.  INTEGRATION.md shows the
 * ctypes stub a maintainer adds on the reference side.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host
 *   - `stream` is a hipStream_t passed as void*; all work is stream-ordered, nothing allocates,
 *     nothing synchronises (except gram_generate's final 4-byte width read when asked)
 *   - "bf16" in a name or a comment = the library's 16-bit operand type, row-major: IEEE HALF in the default build, bfloat16 only in
 *     the `make PIECE=bf16` A/B build (gram_piece_format() tells; the names predate the switch).  Every `_bf16` entry point has an
 *     `_f16` alias (end of this header) that runs only in a library built on IEEE half -- bind those.
 *   - "inner" = n_heads * 64; d_kv must be 64
 *   - return value: 0 on success, >0 a hipError_t, <0 an argument error (GRAM_E_*)
 */
#ifndef GRAM_HIP_H
#define GRAM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GRAM_ABI_VERSION 7

#define GRAM_E_ARG (-1)       /* bad shape / unsupported size                           */
#define GRAM_E_WORKSPACE (-2) /* workspace too small (see gram_workspace_bytes)          */
#define GRAM_E_BEAM (-3)      /* device beam bookkeeping flagged an impossible state      */
#define GRAM_E_NONFINITE (-4) /* a returned score is NaN or +inf: an activation left the range of the 16-bit pieces (IEEE half:
                                 |x| <= 65 504) -- use the bfloat16 build (make PIECE=bf16)             */

#define GRAM_MAX_BEAMS 64      /* K <= 64 (reference default 50, headline 20)             */
#define GRAM_MAX_DEC_LEN 64    /* max_length <= 64 (reference: the longest candidate, 8..12, for the "split" / "t5_token" id
                                  types, 50 for "term": single_runner_gram.py:629-637); also the row stride of dec_bias_f32 */
#define GRAM_MAX_PASSAGE_LEN 128 /* L <= 128 (arguments.py:293-298), L % 32 == 0          */

/* ---- GEMM epilogues ------------------------------------------------------------------ */
enum gram_epilogue {
  GRAM_EPI_BF16 = 0,      /* C_bf16[m][n]  = acc                                          */
  GRAM_EPI_BF16_RELU = 1, /* C_bf16[m][n]  = max(acc, 0)        (T5DenseActDense wi+ReLU) */
  GRAM_EPI_F32_ADD = 2,   /* C_f32[m][n]  += acc                (residual add)            */
  GRAM_EPI_F32 = 3,       /* C_f32[m][n]   = acc                (lm_head logits)          */
  GRAM_EPI_KV_BANK = 4,   /* scatter into the fused K bank / V^T bank (see below)         */
  GRAM_EPI_F32_LSE = 5    /* GRAM_EPI_F32 + per-(row, 64-column block) softmax partials       */
};

/* Fused cross-attention KV bank for B users, S = N*L fused tokens, n_layers decoder layers:
 *   k  : [layer][b][h][s][64]   bf16   (key rows contiguous: streamed 128 B per key)
 *   vt : [layer][b][h][s/32][64][32] bf16 (V transposed and blocked by 32 keys: the 64 x 32 tile of a 32-key step is 4 KiB
 *                                    contiguous, fetched as whole 128-B lines like a K tile; S % 32 == 0)
 * Beam-invariant: ONE copy per user, shared by the user's K beams (the reference replicates it
 * K times and index_select-s it every step: gram_t5_modeling.py:531-549, gram_t5.py:320-348). */
typedef struct {
  void* k;
  void* vt;
  int32_t n_layers, B, H, S;
  /* optional compaction of fully padded passages: GEMM row m belongs to compact passage p = m / L, which is
   * passage n = passage_map[p] % N of user b = passage_map[p] / N (S = N*L).  NULL = rows are b*S + s. */
  const int32_t* passage_map;
  int32_t N, L;
} gram_kv_bank_t;

/* C[M,N] (+)= A[M,K] @ W[N,K]^T, bf16 operands, fp32 accumulate on MFMA.
 * Replaces every nn.Linear on the path (gram_t5_modeling.py:300-301,369-372; gram_t5.py:254).
 * N % 128 == 0, K % 64 == 0, any M >= 1.  For GRAM_EPI_KV_BANK, C is ignored, `bank` is used,
 * rows m = b*S + s and columns n = (layer*2 + which)*inner + h*64 + d. */
int gram_gemm_bf16(const void* A, const void* W, void* C, int M, int N, int K, int lda, int ldc,
                   int epilogue, const gram_kv_bank_t* bank_host, void* stream);

/* T5LayerNorm folded into the GEMMs on either side of it (no separate norm pass over the residual stream):
 *   producer (GRAM_EPI_F32_ADD):  besides x += acc it writes xb = bf16(x) and, per row and 64-column block,
 *                                 the partial sum of squares ss_out[m][n/64] (deterministic: no atomics)
 *   consumer (GRAM_EPI_BF16[_RELU]): A is xb, W has the norm gain folded in (W[n][k] * g[k]), and every output
 *                                 row is scaled by rsqrt(sum_b ss_in[m][b] / d + eps) before ReLU / rounding,
 * which equals Linear(T5LayerNorm(x)) (gram_t5_modeling.py:262-276 followed by :300-301 / :369-372). */
typedef struct {
  void* xb_out;        /* producer: bf16 [M][ldc] copy of the updated residual, or NULL            */
  float* ss_out;       /* producer: f32 [M][N/64] partial sums of squares, or NULL                 */
  const float* ss_in;  /* consumer: f32 [M][nblk_in] partials of the rows of A, or NULL            */
  int32_t nblk_in;     /* consumer: d_model / 64, or 0 if ss_in already holds 1/rms per row        */
  int32_t d;           /* consumer: d_model (the mean is over d elements)                          */
  float eps;
  int32_t quarter;     /* != 0: the partials cover 16 columns each -- ss_out [M][N/16], ss_in [M][4*nblk_in] -- and a block's
                          sum is (q0 + q1) + (q2 + q3), which is how the 64-column epilogues add it up themselves: same bits.
                          The layout of the small-M streaming GEMM (one 16-column n-tile per workgroup); only legal for
                          M <= gram_gemm_stream_max_m() and K % 128 == 0 (GRAM_E_ARG otherwise)                  */
  /* Range of the 16-bit copy.  T5's residual stream is unnormalised and leaves the IEEE-half range in trained checkpoints (the
   * reference's own T5 carries the fp16 clamp for it: gram_t5_modeling.py:773-776,803-808,824-827), so xb is stored as the pieces
   * of x[m] * xs[m], xs[m] a POWER OF TWO per row (exact), and the consumer's row scale is divided by it (exact):
   *   producer: xs_in f32 [M] -- xb_out[m] = pieces(x[m] * xs_in[m]); NULL = 1
   *   consumer: xs_in f32 [M] -- the factor the producer of A applied: every output row is scaled by rsqrt(..) / xs_in[m].  Ignored
   *             when nblk_in == 0 (ss_in then already holds rsqrt(..) / xs: gram_row_rscale_xs).
   *             xs_out f32 [M] or NULL -- the workgroups of the first n-tile also write the factor for the NEXT producer from the
   *             partials they have just added up: with typ^2 = (smallest 64-column partial of row m) / 64 -- the ordinary magnitude
   *             of the row; the few outlier features of a T5 row dominate its rms but not its quietest block -- the power of two
   *             that puts typ * xs into [2^-2, 2^-1), capped so that sqrt(sum of squares) * xs <= 2^10, clamped to [2^-40, 2^20].
   *             Two IEEE-half pieces then hold 22 bits of every element down to half the ordinary magnitude (error floor 2^-23 of
   *             it below) and up to 1.3e5 times it.  xs_out must not alias xs_in. */
  const float* xs_in;
  float* xs_out;
} gram_norm_fusion_t;
/* Largest M the streaming small-M GEMM takes (0: switched off, GRAM_GEMM_STREAM=0 or a forced variant). */
int gram_gemm_stream_max_m(void);
int gram_gemm_bf16_ex(const void* A, const void* W, void* C, int M, int N, int K, int lda, int ldc, int epilogue,
                      const gram_kv_bank_t* bank_host, const gram_norm_fusion_t* nf_host, void* stream);
/* rs[m] = rsqrt(sum_b ss[m][b] / d + eps); a consumer may take it directly with nblk_in = 0 (ss_in = rs). */
int gram_row_rscale(const float* ss, float* rs, int M, int nblk, int d, float eps, void* stream);
/* The same with the row factors of the 16-bit copy (gram_norm_fusion_t): rs[m] = rsqrt(..) / xs_in[m] (xs_in NULL = 1) and, when
 * xs_out is given, xs_out[m] = the factor for the next producer (see xs_out above).  xs_out may not alias xs_in. */
int gram_row_rscale_xs(const float* ss, float* rs, const float* xs_in, float* xs_out, int M, int nblk, int d, float eps, void* stream);
/* embed_tokens for the folded path: x (f32), xb = bf16(x), ss[m][b] = the sum of squares of 64-column block b when nblk == d / 64
 * (any other nblk: the row's total in ss[m][0], zeros behind it). */
int gram_embed_ex(const float* table, const void* ids, int ids_are_i64, float* x, void* xb, float* ss, int nblk, int rows,
                  int d, void* stream);

/* ---- two-piece operands: fp32-class arithmetic on the 16-bit MFMA ("f16x3" / "bf16x3") --------------------------------
 * The reference computes in fp32 end to end (no autocast / half anywhere in src/; fp32 softmax gram_t5_modeling.py:608).
 * One 16-bit piece per value moves Recall@5/NDCG@5 by ~5e-3 (bf16) on a 2 048-user population -- 50x the 1e-4 bound.  With
 * gram_split_t.pieces = 2 a value v travels as TWO 16-bit numbers
 *     p0 = r16(v),  p1 = r16(v - p0)              (r16 = round to the library's 16-bit type, gram_piece_format())
 * and a product a*w is evaluated on the 16-bit MFMA as three piece products accumulated in fp32, smallest first:
 *     a0*w1 + a1*w0 + a0*w0          relative error ~2^-22 with IEEE-half pieces (11 significant bits each), ~2^-18 with bf16
 * Layouts.  GEMM operands are INTERLEAVED: a logical [rows][K] matrix is stored as [rows][K/32][2][32] -- the two pieces of a
 * 32-column block side by side, 128 B -- so that one 64-column k-tile of the physical [rows][2K] matrix carries everything the
 * three products of that block need and is fetched once (inter(n, p) = (n / 32) * 64 + p * 32 + n % 32).  This holds for every
 * activation a GEMM reads (the normed / copied residual stream, attention outputs, the FFN intermediate) and for every weight
 * matrix.  What only attention kernels read -- Q/K/V rows, the KV bank, the self-attention cache -- stays PLANAR: `pieces`
 * copies of the plain layout, `*_pstride` elements apart.  pieces = 1 is the plain path (every *_split entry point accepts
 * split = NULL). */
#define GRAM_MAX_PIECES 2
typedef struct {
  int32_t pieces;         /* 1 or 2                                                                                   */
  int32_t c_interleaved;  /* a 16-bit C (GRAM_EPI_BF16[_RELU]) is written interleaved ([M][ldc >= 2N]) instead of planar  */
  int64_t c_pstride;      /* planar 16-bit C: elements between its pieces                                              */
  int64_t bank_pstride;   /* elements between the pieces of gram_kv_bank_t.k and .vt                                   */
  float out_scale;        /* every result is acc * out_scale; 0 = 1.  Must be a power of two (GRAM_E_ARG otherwise): the
                             inverse of the factor the caller scaled W by (gram_model_desc_t.w_scales)                  */
} gram_split_t;
/* gram_gemm_bf16_ex on two-piece operands: A is the interleaved [M][lda >= 2 kc] matrix, W the interleaved [N][2 kc] one, kc the
 * LOGICAL reduction length; 16-bit results are written as pieces (C planar or interleaved as split says; gram_norm_fusion_t.xb_out
 * always interleaved, [M][2 ldc]; the bank planar), fp32 results (residual stream, logits, LSE) as fp32. */
int gram_gemm_bf16_split(const void* A, const void* W, void* C, int M, int N, int kc, int lda, int ldc, int epilogue,
                         const gram_kv_bank_t* bank_host, const gram_norm_fusion_t* nf_host, const gram_split_t* split_host,
                         void* stream);
int gram_gemm_bf16_lse_split(const void* A, const void* W, float* logits, float* lse_part, int M, int N, int kc, int lda,
                             int ldc, const gram_split_t* split_host, void* stream);

/* The other kernels on two-piece operands.  Masks, bias tables, fp32 tensors and integer state are as in the plain entry points.
 * Outputs that feed a GEMM (xb, the norm's out, the attention outputs) are interleaved ([rows][2 * cols]) when pieces = 2; Q/K/V
 * inputs, the bank and the cache are planar, *_pstride elements apart. */
int gram_embed_ex_split(const float* table, const void* ids, int ids_are_i64, float* x, void* xb, float* ss, int nblk, int rows,
                        int d, int pieces, void* stream);
/* ... with the row factor of the 16-bit copy (gram_norm_fusion_t.xs_in of the consumer that follows): xs_out[m] = the factor of the
 * row's OWN rms, xb = pieces(x * xs_out[m]).  xs_out NULL = gram_embed_ex_split. */
int gram_embed_ex_xs(const float* table, const void* ids, int ids_are_i64, float* x, void* xb, float* ss, float* xs_out, int nblk,
                     int rows, int d, int pieces, void* stream);
int gram_rmsnorm_bf16_split(const float* x, const float* w, void* out_bf16, int rows, int d, float eps, float scale,
                            const float* pos, int N, int L, const int32_t* passage_map, int pieces, void* stream);
int gram_enc_self_attn_split(const void* qkv, const float* bias, const uint8_t* mask, void* out, int P, int L, int H,
                             int pieces, int64_t qkv_pstride, void* stream);
/* users/rowpos NULL: all B users, rows b*K + beam; else the live-row form (gram_cross_attn_decode_live, B = n_users).
 * key_bits: gram_mask_key_bits(mask) computed once per generate (the mask is the same for every head, layer and step), or
 * NULL: every workgroup packs its user's bits from the mask bytes itself. */
int gram_cross_attn_decode_split(const void* q, const void* k_layer, const void* vt_layer, const uint8_t* mask, void* out, int B,
                                 int K, int H, int S, const int32_t* users, const int32_t* rowpos, int pieces,
                                 int64_t q_pstride, int64_t bank_pstride, const uint32_t* key_bits, void* stream);
/* key_bits u32 [B][128]: bit j of word st = mask[b][32*st + j] != 0 (st < S/32). */
int gram_mask_key_bits(const uint8_t* mask, uint32_t* key_bits, int B, int S, void* stream);
/* rows NULL: all R rows; else the live-row form (gram_dec_self_attn_live). */
int gram_dec_self_attn_split(const void* qkv, void* kcache, void* vcache, const int32_t* anc, const float* bias, void* out,
                             int R, int n_rows, const int32_t* rows, int H, int t, int Tmax, int pieces, int64_t qkv_pstride,
                             int64_t cache_pstride, void* stream);
/* lm_head with the log-softmax normaliser fused: logits as GRAM_EPI_F32, plus for every row and every
 * 64-column block the pair (max, sum exp(x - max)) in lse_part f32 [M][N/64][2]; gram_lse_combine folds
 * them into lse[M] = log sum_v exp(logits[m][v]) without re-reading the logits (gram_row_lse does).
 * logits may be NULL: then only the partials are produced (see gram_beam_step_sparse). */
int gram_gemm_bf16_lse(const void* A, const void* W, float* logits, float* lse_part, int M, int N, int K, int lda,
                       int ldc, void* stream);
int gram_lse_combine(const float* lse_part, float* lse, int M, int nblk, void* stream);

/* x[row][:] = table[ids[row]][:]   (embed_tokens, gram_t5_modeling.py:1091). */
int gram_embed_i64(const float* table, const int64_t* ids, float* x, int rows, int d, void* stream);
int gram_embed_i32(const float* table, const int32_t* ids, float* x, int rows, int d, void* stream);

/* T5LayerNorm (gram_t5_modeling.py:262-276) with the two fusions the path needs:
 *   out_bf16[row][i] = bf16( x[row][i] * rsqrt(mean(x^2)+eps) * w[i] * scale
 *                            + (pos ? pos[(row / L) % N][i] : 0) )
 * scale = d_model^-0.5 for the tied lm_head (gram_t5.py:249-252); pos = per-passage position
 * embedding of the late fusion (gram.py:238-249). */
int gram_rmsnorm_bf16(const float* x, const float* w, void* out_bf16, int rows, int d, float eps,
                      float scale, const float* pos, int N, int L, void* stream);
/* same, for compacted passages: the passage of row r is passage_map[r / L] % N (see gram_kv_bank_t) */
int gram_rmsnorm_bf16_map(const float* x, const float* w, void* out_bf16, int rows, int d, float eps, float scale,
                          const float* pos, int N, int L, const int32_t* passage_map, void* stream);

/* Encoder self-attention for P passages of L tokens (T5Attention.forward, bidirectional,
 * gram_t5_modeling.py:479-631): unscaled QK^T + bucketed relative bias + key mask, fp32
 * softmax, @V.  qkv: bf16 [P*L][3*inner] (q|k|v); bias: f32 [H][255] indexed by
 * (key - query + 127); mask: u8 [P][L] (1 = valid); out: bf16 [P*L][inner]. */
int gram_enc_self_attn(const void* qkv, const float* bias, const uint8_t* mask, void* out, int P,
                       int L, int H, void* stream);

/* Late-fusion cross-attention for one decoder layer and one decode step (the fusion read:
 * T5Attention.forward cross branch, gram_t5_modeling.py:531-534,547-549,572-622; zero position
 * bias :577-582; mask :1145-1147).  q: bf16 [B*K][inner] (row = b*K + beam); k/vt: that layer's
 * slices of the bank; mask: u8 [B][S]; out: bf16 [B*K][inner].  S % 32 == 0, K <= 64. */
int gram_cross_attn_decode(const void* q, const void* k_layer, const void* vt_layer,
                           const uint8_t* mask, void* out, int B, int K, int H, int S, void* stream);

/* Decoder causal self-attention for the token at position t with a slot cache and beam-parent
 * indirection instead of the reference's torch.cat + index_select (gram_t5_modeling.py:536-540,
 * 586-593; gram_t5.py:320-348).  qkv: bf16 [R][3*inner]; kcache/vcache: bf16 [Tmax][R][inner]
 * (this layer); anc: i32 [Tmax][R], anc[j][r] = row whose step-j K/V is r's ancestor (j < t);
 * bias: f32 [H][GRAM_MAX_DEC_LEN] indexed by distance t-j.  Writes this step's k,v into slot t. */
int gram_dec_self_attn(const void* qkv, void* kcache, void* vcache, const int32_t* anc,
                       const float* bias, void* out, int R, int H, int t, int Tmax, void* stream);

/* lse[r] = log(sum_v exp(logits[r][v]))   (the normaliser of HF's log_softmax in beam_search). */
int gram_row_lse(const float* logits, float* lse, int R, int V, void* stream);

/* Flat Trie in HBM (CSR), built from generation_trie.py:5-68's nested dict by
 * gram_amd.utils.generation_trie.FlatTrie.  Children of node n are
 * child_tok/child_node[child_off[n] .. child_off[n+1]), sorted by token.  Node 0 is the root. */
typedef struct {
  const int32_t* child_off;
  const int32_t* child_tok;
  const int32_t* child_node;
  int32_t n_nodes, n_edges, max_fanout;
  int32_t min_seq_len; /* shortest candidate sequence, start token and EOS included; 0 = unknown (gram_generate then
                          never tries the live-row compaction below) */
} gram_trie_t;

/* Beam-search state for B users x K beams (HF transformers 4.26 beam_search + BeamSearchScorer
 * + BeamHypotheses state, kept on the device).  All arrays are caller-allocated. */
typedef struct {
  int32_t B, K, Tmax;      /* Tmax = max_length                                           */
  float length_penalty;
  int32_t eos, pad;
  int32_t* tokens;         /* [R]        next decoder input token per row                  */
  int32_t* node;           /* [R]        Trie node of each beam's prefix, -1 = not in Trie */
  float* beam_scores;      /* [R]                                                        */
  int32_t* seq;            /* [R][Tmax]  input_ids of each beam                           */
  int32_t* anc;            /* [Tmax][R]  self-attention ancestor table                    */
  int32_t* done;           /* [B]                                                        */
  int32_t* n_hyps;         /* [B]                                                        */
  double* hyp_score;       /* [B][K+1]   length-normalised score (Python float semantics);
                              one spare slot: BeamHypotheses.add appends, then drops the worst */
  double* worst;           /* [B]                                                        */
  int32_t* hyp_len;        /* [B][K+1]                                                   */
  int32_t* hyp_tok;        /* [B][K+1][Tmax]                                             */
  int32_t* error;          /* [1]        set non-zero on an impossible state              */
  /* optional scratch (NULL / 0 = off): with B <= cand_logits_users (and <= 16) users a sparse search step computes its candidates'
   * logits with a kernel of its own over many CUs before the per-user search kernel runs -- one workgroup per user pulls up to
   * K * fan-out lm_head rows through ONE CU, most of a one-user step.  f32 [cand_logits_users][cand_logits_stride],
   * cand_logits_stride >= the power of two >= K * max_fanout.  Same values either way. */
  float* cand_logits;
  int32_t cand_logits_users;
  int64_t cand_logits_stride;
} gram_beam_state_t;

/* decoder_input_ids = [[start]]*B*K ; beam_scores = [0,-1e9,...] (HF 4.26 beam_search init). */
int gram_beam_init(const gram_beam_state_t* st_host, const gram_trie_t* trie_host, int start_token,
                   void* stream);

/* One search step, cur_len = number of tokens already in each beam:
 *   log_softmax (via lse) -> PrefixConstrainedLogitsProcessor (Trie children; -inf elsewhere)
 *   -> + beam_scores -> top-2K over K*V (ties: lower flat index first) -> BeamSearchScorer.process
 *   -> next tokens / scores / parents, sequences and the ancestor table advanced in place.
 * Replaces HF 4.26 beam_search's per-step body and generation_trie.py:89-95's per-beam Python
 * callback (one D2H sync per beam per step in the reference).
 * rows_per_user = K normally (logits/lse have one row per beam); 1 when the K beams of a user share
 * one logits row and one self-attention cache row (step 0, where all beams are identical). */
int gram_beam_step(const gram_beam_state_t* st_host, const gram_trie_t* trie_host, const float* logits,
                   const float* lse, int V, int cur_len, int rows_per_user, void* stream);

/* Same step without a materialised logits tensor: the logits of the allowed tokens are recomputed inside the
 * kernel as hidden[row] . lm_head[tok] (bf16 operands, fp32 accumulate), `lse` coming from
 * gram_gemm_bf16_lse(logits = NULL) + gram_lse_combine.  hidden: bf16 [rows][d] (the lm_head A operand). */
int gram_beam_step_sparse(const gram_beam_state_t* st_host, const gram_trie_t* trie_host, const void* hidden_bf16,
                          const void* lm_head_bf16, int d, const float* lse, int V, int cur_len, int rows_per_user,
                          void* stream);

/* Live rows.  When a beam emits EOS its hypothesis moves to the heap and HF refills the slot with a -inf candidate
 * (BeamSearchScorer.process); such a row, and every row of a finished user, can only produce -inf candidates from then
 * on, so nothing downstream reads its decoder output -- but the reference still runs the decoder on it.  With item
 * ids of l or l+1 pieces most rows of the last step are in that state.  gram_live_rows compacts the
 * others (ascending, so grouped by user); the *_live variants of the per-row kernels then work on compact rows while
 * the self-attention cache, the ancestor table, the bank and the beam state keep their original indexing.
 * Results are bit-identical to running every row (every kernel of the step is row-independent). */
typedef struct {
  int32_t* rows;   /* [R]  original row of compact row i                                   */
  int32_t* rowpos; /* [R]  compact row of original row r, -1 = not live                    */
  int32_t* users;  /* [B]  users owning at least one live row, ascending                   */
  int32_t* tokens; /* [R]  decoder input token of compact row i                            */
  int32_t* counts; /* [2]  number of live rows, number of such users                       */
} gram_live_rows_t;
int gram_live_rows(const gram_beam_state_t* st_host, const gram_trie_t* trie_host, const gram_live_rows_t* out_host,
                   void* stream);
/* gram_dec_self_attn on n_rows compact rows: qkv/out [n_rows][..] compact, cache/anc rows = rows[i] of R. */
int gram_dec_self_attn_live(const void* qkv, void* kcache, void* vcache, const int32_t* anc, const float* bias,
                            void* out, int R, int n_rows, const int32_t* rows, int H, int t, int Tmax, void* stream);
/* gram_cross_attn_decode for the n_users users in `users`; q/out rows = rowpos[user*K + beam] (skipped if -1). */
int gram_cross_attn_decode_live(const void* q, const void* k_layer, const void* vt_layer, const uint8_t* mask,
                                void* out, int n_users, const int32_t* users, const int32_t* rowpos, int K, int H,
                                int S, void* stream);
/* gram_beam_step_sparse with hidden/lse indexed by compact row (all B users are stepped). */
int gram_beam_step_sparse_live(const gram_beam_state_t* st_host, const gram_trie_t* trie_host, const void* hidden_bf16,
                               const void* lm_head_bf16, int d, const float* lse, int V, int cur_len,
                               const int32_t* rowpos, void* stream);

/* gram_beam_step_sparse[_live] with the hidden state as pieces (interleaved [rows][2 d] when pieces = 2): the allowed logits are
 * h . E[tok] in fp32, h the fp32 sum of the pieces and E the FP32 lm_head table [V][d] (rowpos NULL: rows b*K + beam,
 * rows_per_user as in gram_beam_step). */
int gram_beam_step_sparse_split(const gram_beam_state_t* st_host, const gram_trie_t* trie_host, const void* hidden_bf16,
                                const float* lm_head_f32, int d, const float* lse, int V, int cur_len, int rows_per_user,
                                const int32_t* rowpos, int pieces, void* stream);

/* HF 4.26 greedy_search (generate with num_beams == 1; BASELINE configs[0]) on the same state with K = 1:
 * argmax of the RAW logits over the Trie children (first maximum), finished users emit pad; finalize copies
 * the sequences (i64 [B][max_length]) and reports the width HF would return (it stops once all rows hit EOS). */
int gram_greedy_step(const gram_beam_state_t* st_host, const gram_trie_t* trie_host, const float* logits, int V,
                     int cur_len, void* stream);
int gram_greedy_finalize(const gram_beam_state_t* st_host, int max_length, int64_t* sequences, int32_t* out_width,
                         void* stream);

/* BeamSearchScorer.finalize: sequences int64 [B*nret][Tmax] (0-padded, EOS appended when it
 * fits), scores f32 [B*nret], out_width[0] = min(max hyp len + 1, max_length). */
int gram_beam_finalize(const gram_beam_state_t* st_host, int nret, int max_length, int64_t* sequences,
                       float* scores, int32_t* out_width, void* stream);

/* Item index of each returned sequence.  Every hypothesis of a Trie-constrained search is a root-to-leaf path of the candidate Trie,
 * so what the runner does next per user -- tokenizer.batch_decode of the K generated id rows and a string comparison with the decoded
 * target (single_runner_gram.py:657-673, evaluate.py:5-22) -- needs only WHICH candidate each row spells: the strings come from one
 * batch_decode of the candidate list per evaluation.  sequences i64 [rows][T] as gram_beam_finalize / gram_generate wrote them (start
 * token first, 0-padded); node_item i32 [n_nodes]: candidate index of a leaf node (the first one if several candidates share a
 * sequence), anything for inner nodes.  out_item i32 [rows]: the index, or -1 when the row is not a candidate sequence followed by
 * padding (the -inf filler beams HF returns when fewer than nret hypotheses finished): the caller decodes those rows itself. */
int gram_trie_item_index(const gram_trie_t* trie_host, const int32_t* node_item, const int64_t* sequences, int rows, int T,
                         int32_t* out_item, void* stream);

/* ---- whole-path entry points ---------------------------------------------------------- */
typedef struct {
  int32_t vocab, d_model, d_ff, n_heads, n_enc_layers, n_dec_layers, max_passages;
  int32_t tie_word_embeddings, use_position_embedding;
  int32_t fold_norm; /* 1: enc_wqkv/enc_wi/dec_wqkv/dec_wq_x/dec_wi carry their layer-norm gain (gram_norm_fusion_t path) */
  float eps;
  const float* embed_f32;      /* [V][d]      shared.weight                                */
  const void* lm_head_bf16;    /* [V][d]      lm_head.weight                               */
  const float* pos_emb_f32;    /* [max_passages][d] or NULL                                */
  const float* enc_bias_f32;   /* [H][255]    encoder relative bias, layer-0 table         */
  const float* dec_bias_f32;   /* [H][GRAM_MAX_DEC_LEN] decoder relative bias by distance  */
  const float* enc_final_ln;   /* [d]                                                     */
  const float* dec_final_ln;   /* [d]                                                     */
  /* per-layer arrays (HOST arrays of device pointers) */
  const float* const* enc_ln1; /* [n_enc]  layer.0.layer_norm                              */
  const void* const* enc_wqkv; /* [n_enc]  bf16 [3*inner][d]  (q;k;v rows concatenated)    */
  const void* const* enc_wo;   /* [n_enc]  bf16 [d][inner]                                 */
  const float* const* enc_ln2; /* [n_enc]                                                 */
  const void* const* enc_wi;   /* [n_enc]  bf16 [d_ff][d]                                  */
  const void* const* enc_wo2;  /* [n_enc]  bf16 [d][d_ff]                                  */
  const float* const* dec_ln1; /* [n_dec]                                                 */
  const void* const* dec_wqkv; /* [n_dec]  bf16 [3*inner][d]                               */
  const void* const* dec_wo;   /* [n_dec]  bf16 [d][inner]                                 */
  const float* const* dec_ln2; /* [n_dec]                                                 */
  const void* const* dec_wq_x; /* [n_dec]  bf16 [inner][d]   EncDecAttention.q             */
  const void* const* dec_wo_x; /* [n_dec]  bf16 [d][inner]   EncDecAttention.o             */
  const float* const* dec_ln3; /* [n_dec]                                                 */
  const void* const* dec_wi;   /* [n_dec]  bf16 [d_ff][d]                                  */
  const void* const* dec_wo2;  /* [n_dec]  bf16 [d][d_ff]                                  */
  const void* dec_wkv_x_all;   /* bf16 [n_dec*2*inner][d]: per layer k rows then v rows     */
  /* two-piece mode (gram_split_t): pieces = 2 ("f16x3" / "bf16x3"); 0 / 1 = one piece.  Then EVERY 16-bit weight above is the
   * interleaved two-piece matrix [out][in / 32][2][32] and lm_head_f32 [V][d] must be given too (the beam kernel's sparse
   * logits).  fold_norm must be 1. */
  int32_t pieces;
  const float* lm_head_f32;
  /* Power-of-two factors the 16-bit weight matrices above were multiplied by before they were rounded (HOST array, or NULL = all
   * 1): [enc_wqkv x n_enc][enc_wo x n_enc][enc_wi x n_enc][enc_wo2 x n_enc][dec_wqkv x n_dec][dec_wo][dec_wq_x][dec_wo_x][dec_wi]
   * [dec_wo2 x n_dec][dec_wkv_x_all][lm_head]  (4 n_enc + 6 n_dec + 2 floats).  Every GEMM multiplies its result by the inverse
   * (gram_split_t.out_scale).  Used by the f16 build (gram_piece_format() == 1): an IEEE-half low piece of a small weight would
   * otherwise be subnormal and lose bits; bf16 has fp32's exponent range and needs none. */
  const float* w_scales;
} gram_model_desc_t;

typedef struct gram_model gram_model_t;

/* Copies the descriptor (not the weights).  Mirrors create_model("gram", config) +
 * load_state_dict (src/model/__init__.py:9-24, gram.py:162-165). */
gram_model_t* gram_model_create(const gram_model_desc_t* desc_host);
void gram_model_destroy(gram_model_t* m);

/* Bytes of scratch gram_generate needs for this problem size (256-B aligned carve). */
int64_t gram_workspace_bytes(const gram_model_t* m, int B, int N, int L, int K, int max_length);

/* EncoderWrapper.forward + the fused-bank projection (gram.py:200-256; gram_t5_modeling.py
 * T5Stack encoder role; cross K/V projection :531-534 for every decoder layer at once).
 * input_ids i64 [B][N][L], mask u8 [B][N][L].  Writes the bank into the workspace; when
 * enc_out_bf16 != NULL also copies the fused encoder states there (tests): [B*N*L][d], or in the two-piece mode the
 * interleaved [B*N*L][2 d]. */
int gram_encode_fused(const gram_model_t* m, const int64_t* input_ids, const uint8_t* mask, int B, int N,
                      int L, void* workspace, int64_t workspace_bytes, int K, int max_length,
                      void* enc_out_bf16, void* stream);

/* One cached decoder step for all B*K rows at position t, using the bank a preceding
 * gram_encode_fused left in the same workspace (T5ForConditionalGeneration_GRAM.forward with
 * encoder_outputs given, gram_t5.py:181-254).  tokens i32 [B*K]; anc as in gram_dec_self_attn;
 * logits f32 [B*K][V] (device, caller-owned). */
int gram_decode_step(const gram_model_t* m, const int32_t* tokens, const int32_t* anc, const uint8_t* mask,
                     int B, int N, int L, int K, int max_length, int t, void* workspace,
                     int64_t workspace_bytes, float* logits, void* stream);

/* GRAM.generate (gram.py:74-107) with the runner's kwargs (single_runner_gram.py:641-651):
 * encoder -> late fusion -> bank -> max_length-1 constrained beam-search steps -> finalize.
 * sequences i64 [B*nret][max_length], scores f32 [B*nret]; *width_host receives
 * min(max hyp len + 1, max_length) (the column count HF returns) if non-NULL, which costs one
 * stream synchronise.  Returns GRAM_E_BEAM if the device flagged an impossible beam state.
 * K == 1 follows HF's dispatch to greedy_search: `scores` may be NULL and is not written. */
int gram_generate(const gram_model_t* m, const int64_t* input_ids, const uint8_t* mask, int B, int N, int L,
                  int K, int nret, int max_length, float length_penalty, const gram_trie_t* trie_host,
                  void* workspace, int64_t workspace_bytes, int64_t* sequences, float* scores,
                  int32_t* width_host, void* stream);

/* Ragged batches: the Collator pads every user to the batch's largest passage count with fully masked passages
 * (Collator.py:410-436); the encoder need not run on those.  The caller passes the n_active passages that have
 * at least one valid token, gathered contiguously (ids/mask [n_active][L]), and their flat indices
 * passage_map[p] = b*N + n (ascending).  `mask` stays the full (B,N,L) mask: the cross-attention skips the
 * untouched bank positions of the padded passages.  Every user must keep >= 1 active passage. */
typedef struct {
  int32_t n_active;
  const int32_t* passage_map; /* device, i32 [n_active] */
  const int64_t* ids;         /* device, i64 [n_active - n_cached][L] */
  const uint8_t* mask;        /* device, u8  [n_active - n_cached][L] */
  /* Passage cache (all zero = off).  An item passage's encoder output depends neither on the
   * user nor on the slot it sits in: EncoderWrapper runs every passage on its own and adds the slot's position
   * embedding afterwards (gram.py:238-255).  The LAST n_cached entries of passage_map skip the encoder: their
   * residual-stream rows (gram_encode_passages) are gathered from cache_x[cache_slot[i]] instead; rows at
   * positions >= cache_L are zero-filled (they are padding, masked in the cross-attention).  passage_map need
   * not be ascending.  Results are bit-identical to encoding the passages in place. */
  int32_t n_cached;
  int32_t cache_L;            /* rows per cached passage */
  const float* cache_x;       /* device, f32 [slots][cache_L][d_model] */
  const int32_t* cache_slot;  /* device, i32 [n_cached] */
} gram_compaction_t;
int gram_generate_ex(const gram_model_t* m, const int64_t* input_ids, const uint8_t* mask, int B, int N, int L,
                     int K, int nret, int max_length, float length_penalty, const gram_trie_t* trie_host,
                     const gram_compaction_t* compaction_host, void* workspace, int64_t workspace_bytes,
                     int64_t* sequences, float* scores, int32_t* width_host, void* stream);

/* The encoder layers alone on P independent passages (T5Stack encoder role, gram_t5_modeling.py:1037-1296, up to
 * but excluding final_layer_norm): x_out f32 [P][L][d_model] is the residual stream the final norm, the position
 * embedding and the bank projection consume.  ids i64 [P][L], mask u8 [P][L].  Workspace as for
 * gram_workspace_bytes(m, P, 1, L, 1, 2).  Feeds gram_compaction_t.cache_x. */
int gram_encode_passages(const gram_model_t* m, const int64_t* ids, const uint8_t* mask, int P, int L,
                         void* workspace, int64_t workspace_bytes, float* x_out, void* stream);
/* Byte offset, inside a workspace carved for (B, N, L, K, max_length), of the encoder's fp32 residual stream [passages * L][d_model]
 * (before final_layer_norm) as the last gram_generate[_ex] / gram_encode_fused on that workspace left it: rows [p*L, (p+1)*L) belong to
 * compact passage p (the order of gram_compaction_t.ids; b*N + n without a compaction).  It is what gram_encode_passages copies out,
 * so a caller can add the passages a generate() has just encoded to its passage cache (gram_compaction_t.cache_x) without encoding
 * them a second time.  < 0: GRAM_E_ARG. */
int64_t gram_workspace_encoder_x_offset(const gram_model_t* m, int B, int N, int L, int K, int max_length);
/* x[i][l][:] = l < cache_L ? cache_x[slot[i]][l][:] : 0 for i < n, l < L (d % 4 == 0). */
int gram_gather_passage_x(const float* cache_x, const int32_t* slot, float* x, int n, int L, int cache_L, int d,
                          void* stream);

/* ---- live per-kernel timing (bench.py) ------------------------------------------------ */
enum gram_kernel_kind {
  GRAM_K_GEMM = 0,          /* work = 2*M*N*K flops per launch                              */
  GRAM_K_ENC_ATTN = 1,      /* work = 4*P*H*L*L*64 flops                                   */
  GRAM_K_CROSS_ATTN = 2,    /* work = algorithmic HBM bytes: B*H*S*64*2(K,V)*2 B           */
  GRAM_K_DEC_SELF_ATTN = 3, /* work = bytes of cached K/V read                              */
  GRAM_K_ROWOPS = 4,        /* embed / rmsnorm: work = bytes moved                          */
  GRAM_K_LSE = 5,           /* work = R*V*4 bytes                                          */
  GRAM_K_BEAM = 6,          /* work = 0                                                    */
  GRAM_K_COUNT = 7
};
/* Record a HIP-event pair around every launch of the selected kinds, on the launch stream.
 * kind_mask = OR of (1 << kind); 0 disables and frees the pool.  Not thread-safe. */
int gram_prof_enable(uint32_t kind_mask, int max_events);
int gram_prof_reset(void);
/* Sum of event-measured durations (ms), launch count and summed `work` for one kind; waits for
 * the recorded events.  dropped = launches not recorded because the pool was full. */
int gram_prof_collect(int kind, double* total_ms, int64_t* launches, double* work, int64_t* dropped);

/* Diagnostic, OFF by default (the product path runs without it): on != 0 makes every workgroup of the persistent ping-pong GEMM stamp
 * s_memtime / s_memrealtime around its tile loop (two atomics per workgroup and launch) for gram_prof_pp_clock.  Not thread-safe. */
int gram_prof_pp_clock_enable(int on);
/* Time-weighted shader clock (GHz) the chip held inside the persistent ping-pong GEMM launches since the last reset (while enabled,
 * see above): every workgroup of those kernels adds its s_memtime / s_memrealtime differences around its tile loop to two device counters.  The MFMA peak a
 * power-limited chip can be priced against is 2.5 PFLOP/s x this / 2.4 (bench.py's roofline).  Synchronises the device; reset != 0
 * zeroes the counters after reading.  ghz may be NULL. */
int gram_prof_pp_clock(double* ghz, int reset);

/* Tuning hook: force a GEMM staging variant (0 = register-staged double buffer, 1 = LDS-DMA
 * single buffer, -1 = automatic per problem size).  Used by tests/bench_gemm.py.  1000 + s: start stagger of the persistent kernel.
 * 2000 + n (n <= 63; 2000 = off): TEST hook, honoured by the chaos build only (make CHAOS=1; the product kernels carry no test code) --
 * wave group 1 of the persistent ping-pong kernel enters its prologue n x 512 cycles late, which turns a round-4 race of that prologue
 * (fixed) into a deterministic test. */
int gram_debug_set_gemm_variant(int variant);

/* Calibration probe (bench.py): one streaming read of `bytes` (16-B aligned) through every CU; nothing is
 * written unless a 32-bit fold of the data hits one magic value (sink may be NULL). */
int gram_debug_stream_read(const void* src, size_t bytes, void* sink, void* stream);
/* variants of the probe (tests/bench_stream.py): 0 = gram_debug_stream_read's kernel, 1 contiguous 16-KiB chunks per workgroup, 2 nontemporal,
 * 3 eight loads in flight, 4 LDS-DMA, 5 = 1 + nontemporal; wgs = workgroups of 256 threads */
int gram_debug_stream_read_variant(const void* src, size_t bytes, void* sink, int variant, int wgs, void* stream);

/* A/B hook (bench.py): 0 = decode every row in every step like the reference, 1 = live-row compaction (gram_live_rows_t),
 * -1 = what the GRAM_LIVE_ROWS environment variable says (default 1).  Results are bit-identical either way. */
int gram_debug_set_live_rows(int on);
/* Sensitivity sweeps over the split modes (tests/precision_population.py --sweep): stage s of a generate() computes on the first
 * caps[s] pieces of its operands only (the upper pieces of its activation operands are zeroed before use; the caller zeroes the
 * upper pieces of that stage's weights when it expands them).  caps NULL = no caps.  n must be GRAM_STAGE_COUNT. */
enum gram_stage {
  GRAM_STAGE_ENC_ATTN = 0,  /* encoder QKV GEMM, self-attention, O GEMM                    */
  GRAM_STAGE_ENC_FFN = 1,   /* encoder wi / wo GEMMs                                       */
  GRAM_STAGE_BANK_K = 2,    /* the stored K bank (and the K rows of the bank projection)    */
  GRAM_STAGE_BANK_V = 3,    /* the stored V^T bank (and the V rows of the bank projection)  */
  GRAM_STAGE_DEC_SELF = 4,  /* decoder QKV GEMM, cached self-attention, O GEMM             */
  GRAM_STAGE_DEC_CROSS = 5, /* decoder cross-attention q GEMM, the query side, O GEMM      */
  GRAM_STAGE_DEC_FFN = 6,   /* decoder wi / wo GEMMs                                       */
  GRAM_STAGE_LM_HEAD = 7,   /* final hidden state: lm_head GEMM (LSE) and the sparse logits */
  GRAM_STAGE_COUNT = 8
};
int gram_debug_set_stage_pieces(const int32_t* caps, int n);

/* `_f16` aliases of the entry points whose historical names say `_bf16`: same arguments; they forward in a library built on IEEE half
 * (gram_piece_format() == 1, the default) and return GRAM_E_ARG in the bfloat16 A/B build, so a binding by name cannot hand the
 * kernels the wrong 16-bit type. */
int gram_gemm_f16(const void* A, const void* W, void* C, int M, int N, int K, int lda, int ldc, int epilogue,
                  const gram_kv_bank_t* bank_host, void* stream);
int gram_gemm_f16_ex(const void* A, const void* W, void* C, int M, int N, int K, int lda, int ldc, int epilogue,
                     const gram_kv_bank_t* bank_host, const gram_norm_fusion_t* nf_host, void* stream);
int gram_gemm_f16_split(const void* A, const void* W, void* C, int M, int N, int kc, int lda, int ldc, int epilogue,
                        const gram_kv_bank_t* bank_host, const gram_norm_fusion_t* nf_host, const gram_split_t* split_host, void* stream);
int gram_gemm_f16_lse(const void* A, const void* W, float* logits, float* lse_part, int M, int N, int K, int lda, int ldc, void* stream);
int gram_gemm_f16_lse_split(const void* A, const void* W, float* logits, float* lse_part, int M, int N, int kc, int lda, int ldc,
                            const gram_split_t* split_host, void* stream);
int gram_rmsnorm_f16(const float* x, const float* w, void* out_f16, int rows, int d, float eps, float scale, const float* pos, int N,
                     int L, void* stream);
int gram_rmsnorm_f16_map(const float* x, const float* w, void* out_f16, int rows, int d, float eps, float scale, const float* pos,
                         int N, int L, const int32_t* passage_map, void* stream);
int gram_rmsnorm_f16_split(const float* x, const float* w, void* out_f16, int rows, int d, float eps, float scale, const float* pos,
                           int N, int L, const int32_t* passage_map, int pieces, void* stream);

int gram_abi_version(void);
/* 16-bit operand type this build of the library computes on: 0 = bfloat16, 1 = IEEE half ("f16": make PIECE=f16).  Every "bf16"
 * pointer of this header is a pointer to that type. */
int gram_piece_format(void);

#ifdef __cplusplus
}
#endif
#endif /* GRAM_HIP_H */
