// beam.hip -- Trie-constrained beam search state machine on the device.
//
// Replaces, per decode step and per user (reference call site gram.py:93-99, kwargs
// single_runner_gram.py:641-651; third-party transformers==4.26.0 semantics, restated in
// oracle/gram_oracle.py::beam_search):
//   log_softmax                      -> logits[tok] - lse[row]            (lse from rowops.hip)
//   PrefixConstrainedLogitsProcessor -> children of the beam's Trie node (flat CSR in HBM);
//                                       the per-beam Python callback of generation_trie.py:89-95
//                                       (one D2H sync per beam per step) disappears
//   + beam_scores, topk(2K) over K*V -> bitonic sort in LDS of the <= K*max_fanout finite
//                                       candidates (64-bit keys: orderable score | ~flat index,
//                                       so ties resolve to the lower flat index) + -inf fillers
//   BeamSearchScorer.process         -> one thread walks the 2K ranked candidates
//   input_ids gather / _reorder_cache-> sequences and the self-attention ancestor table are
//                                       advanced in place; K/V caches are never moved
// One workgroup per user; users are independent, so this shards trivially.
#include <stdlib.h>

#include "common.h"
#include "prof.h"

namespace {

__device__ __forceinline__ uint32_t f2ord(float f) {
  uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(uint32_t o) {
  uint32_t u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
  return __uint_as_float(u);
}

// index of tok among node's children, or -1
__device__ int find_child(const gram_trie_t& tr, int node, int tok) {
  if (node < 0) return -1;
  int lo = tr.child_off[node], hi = tr.child_off[node + 1];
  while (lo < hi) {
    int mid = (lo + hi) >> 1;
    int v = tr.child_tok[mid];
    if (v == tok) return mid;
    if (v < tok) lo = mid + 1; else hi = mid;
  }
  return -1;
}

__device__ double norm_len(int len, float lp) {
  if (lp == 1.0f) return (double)len;
  if (lp == 0.0f) return 1.0;
  return pow((double)len, (double)lp);
}

// BeamHypotheses.add (list semantics: append, drop the minimum, later elements shift down)
__device__ void hyp_add(const gram_beam_state_t& st, int b, const int32_t* toks, int len, float sum_logprobs) {
  const int K = st.K, T = st.Tmax;
  double* hs = st.hyp_score + (size_t)b * (K + 1);
  int32_t* hl = st.hyp_len + (size_t)b * (K + 1);
  int32_t* ht = st.hyp_tok + (size_t)b * (K + 1) * T;
  int n = st.n_hyps[b];
  const double score = (double)sum_logprobs / norm_len(len, st.length_penalty);
  // a NaN would lose every comparison below (never kept once the heap is full, evicted first otherwise) and vanish silently; +inf cannot
  // be a sum of log-probabilities: both mean an activation left the range of the 16-bit pieces upstream (GRAM_E_NONFINITE)
  if (!(score < 1.0e300)) st.error[0] = 4;
  if (n < K || score > st.worst[b]) {
    hs[n] = score;
    hl[n] = len;
    for (int i = 0; i < len; ++i) ht[(size_t)n * T + i] = toks[i];
    ++n;
    if (n > K) {
      // sorted([(s, idx)]): minimum score, ties -> lowest index; worst = second smallest
      int imin = 0;
      for (int i = 1; i < n; ++i)
        if (hs[i] < hs[imin]) imin = i;
      double second = 1e300;
      bool have = false;
      for (int i = 0; i < n; ++i) {
        if (i == imin) continue;
        if (!have || hs[i] < second) { second = hs[i]; have = true; }
      }
      for (int i = imin; i + 1 < n; ++i) {
        hs[i] = hs[i + 1];
        hl[i] = hl[i + 1];
        for (int p = 0; p < T; ++p) ht[(size_t)i * T + p] = ht[(size_t)(i + 1) * T + p];
      }
      --n;
      st.worst[b] = second;
    } else {
      st.worst[b] = score < st.worst[b] ? score : st.worst[b];
    }
    st.n_hyps[b] = n;
  }
}

__global__ void beam_init_kernel(gram_beam_state_t st, gram_trie_t tr, int start) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  const int R = st.B * st.K;
  if (r == 0) st.error[0] = 0;
  if (r < st.B) {
    st.done[r] = 0;
    st.n_hyps[r] = 0;
    st.worst[r] = 1e9;
  }
  if (r >= R) return;
  st.tokens[r] = start;
  st.beam_scores[r] = (r % st.K) == 0 ? 0.f : -1e9f;
  for (int p = 0; p < st.Tmax; ++p) {
    st.seq[(size_t)r * st.Tmax + p] = p == 0 ? start : 0;
    st.anc[(size_t)p * R + r] = r;
  }
  const int e = find_child(tr, 0, start);
  st.node[r] = e < 0 ? -1 : tr.child_node[e];
}

// h[row] . E[tok] on 8 lanes (sub = lane & 7): the same sums in the same order wherever a candidate is computed -- inside the search
// step's workgroup or, for a handful of users, by sparse_logits_kernel's many workgroups (launch_beam_step)
__device__ __forceinline__ float sparse_dot(bool act, int lr, int tok, int sub, const p16* __restrict__ hd, const p16* __restrict__ emb,
                                            const float* __restrict__ emb32, int d, int pieces) {
  const int per = d >> 3;  // elements per lane
    float acc = 0.f;
    if (act && emb32) {
      // two-piece mode (gram_split_t): h = fp32 sum of its pieces (smallest first; the row is interleaved, [2 d]), E = the fp32 lm_head row
      const p16* hrow = hd + (size_t)lr * d * pieces;
      const int c0 = sub * per;
      auto hoff = [&](int n, int pc) { return pieces == 2 ? inter_off(n, pc) : n; };
      const f32x4* ep = reinterpret_cast<const f32x4*>(emb32 + (size_t)tok * d + sub * per);
      int i = 0;
      if (pieces <= 2) {
        // 32 elements per lane and trip, every load of the trip in flight together (one 8-element step per trip is a dependent
        // round trip each: 12 of them per candidate at d = 768); same sums in the same order as the loop below
        for (; i + 32 <= per; i += 32) {
          p16x8 hb0[4], hb1[4];
          f32x4 ev[8];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            hb0[u] = ld_global_b128(hrow + hoff(c0 + i + 8 * u, 0));
            hb1[u] = pieces == 2 ? ld_global_b128(hrow + hoff(c0 + i + 8 * u, 1)) : zero_bf16x8();
            ev[2 * u] = ep[(i >> 2) + 2 * u];
            ev[2 * u + 1] = ep[(i >> 2) + 2 * u + 1];
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            float hv[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) hv[e] = pieces == 2 ? (0.f + (float)hb1[u][e]) + (float)hb0[u][e] : 0.f + (float)hb0[u][e];
#pragma unroll
            for (int e = 0; e < 4; ++e) acc += hv[e] * ev[2 * u][e];
#pragma unroll
            for (int e = 0; e < 4; ++e) acc += hv[4 + e] * ev[2 * u + 1][e];
          }
        }
      }
      for (; i < per; i += 8) {
        float hv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int pc = pieces - 1; pc >= 0; --pc) {
          const p16x8 hb = ld_global_b128(hrow + hoff(c0 + i, pc));
#pragma unroll
          for (int e = 0; e < 8; ++e) hv[e] += (float)hb[e];
        }
        const f32x4 e0 = ep[i >> 2], e1 = ep[(i >> 2) + 1];
#pragma unroll
        for (int e = 0; e < 4; ++e) acc += hv[e] * e0[e];
#pragma unroll
        for (int e = 0; e < 4; ++e) acc += hv[4 + e] * e1[e];
      }
    } else if (act) {
      const p16* hp = hd + (size_t)lr * d + sub * per;
      const p16* ep = emb + (size_t)tok * d + sub * per;
      int i = 0;
      for (; i + 32 <= per; i += 32) {  // 8 loads in flight per lane (one load pair per iteration is a dependent round trip each)
        p16x8 hv[4], ev[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          hv[u] = ld_global_b128(hp + i + 8 * u);
          ev[u] = ld_global_b128(ep + i + 8 * u);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int e = 0; e < 8; ++e) acc += (float)hv[u][e] * (float)ev[u][e];  // same order as the scalar loop
      }
      for (; i < per; i += 8) {
        const p16x8 hv = ld_global_b128(hp + i), ev = ld_global_b128(ep + i);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc += (float)hv[e] * (float)ev[e];
      }
    }
    acc += __shfl_xor(acc, 1, 64);
    acc += __shfl_xor(acc, 2, 64);
    acc += __shfl_xor(acc, 4, 64);
    return acc;
  }

// NTHR threads per workgroup (= per user): 256 for batches that fill the chip with workgroups; 1 024 for small batches, where one user's
// sparse logits (a trip computes NTHR / 4 candidates' dot products, each on its 8 lanes) and sort stages are the step's latency
template <int NTHR>
__global__ __launch_bounds__(NTHR) void beam_step_kernel(gram_beam_state_t st, gram_trie_t tr, const float* __restrict__ logits,
                                                        const float* __restrict__ lse, int V, int cur_len, int nc_max, int rows_per_user,
                                                        const p16* __restrict__ hd, const p16* __restrict__ emb, int d,
                                                        const int32_t* __restrict__ rowpos, int pieces,
                                                        const float* __restrict__ emb32, const float* __restrict__ pre) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  unsigned long long* keys = reinterpret_cast<unsigned long long*>(smem);  // [nc_max]
  float* s_log = reinterpret_cast<float*>(keys + nc_max);                   // [max_fanout, rounded up to 4] shared step-0 logits
  // [K][Tmax] + [Tmax][K]: the advanced sequences / ancestor table on their way back to HBM (sized by the call, not by the maxima)
  int* new_seq = reinterpret_cast<int*>(smem + (size_t)nc_max * 8 + ((size_t)tr.max_fanout * 4 + 15) / 16 * 16);
  int* new_anc = new_seq + st.K * st.Tmax;
  __shared__ int s_pre[GRAM_MAX_BEAMS + 1];
  __shared__ int s_C, s_NC, s_isdone;
  __shared__ float sel_score[GRAM_MAX_BEAMS];
  __shared__ int sel_tok[GRAM_MAX_BEAMS], sel_par[GRAM_MAX_BEAMS], sel_node[GRAM_MAX_BEAMS];
  __shared__ int s_edge[2 * GRAM_MAX_BEAMS];
  __shared__ int s_off[GRAM_MAX_BEAMS], s_cnt[GRAM_MAX_BEAMS], s_lr[GRAM_MAX_BEAMS];

  const int b = blockIdx.x, tid = threadIdx.x;
  const int K = st.K, T = st.Tmax, R = st.B * K;
  const int row0 = b * K;
  const int t = cur_len - 1;  // decode step whose K/V were just written

  // per beam: first child edge, child count, hidden-state row -- K threads at once (one thread walking the K beams was K dependent
  // node -> offsets round trips, ~20 us of a 100-us step at K = 20)
  if (tid < K) {
    const int nd = st.node[row0 + tid];
    const int done = st.done[b];
    int o0 = 0, cnt = 0;
    if (!done && nd >= 0) {
      o0 = tr.child_off[nd];
      cnt = tr.child_off[nd + 1] - o0;
    }
    s_off[tid] = o0;
    s_cnt[tid] = cnt;
    s_lr[tid] = rowpos ? rowpos[row0 + tid] : row0 + tid;  // live-row step: hidden/lse are indexed by compact row
    if (tid == 0) s_isdone = done;
  }
  gram_sync();
  if (tid == 0) {
    int acc = 0;
    for (int k = 0; k < K; ++k) {
      s_pre[k] = acc;
      acc += s_cnt[k];
    }
    s_pre[K] = acc;
    s_C = acc;
    int nc = 64;
    while (nc < acc) nc <<= 1;
    s_NC = nc;
    if (acc > nc_max) { st.error[0] = 2; s_C = 0; s_NC = 64; s_pre[K] = 0; }
  }
  gram_sync();
  const int C = s_C, NC = s_NC;
  const bool isdone = s_isdone != 0;

  if (!isdone && logits == nullptr) {
    // SPARSE mode: the lm_head GEMM stored only the softmax partials (lse); the logits of the <= K*fan-out
    // allowed tokens are recomputed here as h[row] . E[tok] (bf16 operands, fp32 accumulate; 8 lanes per
    // candidate, 16-byte loads).  The [rows][V] logits tensor (5 GB per step at B = 2048) is never written.
    for (int ci = C + tid; ci < NC; ci += NTHR) keys[ci] = 0ull;
    const int sub = tid & 7, grp = tid >> 3;
    const bool shared0 = rows_per_user == 1;  // step 0: all K beams sit on the same node and the same row
    const int nuniq = shared0 ? s_pre[1] : C;
    // (beam, token) of every candidate first, all threads at once, parked in the candidate's key slot: the dot products below then
    // start from LDS instead of a node -> edge -> token chain of global loads per batch
    for (int ci = tid; ci < nuniq; ci += NTHR) {
      int k = 0;
      if (!shared0)
        while (s_pre[k + 1] <= ci) ++k;
      const int tok = tr.child_tok[s_off[k] + (ci - s_pre[k])];
      keys[ci] = ((unsigned long long)(uint32_t)k << 32) | (unsigned long long)(uint32_t)tok;
    }
    gram_sync();
    auto dot = [&](bool act, int lr, int tok, int ci) -> float {
      if (pre) return act ? pre[(size_t)b * nc_max + ci] : 0.f;  // (computed by sparse_logits_kernel: the same function, the same bits)
      return sparse_dot(act, lr, tok, sub, hd, emb, emb32, d, pieces);
    };
    // two candidates per 8-lane group and trip (NTHR / 4 per workgroup): their loads are independent and overlap
    constexpr int NG = NTHR / 8;
    for (int base = 0; base < nuniq; base += 2 * NG) {
      int ci2[2], k2[2], tok2[2], lr2[2];
      bool act2[2];
#pragma unroll
      for (int w = 0; w < 2; ++w) {
        ci2[w] = base + NG * w + grp;
        act2[w] = ci2[w] < nuniq;
        const unsigned long long kt = act2[w] ? keys[ci2[w]] : 0ull;
        k2[w] = (int)(kt >> 32);
        tok2[w] = (int)(kt & 0xffffffffull);
        lr2[w] = shared0 ? b : s_lr[k2[w]];
      }
      float acc2[2];
#pragma unroll
      for (int w = 0; w < 2; ++w) acc2[w] = dot(act2[w], lr2[w], tok2[w], ci2[w]);
#pragma unroll
      for (int w = 0; w < 2; ++w) {
        if (act2[w] && sub == 0) {
          if (shared0) {
            s_log[ci2[w]] = acc2[w];
          } else {
            const float sc = (acc2[w] - lse[lr2[w]]) + st.beam_scores[row0 + k2[w]];
            keys[ci2[w]] = ((unsigned long long)f2ord(sc) << 32) | (unsigned long long)(0xffffffffu - (uint32_t)(k2[w] * V + tok2[w]));
          }
        }
      }
    }
    if (shared0) {
      gram_sync();
      const int cnt0 = s_pre[1];
      const int off0 = cnt0 > 0 ? tr.child_off[st.node[row0]] : 0;
      for (int ci = tid; ci < C; ci += NTHR) {
        const int k = ci / cnt0, j = ci - k * cnt0;
        const int tok = tr.child_tok[off0 + j];
        const float sc = (s_log[j] - lse[b]) + st.beam_scores[row0 + k];
        keys[ci] = ((unsigned long long)f2ord(sc) << 32) | (unsigned long long)(0xffffffffu - (uint32_t)(k * V + tok));
      }
    }
    gram_sync();
  }
  if (!isdone && logits != nullptr) {
    // gather: log_softmax at the allowed tokens + running beam score
    for (int ci = tid; ci < NC; ci += NTHR) {
      unsigned long long key = 0ull;
      if (ci < C) {
        int k = 0;
        while (s_pre[k + 1] <= ci) ++k;
        const int r = row0 + k;
        const int e = tr.child_off[st.node[r]] + (ci - s_pre[k]);
        const int tok = tr.child_tok[e];
        // rows_per_user == 1: the K beams of a user share one logits row (step 0: identical beams)
        const int lr = rows_per_user == 1 ? b : r;
        const float sc = (logits[(size_t)lr * V + tok] - lse[lr]) + st.beam_scores[r];
        key = ((unsigned long long)f2ord(sc) << 32) | (unsigned long long)(0xffffffffu - (uint32_t)(k * V + tok));
      }
      keys[ci] = key;
    }
    gram_sync();
  }
  if (!isdone) {
    // Non-finite arithmetic is flagged HERE, at its source (GRAM_E_NONFINITE), not at the returned scores: a NaN candidate sorts above
    // +inf (positive NaN) or below -inf (negative NaN) and would be picked first or never, a row whose normaliser is +inf / NaN turns
    // all its candidates into -inf / NaN -- either way the search would go on and return an ordinary-looking ranking without them.
    // -inf candidates are legitimate (HF's -inf refills of finished beams).
    for (int ci = tid; ci < C; ci += NTHR) {
      const uint32_t o = (uint32_t)(keys[ci] >> 32);
      if (o >= 0xff800000u || o < 0x007fffffu) st.error[0] = 4;  // f2ord(+inf) = 0xff800000, f2ord(-inf) = 0x007fffff
    }
    if (tid < K && s_cnt[tid] > 0) {
      const int lr = rows_per_user == 1 ? b : s_lr[tid];
      if (lr >= 0 && !(fabsf(lse[lr]) < 3.0e38f)) st.error[0] = 4;
    }
    // Only the best 2K candidates are looked at (topk(2K) in beam_search), in descending order: a bitonic TOP-P
    // selection, P = the power of two >= 2K.  Sort every P-block (directions alternating, as in a full bitonic sort
    // stopped at stage P), then halve the array round by round: a descending block followed by an ascending one is a
    // bitonic sequence, so the elementwise maxima of the pair are a bitonic block that holds the pair's P largest keys;
    // re-sort it (log2 P merge stages) and go on until one block is left.  ~3x fewer compare-exchanges than sorting
    // all NC <= 16 384 keys; keys are unique (flat index in the low word), so the result is the full sort's prefix.
    int P = 64;
    while (P < 2 * K) P <<= 1;
    const int top = NC < P ? NC : P;
    auto stage = [&](int n, int kk, int j) {
      for (int i = tid; i < n; i += NTHR) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const unsigned long long a = keys[i], c = keys[ixj];
          const bool desc = (i & kk) == 0;
          if (desc ? (a < c) : (a > c)) {
            keys[i] = c;
            keys[ixj] = a;
          }
        }
      }
      gram_sync();
    };
    for (int kk = 2; kk <= top; kk <<= 1)
      for (int j = kk >> 1; j > 0; j >>= 1) stage(NC, kk, j);
    for (int n = NC; n > P; n >>= 1) {
      // blocks (2q, 2q+1) -> block q of the half-size array; every thread reads its pairs before anyone writes
      const int half = n >> 1;
      constexpr int NMX = 8192 / NTHR;  // half / NTHR <= 8192 / NTHR (K * max_fanout <= 16 384); statically indexed: stays in registers
      unsigned long long mx[NMX];
#pragma unroll
      for (int c = 0; c < NMX; ++c) {
        const int o = tid + c * NTHR;
        if (o < half) {
          const int q = o / P, i = o - q * P;
          const unsigned long long a = keys[(2 * q) * P + i], b2 = keys[(2 * q + 1) * P + i];
          mx[c] = a > b2 ? a : b2;
        }
      }
      gram_sync();
#pragma unroll
      for (int c = 0; c < NMX; ++c) {
        const int o = tid + c * NTHR;
        if (o < half) keys[o] = mx[c];
      }
      gram_sync();
      for (int j = P >> 1; j > 0; j >>= 1) stage(half, P, j);  // bitonic blocks -> sorted, directions alternating again
    }
  }

  // the Trie edge of every ranked candidate, looked up by 2K threads at once (one binary search each; the walk below ran them one
  // after the other: up to K dependent searches of ~6 global loads each on a single thread)
  if (!isdone) {
    for (int rank = tid; rank < 2 * K; rank += NTHR) {
      int e = -1;
      if (rank < C) {
        const unsigned long long key = keys[rank];
        const uint32_t flat = 0xffffffffu - (uint32_t)(key & 0xffffffffull);
        const int k = (int)(flat / (uint32_t)V), tok = (int)(flat % (uint32_t)V);
        if (tok != st.eos) e = find_child(tr, st.node[row0 + k], tok);
      }
      s_edge[rank] = e;
    }
    gram_sync();
  }

  if (tid == 0) {
    if (isdone) {
      // BeamSearchScorer.process pads a finished user
      for (int j = 0; j < K; ++j) {
        sel_score[j] = 0.f;
        sel_tok[j] = st.pad;
        sel_par[j] = j;
        sel_node[j] = -1;
      }
    } else {
      int j = 0;
      int ftok = 0;  // next filler token candidate (beam 0, -inf), ascending flat index
      const int node0 = st.node[row0];
      float best = -INFINITY;
      for (int rank = 0; rank < 2 * K && j < K; ++rank) {
        float sc;
        int k, tok;
        if (rank < C) {
          const unsigned long long key = keys[rank];
          sc = ord2f((uint32_t)(key >> 32));
          const uint32_t flat = 0xffffffffu - (uint32_t)(key & 0xffffffffull);
          k = (int)(flat / (uint32_t)V);
          tok = (int)(flat % (uint32_t)V);
        } else {
          while (ftok < V && find_child(tr, node0, ftok) >= 0) ++ftok;
          sc = -INFINITY;
          k = 0;
          tok = ftok++;
        }
        if (rank == 0) best = sc;
        if (tok == st.eos) {
          if (rank >= K) continue;
          hyp_add(st, b, st.seq + (size_t)(row0 + k) * T, cur_len, sc);
        } else {
          const int e = (rank < C) ? s_edge[rank] : -1;
          sel_score[j] = sc;
          sel_tok[j] = tok;
          sel_par[j] = k;
          sel_node[j] = e < 0 ? -1 : tr.child_node[e];
          ++j;
        }
      }
      if (j < K) {
        st.error[0] = 1;  // HF raises ValueError here
        for (; j < K; ++j) { sel_score[j] = -INFINITY; sel_tok[j] = st.pad; sel_par[j] = 0; sel_node[j] = -1; }
      }
      // BeamHypotheses.is_done(best_sum_logprobs = next_scores.max(), cur_len)
      if (st.n_hyps[b] >= K) {
        const double cur = (double)best / norm_len(cur_len, st.length_penalty);
        if (st.worst[b] >= cur) st.done[b] = 1;
      }
    }
  }
  gram_sync();

  // advance sequences / ancestor table / per-row state (read old -> LDS -> write)
  for (int idx = tid; idx < K * T; idx += NTHR) {
    const int j = idx / T, p = idx - j * T;
    int v = 0;
    if (p < cur_len) v = st.seq[(size_t)(row0 + sel_par[j]) * T + p];
    else if (p == cur_len) v = sel_tok[j];
    new_seq[idx] = v;
  }
  for (int idx = tid; idx < T * K; idx += NTHR) {
    const int p = idx / K, j = idx - p * K;
    int v;
    if (p < t) v = st.anc[(size_t)p * R + row0 + sel_par[j]];
    else if (p == t) v = rows_per_user == 1 ? b : row0 + sel_par[j];  // compact step: slot t holds one row per user
    else v = row0 + j;
    new_anc[idx] = v;
  }
  gram_sync();
  for (int idx = tid; idx < K * T; idx += NTHR) st.seq[(size_t)row0 * T + idx] = new_seq[idx];
  for (int idx = tid; idx < T * K; idx += NTHR) {
    const int p = idx / K, j = idx - p * K;
    st.anc[(size_t)p * R + row0 + j] = new_anc[idx];
  }
  if (tid < K) {
    st.beam_scores[row0 + tid] = sel_score[tid];
    st.tokens[row0 + tid] = sel_tok[tid];
    st.node[row0 + tid] = sel_node[tid];
  }
}

// The sparse logits of a search step, computed by MANY workgroups (a handful of users: beam_step_kernel's one workgroup per user pulls
// up to K * fan-out fp32 lm_head rows -- 15 MB at K = 20, fan-out 255 -- through ONE CU, 70 us on average and up to 290 us of a one-user
// step).  Workgroup (chunk, user) recomputes the user's candidate list exactly as beam_step_kernel does (beams in order, each beam's
// Trie children in token order) and writes h . E[tok] of candidates [32 chunk, 32 chunk + 32) to out[user][candidate]; the search
// step then reads them instead of computing them -- sparse_dot either way: the same bits.
__global__ __launch_bounds__(256) void sparse_logits_kernel(gram_beam_state_t st, gram_trie_t tr, int nc_max, int rows_per_user,
                                                            const p16* __restrict__ hd, const p16* __restrict__ emb, int d,
                                                            const int32_t* __restrict__ rowpos, int pieces,
                                                            const float* __restrict__ emb32, float* __restrict__ out) {
  __shared__ int s_pre[GRAM_MAX_BEAMS + 1], s_off[GRAM_MAX_BEAMS], s_cnt[GRAM_MAX_BEAMS], s_lr[GRAM_MAX_BEAMS];
  const int b = blockIdx.y, tid = threadIdx.x, K = st.K, row0 = b * K;
  if (st.done[b]) return;
  if (tid < K) {
    const int nd = st.node[row0 + tid];
    int o0 = 0, cnt = 0;
    if (nd >= 0) {
      o0 = tr.child_off[nd];
      cnt = tr.child_off[nd + 1] - o0;
    }
    s_off[tid] = o0;
    s_cnt[tid] = cnt;
    s_lr[tid] = rowpos ? rowpos[row0 + tid] : row0 + tid;
  }
  gram_sync();
  if (tid == 0) {
    int acc = 0;
    for (int k = 0; k < K; ++k) {
      s_pre[k] = acc;
      acc += s_cnt[k];
    }
    s_pre[K] = acc;
  }
  gram_sync();
  const bool shared0 = rows_per_user == 1;
  const int nuniq = shared0 ? s_pre[1] : (s_pre[K] <= nc_max ? s_pre[K] : 0);  // (more than nc_max: the search step flags it)
  const int ci = blockIdx.x * 32 + (tid >> 3), sub = tid & 7;
  const bool act = ci < nuniq;
  int k = 0, tok = 0;
  if (act) {
    if (!shared0)
      while (s_pre[k + 1] <= ci) ++k;
    tok = tr.child_tok[s_off[k] + (ci - s_pre[k])];
  }
  const float v = sparse_dot(act, shared0 ? b : s_lr[k], tok, sub, hd, emb, emb32, d, pieces);
  if (act && sub == 0) out[(size_t)b * nc_max + ci] = v;
}

// Live rows of the coming decode step: a beam that left the Trie (its hypothesis went to the heap at EOS and HF refilled
// the slot with a -inf candidate) or belongs to a finished user can only produce -inf candidates, so its decoder row is
// never read by beam_step_kernel.  One workgroup compacts the others (ascending, i.e. grouped by user).
__global__ __launch_bounds__(1024) void live_rows_kernel(gram_beam_state_t st, gram_trie_t tr, gram_live_rows_t out) {
  __shared__ int s_w[16], s_tot;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int K = st.K, B = st.B, R = B * K;
  auto scan = [&](int v) {  // inclusive block scan of a 0/1 flag; returns the inclusive prefix, s_tot = block total
    for (int o = 1; o < 64; o <<= 1) {
      const int n = __shfl_up(v, o, 64);
      if (lane >= o) v += n;
    }
    gram_sync();  // previous round's s_w / s_tot readers are done
    if (lane == 63) s_w[wave] = v;
    gram_sync();
    if (tid == 0) {
      int acc = 0;
      for (int w = 0; w < 16; ++w) {
        const int t = s_w[w];
        s_w[w] = acc;
        acc += t;
      }
      s_tot = acc;
    }
    gram_sync();
    return v + s_w[wave];
  };
  int base = 0;
  for (int c0 = 0; c0 < R; c0 += 1024) {
    const int r = c0 + tid;
    int live = 0;
    if (r < R && !st.done[r / K]) {
      const int nd = st.node[r];
      live = nd >= 0 && tr.child_off[nd + 1] > tr.child_off[nd];
    }
    const int pos = base + scan(live) - 1;
    if (live) {
      out.rows[pos] = r;
      out.tokens[pos] = st.tokens[r];
    }
    if (r < R) out.rowpos[r] = live ? pos : -1;
    base += s_tot;
  }
  gram_sync();  // rowpos written by this workgroup is visible to it
  int ubase = 0;
  for (int c0 = 0; c0 < B; c0 += 1024) {
    const int b = c0 + tid;
    int live = 0;
    if (b < B)
      for (int k = 0; k < K; ++k) live |= out.rowpos[b * K + k] >= 0;
    const int pos = ubase + scan(live) - 1;
    if (live) out.users[pos] = b;
    ubase += s_tot;
  }
  if (tid == 0) {
    out.counts[0] = base;
    out.counts[1] = ubase;
  }
}

__global__ void beam_finalize_kernel(gram_beam_state_t st, int nret, int max_length, int cur_len, int64_t* __restrict__ sequences,
                                     float* __restrict__ scores, int32_t* __restrict__ out_width) {
  const int b = blockIdx.x;
  if (threadIdx.x != 0) return;
  const int K = st.K, T = st.Tmax;
  if (!st.done[b]) {
    for (int k = 0; k < K; ++k) {
      const int r = b * K + k;
      hyp_add(st, b, st.seq + (size_t)r * T, cur_len, st.beam_scores[r]);
    }
  }
  const int n = st.n_hyps[b];
  const double* hs = st.hyp_score + (size_t)b * (K + 1);
  const int32_t* hl = st.hyp_len + (size_t)b * (K + 1);
  const int32_t* ht = st.hyp_tok + (size_t)b * (K + 1) * T;
  unsigned long long taken = 0ull;  // K+1 <= 65 entries; n <= K <= 64 here
  int maxlen = 0;
  for (int j = 0; j < nret; ++j) {
    int64_t* dst = sequences + ((size_t)b * nret + j) * max_length;
    for (int p = 0; p < max_length; ++p) dst[p] = st.pad;
    if (j >= n) {  // sorted_hyps.pop() on an empty list: IndexError in HF
      st.error[0] = 3;
      scores[b * nret + j] = -INFINITY;
      continue;
    }
    // sorted(beams, key=score) ascending + pop(): best score, ties -> highest list index
    int best = -1;
    for (int i = 0; i < n; ++i) {
      if (taken & (1ull << i)) continue;
      if (best < 0 || hs[i] >= hs[best]) best = i;
    }
    taken |= 1ull << best;
    const int len = hl[best];
    for (int p = 0; p < len && p < max_length; ++p) dst[p] = ht[(size_t)best * T + p];
    if (len < max_length) dst[len] = st.eos;
    scores[b * nret + j] = (float)hs[best];
    // log-probabilities are <= 0: a NaN or +inf score means an activation overflowed the 16-bit pieces somewhere upstream (GRAM_E_NONFINITE)
    if (hs[best] != hs[best] || hs[best] > 1.0e30) st.error[0] = 4;
    maxlen = len > maxlen ? len : maxlen;
  }
  int w = maxlen + 1;
  if (w > max_length) w = max_length;
  atomicMax(out_width, w);
}

// HF 4.26 greedy_search step (num_beams == 1): one thread per user.  argmax of the raw logits over the
// Trie children (first maximum wins, like torch.argmax on the -inf-masked row; no children -> index 0),
// finished users emit pad.  done[b] doubles as HF's (1 - unfinished_sequences); n_hyps[b] records the
// sequence length at which the user finished (0 = still running) for the final width.
__global__ void greedy_step_kernel(gram_beam_state_t st, gram_trie_t tr, const float* __restrict__ logits, int V, int cur_len) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= st.B) return;
  const int T = st.Tmax;
  int tok = st.pad, next_node = -1;
  if (!st.done[b]) {
    const int nd = st.node[b];
    tok = 0;  // argmax of an all -inf row
    if (nd >= 0) {
      const int lo = tr.child_off[nd], hi = tr.child_off[nd + 1];
      float best = -INFINITY;
      bool any = false;
      for (int e = lo; e < hi; ++e) {
        const int c = tr.child_tok[e];
        const float v = logits[(size_t)b * V + c];
        // children are sorted by token: strict > keeps the first maximum; a -inf logit at a lower index than
        // an allowed -inf one cannot happen for finite model outputs, NaN is never selected over a number
        if (!any || v > best) {
          best = v;
          tok = c;
          next_node = tr.child_node[e];
          any = true;
        }
      }
    }
    if (tok == st.eos) {
      st.done[b] = 1;
      st.n_hyps[b] = cur_len + 1;
    }
  }
  st.seq[(size_t)b * T + cur_len] = tok;
  st.tokens[b] = tok;
  st.node[b] = next_node;
}

__global__ void greedy_finalize_kernel(gram_beam_state_t st, int max_length, int64_t* __restrict__ sequences,
                                       int32_t* __restrict__ out_width) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= st.B) return;
  const int T = st.Tmax;
  for (int p = 0; p < max_length; ++p) sequences[(size_t)b * max_length + p] = st.seq[(size_t)b * T + p];
  // HF stops as soon as every row is finished: width = the longest finished length (max_length if any row never finished)
  const int len = st.n_hyps[b] > 0 ? st.n_hyps[b] : max_length;
  atomicMax(out_width, len);
}

// Item index of every returned sequence: a hypothesis of a Trie-constrained search is a root-to-leaf path, so walking the CSR
// from the root along the sequence's tokens (start token first) ends on a leaf, and node_item[leaf] is the candidate it spells.
// -1: the row is not a candidate (the -inf filler beams HF pads a user with when fewer than nret hypotheses finished).
__global__ void trie_item_index_kernel(gram_trie_t tr, const int32_t* __restrict__ node_item, const int64_t* __restrict__ sequences,
                                       int rows, int T, int pad, int32_t* __restrict__ out_item) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  const int64_t* seq = sequences + (size_t)r * T;
  int node = 0, p = 0;
  for (; p < T; ++p) {
    if (tr.child_off[node + 1] == tr.child_off[node]) break;  // a leaf: the candidate is complete
    const int64_t tok = seq[p];
    const int e = (tok < 0 || tok > 0x7fffffff) ? -1 : find_child(tr, node, (int)tok);
    if (e < 0) { node = -1; break; }
    node = tr.child_node[e];
  }
  int item = -1;
  if (node > 0 && tr.child_off[node + 1] == tr.child_off[node]) {
    item = node_item[node];
    for (; p < T; ++p)  // what follows a candidate is padding, or the row is some other sequence that merely starts like one
      if (seq[p] != pad) item = -1;
  }
  out_item[r] = item;
}

}  // namespace

static int check_state(const gram_beam_state_t* st) {
  if (!st || st->B < 1 || st->K < 1 || st->K > GRAM_MAX_BEAMS || st->Tmax < 2 || st->Tmax > GRAM_MAX_DEC_LEN) return GRAM_E_ARG;
  return 0;
}

extern "C" int gram_beam_init(const gram_beam_state_t* st, const gram_trie_t* tr, int start_token, void* stream) {
  if (int e = check_state(st)) return e;
  if (!tr) return GRAM_E_ARG;
  const int R = st->B * st->K;
  hipLaunchKernelGGL(beam_init_kernel, dim3((R + 255) / 256), dim3(256), 0, (hipStream_t)stream, *st, *tr, start_token);
  GRAM_CHECK_LAUNCH();
  return 0;
}

static int launch_beam_step(const gram_beam_state_t* st, const gram_trie_t* tr, const float* logits, const float* lse, int V,
                            int cur_len, int rows_per_user, const void* hd, const void* emb, int d, const int32_t* rowpos,
                            void* stream, int pieces = 1, const float* emb32 = nullptr) {
  if (int e = check_state(st)) return e;
  if (!tr || !lse || cur_len < 1 || cur_len >= st->Tmax || V < 2 || (rows_per_user != 1 && rows_per_user != st->K)) return GRAM_E_ARG;
  if (!logits && (!hd || (!emb && !emb32) || d < 64 || (d & 63))) return GRAM_E_ARG;
  if (pieces < 1 || pieces > GRAM_MAX_PIECES || (pieces > 1 && (!emb32 || (d & 63)))) return GRAM_E_ARG;  // (8 lanes x 8-element loads inside 32-column blocks)
  long long need = (long long)st->K * tr->max_fanout;
  int nc = 64;
  while (nc < need) nc <<= 1;
  const size_t smem = (size_t)nc * 8 + ((size_t)tr->max_fanout * 4 + 15) / 16 * 16 + (size_t)2 * st->K * st->Tmax * 4;
  if (smem > 152 * 1024) return GRAM_E_ARG;  // (+ ~3 KB of static arrays: the CU's 160 KB)
  static size_t attr_bytes = 0;
  if (smem > attr_bytes) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(beam_step_kernel<256>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)smem);
    if (e == hipSuccess)
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(beam_step_kernel<1024>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return (int)e;
    attr_bytes = smem;
  }
  gram_prof::Scope prof(GRAM_K_BEAM, (hipStream_t)stream, 0.0);
  // few users: one workgroup per user leaves the chip empty and the step is that workgroup's latency -> 1 024 threads per user
  // (same per-candidate arithmetic, same total order of the keys: identical results; GRAM_BEAM_WIDE_MAXB: A/B hook, 0 = never)
  static const int wide_max_b = getenv("GRAM_BEAM_WIDE_MAXB") ? atoi(getenv("GRAM_BEAM_WIDE_MAXB")) : 128;
  // a handful of users: the sparse logits by their own kernel over many CUs (sparse_logits_kernel), the search step reads them
  // (gram_beam_state_t.cand_logits: caller-provided scratch; GRAM_BEAM_PRE_MAXB: A/B hook, 0 = never)
  static const int pre_max_b = getenv("GRAM_BEAM_PRE_MAXB") ? atoi(getenv("GRAM_BEAM_PRE_MAXB")) : 16;
  const float* pre = nullptr;
  if (!logits && st->cand_logits && st->B <= pre_max_b && st->B <= st->cand_logits_users && (long long)nc <= st->cand_logits_stride) {
    const int chunks = (int)((need + 31) / 32);
    hipLaunchKernelGGL(sparse_logits_kernel, dim3(chunks, st->B), dim3(256), 0, (hipStream_t)stream, *st, *tr, nc,
                       rows_per_user, (const p16*)hd, (const p16*)emb, d, rowpos, pieces, emb32, st->cand_logits);
    GRAM_CHECK_LAUNCH();
    pre = st->cand_logits;
  }
  const int nc_arg = nc;
  if (st->B <= wide_max_b)
    hipLaunchKernelGGL(beam_step_kernel<1024>, dim3(st->B), dim3(1024), smem, (hipStream_t)stream, *st, *tr, logits, lse, V, cur_len, nc_arg,
                       rows_per_user, (const p16*)hd, (const p16*)emb, d, rowpos, pieces, emb32, pre);
  else
    hipLaunchKernelGGL(beam_step_kernel<256>, dim3(st->B), dim3(256), smem, (hipStream_t)stream, *st, *tr, logits, lse, V, cur_len, nc_arg,
                       rows_per_user, (const p16*)hd, (const p16*)emb, d, rowpos, pieces, emb32, pre);
  GRAM_CHECK_LAUNCH();
  return 0;
}

extern "C" int gram_beam_step(const gram_beam_state_t* st, const gram_trie_t* tr, const float* logits, const float* lse, int V,
                              int cur_len, int rows_per_user, void* stream) {
  if (!logits) return GRAM_E_ARG;
  return launch_beam_step(st, tr, logits, lse, V, cur_len, rows_per_user, nullptr, nullptr, 0, nullptr, stream);
}

extern "C" int gram_beam_step_sparse(const gram_beam_state_t* st, const gram_trie_t* tr, const void* hidden_bf16,
                                     const void* lm_head_bf16, int d, const float* lse, int V, int cur_len, int rows_per_user,
                                     void* stream) {
  return launch_beam_step(st, tr, nullptr, lse, V, cur_len, rows_per_user, hidden_bf16, lm_head_bf16, d, nullptr, stream);
}

extern "C" int gram_beam_step_sparse_live(const gram_beam_state_t* st, const gram_trie_t* tr, const void* hidden_bf16,
                                          const void* lm_head_bf16, int d, const float* lse, int V, int cur_len,
                                          const int32_t* rowpos, void* stream) {
  if (!rowpos || !st) return GRAM_E_ARG;
  return launch_beam_step(st, tr, nullptr, lse, V, cur_len, st->K, hidden_bf16, lm_head_bf16, d, rowpos, stream);
}

extern "C" int gram_beam_step_sparse_split(const gram_beam_state_t* st, const gram_trie_t* tr, const void* hidden_bf16,
                                           const float* lm_head_f32, int d, const float* lse, int V, int cur_len, int rows_per_user,
                                           const int32_t* rowpos, int pieces, void* stream) {
  if (!lm_head_f32) return GRAM_E_ARG;
  return launch_beam_step(st, tr, nullptr, lse, V, cur_len, rows_per_user, hidden_bf16, nullptr, d, rowpos, stream, pieces, lm_head_f32);
}

extern "C" int gram_live_rows(const gram_beam_state_t* st, const gram_trie_t* tr, const gram_live_rows_t* out, void* stream) {
  if (int e = check_state(st)) return e;
  if (!tr || !out || !out->rows || !out->rowpos || !out->users || !out->tokens || !out->counts) return GRAM_E_ARG;
  gram_prof::Scope prof(GRAM_K_BEAM, (hipStream_t)stream, 0.0);
  hipLaunchKernelGGL(live_rows_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, *st, *tr, *out);
  GRAM_CHECK_LAUNCH();
  return 0;
}

extern "C" int gram_greedy_step(const gram_beam_state_t* st, const gram_trie_t* tr, const float* logits, int V, int cur_len,
                               void* stream) {
  if (int e = check_state(st)) return e;
  if (!tr || st->K != 1 || cur_len < 1 || cur_len >= st->Tmax || V < 2) return GRAM_E_ARG;
  gram_prof::Scope prof(GRAM_K_BEAM, (hipStream_t)stream, 0.0);
  hipLaunchKernelGGL(greedy_step_kernel, dim3((st->B + 127) / 128), dim3(128), 0, (hipStream_t)stream, *st, *tr, logits, V, cur_len);
  GRAM_CHECK_LAUNCH();
  return 0;
}

extern "C" int gram_greedy_finalize(const gram_beam_state_t* st, int max_length, int64_t* sequences, int32_t* out_width,
                                    void* stream) {
  if (int e = check_state(st)) return e;
  if (st->K != 1 || max_length != st->Tmax) return GRAM_E_ARG;
  hipError_t e = hipMemsetAsync(out_width, 0, sizeof(int32_t), (hipStream_t)stream);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(greedy_finalize_kernel, dim3((st->B + 127) / 128), dim3(128), 0, (hipStream_t)stream, *st, max_length, sequences,
                     out_width);
  GRAM_CHECK_LAUNCH();
  return 0;
}

extern "C" int gram_beam_finalize(const gram_beam_state_t* st, int nret, int max_length, int64_t* sequences, float* scores,
                                  int32_t* out_width, void* stream) {
  if (int e = check_state(st)) return e;
  if (nret < 1 || nret > st->K || max_length != st->Tmax) return GRAM_E_ARG;
  hipError_t e = hipMemsetAsync(out_width, 0, sizeof(int32_t), (hipStream_t)stream);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(beam_finalize_kernel, dim3(st->B), dim3(64), 0, (hipStream_t)stream, *st, nret, max_length, max_length,
                     sequences, scores, out_width);
  GRAM_CHECK_LAUNCH();
  return 0;
}

extern "C" int gram_trie_item_index(const gram_trie_t* tr, const int32_t* node_item, const int64_t* sequences, int rows, int T,
                                    int32_t* out_item, void* stream) {
  if (!tr || !tr->child_off || !node_item || !sequences || !out_item || rows < 0 || T < 1 || tr->n_nodes < 1) return GRAM_E_ARG;
  if (rows == 0) return 0;
  hipLaunchKernelGGL(trie_item_index_kernel, dim3((rows + 255) / 256), dim3(256), 0, (hipStream_t)stream, *tr, node_item, sequences, rows,
                     T, /*pad=*/0, out_item);
  GRAM_CHECK_LAUNCH();
  return 0;
}
