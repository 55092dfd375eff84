// dec_attn.hip -- the two decoder attention kernels of one decode step.
//
// (1) cross_attn_kernel: late-fusion cross-attention, THE HBM-bound kernel of the path.
//     Reference: T5Attention.forward cross branch, gram_t5_modeling.py:531-534,547-549,572-622
//     (zero position bias :577-582, additive mask :1145-1147), called per layer per step on
//     K-times replicated K/V.  Here ONE bank per user is streamed ONCE per (layer, step) and
//     shared by all of the user's beams: algorithmic bytes = 2 * S * 64 * 2 B per (user, head)
//     and bf16 piece of the bank.
//
//     Mapping: one workgroup per (user, head), NW waves (template; 2 for K <= 32, 1 above), wave w owns the valid
//     32-key steps w, w+NW, ...  A step's K rows (32 x 128 B) and V^T rows (64 x 64 B) of every bf16 piece are brought
//     into a wave-PRIVATE ring of R stages in LDS by LDS-DMA (global_load_lds, 16 B per lane, no staging registers);
//     waves never synchronise inside the loop (counted s_waitcnt vmcnt only).  The DMA writes LDS linearly, so the
//     bank-conflict swizzles are applied to the per-lane SOURCE chunk.  S^T = K Q^T puts a beam on a lane column, so
//     the online softmax is in-register + two cross-lane steps, and exp(S^T) is directly the B operand of
//     O^T = V^T P^T (same row-permutation trick as enc_attn.hip).  With NW > 1 the waves merge their (m, l, O)
//     partials through LDS (aliasing the rings) at the end; with NW = 1 the accumulators are the result.
//     (NW, R) per shape were measured, see launch_cross_v() and DESIGN.md §4.2.
//
// (2) dec_self_attn_kernel: causal self-attention of the newest token over <= 32 cached
//     positions with beam-parent indirection (anc table) instead of the reference's
//     torch.cat + index_select of the whole cache (gram_t5_modeling.py:536-540,
//     gram_t5.py:320-348); unidirectional relative bias, last query row (:586-593).
//
// Both kernels take their bf16 operands as 1..3 pieces (gram_split_t in gram_hip.h).
#include <stdlib.h>
#include <type_traits>

#include "common.h"
#include "prof.h"

namespace {

// LDS-DMA through inline asm (see gemm.hip: the builtin makes hipcc wait lgkmcnt(0) in front of every DMA)
__device__ __forceinline__ void dma16(uint32_t lds_addr /*wave-uniform*/, uint32_t voff, const char* base /*uniform*/) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 3\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_addr), "v"(voff), "s"(base) : "memory");
}
// the same with the streaming (nontemporal) hint: the bank is read once per step and never again before the next step's sweep
__device__ __forceinline__ void dma16_nt(uint32_t lds_addr /*wave-uniform*/, uint32_t voff, const char* base /*uniform*/) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 3\n\tglobal_load_lds_dwordx4 %1, %2 nt" ::"s"(lds_addr), "v"(voff), "s"(base) : "memory");
}
#ifndef GRAM_XA_NT
#define GRAM_XA_NT 1  // bit 0 = K tiles, bit 1 = V^T tiles fetched with the nt hint (A/B build hook; in the bench, one box: 168.8 ms of cross-attention per step without, 163.3 with K only, 165.7 with both)
#endif
template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// K tile [32 keys][128 B]: 16-B chunk ch of key row r sits at position ch ^ ksw(r).  A fragment read touches the 16 rows
// 8a + b (+ 4t), a, b = 0..3: (b >> 1, a) gives 8 distinct swizzles, b & 1 the other half of the banks -> conflict-free.
__device__ __forceinline__ int ksw(int row) { return ((row >> 1) & 1) | (((row >> 3) & 3) << 1); }
// V^T tile [64 d][64 B]: chunk ch of row d at position ch ^ F[(d >> 2) & 3], F = {0, 2, 3, 1} (16 consecutive rows, one chunk)
__device__ __forceinline__ int vsw(int row) { return (0x78 >> (((row >> 2) & 3) * 2)) & 3; }

constexpr int XA_TILE = 4096;  // one K tile or one V^T tile of a 32-key step, per piece

// The two places of the cross-attention whose rounding depends on an fma-contraction decision, spelled out once for both kernels below
// (every instantiation of the kernel must return the same bits for the same (user, head) -- a user's result does not depend on the batch
// it is scored in -- and hipcc decides contraction per call site).
__device__ __forceinline__ float xa_l_update(float l, float alpha, float ps) { return __builtin_fmaf(l, alpha, ps); }
// merge of the two partial softmaxes (m, l, O) of a (user, head): weights and 1 / l
__device__ __forceinline__ void xa_merge_w(float m0, float m1, float l0, float l1, float& w0, float& w1, float& inv) {
  const float M = fmaxf(fmaxf(GRAM_FMIN, m0), m1);
  w0 = __expf(m0 - M);
  w1 = __expf(m1 - M);
  inv = 1.f / __builtin_fmaf(l1, w1, l0 * w0);
}
__device__ __forceinline__ f32x4 xa_merge_o(f32x4 o0, f32x4 o1, float w0, float w1, float inv) {
  f32x4 r;
#pragma unroll
  for (int e = 0; e < 4; ++e) r[e] = __builtin_fmaf(o1[e], w1, o0[e] * w0) * inv;
  return r;
}

// NT = 16-beam tiles; LIVE: live-row step (gram_live_rows_t); S = bf16 pieces; NW = waves per workgroup (1: a wave owns the whole
// (user, head) and nothing is merged); R = ring stages per wave
template <int NT, bool LIVE, int S, int NW, int R>
__global__ __launch_bounds__(NW * 64) void cross_attn_kernel(
    const p16* __restrict__ q, const p16* __restrict__ kbank, const p16* __restrict__ vtbank,
    const uint8_t* __restrict__ mask, p16* __restrict__ out, int K, int H, int Sk, const int32_t* __restrict__ users,
    const int32_t* __restrict__ rowpos, long q_pstride, long bank_pstride, const uint32_t* __restrict__ key_bits) {
  using T = SplitTab<S>;
  constexpr int NB = NT * 16;                     // padded beams
  // R == 0: "half slot" -- ONE region per wave that holds a step's K tiles, then (once their fragments are in registers) its V^T tiles,
  // then the next step's K tiles ...: half the LDS per workgroup, twice the workgroups per CU to cover each other's prologue and merge
  constexpr bool HALF = R == 0;
  constexpr int PSTR = HALF ? XA_TILE : 2 * XA_TILE;  // piece stride inside a slot
  constexpr int VOFF = HALF ? 0 : XA_TILE;           // V^T tiles: behind the K tiles, or in their place
  constexpr int STAGE = S * PSTR;                     // per wave and ring slot: S x (K tile | V^T tile)
  constexpr int RING = (HALF ? 1 : R) * STAGE;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sm_m = reinterpret_cast<float*>(smem);   // [NW][NB]     (the merge buffers alias the rings)
  float* sm_l = sm_m + NW * NB;                   // [NW][NB]
  float* sm_o = sm_l + NW * NB;                   // [NW][NB][64]

  // live-row step (users != NULL): workgroup y serves user users[y]; q/out rows are the compact rows rowpos[b*K + beam]
  // (-1 = beam not live: zero query, nothing stored)
  const int h = blockIdx.x, b = LIVE ? users[blockIdx.y] : blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, g = lane >> 4;
  const int inner = H * 64;
  const char* kb = reinterpret_cast<const char*>(kbank + ((size_t)b * H + h) * Sk * 64);
  const char* vt = reinterpret_cast<const char*>(vtbank + ((size_t)b * H + h) * 64 * Sk);
  const uint8_t* mk = mask + (size_t)b * Sk;

  p16x8 qf[S][NT][2];  // query fragments: loaded once the ring is primed (below)
  f32x4 o[4][NT];
  float m[NT], l[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    m[nt] = GRAM_FMIN;
    l[nt] = 0.f;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) o[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }

  if constexpr (R <= 1) {
    // query fragments: requested first, in parallel with the key bits, and retired with them by the wait in front of the ring --
    // the counted waits of the half-stage loop must see DMA instructions only
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int beam = 16 * nt + c;
      int qrow = beam < K ? b * K + beam : -1;
      if constexpr (LIVE) {
        if (qrow >= 0) qrow = rowpos[qrow];
      }
#pragma unroll
      for (int pc = 0; pc < S; ++pc)
#pragma unroll
        for (int kd = 0; kd < 2; ++kd)
          qf[pc][nt][kd] = qrow >= 0 ? ld_global_b128(q + pc * q_pstride + (size_t)qrow * inner + h * 64 + 32 * kd + 8 * g) : zero_bf16x8();
    }
  }
  // Fully masked 32-key steps (padded passages / padded tails) are never fetched: the algorithmic
  // traffic is proportional to the VALID fused keys.  One word of key bits per step (S <= 4096 -> <= 128 steps);
  // the valid steps are dealt round-robin to the waves.  A user with no valid key at all keeps
  // every step: the reference's softmax over all-finfo.min scores is uniform over all S keys.
  const int nsteps = Sk >> 5;
  // this lane's two words of key bits (steps lane and lane + 64), either precomputed once per generate (gram_mask_key_bits) or
  // packed here from the mask bytes; every wave holds all of them, so a step's word is one v_readlane away
  uint32_t kb0 = 0, kb1 = 0;
  if (key_bits) {
    if (lane < nsteps) kb0 = key_bits[(size_t)b * 128 + lane];
    if (lane + 64 < nsteps) kb1 = key_bits[(size_t)b * 128 + 64 + lane];
  } else {
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int st = lane + 64 * half;
      uint32_t bits = 0;
      if (st < nsteps) {
        const uint4* p = reinterpret_cast<const uint4*>(mk + 32 * st);
        const uint4 a = p[0], c2 = p[1];
        const uint32_t w8[8] = {a.x, a.y, a.z, a.w, c2.x, c2.y, c2.z, c2.w};
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) bits |= ((w8[i] >> (8 * j)) & 0xffu) ? (1u << (4 * i + j)) : 0u;
      }
      if (half == 0) kb0 = bits; else kb1 = bits;
    }
  }
  unsigned long long v0 = __ballot(kb0 != 0u), v1 = __ballot(kb1 != 0u);
  if ((v0 | v1) == 0ull) {
    v0 = nsteps >= 64 ? ~0ull : ((1ull << nsteps) - 1ull);
    v1 = nsteps > 64 ? ((nsteps >= 128 ? ~0ull : ((1ull << (nsteps - 64)) - 1ull))) : 0ull;
  }
  v0 = __builtin_amdgcn_readfirstlane((unsigned)v0) | ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(v0 >> 32)) << 32);
  v1 = __builtin_amdgcn_readfirstlane((unsigned)v1) | ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(v1 >> 32)) << 32);
  int ord = 0;  // ordinal of the next valid step (wave-uniform scalar state)
  auto next = [&](int s) -> int {
    for (s = s + 1; s < nsteps; ++s) {
      const unsigned long long bit = (s < 64 ? (v0 >> s) : (v1 >> (s - 64))) & 1ull;
      if (bit) {
        const bool mine = ord == wave;
        ord = ord + 1 == NW ? 0 : ord + 1;
        if (mine) return s;
      }
    }
    return nsteps;
  };

  // per-lane byte offsets of this lane's 16 B in each of the 4 + 4 DMA instructions of a (step, piece)
  uint32_t koff[4], voff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int kr = 8 * i + (lane >> 3);  // key row of the tile; the lane lands at position lane & 7 and fetches chunk pos ^ ksw
    koff[i] = (uint32_t)(kr * 128 + (((lane & 7) ^ ksw(kr)) << 4));
    const int d = 16 * i + (lane >> 2);  // V^T row; position lane & 3
    voff[i] = (uint32_t)(d * 64 + (((lane & 3) ^ vsw(d)) << 4));  // (the bank's V^T is blocked by 32 keys: a step's tile is [64 d][64 B], contiguous)
  }
  const uint32_t ring0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem + wave * RING;
  // 4 * S DMA instructions each: this step's K tiles / V^T tiles of every piece -> ring slot
  auto issue_k = [&](int slot, int step) {
    const uint32_t dst = ring0 + slot * STAGE;
#pragma unroll
    for (int pc = 0; pc < S; ++pc) {
      const char* kbase = kb + (size_t)pc * bank_pstride * 2 + (size_t)step * (32 * 128);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if constexpr ((GRAM_XA_NT & 1) != 0) dma16_nt(dst + pc * PSTR + i * 1024, koff[i], kbase);
        else dma16(dst + pc * PSTR + i * 1024, koff[i], kbase);
      }
    }
  };
  auto issue_v = [&](int slot, int step) {
    const uint32_t dst = ring0 + slot * STAGE;
#pragma unroll
    for (int pc = 0; pc < S; ++pc) {
      const char* vbase = vt + (size_t)pc * bank_pstride * 2 + (size_t)step * 4096;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if constexpr ((GRAM_XA_NT & 2) != 0) dma16_nt(dst + pc * PSTR + VOFF + i * 1024, voff[i], vbase);
        else dma16(dst + pc * PSTR + VOFF + i * 1024, voff[i], vbase);
      }
    }
  };
  auto issue = [&](int slot, int step) {
    issue_k(slot, step);
    issue_v(slot, step);
  };
  const int krow = 8 * (c >> 2) + (c & 3);  // + 4t: key row of S^T tile t this lane feeds
  // A stage is consumed in two phases: read_frags pulls every K and V^T fragment of the step out of the ring slot into registers, and
  // once those reads have returned the slot is re-filled (the next step's DMAs are in flight during the whole of `math`, which
  // works on registers only) -- with one stage per wave and the issue after the math, a wave had nothing in flight while it computed.
  p16x8 kf[S][2][2], vf[S][4];
  auto read_k = [&](int slot) {
    const char* stg = smem + wave * RING + slot * STAGE;
#pragma unroll
    for (int pc = 0; pc < S; ++pc)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int kd = 0; kd < 2; ++kd) {
          const int r = krow + 4 * t;
          kf[pc][t][kd] = *reinterpret_cast<const p16x8*>(stg + pc * PSTR + r * 128 + (((g + 4 * kd) ^ ksw(r)) << 4));
        }
  };
  auto read_v = [&](int slot) {
    const char* stg = smem + wave * RING + slot * STAGE;
#pragma unroll
    for (int pc = 0; pc < S; ++pc)
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        const int d = 16 * mt + c;
        vf[pc][mt] = *reinterpret_cast<const p16x8*>(stg + pc * PSTR + VOFF + d * 64 + ((g ^ vsw(d)) << 4));
      }
  };
  auto read_frags = [&](int slot) {
    read_k(slot);
    read_v(slot);
  };
  p16x8 pf[S][NT];  // exp(S^T - max) of the step, as pieces: the B operand of O^T += V^T P^T
  auto math_qk = [&](int step) {
    const uint32_t kword = step < 64 ? (uint32_t)__builtin_amdgcn_readlane((int)kb0, step) : (uint32_t)__builtin_amdgcn_readlane((int)kb1, step - 64);
    const uint32_t kbits = kword >> (8 * g);  // this lane's keys 8g + 4t + j
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      f32x4 s[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        f32x4 a = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int pr = 0; pr < T::NP; ++pr) {
          a = mfma16(kf[T::A[pr]][t][0], qf[T::B[pr]][nt][0], a);
          a = mfma16(kf[T::A[pr]][t][1], qf[T::B[pr]][nt][1], a);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) a[j] = ((kbits >> (4 * t + j)) & 1u) ? a[j] : GRAM_FMIN;
        s[t] = a;
      }
      float tm = fmaxf(fmaxf(fmaxf(s[0][0], s[0][1]), fmaxf(s[0][2], s[0][3])),
                       fmaxf(fmaxf(s[1][0], s[1][1]), fmaxf(s[1][2], s[1][3])));
      tm = fmaxf(tm, __shfl_xor(tm, 16, 64));
      tm = fmaxf(tm, __shfl_xor(tm, 32, 64));
      const float mn = fmaxf(m[nt], tm);
      const float alpha = __expf(m[nt] - mn);
      m[nt] = mn;
      float ps = 0.f;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float e = __expf(s[t][j] - mn);
          ps += e;
#pragma unroll
          for (int pc = 0; pc < S; ++pc) {
            const p16 eb = (p16)e;
            pf[pc][nt][4 * t + j] = eb;
            e -= (float)eb;
          }
        }
      l[nt] = xa_l_update(l[nt], alpha, ps);
      // (One round-1 build of this kernel returned wrong values in dims 48-63; the cause was not this scaling -- whose packed
      // multiply was blamed at the time -- but accumulators read by the loop's back-edge copies 2-7 wait states after their
      // MFMA, through branches the compiler's hazard padding did not cover: DESIGN.md §4.3.  tools/check_isa_hazards.py
      // checks every build for that, across the control-flow graph.)
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) o[mt][nt] *= alpha;
    }
  };
  auto math_pv = [&]() {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int pr = 0; pr < T::NP; ++pr) o[mt][nt] = mfma16(vf[T::A[pr]][mt], pf[T::B[pr]][nt], o[mt][nt]);
  };
  auto math = [&](int step) {
    math_qk(step);
    math_pv();
  };

  // Everything the first DMAs depend on (the key bits) is in; ordinary loads are retired so that the counted waits below see
  // DMA instructions only.  The query fragments are loaded AFTER the ring is primed: their latency hides behind the first stages
  // (hipcc waits for them with vmcnt(0) at their first use, which the first stage has to reach anyway).
  wait_vm<0>();
  if constexpr (HALF) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int pc = 0; pc < S; ++pc)
#pragma unroll
        for (int kd = 0; kd < 2; ++kd) asm volatile("" : "+v"(qf[pc][nt][kd]));
    // half slot: K(s) -> fragments -> V(s) into the same region (in flight during S^T / softmax) -> fragments -> K(s+1) (in flight during
    // O^T += V^T P^T) ...: one half-stage in flight per wave, twice the waves per CU
    int cur = next(-1);
    if (cur < nsteps) issue_k(0, cur);
    while (cur < nsteps) {
      const int nxt = next(cur);
      wait_vm<0>();
      read_k(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the K fragments are in registers before the region is re-filled
      issue_v(0, cur);
      math_qk(cur);
      wait_vm<0>();
      read_v(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (nxt < nsteps) issue_k(0, nxt);
      math_pv();
      cur = nxt;
    }
  } else if constexpr (R == 1) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int pc = 0; pc < S; ++pc)
#pragma unroll
        for (int kd = 0; kd < 2; ++kd) asm volatile("" : "+v"(qf[pc][nt][kd]));  // (hipcc's own wait for the loads goes HERE)
    // One stage per wave, consumed and re-filled in HALVES: the K tiles of step s+1 are requested as soon as the K fragments of step
    // s are in registers (and fly during S^T, softmax), its V^T tiles as soon as the V^T fragments of step s are (and fly during
    // O^T += V^T P^T and the next S^T): the wave always has a half-stage in flight, where waiting for the whole stage and
    // re-filling it after the reads left nothing in flight for a few hundred cycles of every step.
    int cur = next(-1);
    if (cur < nsteps) issue(0, cur);
    while (cur < nsteps) {
      const int nxt = next(cur);
      wait_vm<4 * S>();  // all but the newest half-stage (this step's V^T tiles): its K tiles have landed
      read_k(0);
      if (nxt < nsteps) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the K fragments are in registers before their tiles are re-filled
        issue_k(0, nxt);
      }
#ifndef GRAM_XA_NOMATH
      math_qk(cur);
#endif
      if (nxt < nsteps) wait_vm<4 * S>();  // (newest: the next step's K tiles)
      else wait_vm<0>();
      read_v(0);
      if (nxt < nsteps) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        issue_v(0, nxt);
      }
#ifndef GRAM_XA_NOMATH
      math_pv();
#else
      o[0][0][0] += (float)kf[0][0][0][0] + (float)vf[0][0][0];
#endif
      cur = nxt;
    }
  } else if constexpr (R >= 2) {
    // ring of R stages: rq[i] = step held by slot i; `head` is the oldest.  All scalar state.
    int rq[R];
    int head = 0, inflight = 0, last = -1;
#pragma unroll
    for (int i = 0; i < R; ++i) {
      rq[i] = nsteps;
      if (last < nsteps) {
        last = next(last);
        if (last < nsteps) {
          rq[i] = last;
          issue(i, last);
          ++inflight;
        }
      }
    }
    // (query loads)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int beam = 16 * nt + c;
      int qrow = beam < K ? b * K + beam : -1;
      if constexpr (LIVE) {
        if (qrow >= 0) qrow = rowpos[qrow];
      }
  #pragma unroll
      for (int pc = 0; pc < S; ++pc)
  #pragma unroll
        for (int kd = 0; kd < 2; ++kd)
          qf[pc][nt][kd] = qrow >= 0 ? ld_global_b128(q + pc * q_pstride + (size_t)qrow * inner + h * 64 + 32 * kd + 8 * g) : zero_bf16x8();
    }

    while (inflight > 0) {
      // all but the (inflight - 1) newer stages' DMAs have landed: the head stage is complete
      switch (inflight - 1) {
        case 0: wait_vm<0>(); break;
        case 1: wait_vm<8 * S>(); break;
        case 2: wait_vm<(R > 2 ? 16 * S : 0)>(); break;
        default: wait_vm<(R > 3 ? 24 * S : 0)>(); break;
      }
      int cur = rq[0];
#pragma unroll
      for (int i = 1; i < R; ++i) cur = head == i ? rq[i] : cur;
      read_frags(head);
      --inflight;
      if (last < nsteps) last = next(last);
      if (last < nsteps) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this slot's fragment reads have returned before it is re-filled
        issue(head, last);
#pragma unroll
        for (int i = 0; i < R; ++i) rq[i] = head == i ? last : rq[i];
        ++inflight;
      }
#ifndef GRAM_XA_NOMATH
      math(cur);
#else
      o[0][0][0] += (float)kf[0][0][0][0] + (float)vf[0][0][0];  // (probe build: the stream without the arithmetic)
#endif
      head = head + 1 == R ? 0 : head + 1;
    }
  }

  if constexpr (NW == 1) {
    // one wave owns the whole (user, head): its accumulators are the result (lane: beam 16 nt + c, dims 16 mt + 4 g ..+3)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      float lt = l[nt];
      lt += __shfl_xor(lt, 16, 64);
      lt += __shfl_xor(lt, 32, 64);
      const float inv = 1.f / lt;
      const int beam = 16 * nt + c;
      int orow = beam < K ? b * K + beam : -1;
      if constexpr (LIVE) {
        if (orow >= 0) orow = rowpos[orow];
      }
      if (orow >= 0) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          f32x4 v = o[mt][nt] * inv;
#pragma unroll
          for (int pc = 0; pc < S; ++pc) {
            p16x4 r;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              r[e] = (p16)v[e];
              v[e] -= (float)r[e];
            }
            const int n = h * 64 + 16 * mt + 4 * g;  // (S == 2: interleaved rows [2 * inner], the O GEMM's A operand)
            *reinterpret_cast<p16x4*>(out + (size_t)orow * inner * S + (S == 2 ? inter_off(n, pc) : n)) = r;
          }
        }
      }
    }
    return;
  }
  // merge the waves' partials (the rings are dead once every wave is past its loop)
  gram_sync();
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    float lt = l[nt];
    lt += __shfl_xor(lt, 16, 64);
    lt += __shfl_xor(lt, 32, 64);
    const int beam = 16 * nt + c;
    if (g == 0) {
      sm_m[wave * NB + beam] = m[nt];
      sm_l[wave * NB + beam] = lt;
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
      *reinterpret_cast<f32x4*>(sm_o + ((size_t)(wave * NB + beam)) * 64 + 16 * mt + 4 * g) = o[mt][nt];
  }
  gram_sync();
  for (int idx = tid; idx < K * 16; idx += NW * 64) {
    const int beam = idx >> 4, d4 = (idx & 15) * 4;
    if constexpr (NW == 2) {
      float w0, w1, inv2;
      xa_merge_w(sm_m[beam], sm_m[NB + beam], sm_l[beam], sm_l[NB + beam], w0, w1, inv2);
      f32x4 v = xa_merge_o(*reinterpret_cast<const f32x4*>(sm_o + ((size_t)beam) * 64 + d4),
                           *reinterpret_cast<const f32x4*>(sm_o + ((size_t)(NB + beam)) * 64 + d4), w0, w1, inv2);
      int orow = b * K + beam;
      if constexpr (LIVE) orow = rowpos[orow];
      if (orow >= 0) {
#pragma unroll
        for (int pc = 0; pc < S; ++pc) {
          p16x4 r;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            r[e] = (p16)v[e];
            v[e] -= (float)r[e];
          }
          const int n = h * 64 + d4;
          *reinterpret_cast<p16x4*>(out + (size_t)orow * inner * S + (S == 2 ? inter_off(n, pc) : n)) = r;
        }
      }
      continue;
    }
    float M = GRAM_FMIN;
#pragma unroll
    for (int w = 0; w < NW; ++w) M = fmaxf(M, sm_m[w * NB + beam]);
    float Lsum = 0.f;
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      const float wgt = __expf(sm_m[w * NB + beam] - M);
      Lsum += sm_l[w * NB + beam] * wgt;
      acc += *reinterpret_cast<const f32x4*>(sm_o + ((size_t)(w * NB + beam)) * 64 + d4) * wgt;
    }
    const float inv = 1.f / Lsum;
    f32x4 v = acc * inv;
    int orow = b * K + beam;
    if constexpr (LIVE) orow = rowpos[orow];
    if (orow >= 0) {
#pragma unroll
      for (int pc = 0; pc < S; ++pc) {
        p16x4 r;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          r[e] = (p16)v[e];
          v[e] -= (float)r[e];
        }
        const int n = h * 64 + d4;
        *reinterpret_cast<p16x4*>(out + (size_t)orow * inner * S + (S == 2 ? inter_off(n, pc) : n)) = r;
      }
    }
  }
}


template <int NT, int S, int NW, int R>
int launch_cross(const void* q, const void* k, const void* vt, const uint8_t* mask, void* out, int B, int K, int H, int Sk,
                 const int32_t* users, const int32_t* rowpos, long q_ps, long bank_ps, const uint32_t* key_bits, hipStream_t st) {
  constexpr int NB = NT * 16;
  constexpr int ring = NW * (R == 0 ? S * XA_TILE : R * S * 2 * XA_TILE), merge = NW == 1 ? 0 : (2 * NW * NB + NW * NB * 64) * 4;
  constexpr int smem = ring > merge ? ring : merge;
  static_assert(smem <= 160 * 1024, "ring does not fit the LDS");
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(cross_attn_kernel<NT, false, S, NW, R>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e == hipSuccess)
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(cross_attn_kernel<NT, true, S, NW, R>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  if (users)
    hipLaunchKernelGGL((cross_attn_kernel<NT, true, S, NW, R>), dim3(H, B), dim3(NW * 64), smem, st, (const p16*)q, (const p16*)k,
                       (const p16*)vt, mask, (p16*)out, K, H, Sk, users, rowpos, q_ps, bank_ps, key_bits);
  else
    hipLaunchKernelGGL((cross_attn_kernel<NT, false, S, NW, R>), dim3(H, B), dim3(NW * 64), smem, st, (const p16*)q, (const p16*)k,
                       (const p16*)vt, mask, (p16*)out, K, H, Sk, users, rowpos, q_ps, bank_ps, key_bits);
  GRAM_CHECK_LAUNCH();
  return 0;
}

// Waves per workgroup and ring depth per piece count (measured, profiles/r02_cross_attn_variants.json); GRAM_XA_VARIANT = 10*NW + R
// selects one of the other instantiated shapes for an A/B run.
#ifndef GRAM_XA_AB
#define GRAM_XA_AB 0
#endif
template <int NT, int S>
int launch_cross_v(const void* q, const void* k, const void* vt, const uint8_t* mask, void* out, int B, int K, int H, int Sk,
                   const int32_t* users, const int32_t* rowpos, long q_ps, long bank_ps, const uint32_t* key_bits, hipStream_t st) {
#define XA_ARGS q, k, vt, mask, out, B, K, H, Sk, users, rowpos, q_ps, bank_ps, key_bits, st
#if GRAM_XA_AB
  static const int v = getenv("GRAM_XA_VARIANT") ? atoi(getenv("GRAM_XA_VARIANT")) : 0;
  if constexpr (NT == 2 && S == 1) {
    switch (v) {
      case 22: return launch_cross<NT, S, 2, 2>(XA_ARGS);
      case 23: return launch_cross<NT, S, 2, 3>(XA_ARGS);
      case 24: return launch_cross<NT, S, 2, 4>(XA_ARGS);
      case 43: return launch_cross<NT, S, 4, 3>(XA_ARGS);
      case 11: return launch_cross<NT, S, 1, 1>(XA_ARGS);
      case 12: return launch_cross<NT, S, 1, 2>(XA_ARGS);
      case 13: return launch_cross<NT, S, 1, 3>(XA_ARGS);
      case 14: return launch_cross<NT, S, 1, 4>(XA_ARGS);
      case 21: return launch_cross<NT, S, 2, 1>(XA_ARGS);
      default: break;
    }
  }
  if constexpr (NT == 4 && S <= 2) {
    switch (v) {
      case 11: return launch_cross<NT, S, 1, 1>(XA_ARGS);
      case 12: return launch_cross<NT, S, 1, 2>(XA_ARGS);
      case 22: return launch_cross<NT, S, 2, 2>(XA_ARGS);
      case 41: return launch_cross<NT, S, 4, 1>(XA_ARGS);
      case 42: return launch_cross<NT, S, 4, 2>(XA_ARGS);
      default: break;
    }
  }
  if constexpr (NT == 2 && S == 2) {
    switch (v) {
      case 42: return launch_cross<NT, S, 4, 2>(XA_ARGS);
      case 22: return launch_cross<NT, S, 2, 2>(XA_ARGS);
      case 23: return launch_cross<NT, S, 2, 3>(XA_ARGS);
      case 11: return launch_cross<NT, S, 1, 1>(XA_ARGS);
      case 12: return launch_cross<NT, S, 1, 2>(XA_ARGS);
      case 13: return launch_cross<NT, S, 1, 3>(XA_ARGS);
      case 14: return launch_cross<NT, S, 1, 4>(XA_ARGS);
      case 21: return launch_cross<NT, S, 2, 1>(XA_ARGS);
      case 20: return launch_cross<NT, S, 2, 0>(XA_ARGS);
      case 10: return launch_cross<NT, S, 1, 0>(XA_ARGS);
      case 40: return launch_cross<NT, S, 4, 0>(XA_ARGS);
      default: break;
    }
  }
#endif
  // two waves per (user, head), ONE stage each: many small workgroups per CU hide the per-workgroup prologue / merge better than
  // deep rings do (in-run A/B at the bench shape, tests/bench_xattn.py: (NW, R) = (2, 1) 5.29 / 5.43 TB/s for 1 / 2 pieces against
  // (1, 2) 5.11 / 5.21, (2, 2) 5.21 / 4.84, (4, 2) 4.49 / 3.88 on the same box)
  // (three or four beam tiles, K > 32: the per-workgroup state is larger and one wave per (user, head) with two stages wins --
  // K = 50, S = 2 688, H = 16: (1, 2) 4.32 / 5.28 TB/s against (2, 1) 3.93 / 4.99, (2, 2) 4.08 / 4.95, (4, 1) 3.66 / 4.75)
  // (a ring of 3 stages per wave for grids of a handful of users was measured too: 12.42 -> 12.67 ms per one-user generate, no gain)
  // (round 3, R = 0 "half slot": one 8-KB region per wave for a step's K tiles, then its V^T tiles -- half the LDS, twice the workgroups per
  // CU: (2, 0) 5.80, (1, 0) 5.80, (4, 0) 5.77 TB/s against 5.75-5.94 for (2, 1) in the same run: no gain, profiles/r03_cross_attn_half_slot_ab.txt)
  // (with the stage consumed and re-filled in halves, ONE stage wins there too: K = 50, S = 2 688, H = 16, same box: (1, 1) 4.65 / 5.75 TB/s
  // against (1, 2) 3.97 / 5.12, (4, 1) 4.19 / 5.37, (2, 2) 3.72 / 5.02 -- profiles/r02m_cross_attn_k50_variants.txt)
  // (round 4, both open ideas of round 3 built and measured, neither kept -- commit e681937, profiles/r04e_*, r04f_*: a PERSISTENT-WAVE
  // kernel -- one wave per strided list of (user, head) items, the next item's first tiles requested before the current one is merged and
  // stored, the two waves as two accumulator sets, bit-identical to this kernel on 8 shapes incl. live rows -- ran the bench's
  // cross-attention in 164.4 ms per step against 160.8 (its 205 VGPRs leave 8 waves per CU where five two-wave workgroups fill the LDS);
  // and the bank addressed as STEP-MAJOR 16-KB RECORDS (read side only) read 5.78 TB/s against 5.86 for the planar bank, with +-5 %
  // between processes for either: the per-item prologue / tail and the four planar streams are not what holds the kernel at ~0.87 of the
  // box's streaming read)
  if constexpr (NT >= 3) return launch_cross<NT, S, 1, 1>(XA_ARGS);
  else return launch_cross<NT, S, 2, 1>(XA_ARGS);
#undef XA_ARGS
}

template <int S>
int launch_cross_nt(const void* q, const void* k, const void* vt, const uint8_t* mask, void* out, int B, int K, int H, int Sk,
                    const int32_t* users, const int32_t* rowpos, long q_ps, long bank_ps, const uint32_t* key_bits, hipStream_t st) {
  switch ((K + 15) / 16) {
    case 1: return launch_cross_v<1, S>(q, k, vt, mask, out, B, K, H, Sk, users, rowpos, q_ps, bank_ps, key_bits, st);
    case 2: return launch_cross_v<2, S>(q, k, vt, mask, out, B, K, H, Sk, users, rowpos, q_ps, bank_ps, key_bits, st);
    case 3: return launch_cross_v<3, S>(q, k, vt, mask, out, B, K, H, Sk, users, rowpos, q_ps, bank_ps, key_bits, st);
    default: return launch_cross_v<4, S>(q, k, vt, mask, out, B, K, H, Sk, users, rowpos, q_ps, bank_ps, key_bits, st);
  }
}

// key_bits[b][st] bit j = mask[b][32 st + j] != 0 (st < S/32 <= 128; rows are 128 words apart): the cross-attention's view of
// the mask, the same for every head, layer and decode step of a generate() -- computed once instead of per workgroup
__global__ __launch_bounds__(128) void mask_key_bits_kernel(const uint8_t* __restrict__ mask, uint32_t* __restrict__ bits, int Sk) {
  const int b = blockIdx.x, st = threadIdx.x;
  uint32_t w = 0;
  if (st < (Sk >> 5)) {
    const uint4* p = reinterpret_cast<const uint4*>(mask + (size_t)b * Sk + 32 * st);
    const uint4 a = p[0], c2 = p[1];
    const uint32_t w8[8] = {a.x, a.y, a.z, a.w, c2.x, c2.y, c2.z, c2.w};
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) w |= ((w8[i] >> (8 * j)) & 0xffu) ? (1u << (4 * i + j)) : 0u;
  }
  bits[(size_t)b * 128 + st] = w;
}

// ---------------------------------------------------------------------------------------------
template <int S>
__global__ __launch_bounds__(256) void dec_self_attn_kernel(const p16* __restrict__ qkv, p16* __restrict__ kcache,
                                                            p16* __restrict__ vcache, const int32_t* __restrict__ anc,
                                                            const float* __restrict__ bias, p16* __restrict__ out, int R,
                                                            int H, int t, const int32_t* __restrict__ rows, long qkv_ps,
                                                            long cache_ps) {
  // live-row step (rows != NULL): qkv/out are indexed by the compact row, the cache and the ancestor table by the
  // original row rows[compact]; R stays the row count of the cache
  // (giving each XCD a contiguous eighth of the rows, so that a user's beams share ancestors' cache rows in ONE L2, was measured:
  // 41.9 ms per step against 41.2 for the round-robin order below)
  const int rc = blockIdx.x, i = threadIdx.x;  // i: 16 threads per head, 4 dims each
  const int r = rows ? rows[rc] : rc;
  const int inner = H * 64, h = i >> 4;
  const p16* row = qkv + (size_t)rc * 3 * inner + 4 * i;
  // values as the fp32 sum of their pieces (exact for two pieces, one rounding for three); the pieces themselves go to the cache
  float qf[4] = {0.f, 0.f, 0.f, 0.f}, kn[4] = {0.f, 0.f, 0.f, 0.f}, vn[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int pc = S - 1; pc >= 0; --pc) {  // smallest piece first
    const p16x4 q4 = *reinterpret_cast<const p16x4*>(row + pc * qkv_ps);
    const p16x4 k4 = *reinterpret_cast<const p16x4*>(row + pc * qkv_ps + inner);
    const p16x4 v4 = *reinterpret_cast<const p16x4*>(row + pc * qkv_ps + 2 * inner);
    *reinterpret_cast<p16x4*>(kcache + pc * cache_ps + ((size_t)t * R + r) * inner + 4 * i) = k4;
    *reinterpret_cast<p16x4*>(vcache + pc * cache_ps + ((size_t)t * R + r) * inner + 4 * i) = v4;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      qf[e] += (float)q4[e];
      kn[e] += (float)k4[e];
      vn[e] += (float)v4[e];
    }
  }
  float m = -INFINITY, l = 0.f, acc[4] = {0.f, 0.f, 0.f, 0.f};
  auto attend = [&](int j, const float (&kj)[4], const float (&vj)[4]) {  // online softmax over the positions, in order
    float s = qf[0] * kj[0] + qf[1] * kj[1] + qf[2] * kj[2] + qf[3] * kj[3];
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    s += __shfl_xor(s, 4, 64);
    s += __shfl_xor(s, 8, 64);
    s += bias[h * GRAM_MAX_DEC_LEN + (t - j)];
    const float mn = fmaxf(m, s);
    const float alpha = __expf(m - mn);
    const float p = __expf(s - mn);
    m = mn;
    l = l * alpha + p;
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] = acc[e] * alpha + p * vj[e];
  };
  // Cached positions in batches of U: the U ancestor indices go out together, then the 2 * S * U row loads they address -- two
  // round trips per batch.  (One position per loop trip was ancestor load -> row loads -> arithmetic, strictly in sequence:
  // 2 t dependent round trips per block, 7 us per block at the bench shape.)  The arithmetic and its order are unchanged.
  constexpr int U = 8;
  for (int j0 = 0; j0 < t; j0 += U) {
    int a[U];
#pragma unroll
    for (int u = 0; u < U; ++u) a[u] = anc[(size_t)min(j0 + u, t - 1) * R + r];
    p16x4 kk[U][S], vv[U][S];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t off = ((size_t)min(j0 + u, t - 1) * R + a[u]) * inner + 4 * i;
#pragma unroll
      for (int pc = 0; pc < S; ++pc) {
        kk[u][pc] = ld_stream_b64(kcache + pc * cache_ps + off);
        vv[u][pc] = ld_stream_b64(vcache + pc * cache_ps + off);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (j0 + u < t) {
        float kj[4] = {0.f, 0.f, 0.f, 0.f}, vj[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int pc = S - 1; pc >= 0; --pc)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            kj[e] += (float)kk[u][pc][e];
            vj[e] += (float)vv[u][pc][e];
          }
        attend(j0 + u, kj, vj);
      }
    }
  }
  attend(t, kn, vn);
  const float inv = 1.f / l;
#pragma unroll
  for (int e = 0; e < 4; ++e) acc[e] *= inv;
#pragma unroll
  for (int pc = 0; pc < S; ++pc) {
    p16x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      o[e] = (p16)acc[e];
      acc[e] -= (float)o[e];
    }
    // (S == 2: interleaved rows [2 * inner], the O GEMM's A operand)
    *reinterpret_cast<p16x4*>(out + (size_t)rc * inner * S + (S == 2 ? inter_off(4 * i, pc) : 4 * i)) = o;
  }
}

}  // namespace

extern "C" int gram_mask_key_bits(const uint8_t* mask, uint32_t* key_bits, int B, int S, void* stream) {
  if (!mask || !key_bits || B < 1 || S < 32 || (S & 31) || S > 4096 || (reinterpret_cast<uintptr_t>(mask) & 15)) return GRAM_E_ARG;
  hipLaunchKernelGGL(mask_key_bits_kernel, dim3(B), dim3(128), 0, (hipStream_t)stream, mask, key_bits, S);
  GRAM_CHECK_LAUNCH();
  return 0;
}

extern "C" int gram_cross_attn_decode_split(const void* q, const void* k_layer, const void* vt_layer, const uint8_t* mask, void* out,
                                            int B, int K, int H, int S, const int32_t* users, const int32_t* rowpos, int pieces,
                                            int64_t q_pstride, int64_t bank_pstride, const uint32_t* key_bits, void* stream) {
  if (B < 1 || K < 1 || K > GRAM_MAX_BEAMS || H < 1 || S < 32 || (S & 31) || S > 4096 || pieces < 1 || pieces > GRAM_MAX_PIECES ||
      (users == nullptr) != (rowpos == nullptr))
    return GRAM_E_ARG;
  if (pieces > 1 && (q_pstride < 1 || bank_pstride < (int64_t)H * S * 64)) return GRAM_E_ARG;
  if ((reinterpret_cast<uintptr_t>(mask) & 15) || (reinterpret_cast<uintptr_t>(k_layer) & 15) || (reinterpret_cast<uintptr_t>(vt_layer) & 15))
    return GRAM_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  gram_prof::Scope prof(GRAM_K_CROSS_ATTN, st, 4.0 * B * H * S * 64 * pieces);  // K + V^T, bf16, every piece
  if (pieces == 2) return launch_cross_nt<2>(q, k_layer, vt_layer, mask, out, B, K, H, S, users, rowpos, q_pstride, bank_pstride, key_bits, st);
  return launch_cross_nt<1>(q, k_layer, vt_layer, mask, out, B, K, H, S, users, rowpos, q_pstride, bank_pstride, key_bits, st);
}

extern "C" int gram_cross_attn_decode(const void* q, const void* k_layer, const void* vt_layer, const uint8_t* mask, void* out,
                                      int B, int K, int H, int S, void* stream) {
  return gram_cross_attn_decode_split(q, k_layer, vt_layer, mask, out, B, K, H, S, nullptr, nullptr, 1, 0, 0, nullptr, stream);
}

extern "C" int gram_cross_attn_decode_live(const void* q, const void* k_layer, const void* vt_layer, const uint8_t* mask, void* out,
                                           int n_users, const int32_t* users, const int32_t* rowpos, int K, int H, int S,
                                           void* stream) {
  if (!users || !rowpos) return GRAM_E_ARG;
  return gram_cross_attn_decode_split(q, k_layer, vt_layer, mask, out, n_users, K, H, S, users, rowpos, 1, 0, 0, nullptr, stream);
}

extern "C" int gram_dec_self_attn_split(const void* qkv, void* kcache, void* vcache, const int32_t* anc, const float* bias, void* out,
                                        int R, int n_rows, const int32_t* rows, int H, int t, int Tmax, int pieces,
                                        int64_t qkv_pstride, int64_t cache_pstride, void* stream) {
  if (R < 1 || n_rows < 1 || n_rows > R || H < 1 || H > 16 || t < 0 || t >= Tmax || Tmax > GRAM_MAX_DEC_LEN || pieces < 1 ||
      pieces > GRAM_MAX_PIECES)
    return GRAM_E_ARG;
  if (pieces > 1 && (qkv_pstride < 1 || cache_pstride < (int64_t)Tmax * R * H * 64)) return GRAM_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  gram_prof::Scope prof(GRAM_K_DEC_SELF_ATTN, st, 4.0 * n_rows * H * 64 * (t + 1) * pieces);
  const dim3 grid(n_rows), block(H * 16);
  if (pieces == 2)
    hipLaunchKernelGGL(dec_self_attn_kernel<2>, grid, block, 0, st, (const p16*)qkv, (p16*)kcache, (p16*)vcache, anc, bias,
                       (p16*)out, R, H, t, rows, (long)qkv_pstride, (long)cache_pstride);
  else
    hipLaunchKernelGGL(dec_self_attn_kernel<1>, grid, block, 0, st, (const p16*)qkv, (p16*)kcache, (p16*)vcache, anc, bias,
                       (p16*)out, R, H, t, rows, (long)qkv_pstride, (long)cache_pstride);
  GRAM_CHECK_LAUNCH();
  return 0;
}

extern "C" int gram_dec_self_attn(const void* qkv, void* kcache, void* vcache, const int32_t* anc, const float* bias, void* out,
                                  int R, int H, int t, int Tmax, void* stream) {
  return gram_dec_self_attn_split(qkv, kcache, vcache, anc, bias, out, R, R, nullptr, H, t, Tmax, 1, 0, 0, stream);
}

extern "C" int gram_dec_self_attn_live(const void* qkv, void* kcache, void* vcache, const int32_t* anc, const float* bias,
                                       void* out, int R, int n_rows, const int32_t* rows, int H, int t, int Tmax, void* stream) {
  if (!rows) return GRAM_E_ARG;
  return gram_dec_self_attn_split(qkv, kcache, vcache, anc, bias, out, R, n_rows, rows, H, t, Tmax, 1, 0, 0, stream);
}
