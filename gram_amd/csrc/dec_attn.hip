// dec_attn.hip -- the two decoder attention kernels of one decode step.
//
// (1) cross_attn_kernel: late-fusion cross-attention, THE HBM-bound kernel of the path.
//     Reference: T5Attention.forward cross branch, gram_t5_modeling.py:531-534,547-549,572-622
//     (zero position bias :577-582, additive mask :1145-1147), called per layer per step on
//     K-times replicated K/V.  Here ONE bank per user is streamed ONCE per (layer, step) and
//     shared by all of the user's beams: algorithmic bytes = 2 * S * 64 * 2 B per (user, head).
//
//     Mapping: one workgroup per (user, head), 4 waves, wave w owns the 32-key steps
//     w, w+4, ...  Per step a wave issues 8 x 16-byte-per-lane global loads (K rows as the
//     MFMA A operand, V^T rows as the A operand of the second product -- both k-contiguous in
//     the bank layouts of gram_hip.h, so nothing is transposed on chip), software-pipelined one
//     step ahead (two named register sets).  S^T = K Q^T puts a beam on a lane column, so the
//     online softmax is in-register + two cross-lane steps, and exp(S^T) is directly the B
//     operand of O^T = V^T P^T (same row-permutation trick as enc_attn.hip).  Waves merge their
//     (m, l, O) partials through LDS at the end.
//
// (2) dec_self_attn_kernel: causal self-attention of the newest token over <= 32 cached
//     positions with beam-parent indirection (anc table) instead of the reference's
//     torch.cat + index_select of the whole cache (gram_t5_modeling.py:536-540,
//     gram_t5.py:320-348); unidirectional relative bias, last query row (:586-593).
#include "common.h"
#include "prof.h"

namespace {

#ifndef STEP_SHIFT
#define STEP_SHIFT 0  // measured: adjacent-step runs per wave (1, 2) are 5-7 % slower than interleaving
#endif
struct StepRegs {
  bf16x8 kf[2][2];
  bf16x8 vf[4];
  uint2 mk;
};

template <int NT, bool LIVE>  // LIVE: live-row step (gram_live_rows_t); a separate instantiation keeps the common kernel's code as it was
__global__ __launch_bounds__(256, (NT <= 2 ? 2 : 1)) void cross_attn_kernel(
    const bf16* __restrict__ q, const bf16* __restrict__ kbank, const bf16* __restrict__ vtbank,
    const uint8_t* __restrict__ mask, bf16* __restrict__ out, int K, int H, int S, const int32_t* __restrict__ users,
    const int32_t* __restrict__ rowpos) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NB = NT * 16;                               // padded beams
  float* sm_m = reinterpret_cast<float*>(smem);             // [4][NB]
  float* sm_l = sm_m + 4 * NB;                              // [4][NB]
  float* sm_o = sm_l + 4 * NB;                              // [4][NB][64]
  unsigned long long* sm_valid = reinterpret_cast<unsigned long long*>(sm_o + 4 * NB * 64);  // [2] valid-step bits

  // live-row step (users != NULL): workgroup y serves user users[y]; q/out rows are the compact rows rowpos[b*K + beam]
  // (-1 = beam not live: zero query, nothing stored)
  const int h = blockIdx.x, b = LIVE ? users[blockIdx.y] : blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, g = lane >> 4;
  const int inner = H * 64;
  const bf16* kb = kbank + ((size_t)b * H + h) * S * 64;
  const bf16* vt = vtbank + ((size_t)b * H + h) * 64 * S;
  const uint8_t* mk = mask + (size_t)b * S;

  bf16x8 qf[NT][2];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int beam = 16 * nt + c;
    int qrow = beam < K ? b * K + beam : -1;
    if constexpr (LIVE) {
      if (qrow >= 0) qrow = rowpos[qrow];
    }
#pragma unroll
    for (int kd = 0; kd < 2; ++kd)
      qf[nt][kd] = qrow >= 0 ? ld_global_b128(q + (size_t)qrow * inner + h * 64 + 32 * kd + 8 * g) : zero_bf16x8();
  }

  f32x4 o[4][NT];
  float m[NT], l[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    m[nt] = GRAM_FMIN;
    l[nt] = 0.f;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) o[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }

  const int krow = 8 * (c >> 2) + (c & 3);  // + 4t: key row of S^T tile t this lane feeds
  auto load = [&](StepRegs& r, int step) {
    const bf16* kp = kb + (size_t)(32 * step + krow) * 64 + 8 * g;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int kd = 0; kd < 2; ++kd) r.kf[t][kd] = ld_global_b128(kp + t * 4 * 64 + 32 * kd);
    const bf16* vp = vt + (size_t)c * S + 32 * step + 8 * g;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) r.vf[mt] = ld_global_b128(vp + (size_t)16 * mt * S);
    r.mk = *reinterpret_cast<const uint2*>(mk + 32 * step + 8 * g);
  };
  auto compute = [&](const StepRegs& r) {
    bf16x8 pf[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      f32x4 s[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        f32x4 a = (f32x4){0.f, 0.f, 0.f, 0.f};
        a = mfma16(r.kf[t][0], qf[nt][0], a);
        a = mfma16(r.kf[t][1], qf[nt][1], a);
        const uint32_t mb = t == 0 ? r.mk.x : r.mk.y;
#pragma unroll
        for (int j = 0; j < 4; ++j) a[j] = ((mb >> (8 * j)) & 0xffu) ? a[j] : GRAM_FMIN;
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
      bf16x8 f;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float e = __expf(s[t][j] - mn);
          ps += e;
          f[4 * t + j] = (bf16)e;
        }
      pf[nt] = f;
      l[nt] = l[nt] * alpha + ps;
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) o[mt][nt] *= alpha;
      // Keep alpha's register live past the scaling.  hipcc (ROCm 7.2) lowers the scaling to
      // v_pk_mul_f32 with alpha broadcast from ONE register (op_sel_hi 0); when alpha dies here the
      // allocator may reuse that register as the low half of a destination pair, and on gfx950 the
      // instruction's high lane then multiplies by the freshly written low RESULT instead of alpha
      // (observed: v_pk_mul_f32 v[138:139], v[46:47], v[138:139] op_sel_hi:[1,0] -> element 1 of one
      // accumulator tile wrong whenever alpha != 1).  tools/check_isa_hazards.py scans every build.
      asm volatile("" ::"v"(alpha));
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) o[mt][nt] = mfma16(r.vf[mt], pf[nt], o[mt][nt]);
  };

  // Fully masked 32-key steps (padded passages / padded tails) are never fetched: the algorithmic
  // traffic is proportional to the VALID fused keys.  One bit per step (S <= 4096 -> <= 128 steps);
  // the valid steps are dealt round-robin to the four waves.  A user with no valid key at all keeps
  // every step: the reference's softmax over all-finfo.min scores is uniform over all S keys.
  const int nsteps = S >> 5;
  {
    bool flag = false;
    if (tid < nsteps) {
      const uint4* p = reinterpret_cast<const uint4*>(mk + 32 * tid);
      const uint4 a = p[0], c2 = p[1];
      flag = (a.x | a.y | a.z | a.w | c2.x | c2.y | c2.z | c2.w) != 0u;
    }
    const unsigned long long bal = __ballot(flag);
    if (lane == 0 && wave < 2) sm_valid[wave] = bal;
  }
  __syncthreads();
  unsigned long long v0 = sm_valid[0], v1 = sm_valid[1];
  if ((v0 | v1) == 0ull) {
    v0 = nsteps >= 64 ? ~0ull : ((1ull << nsteps) - 1ull);
    v1 = nsteps > 64 ? ((nsteps >= 128 ? ~0ull : ((1ull << (nsteps - 64)) - 1ull))) : 0ull;
  }
  v0 = __builtin_amdgcn_readfirstlane((unsigned)v0) | ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(v0 >> 32)) << 32);
  v1 = __builtin_amdgcn_readfirstlane((unsigned)v1) | ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(v1 >> 32)) << 32);
  const int wv = __builtin_amdgcn_readfirstlane(wave);
  int ord = 0;  // ordinal of the next valid step (wave-uniform scalar state)
  auto next = [&](int s) -> int {
    for (s = s + 1; s < nsteps; ++s) {
      const unsigned long long bit = (s < 64 ? (v0 >> s) : (v1 >> (s - 64))) & 1ull;
      if (bit) {
        const bool mine = ((ord >> STEP_SHIFT) & 3) == wv;  // runs of 2^STEP_SHIFT adjacent valid steps per wave
        ++ord;
        if (mine) return s;
      }
    }
    return nsteps;
  };
  // Two named register sets, one step ahead.  Every `load; compute` pair sits in ONE basic block with an
  // unconditional load: with `if (more) load(...)` in front of compute(), hipcc cannot count the loads in
  // flight at the join and waits vmcnt(0) -- i.e. also for the loads it has just issued (no overlap at all
  // in every other iteration).
  StepRegs ra, rb;
  int i = next(-1);
  if (i < nsteps) {
    load(ra, i);
    int nx = next(i);
    while (true) {
      if (nx >= nsteps) {
        compute(ra);
        break;
      }
      load(rb, nx);
      compute(ra);
      i = next(nx);
      if (i >= nsteps) {
        compute(rb);
        break;
      }
      load(ra, i);
      compute(rb);
      nx = next(i);
    }
  }

  // merge the four waves' partials
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
  __syncthreads();
  for (int idx = tid; idx < K * 16; idx += 256) {
    const int beam = idx >> 4, d4 = (idx & 15) * 4;
    float M = GRAM_FMIN;
#pragma unroll
    for (int w = 0; w < 4; ++w) M = fmaxf(M, sm_m[w * NB + beam]);
    float Lsum = 0.f;
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float wgt = __expf(sm_m[w * NB + beam] - M);
      Lsum += sm_l[w * NB + beam] * wgt;
      acc += *reinterpret_cast<const f32x4*>(sm_o + ((size_t)(w * NB + beam)) * 64 + d4) * wgt;
    }
    const float inv = 1.f / Lsum;
    bf16x4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = (bf16)(acc[e] * inv);
    if constexpr (LIVE) {
      const int orow = rowpos[b * K + beam];
      if (orow >= 0) *reinterpret_cast<bf16x4*>(out + (size_t)orow * inner + h * 64 + d4) = r;
    } else {
      *reinterpret_cast<bf16x4*>(out + ((size_t)b * K + beam) * inner + h * 64 + d4) = r;
    }
  }
}

template <int NT>
int launch_cross(const void* q, const void* k, const void* vt, const uint8_t* mask, void* out, int B, int K, int H, int S,
                 const int32_t* users, const int32_t* rowpos, hipStream_t st) {
  const size_t smem = (size_t)(2 * 4 * NT * 16 + 4 * NT * 16 * 64) * sizeof(float) + 16;
  static bool attr_set = false;
  if (!attr_set && smem > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(cross_attn_kernel<NT, false>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e == hipSuccess)
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(cross_attn_kernel<NT, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)smem);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  if (users)
    hipLaunchKernelGGL((cross_attn_kernel<NT, true>), dim3(H, B), dim3(256), smem, st, (const bf16*)q, (const bf16*)k,
                       (const bf16*)vt, mask, (bf16*)out, K, H, S, users, rowpos);
  else
    hipLaunchKernelGGL((cross_attn_kernel<NT, false>), dim3(H, B), dim3(256), smem, st, (const bf16*)q, (const bf16*)k,
                       (const bf16*)vt, mask, (bf16*)out, K, H, S, users, rowpos);
  GRAM_CHECK_LAUNCH();
  return 0;
}

// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dec_self_attn_kernel(const bf16* __restrict__ qkv, bf16* __restrict__ kcache,
                                                            bf16* __restrict__ vcache, const int32_t* __restrict__ anc,
                                                            const float* __restrict__ bias, bf16* __restrict__ out, int R,
                                                            int H, int t, const int32_t* __restrict__ rows) {
  // live-row step (rows != NULL): qkv/out are indexed by the compact row, the cache and the ancestor table by the
  // original row rows[compact]; R stays the row count of the cache
  const int rc = blockIdx.x, i = threadIdx.x;  // i: 16 threads per head, 4 dims each
  const int r = rows ? rows[rc] : rc;
  const int inner = H * 64, h = i >> 4;
  const bf16* row = qkv + (size_t)rc * 3 * inner + 4 * i;
  const bf16x4 q4 = *reinterpret_cast<const bf16x4*>(row);
  const bf16x4 k4 = *reinterpret_cast<const bf16x4*>(row + inner);
  const bf16x4 v4 = *reinterpret_cast<const bf16x4*>(row + 2 * inner);
  *reinterpret_cast<bf16x4*>(kcache + ((size_t)t * R + r) * inner + 4 * i) = k4;
  *reinterpret_cast<bf16x4*>(vcache + ((size_t)t * R + r) * inner + 4 * i) = v4;
  float qf[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) qf[e] = (float)q4[e];
  float m = -INFINITY, l = 0.f, acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int j = 0; j <= t; ++j) {
    bf16x4 kj = k4, vj = v4;
    if (j < t) {
      const int a = anc[(size_t)j * R + r];
      kj = *reinterpret_cast<const bf16x4*>(kcache + ((size_t)j * R + a) * inner + 4 * i);
      vj = *reinterpret_cast<const bf16x4*>(vcache + ((size_t)j * R + a) * inner + 4 * i);
    }
    float s = qf[0] * (float)kj[0] + qf[1] * (float)kj[1] + qf[2] * (float)kj[2] + qf[3] * (float)kj[3];
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
    for (int e = 0; e < 4; ++e) acc[e] = acc[e] * alpha + p * (float)vj[e];
  }
  const float inv = 1.f / l;
  bf16x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) o[e] = (bf16)(acc[e] * inv);
  *reinterpret_cast<bf16x4*>(out + (size_t)rc * inner + 4 * i) = o;
}

}  // namespace

static int cross_attn(const void* q, const void* k_layer, const void* vt_layer, const uint8_t* mask, void* out, int B, int K,
                      int H, int S, const int32_t* users, const int32_t* rowpos, void* stream) {
  if (B < 1 || K < 1 || K > GRAM_MAX_BEAMS || H < 1 || S < 32 || (S & 31) || S > 4096) return GRAM_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  gram_prof::Scope prof(GRAM_K_CROSS_ATTN, st, 4.0 * B * H * S * 64);  // K + V^T, bf16
  switch ((K + 15) / 16) {
    case 1: return launch_cross<1>(q, k_layer, vt_layer, mask, out, B, K, H, S, users, rowpos, st);
    case 2: return launch_cross<2>(q, k_layer, vt_layer, mask, out, B, K, H, S, users, rowpos, st);
    case 3: return launch_cross<3>(q, k_layer, vt_layer, mask, out, B, K, H, S, users, rowpos, st);
    default: return launch_cross<4>(q, k_layer, vt_layer, mask, out, B, K, H, S, users, rowpos, st);
  }
}

extern "C" int gram_cross_attn_decode(const void* q, const void* k_layer, const void* vt_layer, const uint8_t* mask, void* out,
                                      int B, int K, int H, int S, void* stream) {
  return cross_attn(q, k_layer, vt_layer, mask, out, B, K, H, S, nullptr, nullptr, stream);
}

extern "C" int gram_cross_attn_decode_live(const void* q, const void* k_layer, const void* vt_layer, const uint8_t* mask, void* out,
                                           int n_users, const int32_t* users, const int32_t* rowpos, int K, int H, int S,
                                           void* stream) {
  if (!users || !rowpos) return GRAM_E_ARG;
  return cross_attn(q, k_layer, vt_layer, mask, out, n_users, K, H, S, users, rowpos, stream);
}

static int dec_self_attn(const void* qkv, void* kcache, void* vcache, const int32_t* anc, const float* bias, void* out, int R,
                         int n_rows, const int32_t* rows, int H, int t, int Tmax, void* stream) {
  if (R < 1 || n_rows < 1 || n_rows > R || H < 1 || H > 16 || t < 0 || t >= Tmax || Tmax > GRAM_MAX_DEC_LEN) return GRAM_E_ARG;
  gram_prof::Scope prof(GRAM_K_DEC_SELF_ATTN, (hipStream_t)stream, 4.0 * n_rows * H * 64 * (t + 1));
  hipLaunchKernelGGL(dec_self_attn_kernel, dim3(n_rows), dim3(H * 16), 0, (hipStream_t)stream, (const bf16*)qkv, (bf16*)kcache,
                     (bf16*)vcache, anc, bias, (bf16*)out, R, H, t, rows);
  GRAM_CHECK_LAUNCH();
  return 0;
}

extern "C" int gram_dec_self_attn(const void* qkv, void* kcache, void* vcache, const int32_t* anc, const float* bias, void* out,
                                  int R, int H, int t, int Tmax, void* stream) {
  return dec_self_attn(qkv, kcache, vcache, anc, bias, out, R, R, nullptr, H, t, Tmax, stream);
}

extern "C" int gram_dec_self_attn_live(const void* qkv, void* kcache, void* vcache, const int32_t* anc, const float* bias,
                                       void* out, int R, int n_rows, const int32_t* rows, int H, int t, int Tmax, void* stream) {
  if (!rows) return GRAM_E_ARG;
  return dec_self_attn(qkv, kcache, vcache, anc, bias, out, R, n_rows, rows, H, t, Tmax, stream);
}
