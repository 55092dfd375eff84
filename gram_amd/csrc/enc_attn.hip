// enc_attn.hip -- encoder self-attention for one (passage, head) per workgroup, L <= 128.
//
// Reference: T5Attention.forward, bidirectional self-attention branch,
// gram_t5_modeling.py:479-631 (scores UNSCALED :572, bucketed relative bias :452-477 shared from
// layer 0 :1246-1249, additive key mask (1-m)*finfo.min :1130-1132, fp32 softmax :608, @V :620).
//
// MI355X mapping: 4 waves x 32 queries.  K ([L][64]) and V^T ([64][L]) live in LDS, Q comes
// straight from HBM as the MFMA B operand.  The score tile is computed TRANSPOSED
// (S^T = K Q^T) so that a lane owns ONE query column: the softmax reduction is in-register
// plus two cross-lane steps, and the exponentiated tile is already in the B-operand layout of
// the second product O^T = V^T P^T -- P never touches LDS.  For that to hold, the 16 rows of
// S^T tile t of a 32-key step are the keys 8*(row>>2) + 4*t + (row&3) (a row permutation of
// the K fragment read, free of charge).
#include "common.h"
#include "prof.h"

// Ablation builds (`make EABL=n` -> libgram_hip_eabl<n>.so, tests/bench_enc_attn.py; results wrong; the product library has none of it).
// Bits: 1 no QK^T MFMAs, 2 no exp / piece split (P = a constant), 4 no PV MFMAs, 8 no global K/V loads, 16 no V^T scatter into LDS,
// 32 no output stores, 64 no query loads.
#ifndef GRAM_ENC_ABL
#define GRAM_ENC_ABL 0
#endif

namespace {

__device__ __forceinline__ int kswz(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }
__device__ __forceinline__ int vtswz(int d, int key) { return d * 256 + ((((key >> 3) ^ (d & 15))) << 4) + (key & 7) * 2; }

template <int NKS, int S>  // L = 32 * NKS; S = 16-bit pieces per value (1: plain; 2: gram_split_t -- q|k|v planar, the output interleaved)
__global__ __launch_bounds__(256) void enc_attn_kernel(const p16* __restrict__ qkv, const float* __restrict__ bias,
                                                       const uint8_t* __restrict__ mask, p16* __restrict__ out, int H,
                                                       long qkv_pstride) {
  constexpr int L = 32 * NKS;
  using T = SplitTab<S>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* ks = smem;                       // [S][128 * 128]
  char* vts = smem + S * 128 * 128;      // [S][64 * 256]
  // relative bias of head h by (key - query + 127), as FOUR copies shifted by 0..3 floats: a lane's four consecutive keys of a score
  // fragment are then one ALIGNED 16-byte read from the copy matching its query's residue, and the fragment's MFMA chain starts from it
  // (the bias is the accumulator's initial value: no add, 16 ds_read_b128 per lane instead of 64 ds_read_b32)
  float* bias_s = reinterpret_cast<float*>(smem + S * 2 * 128 * 128);  // [4][256]: bias_s[r][i] = bias[i + r]
  float* mask_s = bias_s + 4 * 256;                                    // [128]

  const int h = blockIdx.x, p = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int inner = H * 64;
  const size_t rs = (size_t)3 * inner;  // qkv row stride (elements)
  const p16* base = qkv + (size_t)p * L * rs + h * 64;

  // K and V rows -> LDS: all 2*NKS 16-byte loads of a thread go out first (written as load; store per iteration,
  // hipcc waits for each pair before issuing the next: NKS exposed HBM round trips per workgroup)
#pragma unroll
  for (int pc = 0; pc < S; ++pc) {
    p16x8 kv[NKS], vv[NKS];
#pragma unroll
    for (int it = 0; it < NKS; ++it) {
      const int i = tid + it * 256, row = i >> 3, c = i & 7;
      const p16* src = base + pc * qkv_pstride + (size_t)row * rs + c * 8;
      if constexpr ((GRAM_ENC_ABL & 8) != 0) {
        kv[it] = zero_bf16x8();
        vv[it] = zero_bf16x8();
        asm volatile("" : "+v"(kv[it]), "+v"(vv[it]));
      } else {
        kv[it] = ld_stream_b128(src + inner);
        vv[it] = ld_stream_b128(src + 2 * inner);
      }
    }
#pragma unroll
    for (int it = 0; it < NKS; ++it) {
      const int i = tid + it * 256, row = i >> 3, c = i & 7;
      *reinterpret_cast<p16x8*>(ks + pc * 128 * 128 + kswz(row, c)) = kv[it];
      if constexpr ((GRAM_ENC_ABL & 16) != 0) {
        *reinterpret_cast<p16x8*>(vts + pc * 64 * 256 + kswz(row, c)) = vv[it];  // (same bytes, one 16-B store, wrong layout)
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) *reinterpret_cast<p16*>(vts + pc * 64 * 256 + vtswz(c * 8 + e, row)) = vv[it][e];
      }
    }
  }
  if (tid < 255) {
    const float bv = bias[h * 255 + tid];
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (tid >= r) bias_s[r * 256 + tid - r] = bv;
  }
  bool key_masked = false;
  if (tid < L) {
    key_masked = mask[(size_t)p * L + tid] == 0;
    mask_s[tid] = key_masked ? 1.f : 0.f;
  }
  const bool any_masked = __syncthreads_or(key_masked) != 0;  // (an all-valid passage skips the mask pass: wave-uniform)

  const int q0 = wave * 32;
  if (q0 >= L) return;
  const int c = lane & 15, g = lane >> 4;

  p16x8 qf[S][2][2];
#pragma unroll
  for (int pc = 0; pc < S; ++pc)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int kd = 0; kd < 2; ++kd)
        if constexpr ((GRAM_ENC_ABL & 64) != 0) {
          qf[pc][nt][kd] = zero_bf16x8();
          asm volatile("" : "+v"(qf[pc][nt][kd]));
        } else {
          qf[pc][nt][kd] = ld_stream_b128(base + pc * qkv_pstride + (size_t)(q0 + 16 * nt + c) * rs + 32 * kd + 8 * g);
        }

  f32x4 s[NKS][2][2];
#pragma unroll
  for (int k2 = 0; k2 < NKS; ++k2)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int key = 32 * k2 + 8 * (c >> 2) + 4 * t + (c & 3);  // row permutation (see header)
      p16x8 kf[S][2];
#pragma unroll
      for (int pc = 0; pc < S; ++pc) {
        kf[pc][0] = *reinterpret_cast<const p16x8*>(ks + pc * 128 * 128 + kswz(key, g));
        kf[pc][1] = *reinterpret_cast<const p16x8*>(ks + pc * 128 * 128 + kswz(key, 4 + g));
      }
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        // S^T rows of this lane: keys 32 k2 + 8 g + 4 t + 0..3, query q0 + 16 nt + c -> bias index key - query + 127
        const int idx0 = 32 * k2 + 8 * g + 4 * t + 127 - (q0 + 16 * nt + c), r = idx0 & 3;
        f32x4 a = *reinterpret_cast<const f32x4*>(bias_s + r * 256 + (idx0 - r));
        if constexpr ((GRAM_ENC_ABL & 1) != 0) {
          a[0] += (float)kf[0][0][0] + (float)qf[0][nt][0][0];  // (keep the operands live)
        } else {
#pragma unroll
          for (int pr = 0; pr < T::NP; ++pr) {
            a = mfma16(kf[T::A[pr]][0], qf[T::B[pr]][nt][0], a);
            a = mfma16(kf[T::A[pr]][1], qf[T::B[pr]][nt][1], a);
          }
        }
        s[k2][t][nt] = a;
      }
    }

  // Every wave has read its K fragments: the K tiles' LDS is free from here on -- each wave stages its 32 output rows there at the end,
  // so that the rows leave as whole 128 * S-byte segments (below).  (Waves past L left before the first barrier-counted instruction of
  // this kind; s_barrier counts the waves that are still alive.)
  gram_sync();

  // bias + mask, row maximum over keys for this lane's query column(s)
  float mxq[2], l[2] = {0.f, 0.f};
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    float mx = GRAM_FMIN;
#pragma unroll
    for (int k2 = 0; k2 < NKS; ++k2)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float v = s[k2][t][nt][j];
          if (any_masked) {
            v = (mask_s[32 * k2 + 8 * g + 4 * t + j] != 0.f) ? GRAM_FMIN : v;
            s[k2][t][nt][j] = v;
          }
          mx = fmaxf(mx, v);
        }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    mxq[nt] = mx;
  }

  // O^T = V^T P^T, P = exp(S^T - max) formed 32 keys at a time (as 16-bit pieces) right before its MFMAs
  // (exp(s - m) with the subtraction first: a fully masked row is all finfo.min, and min - min = 0 must give the reference's uniform row)
  f32x4 o[4][2];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) o[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k2 = 0; k2 < NKS; ++k2) {
    p16x8 pf[S][2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      if constexpr ((GRAM_ENC_ABL & 2) != 0) {
        float t = 0.f;
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
          for (int j = 0; j < 4; ++j) t += s[k2][tt][nt][j];
        l[nt] += t;
        const uint32_t wv = __float_as_uint(t) & 0x3c003c00u;
        pf[0][nt] = __builtin_bit_cast(p16x8, make_uint4(wv, wv, wv, wv));
        if constexpr (S == 2) pf[S - 1][nt] = pf[0][nt];
      } else if constexpr (S == 2) {  // both pieces of a pair of probabilities at once (split2_pair: 4 vector instructions per pair)
        uint32_t w[2][4];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int j = 0; j < 4; j += 2) {
            const float e0 = __expf(s[k2][t][nt][j] - mxq[nt]), e1 = __expf(s[k2][t][nt][j + 1] - mxq[nt]);
            l[nt] += e0;
            l[nt] += e1;
            split2_pair(e0, e1, w[0][2 * t + (j >> 1)], w[1][2 * t + (j >> 1)]);
          }
        pf[0][nt] = __builtin_bit_cast(p16x8, make_uint4(w[0][0], w[0][1], w[0][2], w[0][3]));
        pf[1][nt] = __builtin_bit_cast(p16x8, make_uint4(w[1][0], w[1][1], w[1][2], w[1][3]));
      } else {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float e = __expf(s[k2][t][nt][j] - mxq[nt]);
            l[nt] += e;
            pf[0][nt][4 * t + j] = (p16)e;
          }
      }
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const int d = 16 * mt + c;
      p16x8 vf[S];
#pragma unroll
      for (int pc = 0; pc < S; ++pc) vf[pc] = *reinterpret_cast<const p16x8*>(vts + pc * 64 * 256 + vtswz(d, 32 * k2 + 8 * g));
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int pr = 0; pr < T::NP; ++pr) {
          if constexpr ((GRAM_ENC_ABL & 4) != 0) o[mt][nt][0] += (float)vf[T::A[pr]][0] + (float)pf[T::B[pr]][nt][0];
          else o[mt][nt] = mfma16(vf[T::A[pr]], pf[T::B[pr]][nt], o[mt][nt]);
        }
    }
  }

  // Output: a lane holds 4 consecutive dims of ONE query per (nt, mt) -- stored from the registers that is 8 B per lane and piece in
  // 32-byte pieces of 16 different rows per instruction, which costs the memory system twice what the same bytes cost as whole lines
  // (profiles/r03r_enc_attn_ablation.txt: 4.7 ms with the stores, 3.5 without).  The wave's 32 rows x (128 * S) bytes -- one head's
  // slice of the O GEMM's A operand, contiguous per row -- go through its 8 * S KB of the (dead) K tiles instead and leave as 16 B per
  // lane, four (S = 2) or eight whole row segments per instruction.  Chunks are XOR-swizzled by the row: conflict-free both ways.
  constexpr int RB = 128 * S, CPR = RB / 16;  // bytes and 16-B chunks per staged row
  char* stage = ks + wave * (32 * RB);
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    float lt = l[nt];
    lt += __shfl_xor(lt, 16, 64);
    lt += __shfl_xor(lt, 32, 64);
    const float inv_l = 1.f / lt;
    const int ql = 16 * nt + c;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const f32x4 v = o[mt][nt] * inv_l;
      if constexpr (S == 2) {
        // the head's 256 B of an interleaved row: [dims 0..31: piece 0 | piece 1][dims 32..63: piece 0 | piece 1]
        uint2 hi, lo;
        split2x4(v, hi, lo);
        const int ch = (mt >> 1) * 8 + (mt & 1) * 2 + (g >> 1);  // 16-B chunk of piece 0; piece 1 sits 4 chunks on
        *reinterpret_cast<uint2*>(stage + ql * RB + (((ch) ^ (ql & 15)) << 4) + (g & 1) * 8) = hi;
        *reinterpret_cast<uint2*>(stage + ql * RB + (((ch + 4) ^ (ql & 15)) << 4) + (g & 1) * 8) = lo;
      } else {
        p16x4 r;
#pragma unroll
        for (int j = 0; j < 4; ++j) r[j] = (p16)v[j];
        const int ch = 2 * mt + (g >> 1);
        *reinterpret_cast<p16x4*>(stage + ql * RB + ((ch ^ (ql & 7)) << 4) + (g & 1) * 8) = r;
      }
    }
  }
  __builtin_amdgcn_wave_barrier();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  char* obase = reinterpret_cast<char*>(out) + (((size_t)p * L + q0) * inner * S + (size_t)h * 64 * S) * sizeof(p16);
#pragma unroll
  for (int it = 0; it < RB / 32; ++it) {
    const int row = it * (64 / CPR) + lane / CPR, chunk = lane % CPR;
    const uint4 val = *reinterpret_cast<const uint4*>(stage + row * RB + ((chunk ^ (row & (CPR - 1))) << 4));
    if (!(GRAM_ENC_ABL & 32) || val.x == 0x12345678u)
      *reinterpret_cast<uint4*>(obase + (size_t)row * inner * S * sizeof(p16) + chunk * 16) = val;
  }
}

template <int S>
int launch_enc(const void* qkv, const float* bias, const uint8_t* mask, void* out, int P, int L, int H, long qkv_pstride,
               hipStream_t st) {
  const dim3 grid(H, P), block(256);
  constexpr int smem = S * 2 * 128 * 128 + (4 * 256 + 128) * 4;
  static bool attr_set = false;
  if (!attr_set && smem > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(enc_attn_kernel<1, S>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(enc_attn_kernel<2, S>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(enc_attn_kernel<3, S>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(enc_attn_kernel<4, S>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  switch (L / 32) {
    case 1: hipLaunchKernelGGL((enc_attn_kernel<1, S>), grid, block, smem, st, (const p16*)qkv, bias, mask, (p16*)out, H, qkv_pstride); break;
    case 2: hipLaunchKernelGGL((enc_attn_kernel<2, S>), grid, block, smem, st, (const p16*)qkv, bias, mask, (p16*)out, H, qkv_pstride); break;
    case 3: hipLaunchKernelGGL((enc_attn_kernel<3, S>), grid, block, smem, st, (const p16*)qkv, bias, mask, (p16*)out, H, qkv_pstride); break;
    default: hipLaunchKernelGGL((enc_attn_kernel<4, S>), grid, block, smem, st, (const p16*)qkv, bias, mask, (p16*)out, H, qkv_pstride); break;
  }
  GRAM_CHECK_LAUNCH();
  return 0;
}

}  // namespace

extern "C" int gram_enc_self_attn(const void* qkv, const float* bias, const uint8_t* mask, void* out, int P, int L, int H,
                                  void* stream) {
  return gram_enc_self_attn_split(qkv, bias, mask, out, P, L, H, 1, 0, stream);
}

extern "C" int gram_enc_self_attn_split(const void* qkv, const float* bias, const uint8_t* mask, void* out, int P, int L, int H,
                                        int pieces, int64_t qkv_pstride, void* stream) {
  if (P < 1 || H < 1 || L < 32 || L > GRAM_MAX_PASSAGE_LEN || (L & 31) || pieces < 1 || pieces > GRAM_MAX_PIECES) return GRAM_E_ARG;
  if (pieces > 1 && qkv_pstride < (int64_t)P * L * 3 * H * 64) return GRAM_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  gram_prof::Scope prof(GRAM_K_ENC_ATTN, st, 4.0 * P * H * L * L * 64 * (pieces == 2 ? 3 : 1));
  if (pieces == 2) return launch_enc<2>(qkv, bias, mask, out, P, L, H, qkv_pstride, st);
  return launch_enc<1>(qkv, bias, mask, out, P, L, H, qkv_pstride, st);
}
