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

namespace {

__device__ __forceinline__ int kswz(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }
__device__ __forceinline__ int vtswz(int d, int key) { return d * 256 + ((((key >> 3) ^ (d & 15))) << 4) + (key & 7) * 2; }

template <int NKS>  // L = 32 * NKS
__global__ __launch_bounds__(256) void enc_attn_kernel(const bf16* __restrict__ qkv, const float* __restrict__ bias,
                                                       const uint8_t* __restrict__ mask, bf16* __restrict__ out, int H) {
  constexpr int L = 32 * NKS;
  __shared__ __attribute__((aligned(16))) char ks[128 * 128];
  __shared__ __attribute__((aligned(16))) char vts[64 * 256];
  __shared__ float bias_s[256];
  __shared__ float mask_s[128];

  const int h = blockIdx.x, p = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int inner = H * 64;
  const size_t rs = (size_t)3 * inner;  // qkv row stride (elements)
  const bf16* base = qkv + (size_t)p * L * rs + h * 64;

  // K and V rows -> LDS: all 2*NKS 16-byte loads of a thread go out first (written as load; store per iteration,
  // hipcc waits for each pair before issuing the next: NKS exposed HBM round trips per workgroup)
  {
    bf16x8 kv[NKS], vv[NKS];
#pragma unroll
    for (int it = 0; it < NKS; ++it) {
      const int i = tid + it * 256, row = i >> 3, c = i & 7;
      const bf16* src = base + (size_t)row * rs + c * 8;
      kv[it] = ld_global_b128(src + inner);
      vv[it] = ld_global_b128(src + 2 * inner);
    }
#pragma unroll
    for (int it = 0; it < NKS; ++it) {
      const int i = tid + it * 256, row = i >> 3, c = i & 7;
      *reinterpret_cast<bf16x8*>(ks + kswz(row, c)) = kv[it];
#pragma unroll
      for (int e = 0; e < 8; ++e) *reinterpret_cast<bf16*>(vts + vtswz(c * 8 + e, row)) = vv[it][e];
    }
  }
  if (tid < 255) bias_s[tid] = bias[h * 255 + tid];
  if (tid < L) mask_s[tid] = mask[(size_t)p * L + tid] ? 0.f : 1.f;
  __syncthreads();

  const int q0 = wave * 32;
  if (q0 >= L) return;
  const int c = lane & 15, g = lane >> 4;

  bf16x8 qf[2][2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int kd = 0; kd < 2; ++kd) qf[nt][kd] = ld_global_b128(base + (size_t)(q0 + 16 * nt + c) * rs + 32 * kd + 8 * g);

  f32x4 s[NKS][2][2];
#pragma unroll
  for (int k2 = 0; k2 < NKS; ++k2)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int key = 32 * k2 + 8 * (c >> 2) + 4 * t + (c & 3);  // row permutation (see header)
      bf16x8 kf0 = *reinterpret_cast<const bf16x8*>(ks + kswz(key, g));
      bf16x8 kf1 = *reinterpret_cast<const bf16x8*>(ks + kswz(key, 4 + g));
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        f32x4 a = (f32x4){0.f, 0.f, 0.f, 0.f};
        a = mfma16(kf0, qf[nt][0], a);
        a = mfma16(kf1, qf[nt][1], a);
        s[k2][t][nt] = a;
      }
    }

  // bias + mask, softmax over keys for this lane's query column(s)
  float inv_l[2];
  bf16x8 pf[NKS][2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int query = q0 + 16 * nt + c;
    float mx = GRAM_FMIN;
#pragma unroll
    for (int k2 = 0; k2 < NKS; ++k2)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int key = 32 * k2 + 8 * g + 4 * t + j;
          float v = s[k2][t][nt][j] + bias_s[key - query + 127];
          v = (mask_s[key] != 0.f) ? GRAM_FMIN : v;
          s[k2][t][nt][j] = v;
          mx = fmaxf(mx, v);
        }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float l = 0.f;
#pragma unroll
    for (int k2 = 0; k2 < NKS; ++k2) {
      bf16x8 f;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float e = __expf(s[k2][t][nt][j] - mx);
          l += e;
          f[4 * t + j] = (bf16)e;
        }
      pf[k2][nt] = f;
    }
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    inv_l[nt] = 1.f / l;
  }

  // O^T = V^T P^T
  f32x4 o[4][2];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) o[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k2 = 0; k2 < NKS; ++k2)
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const int d = 16 * mt + c;
      bf16x8 vf = *reinterpret_cast<const bf16x8*>(vts + vtswz(d, 32 * k2 + 8 * g));
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) o[mt][nt] = mfma16(vf, pf[k2][nt], o[mt][nt]);
    }

#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int query = q0 + 16 * nt + c;
    bf16* orow = out + ((size_t)p * L + query) * inner + h * 64;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      bf16x4 r;
#pragma unroll
      for (int j = 0; j < 4; ++j) r[j] = (bf16)(o[mt][nt][j] * inv_l[nt]);
      *reinterpret_cast<bf16x4*>(orow + 16 * mt + 4 * g) = r;
    }
  }
}

}  // namespace

extern "C" int gram_enc_self_attn(const void* qkv, const float* bias, const uint8_t* mask, void* out, int P, int L, int H,
                                  void* stream) {
  if (P < 1 || H < 1 || L < 32 || L > GRAM_MAX_PASSAGE_LEN || (L & 31)) return GRAM_E_ARG;
  dim3 grid(H, P), block(256);
  hipStream_t st = (hipStream_t)stream;
  gram_prof::Scope prof(GRAM_K_ENC_ATTN, st, 4.0 * P * H * L * L * 64);
  switch (L / 32) {
    case 1: hipLaunchKernelGGL(enc_attn_kernel<1>, grid, block, 0, st, (const bf16*)qkv, bias, mask, (bf16*)out, H); break;
    case 2: hipLaunchKernelGGL(enc_attn_kernel<2>, grid, block, 0, st, (const bf16*)qkv, bias, mask, (bf16*)out, H); break;
    case 3: hipLaunchKernelGGL(enc_attn_kernel<3>, grid, block, 0, st, (const bf16*)qkv, bias, mask, (bf16*)out, H); break;
    default: hipLaunchKernelGGL(enc_attn_kernel<4>, grid, block, 0, st, (const bf16*)qkv, bias, mask, (bf16*)out, H); break;
  }
  GRAM_CHECK_LAUNCH();
  return 0;
}
