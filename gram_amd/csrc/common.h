// common.h -- shared device helpers for libgram_hip (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/gram_hip.h"

typedef __bf16 bf16;
typedef bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define GRAM_FMIN (-3.4028234663852886e38f)  // torch.finfo(float32).min, gram_t5_modeling.py:1130-1132
#define WAVE 64

#define GRAM_CHECK_LAUNCH()                       \
  do {                                            \
    hipError_t e__ = hipGetLastError();           \
    if (e__ != hipSuccess) return (int)e__;       \
  } while (0)

// D[16x16] += A[16x32] * B[32x16]; lane l supplies A[row l&15][k 8(l>>4)..+7], B[k 8(l>>4)..+7][col l&15];
// lane l receives D[row 4(l>>4)+j][col l&15] in element j.
__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ bf16x8 ld_global_b128(const bf16* p) {
  return *reinterpret_cast<const bf16x8*>(p);
}

// streaming (nontemporal) loads for data that is read once per sweep (nothing to keep in the L2 / MALL for)
#ifndef GRAM_ATTN_NT
#define GRAM_ATTN_NT 0  // A/B build hook: 1 = the self-attention kernels read q/k/v and the cache with the nt hint (measured slower: encoder 56.2 -> 58.0 ms, decoder 42.4 -> 47.4 ms per step -- the K beams of a user share cache rows through the L2)
#endif
__device__ __forceinline__ bf16x8 ld_stream_b128(const bf16* p) {
  if constexpr (GRAM_ATTN_NT != 0) return __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(p));
  else return *reinterpret_cast<const bf16x8*>(p);
}
__device__ __forceinline__ bf16x4 ld_stream_b64(const bf16* p) {
  if constexpr (GRAM_ATTN_NT != 0) return __builtin_nontemporal_load(reinterpret_cast<const bf16x4*>(p));
  else return *reinterpret_cast<const bf16x4*>(p);
}

__device__ __forceinline__ bf16x8 zero_bf16x8() {
  bf16x8 z;
#pragma unroll
  for (int i = 0; i < 8; ++i) z[i] = (bf16)0.0f;
  return z;
}

// products of the split-bf16 modes (gram_hip.h: GRAM_SPLIT_A_PIECE / GRAM_SPLIT_W_PIECE), smallest first
template <int S> struct SplitTab;
template <> struct SplitTab<1> { static constexpr int NP = 1; static constexpr int A[6] = {0, 0, 0, 0, 0, 0}; static constexpr int B[6] = {0, 0, 0, 0, 0, 0}; };
template <> struct SplitTab<2> { static constexpr int NP = 3; static constexpr int A[6] = {0, 1, 0, 0, 0, 0}; static constexpr int B[6] = {1, 0, 0, 0, 0, 0}; };
template <> struct SplitTab<3> { static constexpr int NP = 6; static constexpr int A[6] = {0, 2, 1, 1, 0, 0}; static constexpr int B[6] = {2, 0, 1, 0, 1, 0}; };

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
