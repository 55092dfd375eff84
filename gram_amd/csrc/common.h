// common.h -- shared device helpers for libgram_hip (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/gram_hip.h"

// The 16-bit operand type of every MFMA on the path: IEEE half.  11 significant bits per piece instead of bfloat16's 8, so the
// two-piece mode reaches ~2^-22 instead of ~2^-18 at the same MFMA rate (v_mfma_f32_16x16x32_f16 runs at the bf16 rate and keeps
// f16 subnormal inputs: profiles/r03_mfma_f16_denorm_probe.json) -- measured: no rank flip against the fp32 reference on either
// test population, where two bf16 pieces flip (profiles/r03a_precision_*).  The narrower exponent range is handled by power-of-two
// weight scales (gram_split_t.out_scale).  `make PIECE=bf16` (GRAM_PIECE_BF16) builds the same sources on bfloat16 for A/B runs.
// The type is spelled `p16` ("piece, 16 bits") in the sources; the C ABI's entry points carry `_bf16` in their names (they predate the
// switch) AND `_f16` aliases (gram_hip.h) that refuse to run in a library built on the other type.
#ifdef GRAM_PIECE_BF16
typedef __bf16 p16;
#define GRAM_PIECE_FORMAT 0
#else
typedef _Float16 p16;
#define GRAM_PIECE_FORMAT 1
#define GRAM_F16 1
#endif
typedef p16 p16x8 __attribute__((ext_vector_type(8)));
typedef p16 p16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// largest M the streaming small-M GEMM ever takes (gemm.hip clamps GRAM_GEMM_STREAM_MAXM to it; generate.hip sizes the 16-column
// sum-of-squares partial buffers for it)
constexpr int GRAM_STREAM_MAX_M_LIMIT = 4096;

#define GRAM_FMIN (-3.4028234663852886e38f)  // torch.finfo(float32).min, gram_t5_modeling.py:1130-1132
#define WAVE 64

// ---- workgroup barriers.  Every kernel calls gram_sync() (pp_barrier() in the ping-pong GEMM) instead of __syncthreads(): in the
// product build it IS __syncthreads().  `make CHAOS=1` (libgram_hip_chaos.so, a diagnostic build that is never shipped) lets a wave
// sleep up to ~3.5 us behind a barrier now and then, so that the waves of a workgroup run through the region behind it far apart --
// what a protocol with a barrier missing survives only by timing.  Round 4's prologue race of the ping-pong GEMM is
// the case in point: three rounds of tests never saw it, this build shows it in the first seconds
// (tools/r04_gpu_calls/r04r_chaos.sh: the kernel parity tests against the chaos library).
#ifdef GRAM_CHAOS
__device__ __forceinline__ void gram_chaos_point() {
  const unsigned t = (unsigned)__builtin_amdgcn_s_memtime();
  const unsigned w = (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) + blockIdx.x * 7u;
  const unsigned h = (t ^ (w * 0x9E3779B1u)) * 0x85EBCA6Bu;
  if ((h >> 29) == 0u) {  // one barrier in eight, per wave
    const int n = (int)((h >> 20) & 15u);
    for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(8);  // 512 cycles each
  }
}
#else
__device__ __forceinline__ void gram_chaos_point() {}
#endif
__device__ __forceinline__ void gram_sync() {
  __syncthreads();
  gram_chaos_point();
}

#define GRAM_CHECK_LAUNCH()                       \
  do {                                            \
    hipError_t e__ = hipGetLastError();           \
    if (e__ != hipSuccess) return (int)e__;       \
  } while (0)

// D[16x16] += A[16x32] * B[32x16]; lane l supplies A[row l&15][k 8(l>>4)..+7], B[k 8(l>>4)..+7][col l&15];
// lane l receives D[row 4(l>>4)+j][col l&15] in element j.
__device__ __forceinline__ f32x4 mfma16(p16x8 a, p16x8 b, f32x4 c) {
#ifdef GRAM_F16
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
#else
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
#endif
}

__device__ __forceinline__ p16x8 ld_global_b128(const p16* p) {
  return *reinterpret_cast<const p16x8*>(p);
}

// streaming (nontemporal) loads for data that is read once per sweep (nothing to keep in the L2 / MALL for)
#ifndef GRAM_ATTN_NT
#define GRAM_ATTN_NT 0  // A/B build hook: 1 = the self-attention kernels read q/k/v and the cache with the nt hint (measured slower: encoder 56.2 -> 58.0 ms, decoder 42.4 -> 47.4 ms per step -- the K beams of a user share cache rows through the L2)
#endif
__device__ __forceinline__ p16x8 ld_stream_b128(const p16* p) {
  if constexpr (GRAM_ATTN_NT != 0) return __builtin_nontemporal_load(reinterpret_cast<const p16x8*>(p));
  else return *reinterpret_cast<const p16x8*>(p);
}
__device__ __forceinline__ p16x4 ld_stream_b64(const p16* p) {
  if constexpr (GRAM_ATTN_NT != 0) return __builtin_nontemporal_load(reinterpret_cast<const p16x4*>(p));
  else return *reinterpret_cast<const p16x4*>(p);
}

__device__ __forceinline__ p16x8 zero_bf16x8() {
  p16x8 z;
#pragma unroll
  for (int i = 0; i < 8; ++i) z[i] = (p16)0.0f;
  return z;
}

// v -> its two 16-bit pieces hi = r16(v), lo = r16(v - hi), two values per call, each piece packed into one dword.  IEEE-half build:
// v_cvt_pk_f16_f32 for the pairs and v_fma_mix_f32 (f16 source x -1 + f32) for the remainders: 4 vector instructions per pair where the
// plain C++ (convert, convert back, subtract, convert, pack) compiles to 6-7 -- the tile-end epilogues of the two-piece GEMMs and the
// encoder attention's softmax are bound by their vector-instruction count.  v - hi is exact in fp32 either way (hi is within half a
// 16-bit ulp of v), so the pieces are the same bits as the plain form's.
__device__ __forceinline__ void split2_pair(float a, float b, uint32_t& hi, uint32_t& lo) {
#ifdef GRAM_F16
  typedef float f32x2_ __attribute__((ext_vector_type(2)));
  typedef _Float16 h16x2_ __attribute__((ext_vector_type(2)));
  const f32x2_ ab = {a, b};
  hi = __builtin_bit_cast(uint32_t, __builtin_convertvector(ab, h16x2_));
  float ra, rb;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(ra) : "v"(hi), "v"(a));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(rb) : "v"(hi), "v"(b));
  const f32x2_ r = {ra, rb};
  lo = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, h16x2_));
#else
  typedef p16 b16x2_ __attribute__((ext_vector_type(2)));
  b16x2_ h = {(p16)a, (p16)b};
  b16x2_ l = {(p16)(a - (float)h[0]), (p16)(b - (float)h[1])};
  hi = __builtin_bit_cast(uint32_t, h);
  lo = __builtin_bit_cast(uint32_t, l);
#endif
}
__device__ __forceinline__ void split2x4(f32x4 v, uint2& hi, uint2& lo) {
  split2_pair(v[0], v[1], hi.x, lo.x);
  split2_pair(v[2], v[3], hi.y, lo.y);
}

// Sum of squares of four values (a partial of the folded T5LayerNorm) in ONE spelled-out arithmetic: every GEMM kernel that produces
// these partials must round the same way (a user's result does not depend on which kernel its batch size selects), and hipcc decides
// per call site whether a * a + b * b becomes an fma -- a decision that changed when unrelated code next to it did.
__device__ __forceinline__ float sumsq4(f32x4 v) {
  const float a = __builtin_fmaf(v[1], v[1], v[0] * v[0]);
  const float b = __builtin_fmaf(v[3], v[3], v[2] * v[2]);
  return a + b;
}

// 1/rms of a row from its sum of squares, in ONE spelled-out arithmetic (an explicit fma, then v_rsq_f32): the GEMM kernels that add
// the partials up themselves and gram_row_rscale must return the same bits (batch invariance), and hipcc decides per call site
// whether s * inv_d + eps contracts.
__device__ __forceinline__ float row_rs(float s, float inv_d, float eps) { return rsqrtf(__builtin_fmaf(s, inv_d, eps)); }

// Power-of-two factor of a row's 16-bit copy (gram_norm_fusion_t.xs_out) from the row's sum of squares and the SMALLEST of its
// 64-column partial sums.  Two IEEE-half pieces hold 22 bits of an element whose scaled magnitude is in [2^-3, 65 504]; below that
// the low piece goes subnormal (absolute error 2^-25), above it the high piece overflows.  T5 rows carry a few outlier features
// 10^2 .. 10^4 times the ordinary ones, which dominate the row's rms -- so the factor is set by the ordinary magnitude, estimated
// from the quietest 64-column block (typ^2 = min block sum / 64): typ * xs lands in [2^-2, 2^-1), the error floor stays 2^-23 of
// an ordinary element, and an element may be 1.3e5 x typ before it overflows.  A second bound keeps sqrt(sum) * xs <= 2^10 (every
// element <= 2^10: a row whose blocks all hold outliers, or whose quietest block is nearly empty).  Clamped to [2^-40, 2^20].
// Spelled once: the embedding, the GEMM kernels and gram_row_rscale_xs must agree bit for bit (batch invariance).
__device__ __forceinline__ float row_xscale(float sum, float minblk) {
  const float want = 0.25f * rsqrtf(minblk * (1.0f / 64.0f));  // (+inf for an all-zero block; powers of two: exact products)
  const float cap = 1024.0f * rsqrtf(sum);
  const float f = fminf(want, cap);
  int e = (int)((__float_as_uint(f) >> 23) & 0xffu) - 127;  // floor(log2 f); inf -> 128, 0 / subnormal -> -127: both clamped
  e = e < -40 ? -40 : (e > 20 ? 20 : e);
  return __uint_as_float((uint32_t)(e + 127) << 23);
}

// products of the two-piece mode (gram_hip.h, gram_split_t): (A piece, B piece) pairs, smallest first
template <int S> struct SplitTab;
template <> struct SplitTab<1> { static constexpr int NP = 1; static constexpr int A[3] = {0, 0, 0}; static constexpr int B[3] = {0, 0, 0}; };
template <> struct SplitTab<2> { static constexpr int NP = 3; static constexpr int A[3] = {0, 1, 0}; static constexpr int B[3] = {1, 0, 0}; };

// element offset of (column n, piece pc) inside a row of an INTERLEAVED two-piece matrix [rows][cols / 32][2][32] (gram_hip.h): the
// layout of every 16-bit operand a GEMM reads in the two-piece mode
__device__ __forceinline__ int inter_off(int n, int pc) { return ((n >> 5) << 6) + pc * 32 + (n & 31); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
