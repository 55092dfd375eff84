// rowops.hip -- HBM-bound row kernels: embedding gather, T5 RMSNorm (+late-fusion position
// embedding / lm_head rescale fused), row log-sum-exp.  One wave per row, 16-byte accesses.
#include "common.h"
#include "prof.h"

namespace {

// 16-bit pieces of the 4 values at columns n .. n+3 of a row (gram_split_t): piece p = r16(v - p0 - .. - p_{p-1}); one piece: the
// plain row [d]; two pieces: the interleaved row [d / 32][2][32] (the layout a GEMM reads its A operand in)
__device__ __forceinline__ void store_pieces4(p16* row, int n, f32x4 v, int pieces) {
  for (int pc = 0; pc < pieces; ++pc) {
    p16x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      o[e] = (p16)v[e];
      v[e] -= (float)o[e];
    }
    *reinterpret_cast<p16x4*>(row + (pieces == 2 ? inter_off(n, pc) : n)) = o;
  }
}

template <typename IdT>
__global__ __launch_bounds__(256) void embed_kernel(const float* __restrict__ table, const IdT* __restrict__ ids,
                                                    float* __restrict__ x, int rows, int d) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const f32x4* src = reinterpret_cast<const f32x4*>(table + (size_t)ids[row] * d);
  f32x4* dst = reinterpret_cast<f32x4*>(x + (size_t)row * d);
  for (int i = lane; i < d / 4; i += 64) dst[i] = src[i];
}

// embed_tokens for the folded-norm path: x (f32) + xb = bf16(x) + the row's sum of squares in ss[row][0]
// embed_tokens for the folded-norm path: x (f32), its 16-bit copy xb and the row's 64-column partial sums of squares ss[row][d/64]
// (nblk == d / 64; fewer: the row's total in ss[row][0] and zeros, the layout of round 1 -- no row factor then).
// xs_out != NULL: the copy is x * xs, xs the power-of-two factor of the row itself (row_xscale), written to xs_out[row]
template <typename IdT>
__global__ __launch_bounds__(256) void embed_ex_kernel(const float* __restrict__ table, const IdT* __restrict__ ids,
                                                       float* __restrict__ x, p16* __restrict__ xb, float* __restrict__ ss,
                                                       float* __restrict__ xs_out, int nblk, int rows, int d, int pieces) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const f32x4* src = reinterpret_cast<const f32x4*>(table + (size_t)ids[row] * d);
  f32x4* dst = reinterpret_cast<f32x4*>(x + (size_t)row * d);
  p16* dstb = xb + (size_t)row * d * pieces;
  const bool blocks = nblk == d / 64;  // (d % 64 == 0 then: checked by the launcher)
  f32x4 v[4];  // d <= 1024
  float total = 0.f, minblk = INFINITY;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int i = lane + it * 64;  // f32x4 index: columns 4 i .. 4 i + 3, i.e. 64-column block i / 16 = 4 it + lane / 16
    float s = 0.f;
    if (i < d / 4) {
      v[it] = src[i];
      dst[i] = v[it];
      s = (v[it][0] * v[it][0] + v[it][1] * v[it][1]) + (v[it][2] * v[it][2] + v[it][3] * v[it][3]);
    }
    if (blocks) {
      s += __shfl_xor(s, 1, 64);
      s += __shfl_xor(s, 2, 64);
      s += __shfl_xor(s, 4, 64);
      s += __shfl_xor(s, 8, 64);  // the 16 lanes of a block all hold its sum
      const int blk = 4 * it + (lane >> 4);
      if (blk < nblk) {
        if ((lane & 15) == 0) ss[(size_t)row * nblk + blk] = s;
        minblk = fminf(minblk, s);
      }
      // the row total in block order, pairs first -- the order the consumers add the partials up in (row_rscale)
      const float s1 = __shfl(s, 16, 64), s2 = __shfl(s, 32, 64), s3 = __shfl(s, 48, 64), s0 = __shfl(s, 0, 64);
      if (4 * it < nblk) total += s0 + s1;
      if (4 * it + 2 < nblk) total += s2 + s3;
    } else {
      total += s;
    }
  }
  if (blocks) {
    minblk = wave_min(minblk);
  } else {
    total = wave_sum(total);
    if (lane < nblk) ss[(size_t)row * nblk + lane] = lane == 0 ? total : 0.f;
  }
  const float xs = xs_out && blocks ? row_xscale(total, minblk) : 1.f;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int i = lane + it * 64;
    if (i < d / 4) store_pieces4(dstb, 4 * i, v[it] * xs, pieces);
  }
  if (xs_out && lane == 0) xs_out[row] = xs;
}

// T5LayerNorm (gram_t5_modeling.py:262-276): fp32 variance, no mean subtraction, no bias.
__global__ __launch_bounds__(256) void rmsnorm_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                      p16* __restrict__ out, int rows, int d, float eps, float scale,
                                                      const float* __restrict__ pos, int N, int L,
                                                      const int32_t* __restrict__ pmap, int pieces) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const f32x4* xr = reinterpret_cast<const f32x4*>(x + (size_t)row * d);
  const int nv = d / 4;
  f32x4 v[4];  // d <= 1024
  float ss = 0.f;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    int i = lane + it * 64;
    if (i < nv) {
      v[it] = xr[i];
      ss += v[it][0] * v[it][0] + v[it][1] * v[it][1] + v[it][2] * v[it][2] + v[it][3] * v[it][3];
    }
  }
  ss = wave_sum(ss);
  const float rs = rsqrtf(ss / (float)d + eps);
  const f32x4* wr = reinterpret_cast<const f32x4*>(w);
  // passage index of this row: row / L, or through the compaction map (flat index b*N + n)
  const int pn = pos ? ((pmap ? pmap[row / L] : row / L) % N) : 0;
  const f32x4* pr = pos ? reinterpret_cast<const f32x4*>(pos + (size_t)pn * d) : nullptr;
  p16* o = out + (size_t)row * d * pieces;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    int i = lane + it * 64;
    if (i < nv) {
      f32x4 g = wr[i];
      f32x4 p = pr ? pr[i] : (f32x4){0.f, 0.f, 0.f, 0.f};
      f32x4 r;
#pragma unroll
      for (int e = 0; e < 4; ++e) r[e] = g[e] * (v[it][e] * rs) * scale + p[e];
      store_pieces4(o, 4 * i, r, pieces);
    }
  }
}

// lse[r] = log sum_v exp(logits[r][v]); one block per row, online (max,sum) per thread.
__global__ __launch_bounds__(256) void row_lse_kernel(const float* __restrict__ logits, float* __restrict__ lse, int V) {
  __shared__ float sm[4], ss[4];
  const int row = blockIdx.x, tid = threadIdx.x;
  const f32x4* p = reinterpret_cast<const f32x4*>(logits + (size_t)row * V);
  float m = -INFINITY, s = 0.f;
  for (int i = tid; i < V / 4; i += 256) {
    f32x4 v = p[i];
    float vm = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
    if (vm > m) {
      s *= __expf(m - vm);
      m = vm;
    }
    s += __expf(v[0] - m) + __expf(v[1] - m) + __expf(v[2] - m) + __expf(v[3] - m);
  }
  float wm = wave_max(m);
  s *= (m == -INFINITY) ? 0.f : __expf(m - wm);
  s = wave_sum(s);
  if ((tid & 63) == 0) {
    sm[tid >> 6] = wm;
    ss[tid >> 6] = s;
  }
  gram_sync();
  if (tid == 0) {
    float M = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
    float S = 0.f;
    for (int i = 0; i < 4; ++i) S += (sm[i] == -INFINITY) ? 0.f : ss[i] * __expf(sm[i] - M);
    lse[row] = M + logf(S);
  }
}

// lse[row] from the lm_head epilogue's per-64-column partials (max, sum exp(x - max)): one wave per row
__global__ __launch_bounds__(256) void lse_combine_kernel(const float* __restrict__ part, float* __restrict__ lse, int M, int nblk) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int lane = threadIdx.x & 63;
  const float2* p = reinterpret_cast<const float2*>(part + (size_t)row * nblk * 2);
  float m = -INFINITY, s = 0.f;
  for (int i = lane; i < nblk; i += 64) {
    const float2 v = p[i];
    if (v.x > m) {
      s *= __expf(m - v.x);
      m = v.x;
    }
    s += v.y * __expf(v.x - m);
  }
  const float wm = wave_max(m);
  s *= (m == -INFINITY) ? 0.f : __expf(m - wm);
  s = wave_sum(s);
  if (lane == 0) lse[row] = wm + logf(s);
}

// 1/rms per row from the residual GEMM's partial sums of squares (fixed summation order)
__global__ __launch_bounds__(256) void row_rscale_kernel(const float* __restrict__ ss, float* __restrict__ rs,
                                                         const float* __restrict__ xs_in, float* __restrict__ xs_out, int M, int nblk,
                                                         float inv_d, float eps) {
  const int m = blockIdx.x * 256 + threadIdx.x;
  if (m >= M) return;
  const float2* p = reinterpret_cast<const float2*>(ss + (size_t)m * nblk);
  float s = 0.f, mn = INFINITY;
  for (int i = 0; i < nblk / 2; ++i) {
    const float2 v = p[i];
    s += v.x + v.y;
    mn = fminf(mn, fminf(v.x, v.y));
  }
  const float r = row_rs(s, inv_d, eps);
  rs[m] = xs_in ? r / xs_in[m] : r;  // (a power of two: exact)
  if (xs_out) xs_out[m] = row_xscale(s, mn);
}

// Calibration probe: stream `bytes` once through every CU (16-B loads, 4 in flight per lane) and fold them
// into a value that is never stored.  bench.py quotes its rate beside the cross-attention roofline as the
// read rate this box reaches on a plain sweep.
__global__ __launch_bounds__(256) void stream_read_kernel(const uint4* __restrict__ p, size_t n16, uint32_t* sink) {
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  uint4 a = {0, 0, 0, 0};
  for (; i + 3 * stride < n16; i += 4 * stride) {
    const uint4 v0 = p[i], v1 = p[i + stride], v2 = p[i + 2 * stride], v3 = p[i + 3 * stride];
    a.x ^= v0.x ^ v1.y ^ v2.z ^ v3.w;
    a.y ^= v0.y ^ v1.z ^ v2.w ^ v3.x;
    a.z ^= v0.z ^ v1.w ^ v2.x ^ v3.y;
    a.w ^= v0.w ^ v1.x ^ v2.y ^ v3.z;
  }
  for (; i < n16; i += stride) {
    const uint4 v = p[i];
    a.x ^= v.x;
    a.y ^= v.y;
    a.z ^= v.z;
    a.w ^= v.w;
  }
  if ((a.x ^ a.y ^ a.z ^ a.w) == 0x9e3779b9u && sink) *sink = a.x;  // keeps the loads alive; practically never taken
}

// Variants of the probe (tests/bench_stream.py): V = 1 each workgroup sweeps contiguous 16-KiB chunks, 2 nontemporal loads,
// 3 eight loads in flight per lane, 4 LDS-DMA (no registers), 5 = 1 with nontemporal loads
template <int V>
__global__ __launch_bounds__(256) void stream_read_var_kernel(const uint4* __restrict__ p, size_t n16, uint32_t* sink) {
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  const u32x4* q = reinterpret_cast<const u32x4*>(p);
  u32x4 a = {0, 0, 0, 0};
  if constexpr (V == 1 || V == 5) {
    const size_t nchunk = n16 / 1024;  // 16-KiB chunks
    for (size_t c = blockIdx.x; c < nchunk; c += gridDim.x) {
      const u32x4* b = q + c * 1024 + threadIdx.x;
      u32x4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = V == 5 ? __builtin_nontemporal_load(b + 256 * u) : b[256 * u];
#pragma unroll
      for (int u = 0; u < 4; ++u) a ^= v[u];
    }
  } else if constexpr (V == 2 || V == 3) {
    constexpr int U = V == 3 ? 8 : 4;
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n16; i += U * stride) {
      u32x4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = V == 2 ? __builtin_nontemporal_load(q + i + u * stride) : q[i + u * stride];
#pragma unroll
      for (int u = 0; u < U; ++u) a ^= v[u];
    }
  } else if constexpr (V == 4) {
    __shared__ __attribute__((aligned(16))) char buf[4][8][1024];  // per wave: 8 DMA slots
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t nchunk = n16 / 2048;  // 32-KiB chunks: 8 KiB per wave
    for (size_t c = blockIdx.x; c < nchunk; c += gridDim.x) {
      const char* b = reinterpret_cast<const char*>(q + c * 2048 + wave * 512);
#pragma unroll
      for (int u = 0; u < 8; ++u)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b + u * 1024 + lane * 16),
                                         (__attribute__((address_space(3))) void*)(&buf[wave][u][0]), 16, 0, 0);
    }
    gram_sync();
    a.x = *reinterpret_cast<const uint32_t*>(&buf[wave][0][lane * 4]);
  }
  if constexpr (V == 6 || V == 7) {  // LDS-DMA through inline asm, with the nt hint (6) / without (7); 16 KiB per wave in flight, counted waits
    __shared__ __attribute__((aligned(16))) char buf2[4][2][8][1024];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)&buf2[wave][0][0][0];
    const size_t nchunk = n16 / 2048;
    int par = 0;
    for (size_t c = blockIdx.x; c < nchunk; c += gridDim.x) {
      const char* b = reinterpret_cast<const char*>(q + c * 2048 + wave * 512);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if constexpr (V == 6)
          asm volatile("s_mov_b32 m0, %0\n\ts_nop 3\n\tglobal_load_lds_dwordx4 %1, %2 nt" ::"s"(lds0 + par * 8192 + u * 1024), "v"((uint32_t)(lane * 16)), "s"(b + u * 1024) : "memory");
        else
          asm volatile("s_mov_b32 m0, %0\n\ts_nop 3\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds0 + par * 8192 + u * 1024), "v"((uint32_t)(lane * 16)), "s"(b + u * 1024) : "memory");
      }
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // the previous chunk has landed
      par ^= 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    a.x = *reinterpret_cast<const uint32_t*>(&buf2[wave][0][0][lane * 4]);
  }
  if ((a.x ^ a.y ^ a.z ^ a.w) == 0x9e3779b9u && sink) *sink = a.x;
}

// passage cache -> residual stream: one 16-byte chunk per thread, consecutive threads on consecutive chunks of a row
__global__ __launch_bounds__(256) void gather_passage_x_kernel(const float* __restrict__ cache, const int32_t* __restrict__ slot,
                                                               float* __restrict__ x, int64_t nchunks, int L, int cache_L, int d4) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= nchunks) return;
  const int c = (int)(i % d4);
  const int64_t row = i / d4;
  const int l = (int)(row % L);
  const int64_t p = row / L;
  f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (l < cache_L) v = reinterpret_cast<const f32x4*>(cache)[((int64_t)slot[p] * cache_L + l) * d4 + c];
  reinterpret_cast<f32x4*>(x)[i] = v;
}
}  // namespace

extern "C" int gram_debug_stream_read(const void* src, size_t bytes, void* sink, void* stream) {
  if (!src || bytes < 16 || (reinterpret_cast<uintptr_t>(src) & 15)) return GRAM_E_ARG;
  // the fastest access shape found (tests/bench_stream.py, profiles/r02j_stream_read_variants.jsonl): contiguous 16-KiB chunks per
  // workgroup, nontemporal loads -- 7.0 TB/s where the grid-stride sweep of round 1 reads 5.7
  if (bytes >= (1u << 20))
    hipLaunchKernelGGL(stream_read_var_kernel<5>, dim3(256 * 16), dim3(256), 0, (hipStream_t)stream, (const uint4*)src, bytes / 16, (uint32_t*)sink);
  else
    hipLaunchKernelGGL(stream_read_kernel, dim3(256 * 8), dim3(256), 0, (hipStream_t)stream, (const uint4*)src, bytes / 16, (uint32_t*)sink);
  GRAM_CHECK_LAUNCH();
  return 0;
}

extern "C" int gram_debug_stream_read_variant(const void* src, size_t bytes, void* sink, int variant, int wgs, void* stream) {
  if (!src || bytes < (1 << 20) || (reinterpret_cast<uintptr_t>(src) & 15) || wgs < 1) return GRAM_E_ARG;
  const dim3 g(wgs), b(256);
  hipStream_t st = (hipStream_t)stream;
  switch (variant) {
    case 0: hipLaunchKernelGGL(stream_read_kernel, g, b, 0, st, (const uint4*)src, bytes / 16, (uint32_t*)sink); break;
    case 1: hipLaunchKernelGGL(stream_read_var_kernel<1>, g, b, 0, st, (const uint4*)src, bytes / 16, (uint32_t*)sink); break;
    case 2: hipLaunchKernelGGL(stream_read_var_kernel<2>, g, b, 0, st, (const uint4*)src, bytes / 16, (uint32_t*)sink); break;
    case 3: hipLaunchKernelGGL(stream_read_var_kernel<3>, g, b, 0, st, (const uint4*)src, bytes / 16, (uint32_t*)sink); break;
    case 4: hipLaunchKernelGGL(stream_read_var_kernel<4>, g, b, 0, st, (const uint4*)src, bytes / 16, (uint32_t*)sink); break;
    case 5: hipLaunchKernelGGL(stream_read_var_kernel<5>, g, b, 0, st, (const uint4*)src, bytes / 16, (uint32_t*)sink); break;
    case 6: hipLaunchKernelGGL(stream_read_var_kernel<6>, g, b, 0, st, (const uint4*)src, bytes / 16, (uint32_t*)sink); break;
    case 7: hipLaunchKernelGGL(stream_read_var_kernel<7>, g, b, 0, st, (const uint4*)src, bytes / 16, (uint32_t*)sink); break;
    default: return GRAM_E_ARG;
  }
  GRAM_CHECK_LAUNCH();
  return 0;
}

extern "C" int gram_row_rscale_xs(const float* ss, float* rs, const float* xs_in, float* xs_out, int M, int nblk, int d, float eps,
                                  void* stream) {
  if (M < 1 || nblk < 2 || (nblk & 1) || d < 64 || !ss || !rs || (xs_out && xs_out == xs_in)) return GRAM_E_ARG;
  gram_prof::Scope prof(GRAM_K_ROWOPS, (hipStream_t)stream, 4.0 * M * (nblk + 1 + (xs_in != nullptr) + (xs_out != nullptr)));
  hipLaunchKernelGGL(row_rscale_kernel, dim3((M + 255) / 256), dim3(256), 0, (hipStream_t)stream, ss, rs, xs_in, xs_out, M, nblk,
                     1.0f / (float)d, eps);
  GRAM_CHECK_LAUNCH();
  return 0;
}
extern "C" int gram_row_rscale(const float* ss, float* rs, int M, int nblk, int d, float eps, void* stream) {
  return gram_row_rscale_xs(ss, rs, nullptr, nullptr, M, nblk, d, eps, stream);
}
extern "C" int gram_lse_combine(const float* lse_part, float* lse, int M, int nblk, void* stream) {
  if (M < 1 || nblk < 1) return GRAM_E_ARG;
  gram_prof::Scope prof(GRAM_K_LSE, (hipStream_t)stream, 8.0 * M * nblk);
  hipLaunchKernelGGL(lse_combine_kernel, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, lse_part, lse, M, nblk);
  GRAM_CHECK_LAUNCH();
  return 0;
}

extern "C" int gram_embed_ex(const float* table, const void* ids, int ids_are_i64, float* x, void* xb, float* ss, int nblk,
                             int rows, int d, void* stream) {
  return gram_embed_ex_split(table, ids, ids_are_i64, x, xb, ss, nblk, rows, d, 1, stream);
}

extern "C" int gram_embed_ex_split(const float* table, const void* ids, int ids_are_i64, float* x, void* xb, float* ss, int nblk,
                                   int rows, int d, int pieces, void* stream) {
  return gram_embed_ex_xs(table, ids, ids_are_i64, x, xb, ss, nullptr, nblk, rows, d, pieces, stream);
}

extern "C" int gram_embed_ex_xs(const float* table, const void* ids, int ids_are_i64, float* x, void* xb, float* ss, float* xs_out,
                                int nblk, int rows, int d, int pieces, void* stream) {
  if (rows < 1 || (d & 3) || d > 1024 || !xb || !ss || nblk < 1 || nblk > 64 || pieces < 1 || pieces > GRAM_MAX_PIECES || (pieces > 1 && (d & 31)))
    return GRAM_E_ARG;
  if (xs_out && (d % 64 != 0 || nblk != d / 64)) return GRAM_E_ARG;  // the row factor needs the true 64-column partials
  gram_prof::Scope prof(GRAM_K_ROWOPS, (hipStream_t)stream, (8.0 + 2.0 * pieces) * rows * d);
  if (ids_are_i64)
    hipLaunchKernelGGL(embed_ex_kernel<int64_t>, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, table, (const int64_t*)ids,
                       x, (p16*)xb, ss, xs_out, nblk, rows, d, pieces);
  else
    hipLaunchKernelGGL(embed_ex_kernel<int32_t>, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, table, (const int32_t*)ids,
                       x, (p16*)xb, ss, xs_out, nblk, rows, d, pieces);
  GRAM_CHECK_LAUNCH();
  return 0;
}

extern "C" int gram_embed_i64(const float* table, const int64_t* ids, float* x, int rows, int d, void* stream) {
  if (rows < 1 || (d & 3)) return GRAM_E_ARG;
  gram_prof::Scope prof(GRAM_K_ROWOPS, (hipStream_t)stream, 8.0 * rows * d);
  hipLaunchKernelGGL(embed_kernel<int64_t>, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, table, ids, x, rows, d);
  GRAM_CHECK_LAUNCH();
  return 0;
}
extern "C" int gram_embed_i32(const float* table, const int32_t* ids, float* x, int rows, int d, void* stream) {
  if (rows < 1 || (d & 3)) return GRAM_E_ARG;
  gram_prof::Scope prof(GRAM_K_ROWOPS, (hipStream_t)stream, 8.0 * rows * d);
  hipLaunchKernelGGL(embed_kernel<int32_t>, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, table, ids, x, rows, d);
  GRAM_CHECK_LAUNCH();
  return 0;
}
extern "C" int gram_rmsnorm_bf16(const float* x, const float* w, void* out, int rows, int d, float eps, float scale,
                                 const float* pos, int N, int L, void* stream) {
  return gram_rmsnorm_bf16_map(x, w, out, rows, d, eps, scale, pos, N, L, nullptr, stream);
}

extern "C" int gram_rmsnorm_bf16_map(const float* x, const float* w, void* out, int rows, int d, float eps, float scale,
                                     const float* pos, int N, int L, const int32_t* passage_map, void* stream) {
  return gram_rmsnorm_bf16_split(x, w, out, rows, d, eps, scale, pos, N, L, passage_map, 1, stream);
}

extern "C" int gram_rmsnorm_bf16_split(const float* x, const float* w, void* out, int rows, int d, float eps, float scale,
                                       const float* pos, int N, int L, const int32_t* passage_map, int pieces, void* stream) {
  if (rows < 1 || (d & 3) || d > 1024 || (pos && (N < 1 || L < 1)) || pieces < 1 || pieces > GRAM_MAX_PIECES || (pieces > 1 && (d & 31)))
    return GRAM_E_ARG;
  if (!pos) {
    N = 1;
    L = 1;
  }
  gram_prof::Scope prof(GRAM_K_ROWOPS, (hipStream_t)stream, (4.0 + 2.0 * pieces) * rows * d);
  hipLaunchKernelGGL(rmsnorm_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, w, (p16*)out, rows, d, eps,
                     scale, pos, N, L, passage_map, pieces);
  GRAM_CHECK_LAUNCH();
  return 0;
}
extern "C" int gram_row_lse(const float* logits, float* lse, int R, int V, void* stream) {
  if (R < 1 || (V & 3)) return GRAM_E_ARG;
  gram_prof::Scope prof(GRAM_K_LSE, (hipStream_t)stream, 4.0 * R * V);
  hipLaunchKernelGGL(row_lse_kernel, dim3(R), dim3(256), 0, (hipStream_t)stream, logits, lse, V);
  GRAM_CHECK_LAUNCH();
  return 0;
}

extern "C" int gram_gather_passage_x(const float* cache_x, const int32_t* slot, float* x, int n, int L, int cache_L, int d,
                                     void* stream) {
  if (!cache_x || !slot || !x || n < 1 || L < 1 || cache_L < 1 || d < 4 || (d & 3)) return GRAM_E_ARG;
  const int64_t nchunks = (int64_t)n * L * (d / 4);
  if ((nchunks + 255) / 256 > 0x7fffffffLL) return GRAM_E_ARG;
  gram_prof::Scope prof(GRAM_K_ROWOPS, (hipStream_t)stream, 8.0 * n * L * d);
  hipLaunchKernelGGL(gather_passage_x_kernel, dim3((unsigned)((nchunks + 255) / 256)), dim3(256), 0, (hipStream_t)stream, cache_x,
                     slot, x, nchunks, L, cache_L, d / 4);
  GRAM_CHECK_LAUNCH();
  return 0;
}
