// gemm.hip -- C[M,N] (+)= A[M,K] @ W[N,K]^T on MFMA (bf16 in, fp32 accumulate), gfx950.
//
// Replaces every nn.Linear of the path (reference: gram_t5_modeling.py:300-301,369-372,
// gram_t5.py:254).  Both operands are K-contiguous ("NT" GEMM), which is exactly the MFMA
// fragment shape: lane (r = l&15, g = l>>4) reads 8 consecutive k of one row = one 16-byte
// LDS read, no transposition anywhere.
//
// Tiling: 128x128x64 block tile, 256 threads = 4 waves (2x2), each wave a 64x64 sub-tile =
// 4x4 MFMA 16x16x32 tiles.  The MFMA is issued "swapped" (W fragment as the A operand, the
// activation fragment as B) so that each lane ends up with 4 CONSECUTIVE output columns of one
// output row -> 8-byte bf16 / 16-byte fp32 vector epilogue accesses.
//
// LDS tiles are [128 rows][64 k] bf16 (128-B rows), XOR-swizzled in 16-byte chunks
// (chunk ^= (row>>1)&7) so the ds_read_b128 lane groups of gfx950 are conflict-free.
//
// Staging variants (picked per problem by pick_variant(); measured in tests/bench_gemm.py):
//   V_REG2  registers -> LDS, 2 LDS stages (64 KiB), loads issued one k-tile ahead, one barrier
//           per k-tile, 2 workgroups/CU.  Best when the grid cannot fill the chip (decoder at
//           small batch): the in-block prefetch is the only latency hiding there is.
//   V_DMA   global_load_lds (LDS-DMA, 16 B/lane, no staging VGPRs), 1 LDS stage (32 KiB),
//           two barriers per k-tile, 4 workgroups/CU: latency is hidden by the other resident
//           workgroups (thread-level parallelism) instead of by in-block software pipelining.
//           The swizzle is applied to the per-lane SOURCE address (the DMA writes LDS linearly).
// Workgroup ids are remapped XCD-aware (ids i and i+8 share an XCD and its 4 MiB L2): every
// XCD walks a contiguous range of tiles, n-tile fastest, so an A row-panel is fetched by one
// XCD only and the (small) weight matrix stays L2-resident.
#include <stdlib.h>

#include "common.h"
#include "prof.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;  // 16 KiB per operand per stage

enum { V_REG2 = 0, V_DMA = 1, V_DMA2 = 2, V_DMA_M256 = 3, V_DMA2_M256 = 4, V_DMA2_256SQ = 5, V_DMA_256SQ = 6, V_RING = 7, V_IL = 8, V_W4 = 13 };
int g_force_variant = -1;  // tuning hook (gram_debug_set_gemm_variant)
int g_stagger = 0;         // start stagger of the persistent kernel (measured: no gain), see gemm_il_kernel


__device__ __forceinline__ int swz(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

struct EpiArgs {
  void* C;
  int ldc;
  float* lse_part;  // GRAM_EPI_F32_LSE: [M][N/64][2]
  int lse_nblk;
  // T5LayerNorm folding (gram_norm_fusion_t)
  bf16* xb_out;        // producer: bf16 copy of the updated residual
  float* ss_out;       // producer: [M][N/64] partial sums of squares
  const float* ss_in;  // consumer: [M][ss_nblk] partials of A's rows
  int ss_nblk, ss_out_nblk;
  float inv_d, eps;
  // KV bank
  bf16* bank_k;
  bf16* bank_vt;
  int S, H, B, inner;
  const int32_t* pmap;  // compacted encoder rows: passage p = m / pL is flat passage pmap[p] = b*pN + n (NULL: identity)
  int pL, pN;
};

// XCD-aware bijective remap of the 1-D workgroup id -> (m-tile, n-tile), n fastest.
__device__ __forceinline__ void tile_of_block(int ntn, int& mt, int& nt) {
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, local = bid >> 3;
  const int q = nwg >> 3, r = nwg & 7;
  const int id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
  mt = id / ntn;
  nt = id - mt * ntn;
}

// one k-tile of MFMAs from a staged LDS tile pair
template <int TNW>
__device__ __forceinline__ void compute_tile(const char* sa, const char* sw, int wm, int wn, int r16, int g,
                                             f32x4 (&acc)[TNW][4]) {
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    bf16x8 fw[TNW], fa[4];
#pragma unroll
    for (int i = 0; i < TNW; ++i) fw[i] = *reinterpret_cast<const bf16x8*>(sw + swz(wn * 16 * TNW + i * 16 + r16, ks * 4 + g));
#pragma unroll
    for (int i = 0; i < 4; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(sa + swz(wm * 64 + i * 16 + r16, ks * 4 + g));
#pragma unroll
    for (int i = 0; i < TNW; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = mfma16(fw[i], fa[j], acc[i][j]);
  }
}

// acc[i][j] lane (r16,g) element e = C[m = m0+wm*64+j*16+r16][n = n0+wn*16*TNW+i*16+4g+e]
//
// bf16 outputs: a lane owns 4 consecutive columns (8 B) of two adjacent n-tiles; lanes g and g^1
// swap halves (one cross-lane exchange per tile pair) so that every lane issues ONE 16-byte store
// per tile pair instead of two 8-byte ones -- the store tail of these K=768 GEMMs is issue-bound.
// consumer side of the folded T5LayerNorm: 1/rms of row m of A from the producer's per-64-column partial sums
// of squares, added in a fixed order (deterministic)
__device__ __forceinline__ float row_rscale(const EpiArgs& ep, int m) {
  if (!ep.ss_in) return 1.f;
  if (ep.ss_nblk == 0) return ep.ss_in[m];  // already 1/rms (gram_row_rscale)
  const float2* p = reinterpret_cast<const float2*>(ep.ss_in + (size_t)m * ep.ss_nblk);
  float s = 0.f;
  for (int i = 0; i < ep.ss_nblk / 2; ++i) {
    const float2 v = p[i];
    s += v.x + v.y;
  }
  return rsqrtf(s * ep.inv_d + ep.eps);
}

// The four output rows a lane owns (m = mbase + j*16 + r16).  Called BEFORE a tile's k-loop so that the
// dependent loads of the partials are hidden behind the main loop instead of stalling the epilogue.
__device__ __forceinline__ void load_row_scales(const EpiArgs& ep, int mbase, int r16, int M, float (&rs4)[4]) {
#pragma unroll
  for (int j = 0; j < 4; ++j) rs4[j] = row_rscale(ep, min(mbase + j * 16 + r16, M - 1));
  // pin the values HERE: without this hipcc sinks the dependent loads down to their use in the epilogue
  // (registers are tight), where they stall every tile by ~3 us
  if (ep.ss_in) {
#pragma unroll
    for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(rs4[j]));
  }
}

__device__ __forceinline__ uint2 pack_bf16x4(f32x4 v) {
  bf16x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) o[e] = (bf16)v[e];
  return __builtin_bit_cast(uint2, o);
}

template <int EPI, int TNW>
__device__ __forceinline__ void epilogue(f32x4 (&acc)[TNW][4], int m0, int n0, int wm, int wn, int r16, int g, int M,
                                         const EpiArgs& ep, const float (&rs4)[4]) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int m = m0 + wm * 64 + j * 16 + r16;
    const bool row_ok = m < M;
    if constexpr (EPI == GRAM_EPI_BF16 || EPI == GRAM_EPI_BF16_RELU) {
      const float rs = rs4[j];
#pragma unroll
      for (int i = 0; i < TNW; i += 2) {
        f32x4 v0 = acc[i][j] * rs, v1 = acc[i + 1][j] * rs;
        if constexpr (EPI == GRAM_EPI_BF16_RELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v0[e] = fmaxf(v0[e], 0.f);
            v1[e] = fmaxf(v1[e], 0.f);
          }
        }
        const uint2 lo = pack_bf16x4(v0), hi = pack_bf16x4(v1);
        const bool odd = g & 1;
        const uint2 send = odd ? lo : hi;
        uint2 recv;
        recv.x = __shfl_xor(send.x, 16, 64);
        recv.y = __shfl_xor(send.y, 16, 64);
        // even g: tile i, columns 8*(g/2)..+7 = [own lo | partner's lo]; odd g: tile i+1 = [partner's hi | own hi]
        const uint4 out = odd ? make_uint4(recv.x, recv.y, hi.x, hi.y) : make_uint4(lo.x, lo.y, recv.x, recv.y);
        const int n = n0 + wn * 16 * TNW + (i + (odd ? 1 : 0)) * 16 + 8 * (g >> 1);
        if (row_ok) *reinterpret_cast<uint4*>(reinterpret_cast<bf16*>(ep.C) + (size_t)m * ep.ldc + n) = out;
      }
    } else if constexpr (EPI == GRAM_EPI_F32_LSE) {
      // logits + softmax partials of this wave's 64-column block (TNW == 4): a lane holds 16 of the
      // 64 values of row m (4 per n-tile), the other 48 sit in the lanes g^1, g^2, g^3 of the same r16
      static_assert(TNW == 4, "F32_LSE epilogue is written for 64-column wave tiles");
      float mx = -INFINITY;
#pragma unroll
      for (int i = 0; i < TNW; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) mx = fmaxf(mx, acc[i][j][e]);
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      float sm = 0.f;
#pragma unroll
      for (int i = 0; i < TNW; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) sm += __expf(acc[i][j][e] - mx);
      sm += __shfl_xor(sm, 16, 64);
      sm += __shfl_xor(sm, 32, 64);
      if (row_ok) {
        if (ep.C) {  // logits == NULL: partials only (the beam kernel recomputes the few logits it needs)
#pragma unroll
          for (int i = 0; i < TNW; ++i) {
            const int n = n0 + wn * 16 * TNW + i * 16 + 4 * g;
            *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(ep.C) + (size_t)m * ep.ldc + n) = acc[i][j];
          }
        }
        if (g == 0) {
          const int blk = (n0 >> 6) + wn;
          float2 pr = make_float2(mx, sm);
          *reinterpret_cast<float2*>(ep.lse_part + ((size_t)m * ep.lse_nblk + blk) * 2) = pr;
        }
      }
    } else {
      // no early-out for rows past M: the sum-of-squares reduction below is a wave-wide shuffle
      float ssq[TNW / 4];
#pragma unroll
      for (int q = 0; q < TNW / 4; ++q) ssq[q] = 0.f;
#pragma unroll
      for (int i = 0; i < TNW; ++i) {
        const int n = n0 + wn * 16 * TNW + i * 16 + 4 * g;
        f32x4 v = acc[i][j];
        if (row_ok) {
        if constexpr (EPI == GRAM_EPI_F32_ADD) {
          f32x4* p = reinterpret_cast<f32x4*>(reinterpret_cast<float*>(ep.C) + (size_t)m * ep.ldc + n);
          const f32x4 nv = *p + v;
          *p = nv;
          if (ep.xb_out) {
            *reinterpret_cast<uint2*>(ep.xb_out + (size_t)m * ep.ldc + n) = pack_bf16x4(nv);
            ssq[i / 4] += (nv[0] * nv[0] + nv[1] * nv[1]) + (nv[2] * nv[2] + nv[3] * nv[3]);
          }
        } else if constexpr (EPI == GRAM_EPI_F32) {
          *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(ep.C) + (size_t)m * ep.ldc + n) = v;
        } else {  // GRAM_EPI_KV_BANK
          int b, s;
          if (ep.pmap) {
            const int p = m / ep.pL, l = m - p * ep.pL, flat = ep.pmap[p];
            b = flat / ep.pN;
            s = (flat - b * ep.pN) * ep.pL + l;
          } else {
            b = m / ep.S;
            s = m - b * ep.S;
          }
          const int lw = n / ep.inner;  // layer*2 + which   (uniform per block: inner % 128 == 0)
          const int layer = lw >> 1, which = lw & 1;
          const int rem = n - lw * ep.inner;
          const int h = rem >> 6, d = rem & 63;
          const size_t head = ((size_t)layer * ep.B + b) * ep.H + h;
          if (which == 0) {
            const uint2 o = pack_bf16x4(v);
            *reinterpret_cast<uint2*>(ep.bank_k + (head * ep.S + s) * 64 + d) = o;
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) ep.bank_vt[(head * 64 + d + e) * ep.S + s] = (bf16)v[e];
          }
        }
        }
      }
      if constexpr (EPI == GRAM_EPI_F32_ADD) {
        if (ep.ss_out) {  // wave-uniform: per (row, 64-column block) partial sum of squares of the NEW residual
#pragma unroll
          for (int q = 0; q < TNW / 4; ++q) {
            float t = ssq[q];
            t += __shfl_xor(t, 16, 64);
            t += __shfl_xor(t, 32, 64);
            if (g == 0 && row_ok) ep.ss_out[(size_t)m * ep.ss_out_nblk + ((n0 + wn * 16 * TNW) >> 6) + q] = t;
          }
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_reg2_kernel(const bf16* __restrict__ A, const bf16* __restrict__ W, int M,
                                                           int N, int K, int lda, EpiArgs ep) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;  // 2x2 waves
  int mt, nt;
  tile_of_block(N / BN, mt, nt);
  const int m0 = mt * BM, n0 = nt * BN;
  const int r16 = lane & 15, g = lane >> 4;

  // staging map: 4 passes, thread -> (row = tid>>3 + 32*i, chunk = tid&7)
  const int srow = tid >> 3, schunk = tid & 7;
  const bf16* a_src[4];
  const bf16* w_src[4];
  bool a_ok[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int row = srow + 32 * i;
    a_ok[i] = (m0 + row) < M;
    a_src[i] = A + (size_t)(a_ok[i] ? (m0 + row) : 0) * lda + schunk * 8;
    w_src[i] = W + (size_t)(n0 + row) * K + schunk * 8;
  }
  bf16x8 ra[4], rw[4];
  auto load_tile = [&](int kt) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ra[i] = a_ok[i] ? ld_global_b128(a_src[i] + kt * BK) : zero_bf16x8();
      rw[i] = ld_global_b128(w_src[i] + kt * BK);
    }
  };
  auto store_tile = [&](int stage) {
    char* sa = smem + stage * 2 * TILE_BYTES;
    char* sw = sa + TILE_BYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int row = srow + 32 * i;
      *reinterpret_cast<bf16x8*>(sa + swz(row, schunk)) = ra[i];
      *reinterpret_cast<bf16x8*>(sw + swz(row, schunk)) = rw[i];
    }
  };

  f32x4 acc[4][4];  // [n-tile][m-tile]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nkt = K / BK;
  float rs4[4];
  load_row_scales(ep, m0 + wm * 64, r16, M, rs4);
  load_tile(0);
  store_tile(0);
  __syncthreads();
  for (int kt = 0; kt < nkt; ++kt) {
    const int stage = kt & 1;
    if (kt + 1 < nkt) load_tile(kt + 1);
    const char* sa = smem + stage * 2 * TILE_BYTES;
    compute_tile<4>(sa, sa + TILE_BYTES, wm, wn, r16, g, acc);
    if (kt + 1 < nkt) store_tile(stage ^ 1);
    __syncthreads();
  }
  epilogue<EPI, 4>(acc, m0, n0, wm, wn, r16, g, M, ep, rs4);
}

// ---------------------------------------------------------------------------------------------
template <int EPI, int WM, int NST, int TNW>
__global__ __launch_bounds__(WM * 128, (TNW == 8 ? 2 : (NST == 1 ? 4 : 2))) void gemm_dma_kernel(
    const bf16* __restrict__ A, const bf16* __restrict__ W, int M, int N, int K, int lda, EpiArgs ep) {
  // WM x 2 waves, block tile (64*WM) x 128.  NST = 1: single LDS stage, latency hidden by the other
  // resident workgroups.  NST = 2: the DMA of k-tile kt+1 is issued before the MFMAs of k-tile kt and
  // drained by the (single) barrier after them.
  constexpr int TBM = 64 * WM, TBN = 32 * TNW;
  constexpr int A_BYTES = TBM * BK * 2, W_BYTES = TBN * BK * 2, STAGE = A_BYTES + W_BYTES;
  constexpr int NW = 2 * WM;                       // waves per workgroup
  constexpr int AG = (TBM / 8) / NW, WG = (TBN / 8) / NW;  // 8-row DMA groups per wave
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  int mt, nt;
  tile_of_block(N / TBN, mt, nt);
  const int m0 = mt * TBM, n0 = nt * TBN;
  const int r16 = lane & 15, g = lane >> 4;

  // LDS-DMA map: a wave instruction fills one 8-row x 128-B group (1 KiB, lane-linear).
  // Lane l lands at (row 8*grp + l>>3, slot l&7), so it must FETCH chunk (l&7) ^ ((row>>1)&7)
  // for the read-side swizzle to find it.
  const bf16* a_src[AG];
  const bf16* w_src[WG];
#pragma unroll
  for (int i = 0; i < AG; ++i) {
    const int row = (wave * AG + i) * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    const int am = min(m0 + row, M - 1);  // rows past M: any valid address (never stored)
    a_src[i] = A + (size_t)am * lda + chunk * 8;
  }
#pragma unroll
  for (int i = 0; i < WG; ++i) {
    const int row = (wave * WG + i) * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    w_src[i] = W + (size_t)(n0 + row) * K + chunk * 8;
  }
  auto dma = [&](int kt, int stage) {
    char* sa = smem + stage * STAGE;
    char* sw = sa + A_BYTES;
#pragma unroll
    for (int i = 0; i < AG; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_src[i] + kt * BK),
                                       (__attribute__((address_space(3))) void*)(sa + (wave * AG + i) * 1024), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < WG; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(w_src[i] + kt * BK),
                                       (__attribute__((address_space(3))) void*)(sw + (wave * WG + i) * 1024), 16, 0, 0);
  };

  f32x4 acc[TNW][4];
#pragma unroll
  for (int i = 0; i < TNW; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nkt = K / BK;
  float rs4[4];
  load_row_scales(ep, m0 + wm * 64, r16, M, rs4);
  if constexpr (NST == 1) {
    for (int kt = 0; kt < nkt; ++kt) {
      dma(kt, 0);
      __syncthreads();  // hipcc drains the DMA (vmcnt(0)) ahead of the barrier
      compute_tile<TNW>(smem, smem + A_BYTES, wm, wn, r16, g, acc);
      __syncthreads();
    }
  } else {
    dma(0, 0);
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
      const int st = kt & 1;
      if (kt + 1 < nkt) dma(kt + 1, st ^ 1);
      compute_tile<TNW>(smem + st * STAGE, smem + st * STAGE + A_BYTES, wm, wn, r16, g, acc);
      __syncthreads();  // drains DMA(kt+1) and fences the reads of stage st
    }
  }
  epilogue<EPI, TNW>(acc, m0, n0, wm, wn, r16, g, M, ep, rs4);
}

// ---------------------------------------------------------------------------------------------
// V_RING: 256x256 tile, 8 waves (4 x 2, wave tile 64 x 128), k-stages of 32 in a 4-slot LDS ring
// (4 x 32 KiB = 128 KiB): three stages (96 KiB) are always in flight while the fourth is consumed.
// Counter profiles of the simpler variants showed the MFMA pipe ~37 % busy, zero bank conflicts,
// and an L2->LDS fill rate pinned at bytes-in-flight / DMA latency; this variant raises the bytes
// in flight per CU instead of the occupancy.  DMA completion is tracked with COUNTED s_waitcnt
// vmcnt(N) (each wave issues exactly 4 DMA instructions per stage) and a raw s_barrier, so the
// prefetch stays in flight across the barrier (__syncthreads() would drain it).
// Stage tiles are [256 rows][32 k] bf16 = 64-B rows; 16-B chunk swizzle chunk ^= F[(row>>2)&3],
// F = {0,2,3,1}, conflict-free for the ds_read_b128 lane groups (derivation in DESIGN.md).
__device__ __forceinline__ int fsw(int row) { return (0x78 >> (((row >> 2) & 3) * 2)) & 3; }
__device__ __forceinline__ int swz32(int row, int chunk) { return row * 64 + ((chunk ^ fsw(row)) << 4); }

template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm_ring_kernel(const bf16* __restrict__ A, const bf16* __restrict__ W, int M, int N,
                                                           int K, int lda, EpiArgs ep) {
  constexpr int TB = 256, SK = 32, OPB = TB * SK * 2 /*16 KiB*/, STAGE = 2 * OPB, NSLOT = 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  int mt, nt;
  tile_of_block(N / TB, mt, nt);
  const int m0 = mt * TB, n0 = nt * TB;
  const int r16 = lane & 15, g = lane >> 4;

  // DMA map: one wave instruction = 16 rows x 64 B (1 KiB, lane-linear): lane l -> row 16*grp + (l>>2),
  // slot l&3, fetching chunk slot ^ F.  Each operand stage is 16 groups; wave w owns groups 2w, 2w+1.
  const bf16* a_src[2];
  const bf16* w_src[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = (wave * 2 + i) * 16 + (lane >> 2);
    const int chunk = (lane & 3) ^ fsw(row);
    a_src[i] = A + (size_t)min(m0 + row, M - 1) * lda + chunk * 8;
    w_src[i] = W + (size_t)(n0 + row) * K + chunk * 8;
  }
  auto dma = [&](int kt) {
    char* sa = smem + (kt & (NSLOT - 1)) * STAGE;
    char* sw = sa + OPB;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_src[i] + kt * SK),
                                       (__attribute__((address_space(3))) void*)(sa + (wave * 2 + i) * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(w_src[i] + kt * SK),
                                       (__attribute__((address_space(3))) void*)(sw + (wave * 2 + i) * 1024), 16, 0, 0);
    }
  };

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nk = K / SK;
  float rs4[4];
  load_row_scales(ep, m0 + wm * 64, r16, M, rs4);
  dma(0);
  if (nk > 1) dma(1);
  if (nk > 2) dma(2);
  for (int kt = 0; kt < nk; ++kt) {
    // retire this wave's share of stage kt, leaving the younger stages in flight
    const int ahead = min(2, nk - 1 - kt);
    if (ahead == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (ahead == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // stage kt complete for every wave; slot (kt-1)&3 no longer read
    if (kt + 3 < nk) dma(kt + 3);
    const char* sa = smem + (kt & (NSLOT - 1)) * STAGE;
    const char* sw = sa + OPB;
    bf16x8 fw[8], fa[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) fw[i] = *reinterpret_cast<const bf16x8*>(sw + swz32(wn * 128 + i * 16 + r16, g));
#pragma unroll
    for (int i = 0; i < 4; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(sa + swz32(wm * 64 + i * 16 + r16, g));
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = mfma16(fw[i], fa[j], acc[i][j]);
  }
  epilogue<EPI, 8>(acc, m0, n0, wm, wn, r16, g, M, ep, rs4);
}

template <int EPI>
int launch_ring(const void* A, const void* W, int M, int N, int K, int lda, EpiArgs ep, hipStream_t st) {
  constexpr int smem = 4 * 2 * 256 * 32 * 2;  // 128 KiB
  if (N % 256 || K % 32) return GRAM_E_ARG;
  const int nblocks = (N / 256) * ((M + 255) / 256);
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_ring_kernel<EPI>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  hipLaunchKernelGGL(gemm_ring_kernel<EPI>, dim3(nblocks), dim3(512), smem, st, (const bf16*)A, (const bf16*)W, M, N, K, lda, ep);
  GRAM_CHECK_LAUNCH();
  return 0;
}


// ---------------------------------------------------------------------------------------------
// Row-contiguous epilogue for the 256x256 kernel (wave tile 64 rows x 128 columns).  The direct
// epilogue touches 16 half-lines per store instruction (16 rows x 64 B) and is store-issue-bound
// (ablation in tests/bench_gemm.py: ~60 % of the store cost remains even when the stores hit a
// cache-resident region).  Here each wave transposes its accumulators through a PRIVATE 8-KiB LDS
// patch (the second DMA stage is idle during the epilogue) and then issues stores/RMWs that cover
// whole rows: 4 rows x 256 B (bf16) or 2 rows x 512 B (fp32) per instruction, full 128-B lines only.
// 16-byte chunks are XOR-swizzled by row so both the transposing writes and the row reads are
// conflict-free.  Wave-private: no workgroup barrier, only the wave's own LDS ordering.
template <int EPI>
__device__ __forceinline__ void epilogue_rows(f32x4 (&acc)[8][4], char* patch /* this wave's 8 KiB */, int m0, int n0, int wm,
                                              int wn, int lane, int M, const EpiArgs& ep, const float (&rs4)[4]) {
  const int r16 = lane & 15, g = lane >> 4;
  if constexpr (EPI == GRAM_EPI_BF16 || EPI == GRAM_EPI_BF16_RELU) {
    // two passes of 32 rows x 128 cols bf16: patch[32][256 B], chunk c (16 B) stored at c ^ (row & 15)
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const int j = pass * 2 + jj;
        const int row = jj * 16 + r16;
        const float rs = rs4[j];  // folded T5LayerNorm (1 if unused)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          f32x4 v = acc[i][j] * rs;
          if constexpr (EPI == GRAM_EPI_BF16_RELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
          }
          // 8-byte piece: columns i*16 + 4g .. +3  -> chunk (i*2 + g/2), half (g&1)
          const int chunk = (i * 2 + (g >> 1)) ^ (row & 15);
          *reinterpret_cast<uint2*>(patch + row * 256 + chunk * 16 + (g & 1) * 8) = pack_bf16x4(v);
        }
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int row = it * 4 + (lane >> 4), c = lane & 15;
        const uint4 val = *reinterpret_cast<const uint4*>(patch + row * 256 + ((c ^ (row & 15)) * 16));
        const int m = m0 + wm * 64 + pass * 32 + row;
        if (m < M) *reinterpret_cast<uint4*>(reinterpret_cast<bf16*>(ep.C) + (size_t)m * ep.ldc + n0 + wn * 128 + c * 8) = val;
      }
      __builtin_amdgcn_wave_barrier();
    }
  } else {
    // fp32: four passes of 16 rows x 128 cols: patch[16][512 B], chunk c (16 B, 32 per row) at c ^ (row & 15)*2
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int chunk = (i * 4 + g) ^ ((r16 & 15) << 1);
        *reinterpret_cast<f32x4*>(patch + r16 * 512 + chunk * 16) = acc[i][j];
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int row = it * 2 + (lane >> 5), c = lane & 31;
        f32x4 val = *reinterpret_cast<const f32x4*>(patch + row * 512 + ((c ^ ((row & 15) << 1)) * 16));
        const int m = m0 + wm * 64 + j * 16 + row;
        float ssq = 0.f;
        if (m < M) {
          f32x4* pc = reinterpret_cast<f32x4*>(reinterpret_cast<float*>(ep.C) + (size_t)m * ep.ldc + n0 + wn * 128 + c * 4);
          if constexpr (EPI == GRAM_EPI_F32_ADD) val += *pc;
          *pc = val;
          if constexpr (EPI == GRAM_EPI_F32_ADD) {
            if (ep.xb_out) {
              *reinterpret_cast<uint2*>(ep.xb_out + (size_t)m * ep.ldc + n0 + wn * 128 + c * 4) = pack_bf16x4(val);
              ssq = (val[0] * val[0] + val[1] * val[1]) + (val[2] * val[2] + val[3] * val[3]);
            }
          }
        }
        if constexpr (EPI == GRAM_EPI_F32_ADD) {
          if (ep.ss_out) {  // 16 lanes cover one 64-column block of one row
            ssq += __shfl_xor(ssq, 1, 64);
            ssq += __shfl_xor(ssq, 2, 64);
            ssq += __shfl_xor(ssq, 4, 64);
            ssq += __shfl_xor(ssq, 8, 64);
            if ((c & 15) == 0 && m < M) ep.ss_out[(size_t)m * ep.ss_out_nblk + ((n0 + wn * 128) >> 6) + (c >> 4)] = ssq;
          }
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
}

// ---------------------------------------------------------------------------------------------
// V_IL: the 256x256 double-buffered DMA variant with the next k-tile's 8 DMA instructions
// INTERLEAVED between MFMA groups (one DMA after every 8 MFMAs) instead of issued back-to-back
// after the barrier: an LDS-DMA instruction costs ~100+ issue cycles inside a busy phase, and in the
// plain variant all 8 waves pay 8 of them at the same moment while the MFMA pipe idles.
template <int EPI, int ABL = 0>  // ABL (ablation, microbench only): 1 = DMA only, 2 = MFMA only, 3 = no stores
__global__ __launch_bounds__(512, 2) void gemm_il_kernel(const bf16* __restrict__ A, const bf16* __restrict__ W, int M, int N,
                                                         int K, int lda, EpiArgs ep, int ntiles, int stagger) {
  // PERSISTENT: one workgroup per CU walks tiles t = round*G + slot.  With one 128-KiB workgroup per
  // CU all CUs run in lockstep, so a per-tile launch leaves the output stream (up to 1.2 GB per GEMM)
  // un-overlapped with MFMA work; here the stores of tile i drain while tile i+1 is fetched/computed,
  // and the first DMA stage of tile i+1 is issued BEFORE the epilogue of tile i.
  constexpr int TB = 256, OPB = TB * BK * 2 /*32 KiB*/, STAGE = 2 * OPB;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int r16 = lane & 15, g = lane >> 4;
  const int ntn = N / TB;
  // XCD-aware slot: workgroups b, b+8, ... share an XCD; give each XCD a contiguous run of every round
  const int G = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, local = bid >> 3, q = G >> 3, rr = G & 7;
  const int slot = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + local;

  const bf16* src[8];  // pieces 0-3: A groups 4w..4w+3, pieces 4-7: W groups 4w..4w+3
  auto set_tile = [&](int tile, int& m0, int& n0) {
    const int mt = tile / ntn, nt = tile - mt * ntn;
    m0 = mt * TB;
    n0 = nt * TB;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = (wave * 4 + i) * 8 + (lane >> 3);
      const int chunk = (lane & 7) ^ ((row >> 1) & 7);
      src[i] = A + (size_t)min(m0 + row, M - 1) * lda + chunk * 8;
      src[4 + i] = W + (size_t)(n0 + row) * K + chunk * 8;
    }
  };
  auto dma_piece = [&](int kt, int stage, int p) {
    char* dst = smem + stage * STAGE + (p >= 4 ? OPB : 0) + (wave * 4 + (p & 3)) * 1024;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[p] + kt * BK),
                                     (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
  };

  f32x4 acc[8][4];
  bf16x8 fw0[8], fa0[4];
  auto ldfrag = [&](bf16x8 (&fw)[8], bf16x8 (&fa)[4], const char* sa, const char* sw, int ks) {
#pragma unroll
    for (int i = 0; i < 8; ++i) fw[i] = *reinterpret_cast<const bf16x8*>(sw + swz(wn * 128 + i * 16 + r16, ks * 4 + g));
#pragma unroll
    for (int i = 0; i < 4; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(sa + swz(wm * 64 + i * 16 + r16, ks * 4 + g));
  };
  auto mma = [&](bf16x8 (&fw)[8], bf16x8 (&fa)[4], int kt, int st, int ks, bool more) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if constexpr (ABL != 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = mfma16(fw[i], fa[j], acc[i][j]);
      }
      if ((i & 1) == 1) {
        if (more && ABL != 2) dma_piece(kt + 1, st ^ 1, ks * 4 + (i >> 1));
        __builtin_amdgcn_sched_barrier(0);  // keep the DMA between these MFMA groups
      }
    }
  };

  const int nkt = K / BK;  // nkt is even or odd; the stage parity restarts at 0 for every tile
  // De-synchronise the CUs: identical tiles keep all 256 workgroups in lockstep, so their epilogues
  // would hit HBM as one 32-MB burst that every workgroup then waits out at its next barrier.  A start
  // stagger of (slot % 8) * ~1/8 tile period spreads the store traffic over the whole period.
  if (stagger > 0) {
    const int units = ((slot & 7) * stagger * nkt) >> 3;
    for (int i = 0; i < units; ++i) __builtin_amdgcn_s_sleep(8);  // 8 * 64 = 512 cycles
  }
  int tile = slot;
  int m0 = 0, n0 = 0;
  if (tile < ntiles) {
    set_tile(tile, m0, n0);
#pragma unroll
    for (int p = 0; p < 8; ++p) dma_piece(0, 0, p);
  }
  while (tile < ntiles) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float rs4[4];
    load_row_scales(ep, m0 + wm * 64, r16, M, rs4);  // consumed by the epilogue; latency hidden by the k-loop
    __syncthreads();  // stage 0 of this tile landed (and the previous tile's stores are issued/drained)
    for (int kt = 0; kt < nkt; ++kt) {
      const int st = kt & 1;
      const bool more = kt + 1 < nkt;
      const char* sa = smem + st * STAGE;
      const char* sw = sa + OPB;
      ldfrag(fw0, fa0, sa, sw, 0);
      mma(fw0, fa0, kt, st, 0, more);
      ldfrag(fw0, fa0, sa, sw, 1);
      mma(fw0, fa0, kt, st, 1, more);
      __syncthreads();  // drains DMA(kt+1), fences the reads of stage st
    }
    // next tile: start its first stage now (both LDS stages are free), then store this tile
    const int cur_m0 = m0, cur_n0 = n0;
    tile += G;
    if (tile < ntiles) {
      set_tile(tile, m0, n0);
#pragma unroll
      for (int p = 0; p < 8; ++p) dma_piece(0, 0, p);
    }
    if constexpr (ABL == 3) {  // ablation: no stores (keep acc live)
      float keep = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) keep += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
      if (keep == 123.456f) reinterpret_cast<float*>(ep.C)[0] = keep;
    } else if constexpr (ABL == 4) {  // ablation: same stores, but always into this workgroup's first tile (cache-resident)
      epilogue<EPI, 8>(acc, (slot / ntn) * TB, (slot % ntn) * TB, wm, wn, r16, g, M, ep, rs4);
    } else if constexpr (EPI == GRAM_EPI_KV_BANK) {
      epilogue<EPI, 8>(acc, cur_m0, cur_n0, wm, wn, r16, g, M, ep, rs4);
    } else {
      // stage 1 is idle here (the next tile's first DMA went to stage 0): 8 waves x 8 KiB patches
      epilogue_rows<EPI>(acc, smem + STAGE + wave * 8192, cur_m0, cur_n0, wm, wn, lane, M, ep, rs4);
    }
  }
}

template <int EPI, int ABL = 0>
int launch_il(const void* A, const void* W, int M, int N, int K, int lda, EpiArgs ep, hipStream_t st) {
  constexpr int smem = 2 * 2 * 256 * 64 * 2;  // 128 KiB
  if (N % 256) return GRAM_E_ARG;
  const int ntiles = (N / 256) * ((M + 255) / 256);
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return GRAM_E_ARG;
    n_cu = prop.multiProcessorCount;
  }
  const int nblocks = ntiles < n_cu ? ntiles : n_cu;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_il_kernel<EPI, ABL>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  hipLaunchKernelGGL((gemm_il_kernel<EPI, ABL>), dim3(nblocks), dim3(512), smem, st, (const bf16*)A, (const bf16*)W, M, N, K, lda, ep,
                     ntiles, g_stagger);
  GRAM_CHECK_LAUNCH();
  return 0;
}

// ---------------------------------------------------------------------------------------------
// V_W4: persistent 256x256 tile computed by FOUR waves (one per SIMD, 512 VGPRs each), wave tile
// 128x128 = 8x8 MFMA tiles (256 accumulator registers).  The 8-wave kernel above is bounded by LDS ->
// register traffic: a 64x128 wave tile reads (64+128)*32*2 B = 12 KiB per 32 MFMAs, 96 KiB per k-step
// per CU ~ 768 LDS cycles against 1024 MFMA cycles.  128x128 wave tiles read 16 KiB per 64 MFMAs =
// 64 KiB per k-step per CU (-33 %), and the register budget allows double-buffered fragments so the
// ds_reads of k-step s+1 overlap the MFMAs of k-step s inside the single wave each SIMD runs.
template <int EPI>
__device__ __forceinline__ void epilogue_rows_w4(f32x4 (&acc)[8][8], char* patch /* this wave's 16 KiB */, int m0, int n0, int wm,
                                                 int wn, int lane, int M, const EpiArgs& ep, const float (&rs8)[8]) {
  const int r16 = lane & 15, g = lane >> 4;
  if constexpr (EPI == GRAM_EPI_BF16 || EPI == GRAM_EPI_BF16_RELU) {
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {  // 64 rows x 128 cols bf16: patch[64][256 B]
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const int j = pass * 4 + jj;
        const int row = jj * 16 + r16;
        const float rs = rs8[j];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          f32x4 v = acc[i][j] * rs;
          if constexpr (EPI == GRAM_EPI_BF16_RELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
          }
          const int chunk = (i * 2 + (g >> 1)) ^ (row & 15);
          *reinterpret_cast<uint2*>(patch + row * 256 + chunk * 16 + (g & 1) * 8) = pack_bf16x4(v);
        }
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int it = 0; it < 16; ++it) {
        const int row = it * 4 + (lane >> 4), c = lane & 15;
        const uint4 val = *reinterpret_cast<const uint4*>(patch + row * 256 + ((c ^ (row & 15)) * 16));
        const int m = m0 + wm * 128 + pass * 64 + row;
        if (m < M) *reinterpret_cast<uint4*>(reinterpret_cast<bf16*>(ep.C) + (size_t)m * ep.ldc + n0 + wn * 128 + c * 8) = val;
      }
      __builtin_amdgcn_wave_barrier();
    }
  } else {
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {  // 32 rows x 128 cols fp32: patch[32][512 B]
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const int j = pass * 2 + jj;
        const int row = jj * 16 + r16;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int chunk = (i * 4 + g) ^ ((row & 15) << 1);
          *reinterpret_cast<f32x4*>(patch + row * 512 + chunk * 16) = acc[i][j];
        }
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int it = 0; it < 16; ++it) {
        const int row = it * 2 + (lane >> 5), c = lane & 31;
        f32x4 val = *reinterpret_cast<const f32x4*>(patch + row * 512 + ((c ^ ((row & 15) << 1)) * 16));
        const int m = m0 + wm * 128 + pass * 32 + row;
        float ssq = 0.f;
        if (m < M) {
          f32x4* pc = reinterpret_cast<f32x4*>(reinterpret_cast<float*>(ep.C) + (size_t)m * ep.ldc + n0 + wn * 128 + c * 4);
          if constexpr (EPI == GRAM_EPI_F32_ADD) val += *pc;
          *pc = val;
          if constexpr (EPI == GRAM_EPI_F32_ADD) {
            if (ep.xb_out) {
              *reinterpret_cast<uint2*>(ep.xb_out + (size_t)m * ep.ldc + n0 + wn * 128 + c * 4) = pack_bf16x4(val);
              ssq = (val[0] * val[0] + val[1] * val[1]) + (val[2] * val[2] + val[3] * val[3]);
            }
          }
        }
        if constexpr (EPI == GRAM_EPI_F32_ADD) {
          if (ep.ss_out) {
            ssq += __shfl_xor(ssq, 1, 64);
            ssq += __shfl_xor(ssq, 2, 64);
            ssq += __shfl_xor(ssq, 4, 64);
            ssq += __shfl_xor(ssq, 8, 64);
            if ((c & 15) == 0 && m < M) ep.ss_out[(size_t)m * ep.ss_out_nblk + ((n0 + wn * 128) >> 6) + (c >> 4)] = ssq;
          }
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
}

// LDS-DMA issued through inline asm: with the builtin, hipcc treats the DMA as an LDS store that may
// alias every pending ds_read and puts s_waitcnt lgkmcnt(0) in front of it, which serialises the fragment
// prefetch this kernel is built around.  The asm form is invisible to the waitcnt pass, so the kernel waits
// for its DMA explicitly (dma_wait) before the barrier that publishes a stage.
__device__ __forceinline__ void dma16_asm(uint32_t lds_addr /*wave-uniform*/, uint32_t voff, const char* base /*uniform*/) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_addr), "v"(voff), "s"(base));
}
__device__ __forceinline__ void dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

template <int EPI, int ABL = 0>
__global__ __launch_bounds__(256, 1) void gemm_w4_kernel(const bf16* __restrict__ A, const bf16* __restrict__ W, int M, int N,
                                                         int K, int lda, EpiArgs ep, int ntiles) {
  // The k-tiles of all the tiles a workgroup walks form ONE stream: the DMA issued during k-tile s always
  // fetches k-tile s+1 of the stream (the first k-tile of the next output tile when s is a tile's last), so
  // the k-loop body is identical for every k-tile and the LDS stage simply alternates.
  constexpr int TB = 256, OPB = TB * BK * 2 /*32 KiB*/, STAGE = 2 * OPB;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int r16 = lane & 15, g = lane >> 4;
  const int ntn = N / TB;
  const int G = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, local = bid >> 3, q = G >> 3, rr = G & 7;
  const int slot = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + local;
  if (slot >= ntiles) return;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const uint32_t wave_lds = __builtin_amdgcn_readfirstlane(lds0 + wave * 8192);  // this wave's 8 row groups of either operand

  // per-lane BYTE offsets of this wave's 16 DMA pieces (8 A row groups, 8 W row groups of 8 rows x 128 B);
  // 32-bit (the launcher checks the operands are < 4 GiB): the uniform base + k offset stays in SGPRs
  uint32_t offA[8], offW[8];
  auto set_tile = [&](int tile, int& m0, int& n0) {
    const int mt = tile / ntn, nt = tile - mt * ntn;
    m0 = mt * TB;
    n0 = nt * TB;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int row = (wave * 8 + i) * 8 + (lane >> 3);
      const int chunk = (lane & 7) ^ ((row >> 1) & 7);
      offA[i] = ((uint32_t)min(m0 + row, M - 1) * (uint32_t)lda + chunk * 8) * 2u;
      offW[i] = ((uint32_t)(n0 + row) * (uint32_t)K + chunk * 8) * 2u;
    }
  };
  auto dma_piece = [&](const char* gA, const char* gW, int stage, int p) {
    const uint32_t dst = wave_lds + stage * STAGE + (p >= 8 ? OPB : 0) + (p & 7) * 1024;
    if (p >= 8) dma16_asm(dst, offW[p & 7], gW);
    else dma16_asm(dst, offA[p & 7], gA);
  };

  f32x4 acc[8][8];
  bf16x8 fw[2][8], fa[2][8];
  auto ldfrag = [&](int buf, const char* sa, const char* sw, int ks) {
#pragma unroll
    for (int i = 0; i < 8; ++i) fw[buf][i] = *reinterpret_cast<const bf16x8*>(sw + swz(wn * 128 + i * 16 + r16, ks * 4 + g));
#pragma unroll
    for (int i = 0; i < 8; ++i) fa[buf][i] = *reinterpret_cast<const bf16x8*>(sa + swz(wm * 128 + i * 16 + r16, ks * 4 + g));
  };

  const int nkt = K / BK;
  const char* const Ab = reinterpret_cast<const char*>(A);
  const char* const Wb = reinterpret_cast<const char*>(W);
  int tile = slot;
  int m0, n0;
  set_tile(tile, m0, n0);
#pragma unroll
  for (int p = 0; p < 16; ++p) dma_piece(Ab, Wb, 0, p);
  dma_wait();
  __syncthreads();
  int st = 0;
  ldfrag(0, smem, smem + OPB, 0);
  while (true) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float rs8[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) rs8[j] = row_rscale(ep, min(m0 + wm * 128 + j * 16 + r16, M - 1));
    if (ep.ss_in) {
#pragma unroll
      for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(rs8[j]));
    }
    const int cur_m0 = m0, cur_n0 = n0;
    const int next = tile + G;
    __syncthreads();  // every wave's epilogue patch (in the stage the first DMA below overwrites) is free again
    for (int kt = 0; kt < nkt; ++kt) {
      // where the next k-tile of the stream lives
      const char *gA, *gW;
      if (kt + 1 < nkt) {
        gA = Ab + (kt + 1) * (BK * 2);
        gW = Wb + (kt + 1) * (BK * 2);
      } else {  // first k-tile of the next tile (of this tile again if there is none: harmless, keeps the body uniform)
        set_tile(next < ntiles ? next : tile, m0, n0);
        gA = Ab;
        gW = Wb;
      }
      const char* sa = smem + st * STAGE;
      if constexpr (ABL != 5) ldfrag(1, sa, sa + OPB, 1);
      __builtin_amdgcn_sched_barrier(0);  // k-step-1 fragment loads go out BEFORE the k-step-0 MFMAs
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if constexpr (ABL != 1) {
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[i][j] = mfma16(fw[0][i], fa[0][j], acc[i][j]);
        }
        if constexpr (ABL != 2 && ABL != 5 && ABL != 6 && ABL != 7) {
          dma_piece(gA, gW, st ^ 1, 2 * i);
          dma_piece(gA, gW, st ^ 1, 2 * i + 1);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if constexpr (ABL != 1) {
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[i][j] = mfma16(fw[1][i], fa[1][j], acc[i][j]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);  // keep the 32 MFMAs above between the last DMA issue and its wait
      if constexpr (ABL != 9) dma_wait();
      if constexpr (ABL != 6 && ABL != 8) __syncthreads();  // the DMA landed; every wave has finished its LDS reads of stage st
      st ^= 1;
      if constexpr (ABL != 5) ldfrag(0, smem + st * STAGE, smem + st * STAGE + OPB, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 4; i < 8; ++i) {  // hides the fragment loads above
        if constexpr (ABL != 1) {
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[i][j] = mfma16(fw[1][i], fa[1][j], acc[i][j]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // st = stage of the next tile's first k-tile (its k-step-0 fragments are already in registers); the other
    // stage is idle until the next tile's first DMA, which the barrier at the loop top orders after this epilogue
    if constexpr (ABL == 3 || ABL >= 5) {
      float keep = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) keep += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
      if (keep == 123.456f) reinterpret_cast<float*>(ep.C)[0] = keep;
    } else {
      epilogue_rows_w4<EPI>(acc, smem + (st ^ 1) * STAGE + wave * 16384, cur_m0, cur_n0, wm, wn, lane, M, ep, rs8);
    }
    if (next >= ntiles) break;
    tile = next;
  }
}

template <int EPI, int ABL = 0>
int launch_w4(const void* A, const void* W, int M, int N, int K, int lda, EpiArgs ep, hipStream_t st) {
  constexpr int smem = 2 * 2 * 256 * 64 * 2;  // 128 KiB
  if (N % 256) return GRAM_E_ARG;
  if ((size_t)M * lda * 2 >= (1ull << 32) || (size_t)N * K * 2 >= (1ull << 32)) return GRAM_E_ARG;  // 32-bit DMA offsets
  if constexpr (EPI == GRAM_EPI_KV_BANK || EPI == GRAM_EPI_F32_LSE) {
    return GRAM_E_ARG;
  } else {
    const int ntiles = (N / 256) * ((M + 255) / 256);
    static int n_cu = 0;
    if (n_cu == 0) {
      int dev = 0;
      hipDeviceProp_t prop;
      if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return GRAM_E_ARG;
      n_cu = prop.multiProcessorCount;
    }
    const int nblocks = ntiles < n_cu ? ntiles : n_cu;
    static bool attr_set = false;
    if (!attr_set) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_w4_kernel<EPI, ABL>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, smem);
      if (e != hipSuccess) return (int)e;
      attr_set = true;
    }
    hipLaunchKernelGGL((gemm_w4_kernel<EPI, ABL>), dim3(nblocks), dim3(256), smem, st, (const bf16*)A, (const bf16*)W, M, N, K, lda,
                       ep, ntiles);
    GRAM_CHECK_LAUNCH();
    return 0;
  }
}

// Measured on MI355X (tests/bench_gemm.py, B = 512 shapes; TFLOP/s):
//   M >= 32768 (encoder, bank):  V_IL (persistent 256x256, row-contiguous epilogue) 935-948 (521 for the
//                                HBM-bound N=768,K=768 residual GEMM) > V_DMA_M256 780-815 > V_DMA
//   lm_head (N = 32128, not a multiple of 256):  V_DMA_M256 783 > V_DMA 614
//   M ~ 10240 (decoder):         V_DMA 475-832 > V_DMA_M256 418-801 > V_IL 299-696 (too few tiles per CU)
int pick_variant(int M, int N, int K) {
  if (g_force_variant >= 0) return g_force_variant;
  const long tiles256 = (long)((M + 255) / 256) * (N / BN);
  static const int big = getenv("GRAM_GEMM_BIG") ? atoi(getenv("GRAM_GEMM_BIG")) : V_IL;  // A/B hook
  // persistent 256x256: needs >= ~2 tiles per CU to amortise its fill/drain; with a long K (FFN-wo, K = 4*d) it
  // already wins at 480 tiles (M = 40960, N = 768: 230 us vs 277 us for the 128x128 variant)
  if (N % 256 == 0 && M >= 32768 && (tiles256 >= 2048 || K >= 2048)) return big;
  if (tiles256 >= 2048) return V_DMA_M256;
  return V_DMA;
}

template <int EPI, int WM, int NST, int TNW = 4>
int launch_dma(const void* A, const void* W, int M, int N, int K, int lda, EpiArgs ep, hipStream_t st) {
  constexpr int TBM = 64 * WM, TBN = 32 * TNW;
  constexpr int smem = NST * (TBM + TBN) * BK * 2;
  if (N % TBN) return GRAM_E_ARG;
  const int nblocks = (N / TBN) * ((M + TBM - 1) / TBM);
  static bool attr_set = false;
  if (!attr_set && smem > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_dma_kernel<EPI, WM, NST, TNW>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  hipLaunchKernelGGL((gemm_dma_kernel<EPI, WM, NST, TNW>), dim3(nblocks), dim3(WM * 128), smem, st, (const bf16*)A, (const bf16*)W,
                     M, N, K, lda, ep);
  GRAM_CHECK_LAUNCH();
  return 0;
}

template <int EPI>
int launch(const void* A, const void* W, int M, int N, int K, int lda, EpiArgs ep, hipStream_t st) {
  gram_prof::Scope prof(GRAM_K_GEMM, st, 2.0 * M * N * K);
  if constexpr (EPI == GRAM_EPI_F32_LSE) {  // written for the 64-column wave tiles only
    return pick_variant(M, N, K) == V_DMA ? launch_dma<EPI, 2, 1>(A, W, M, N, K, lda, ep, st)
                                          : launch_dma<EPI, 4, 1>(A, W, M, N, K, lda, ep, st);
  } else {
  switch (pick_variant(M, N, K)) {
    case V_DMA: return launch_dma<EPI, 2, 1>(A, W, M, N, K, lda, ep, st);
    case V_DMA2: return launch_dma<EPI, 2, 2>(A, W, M, N, K, lda, ep, st);
    case V_DMA_M256: return launch_dma<EPI, 4, 1>(A, W, M, N, K, lda, ep, st);
    case V_DMA2_M256: return launch_dma<EPI, 4, 2>(A, W, M, N, K, lda, ep, st);
    case V_DMA2_256SQ: return launch_dma<EPI, 4, 2, 8>(A, W, M, N, K, lda, ep, st);
    case V_DMA_256SQ: return launch_dma<EPI, 4, 1, 8>(A, W, M, N, K, lda, ep, st);
    case V_RING: return launch_ring<EPI>(A, W, M, N, K, lda, ep, st);
    case V_IL: return launch_il<EPI>(A, W, M, N, K, lda, ep, st);
    case 9: return launch_il<EPI, 1>(A, W, M, N, K, lda, ep, st);
    case 10: return launch_il<EPI, 2>(A, W, M, N, K, lda, ep, st);
    case 11: return launch_il<EPI, 3>(A, W, M, N, K, lda, ep, st);
    case 12: return launch_il<EPI, 4>(A, W, M, N, K, lda, ep, st);
    case V_W4: return launch_w4<EPI>(A, W, M, N, K, lda, ep, st);
    case 14: return launch_w4<EPI, 1>(A, W, M, N, K, lda, ep, st);
    case 15: return launch_w4<EPI, 2>(A, W, M, N, K, lda, ep, st);
    case 16: return launch_w4<EPI, 3>(A, W, M, N, K, lda, ep, st);
    case 17: return launch_w4<EPI, 5>(A, W, M, N, K, lda, ep, st);
    case 18: return launch_w4<EPI, 6>(A, W, M, N, K, lda, ep, st);
    case 19: return launch_w4<EPI, 7>(A, W, M, N, K, lda, ep, st);
    case 20: return launch_w4<EPI, 8>(A, W, M, N, K, lda, ep, st);
    case 21: return launch_w4<EPI, 9>(A, W, M, N, K, lda, ep, st);
    default: break;
  }
  const int nblocks = (N / BN) * ((M + BM - 1) / BM);
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_reg2_kernel<EPI>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 4 * TILE_BYTES);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  hipLaunchKernelGGL(gemm_reg2_kernel<EPI>, dim3(nblocks), dim3(256), 4 * TILE_BYTES, st, (const bf16*)A, (const bf16*)W, M, N,
                     K, lda, ep);
  GRAM_CHECK_LAUNCH();
  return 0;
  }
}

}  // namespace

extern "C" int gram_debug_set_gemm_variant(int v) {
  if (v >= 1000) {  // 1000 + s: set the persistent kernel's start stagger instead
    g_stagger = v - 1000;
    return 0;
  }
  g_force_variant = v;
  return 0;
}

extern "C" int gram_gemm_bf16_lse(const void* A, const void* W, float* logits, float* lse_part, int M, int N, int K, int lda,
                                  int ldc, void* stream) {
  if (M < 1 || N % BN != 0 || K % BK != 0 || lda < K || (lda & 7) || !lse_part || (logits && (ldc & 3))) return GRAM_E_ARG;
  EpiArgs ep{};
  ep.C = logits;
  ep.ldc = ldc;
  ep.lse_part = lse_part;
  ep.lse_nblk = N / 64;
  return launch<GRAM_EPI_F32_LSE>(A, W, M, N, K, lda, ep, (hipStream_t)stream);
}

extern "C" int gram_gemm_bf16(const void* A, const void* W, void* C, int M, int N, int K, int lda, int ldc, int epilogue,
                              const gram_kv_bank_t* bank, void* stream) {
  return gram_gemm_bf16_ex(A, W, C, M, N, K, lda, ldc, epilogue, bank, nullptr, stream);
}

extern "C" int gram_gemm_bf16_ex(const void* A, const void* W, void* C, int M, int N, int K, int lda, int ldc, int epilogue,
                                 const gram_kv_bank_t* bank, const gram_norm_fusion_t* nf, void* stream) {
  if (M < 1 || N % BN != 0 || K % BK != 0 || lda < K || (lda & 7)) return GRAM_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  EpiArgs ep{};
  ep.C = C;
  ep.ldc = ldc;
  if (nf) {
    if (epilogue == GRAM_EPI_F32_ADD) {
      if ((nf->xb_out == nullptr) != (nf->ss_out == nullptr)) return GRAM_E_ARG;
      ep.xb_out = (bf16*)nf->xb_out;
      ep.ss_out = nf->ss_out;
      ep.ss_out_nblk = N / 64;
    } else if (epilogue == GRAM_EPI_BF16 || epilogue == GRAM_EPI_BF16_RELU) {
      if (nf->ss_in && nf->nblk_in != 0 && (nf->nblk_in < 2 || (nf->nblk_in & 1) || nf->d < 64)) return GRAM_E_ARG;
      ep.ss_in = nf->ss_in;
      ep.ss_nblk = nf->nblk_in;
      ep.inv_d = 1.0f / (float)nf->d;
      ep.eps = nf->eps;
    } else if (nf->xb_out || nf->ss_out || nf->ss_in) {
      return GRAM_E_ARG;
    }
  }
  switch (epilogue) {
    case GRAM_EPI_BF16:
      if (!C || (ldc & 7)) return GRAM_E_ARG;  // 16-byte row-aligned bf16 stores
      return launch<GRAM_EPI_BF16>(A, W, M, N, K, lda, ep, st);
    case GRAM_EPI_BF16_RELU:
      if (!C || (ldc & 7)) return GRAM_E_ARG;
      return launch<GRAM_EPI_BF16_RELU>(A, W, M, N, K, lda, ep, st);
    case GRAM_EPI_F32_ADD:
      if (!C || (ldc & 3)) return GRAM_E_ARG;
      return launch<GRAM_EPI_F32_ADD>(A, W, M, N, K, lda, ep, st);
    case GRAM_EPI_F32:
      if (!C || (ldc & 3)) return GRAM_E_ARG;
      return launch<GRAM_EPI_F32>(A, W, M, N, K, lda, ep, st);
    case GRAM_EPI_KV_BANK: {
      if (!bank || !bank->k || !bank->vt) return GRAM_E_ARG;
      const int inner = bank->H * 64;
      if (inner % BN != 0 || N != bank->n_layers * 2 * inner || bank->S % 16 != 0) return GRAM_E_ARG;
      if (bank->passage_map) {
        if (bank->L < 16 || bank->N < 1 || bank->N * bank->L != bank->S || M % bank->L != 0 || M > bank->B * bank->S) return GRAM_E_ARG;
      } else if (M != bank->B * bank->S) {
        return GRAM_E_ARG;
      }
      ep.bank_k = (bf16*)bank->k;
      ep.bank_vt = (bf16*)bank->vt;
      ep.S = bank->S;
      ep.H = bank->H;
      ep.B = bank->B;
      ep.inner = inner;
      ep.pmap = bank->passage_map;
      ep.pL = bank->L;
      ep.pN = bank->N;
      return launch<GRAM_EPI_KV_BANK>(A, W, M, N, K, lda, ep, st);
    }
    default:
      return GRAM_E_ARG;
  }
}
