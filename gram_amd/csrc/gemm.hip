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
// Kernels (picked per problem by pick_variant(); measured in tests/bench_gemm.py):
//   gemm_pp_kernel     persistent 256x256 "ping-pong" 8-phase kernel (M >= 32 768): see its own header below.
//   gemm_dma_kernel    global_load_lds (LDS-DMA, 16 B/lane, no staging VGPRs), 1 LDS stage, two barriers per k-tile, up to
//                      4 workgroups/CU: latency is hidden by the other resident workgroups.  256x128, 128x128 and 64x128
//                      tiles.  The swizzle is applied to the per-lane SOURCE address (the DMA writes LDS linearly).
//                      With at most one workgroup per CU (gridDim <= 256: a decoder step of 30 .. 200 users) nothing else is
//                      resident, and the NST >= 3 instantiations run a ring of 4-6 stages instead (counted vmcnt, one barrier
//                      per k-tile).
//   gemm_stream_kernel M <= 512 (one user .. ~25): one 16-column n-tile per WORKGROUP, W and A streamed through a 150-KiB LDS
//                      ring by LDS-DMA; bf16 / bf16+ReLU / fp32-residual epilogues (16-column sum-of-squares partials).
//   gemm_skinny_kernel M <= 64: one 16-column n-tile per wave over the whole K, operands straight from global memory
//                      (what is left on it: the lm_head of a small batch and producers of 64-column partials).
// All of them take two-piece ("x3") operands as well (gram_split_t, template parameter X3): A and W are then INTERLEAVED matrices
// [rows][K/32][piece 0: 32 columns | piece 1: 32 columns], i.e. a 64-column physical k-tile holds both pieces of one 32-column block
// of K, fetched ONCE, and yields three MFMA products per fragment pair -- a0*w1, a1*w0, a0*w0, in this order, block after block --
// instead of one product-major pass over K per product (round 2: every operand tile travelled L2 -> LDS -> registers once per
// product).  The MFMA sequence per accumulator is the same in every kernel, so their outputs are bit-identical.
// Workgroup ids are remapped XCD-aware (ids i and i+8 share an XCD and its 4 MiB L2): every
// XCD walks a contiguous range of tiles.
#include <stdlib.h>
#include <type_traits>

#include "common.h"
#include "prof.h"

namespace {

#ifndef GRAM_PP_RES_NT
#define GRAM_PP_RES_NT 0  // A/B build hook: the ping-pong kernel's residual read with the nt hint
#endif
// Ablation builds of the ping-pong kernel (`make ABL=n` -> libgram_hip_abl<n>.so, loaded by tests/bench_gemm_x3.py through GRAM_LIB;
// the product library is built with 0 and contains none of it).  Bits: 1 = no tile-end epilogue (results wrong), 2 = no operand DMA
// after the prologue, 4 = no LDS fragment reads after the prologue, (8: was the clock stamps, now always on: gram_prof_pp_clock), 16 = the tile-end epilogue without its
// global stores (fp32: without the stores, the bf16 copy and the partials; the residual loads stay), 32 = the fp32-residual tile-end
// epilogue without its fp32 store (the 16-bit copy, the partials and the residual load stay).
#ifndef GRAM_PP_ABL
#define GRAM_PP_ABL 0
#endif
#ifndef GRAM_PP_PASS_PRIO
#define GRAM_PP_PASS_PRIO 2  // A/B build hook: wave priority of the in-load-slot epilogue's passes (the MFMA slots run at 1)
#endif
#ifndef GRAM_PP_INSL
#define GRAM_PP_INSL 0  // A/B build hook: 1 = two-piece 16-bit outputs from inside the pipeline (measured: no gain, profiles/r03h, r03i)
#endif
// In-kernel clock of the ping-pong GEMM (MI355X_MICROARCH.md, DVFS item 6) -- a DIAGNOSTIC, off unless gram_prof_pp_clock_enable(1)
// (bench.py switches it on for its timed region): every workgroup then stamps s_memtime (shader cycles) and s_memrealtime (100-MHz
// ticks) around its tile loop and adds the two differences to these sums -- two atomics per workgroup and launch; switched off, the
// workgroup still reads the two counters (see the kernel) and skips the atomics.
// gram_prof_pp_clock() = sum / sum x 0.1 GHz: the time-weighted clock the chip held inside these kernels since the last reset (the MFMA
// peak it can be priced against: the chip is power-limited in them).
__device__ unsigned long long g_pp_clk[2];
constexpr int BN = 128, BK = 64;

enum { V_DMA_M64 = 31, V_DMA = 1, V_DMA_M256 = 3, V_PP = 22, V_RING_M64 = 33, V_RING_M128 = 34 };  // (ids kept from the variant table of round 1)
int g_force_variant = -1;  // tuning hook (gram_debug_set_gemm_variant)
int g_stagger = 0;         // start stagger of the persistent kernel (measured: no gain)
int g_pp_entry_delay = 0;  // test hook (gram_debug_set_gemm_variant(2000 + n)): in the CHAOS build wave group 1 of the ping-pong kernel sleeps n x 512 cycles in the prologue
int g_pp_clk_on = 0;       // gram_prof_pp_clock_enable: the ping-pong kernel's clock stamps (diagnostic; the product path runs without)


__device__ __forceinline__ int swz(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

struct EpiArgs {
  void* C;
  int ldc;
  float* lse_part;  // GRAM_EPI_F32_LSE: [M][N/64][2]
  int lse_nblk;
  // T5LayerNorm folding (gram_norm_fusion_t)
  p16* xb_out;        // producer: bf16 copy of the updated residual
  float* ss_out;       // producer: [M][N/64] partial sums of squares
  const float* ss_in;  // consumer: [M][ss_nblk] partials of A's rows
  int ss_nblk, ss_out_nblk;
  int ss_quarter;  // partials per 16 columns ([M][4 * nblk]) instead of per 64: the layout of gemm_stream_kernel (gram_norm_fusion_t.quarter)
  float inv_d, eps;
  // per-row power-of-two factor of the 16-bit copy of the residual stream (gram_norm_fusion_t.xs_in / xs_out)
  const float* xs_in;  // producer: xb = pieces(x * xs_in[m]); consumer: the row scale is divided by it
  float* xs_out;       // consumer: the first n-tile's workgroups write the next producer's factor
  // KV bank
  p16* bank_k;
  p16* bank_vt;
  int S, H, B, inner;
  const int32_t* pmap;  // compacted encoder rows: passage p = m / pL is flat passage pmap[p] = b*pN + n (NULL: identity)
  int pL, pN;
  int nt;  // streaming (nt) stores in the persistent kernel's row-contiguous epilogue (A/B hook GRAM_GEMM_NT)
  // ping-pong KV-bank epilogue: divisions by runtime values as multiply-high (scalar ALU; a division sequence costs
  // a dozen VGPRs the kernel does not have).  mg_x = floor(2^32 / x) + 1; exact for the dividends documented at use.
  uint32_t mg_pL32, mg_S32, mg_pN, mg_it;  // x = pL/32, S/32, pN, inner/256
  // split-bf16 outputs (gram_split_t): every bf16 result (C of the bf16 epilogues, xb_out, the bank) is written as `split` pieces
  // p0 = bf16(v), p1 = bf16(v - p0), ...; piece p lives `*_pstride` elements after piece p - 1
  int split;
  long c_pstride, bank_pstride;
  // split == 2: xb_out is interleaved ([M][2 * ldc], a 32-column block's two pieces side by side: the next GEMM's A operand); a bf16
  // C is interleaved too when c_inter != 0 (ldc is then its physical row stride, >= 2 N) and `c_pstride` apart otherwise (the
  // attention kernels read planar pieces)
  int c_inter;
  // every result is acc * out_scale (gram_split_t.out_scale): the inverse of the power-of-two factor the caller scaled W by, so that
  // the low pieces of small weights stay out of the f16 subnormal range.  A power of two (checked on the host): exact wherever it is
  // applied, so every kernel may fold it where it is cheapest (into the row scale, into the residual add as an fma) -- same bits.
  float out_scale;
};
inline uint32_t magic_u32(uint32_t d) { return d <= 1 ? 0u : (uint32_t)((1ull << 32) / d) + 1u; }
__device__ __forceinline__ uint32_t udiv_magic(uint32_t x, uint32_t d, uint32_t mg) { return d <= 1 ? x : __umulhi(x, mg); }

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void store16(void* p, uint4 v, bool nt) {
  if (nt) __builtin_nontemporal_store(__builtin_bit_cast(u32x4_t, v), reinterpret_cast<u32x4_t*>(p));
  else *reinterpret_cast<uint4*>(p) = v;
}
__device__ __forceinline__ void store16(void* p, f32x4 v, bool nt) {
  if (nt) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p));
  else *reinterpret_cast<f32x4*>(p) = v;
}
__device__ __forceinline__ void store8(void* p, uint2 v, bool nt) {
  if (nt) __builtin_nontemporal_store(__builtin_bit_cast(u32x2_t, v), reinterpret_cast<u32x2_t*>(p));
  else *reinterpret_cast<uint2*>(p) = v;
}

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
template <int TNW, bool X3>
__device__ __forceinline__ void compute_tile(const char* sa, const char* sw, int wm, int wn, int r16, int g,
                                             f32x4 (&acc)[TNW][4]) {
  if constexpr (X3) {
    // the tile's 64 physical columns = both pieces of one 32-column block of K (chunks 0..3: piece 0, chunks 4..7: piece 1):
    // a0*w1, a1*w0, a0*w0 -- smallest first -- from ONE fetch of the four fragment sets
    // (the A fragments of both pieces stay in registers, the W fragments come one n-tile at a time: 10 fragments live, not 16 --
    // the single-stage instantiations run at 128 registers per lane; an accumulator meets its three products 4 MFMAs apart)
    p16x8 fa0[4], fa1[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) fa0[i] = *reinterpret_cast<const p16x8*>(sa + swz(wm * 64 + i * 16 + r16, g));
#pragma unroll
    for (int i = 0; i < 4; ++i) fa1[i] = *reinterpret_cast<const p16x8*>(sa + swz(wm * 64 + i * 16 + r16, 4 + g));
#pragma unroll
    for (int i = 0; i < TNW; ++i) {
      const p16x8 fw1 = *reinterpret_cast<const p16x8*>(sw + swz(wn * 16 * TNW + i * 16 + r16, 4 + g));
      const p16x8 fw0 = *reinterpret_cast<const p16x8*>(sw + swz(wn * 16 * TNW + i * 16 + r16, g));
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = mfma16(fw1, fa0[j], acc[i][j]);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = mfma16(fw0, fa1[j], acc[i][j]);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = mfma16(fw0, fa0[j], acc[i][j]);
    }
    return;
  }
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    p16x8 fw[TNW], fa[4];
#pragma unroll
    for (int i = 0; i < TNW; ++i) fw[i] = *reinterpret_cast<const p16x8*>(sw + swz(wn * 16 * TNW + i * 16 + r16, ks * 4 + g));
#pragma unroll
    for (int i = 0; i < 4; ++i) fa[i] = *reinterpret_cast<const p16x8*>(sa + swz(wm * 64 + i * 16 + r16, ks * 4 + g));
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
// mn (optional): the smallest of the row's 64-column partials, sum: their total -- what the next producer's row factor is made of
// (row_xscale); only meaningful when the partials are added up here (ss_nblk != 0)
__device__ __forceinline__ float row_rscale(const EpiArgs& ep, int m, float* sum = nullptr, float* mn = nullptr) {
  if (!ep.ss_in) return 1.f;
  if (ep.ss_nblk == 0) return ep.ss_in[m];  // already 1/rms (gram_row_rscale)
  const float2* p = reinterpret_cast<const float2*>(ep.ss_in + (size_t)m * ep.ss_nblk);
  float s = 0.f, lo = INFINITY;
  for (int i = 0; i < ep.ss_nblk / 2; ++i) {
    const float2 v = p[i];
    s += v.x + v.y;
    lo = fminf(lo, fminf(v.x, v.y));
  }
  if (sum) *sum = s;
  if (mn) *mn = lo;
  return row_rs(s, ep.inv_d, ep.eps);
}

// The four output rows a lane owns (m = mbase + j*16 + r16).  Called BEFORE a tile's k-loop so that the
// dependent loads of the partials are hidden behind the main loop instead of stalling the epilogue.
// lead: this wave's tile starts at column 0 -- its lanes g == 0 also publish the next producer's row factors (gram_norm_fusion_t.xs_out)
__device__ __forceinline__ void load_row_scales(const EpiArgs& ep, int mbase, int r16, int M, float (&rs4)[4], bool lead = false) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int m = min(mbase + j * 16 + r16, M - 1);
    float sum = 0.f, mn = 0.f;
    const float r = row_rscale(ep, m, &sum, &mn);
    // (ss_nblk == 0: ss_in already holds rsqrt(..) / xs, gram_row_rscale_xs; a power of two: the division is exact)
    rs4[j] = (ep.xs_in && ep.ss_in && ep.ss_nblk != 0 ? r / ep.xs_in[m] : r) * ep.out_scale;
    if (lead && ep.xs_out && ep.ss_in && ep.ss_nblk != 0 && mbase + j * 16 + r16 < M) ep.xs_out[m] = row_xscale(sum, mn);
  }
  // pin the values HERE: without this hipcc sinks the dependent loads down to their use in the epilogue
  // (registers are tight), where they stall every tile by ~3 us
  if (ep.ss_in) {
#pragma unroll
    for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(rs4[j]));
  }
}

__device__ __forceinline__ uint2 pack_bf16x4(f32x4 v) {
  p16x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) o[e] = (p16)v[e];
  return __builtin_bit_cast(uint2, o);
}
__device__ __forceinline__ f32x4 unpack_bf16x4(uint2 u) {
  f32x4 r;
#ifdef GRAM_F16
  const p16x4 h = __builtin_bit_cast(p16x4, u);
#pragma unroll
  for (int e = 0; e < 4; ++e) r[e] = (float)h[e];
#else
  r[0] = __uint_as_float(u.x << 16);
  r[1] = __uint_as_float(u.x & 0xffff0000u);
  r[2] = __uint_as_float(u.y << 16);
  r[3] = __uint_as_float(u.y & 0xffff0000u);
#endif
  return r;
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
        const bool odd = g & 1;
        const int n = n0 + wn * 16 * TNW + (i + (odd ? 1 : 0)) * 16 + 8 * (g >> 1);
        for (int pc = 0; pc < ep.split; ++pc) {  // (one pass unless the output is split into bf16 pieces)
          const uint2 lo = pack_bf16x4(v0), hi = pack_bf16x4(v1);
          const uint2 send = odd ? lo : hi;
          uint2 recv;
          recv.x = __shfl_xor(send.x, 16, 64);
          recv.y = __shfl_xor(send.y, 16, 64);
          // even g: tile i, columns 8*(g/2)..+7 = [own lo | partner's lo]; odd g: tile i+1 = [partner's hi | own hi]
          const uint4 out = odd ? make_uint4(recv.x, recv.y, hi.x, hi.y) : make_uint4(lo.x, lo.y, recv.x, recv.y);
          p16* dst = reinterpret_cast<p16*>(ep.C) + (size_t)m * ep.ldc + (ep.c_inter ? inter_off(n, pc) : pc * ep.c_pstride + n);
          if (row_ok) *reinterpret_cast<uint4*>(dst) = out;
          if (pc + 1 < ep.split) {
            v0 -= unpack_bf16x4(lo);
            v1 -= unpack_bf16x4(hi);
          }
        }
      }
    } else if constexpr (EPI == GRAM_EPI_F32_LSE) {
      // logits + softmax partials of this wave's 64-column block (TNW == 4): a lane holds 16 of the
      // 64 values of row m (4 per n-tile), the other 48 sit in the lanes g^1, g^2, g^3 of the same r16
      static_assert(TNW == 4, "F32_LSE epilogue is written for 64-column wave tiles");
#pragma unroll
      for (int i = 0; i < TNW; ++i) acc[i][j] *= ep.out_scale;
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
        f32x4 v = acc[i][j] * ep.out_scale;
        if (row_ok) {
        if constexpr (EPI == GRAM_EPI_F32_ADD) {
          f32x4* p = reinterpret_cast<f32x4*>(reinterpret_cast<float*>(ep.C) + (size_t)m * ep.ldc + n);
          const f32x4 nv = *p + v;
          *p = nv;
          if (ep.xb_out) {
            f32x4 rem = ep.xs_in ? nv * ep.xs_in[m] : nv;
            p16* xrow = ep.xb_out + (size_t)m * ep.ldc * ep.split;
            for (int pc = 0; pc < ep.split; ++pc) {
              const uint2 pk = pack_bf16x4(rem);
              *reinterpret_cast<uint2*>(xrow + (ep.split == 2 ? inter_off(n, pc) : n)) = pk;
              rem -= unpack_bf16x4(pk);
            }
            ssq[i / 4] += sumsq4(nv);
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
          for (int pc = 0; pc < ep.split; ++pc) {
            const uint2 o = pack_bf16x4(v);
            if (which == 0) {
              *reinterpret_cast<uint2*>(ep.bank_k + pc * ep.bank_pstride + (head * ep.S + s) * 64 + d) = o;
            } else {
              const p16x4 ob = __builtin_bit_cast(p16x4, o);
#pragma unroll
              for (int e = 0; e < 4; ++e)  // V^T blocked by 32 keys: [head][s / 32][64 d][32]
                ep.bank_vt[pc * ep.bank_pstride + ((head * (ep.S >> 5) + (s >> 5)) * 64 + d + e) * 32 + (s & 31)] = ob[e];
            }
            v -= unpack_bf16x4(o);
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

// Row-contiguous fp32 epilogue for the 64x64 wave tiles of gemm_dma_kernel: the wave's accumulators go through a
// 4-KiB LDS patch (the k-loop's stage is idle by then), 16 rows per pass, so the residual read-modify-write moves
// whole 256-B row segments (and the bf16 copy whole 128-B lines) instead of 64-B / 32-B pieces.
template <int EPI>
__device__ __forceinline__ void epilogue_rows64(f32x4 (&acc)[4][4], char* patch /* this wave's 4 KiB */, int m0, int n0, int wm, int wn,
                                                int lane, int M, const EpiArgs& ep) {
  const int r16 = lane & 15, g = lane >> 4;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    f32x4 res[4];  // the pass's residual loads go out together, ahead of the LDS round trip (see epilogue_rows)
    if constexpr (EPI == GRAM_EPI_F32_ADD) {
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int m = min(m0 + wm * 64 + j * 16 + it * 4 + (lane >> 4), M - 1);
        res[it] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(ep.C) + (size_t)m * ep.ldc + n0 + wn * 64 + (lane & 15) * 4);
      }
    }
    // patch[16][256 B], chunk c (16 B) at c ^ row
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(patch + r16 * 256 + (((i * 4 + g) ^ r16) * 16)) = acc[i][j];
    __builtin_amdgcn_wave_barrier();
    float ssq4[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int row = it * 4 + (lane >> 4), c = lane & 15;
      f32x4 val = *reinterpret_cast<const f32x4*>(patch + row * 256 + ((c ^ row) * 16)) * ep.out_scale;
      const int m = m0 + wm * 64 + j * 16 + row;
      float ssq = 0.f;
      if (m < M) {
        f32x4* pc = reinterpret_cast<f32x4*>(reinterpret_cast<float*>(ep.C) + (size_t)m * ep.ldc + n0 + wn * 64 + c * 4);
        if constexpr (EPI == GRAM_EPI_F32_ADD) val += res[it];
        *pc = val;
        if constexpr (EPI == GRAM_EPI_F32_ADD) {
          if (ep.xb_out) {
            f32x4 rem = ep.xs_in ? val * ep.xs_in[m] : val;
            p16* xrow = ep.xb_out + (size_t)m * ep.ldc * ep.split;
            const int n = n0 + wn * 64 + c * 4;
            for (int pc = 0; pc < ep.split; ++pc) {
              const uint2 pk = pack_bf16x4(rem);
              *reinterpret_cast<uint2*>(xrow + (ep.split == 2 ? inter_off(n, pc) : n)) = pk;
              rem -= unpack_bf16x4(pk);
            }
            ssq = sumsq4(val);
          }
        }
      }
      ssq4[it] = ssq;
    }
    if constexpr (EPI == GRAM_EPI_F32_ADD) {
      if (ep.ss_out) {  // the 16 lanes of a row cover exactly one 64-column block; 4 independent butterflies
#pragma unroll
        for (int sh = 1; sh < 16; sh <<= 1)
#pragma unroll
          for (int it = 0; it < 4; ++it) ssq4[it] += __shfl_xor(ssq4[it], sh, 64);
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int m = m0 + wm * 64 + j * 16 + it * 4 + (lane >> 4);
          if ((lane & 15) == 0 && m < M) ep.ss_out[(size_t)m * ep.ss_out_nblk + ((n0 + wn * 64) >> 6)] = ssq4[it];
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// ---------------------------------------------------------------------------------------------
// M <= 64 (one user, or a few: the decoder at the reference's default --eval_batch_size 1 is a chain of ~1 000 such
// GEMMs).  A tiled kernel gives such a problem N/128 workgroups, each pulling a 128-column slab of W through one CU:
// the weights arrive at a fraction of the HBM rate.  Here a workgroup owns 64 columns and each of its 4 waves ONE
// 16-column n-tile over the whole K, operands straight from global memory (W rows are K-contiguous: a lane's 16 B are
// the MFMA operand; A is a few KiB, L2-resident), 8 k-blocks of loads in flight.  The accumulators then meet in LDS and
// wave 0 runs the common epilogue on the 64 x 64 tile, so every output bit equals the tiled kernels' (same MFMA
// sequence per accumulator, same epilogue code).
template <int EPI, int MT, bool X3>  // MT = m-tiles of 16 rows per workgroup that exist (M <= 16 * MT, or MT = 4 and a grid row per 64 rows)
__global__ __launch_bounds__(256) void gemm_skinny_kernel(const p16* __restrict__ A, const p16* __restrict__ W, int M, int N,
                                                          int K, int lda, EpiArgs ep) {
  __shared__ f32x4 xch[4][MT][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, g = lane >> 4;
  const int n0 = blockIdx.x * 64, m0 = blockIdx.y * 64;
  const p16* wp = W + (size_t)(n0 + wave * 16 + r16) * K + g * 8;
  const p16* ap[MT];
  bool a_ok[MT];
#pragma unroll
  for (int j = 0; j < MT; ++j) {
    a_ok[j] = m0 + j * 16 + r16 < M;
    ap[j] = A + (size_t)(a_ok[j] ? m0 + j * 16 + r16 : 0) * lda + g * 8;
  }
  f32x4 acc1[MT];
#pragma unroll
  for (int j = 0; j < MT; ++j) acc1[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // U k-blocks (32 wide) of loads in flight, then their MFMAs (a second register set with the next loads already under
  // way was measured slower: 13.0 vs 11.4 ms of GEMM time per generate at B = 1)
  constexpr int U = 8;
  const int nkb = K >> 5;  // physical 32-column k-blocks (X3: an even/odd pair = piece 0 / piece 1 of one block of K)
  for (int kb = 0; kb < nkb; kb += U) {
    p16x8 fw[U], fa[U][MT];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = min(kb + u, nkb - 1) << 5;  // (past the end: a valid re-read, not used)
      fw[u] = ld_global_b128(wp + k);
#pragma unroll
      for (int j = 0; j < MT; ++j) fa[u][j] = a_ok[j] ? ld_global_b128(ap[j] + k) : zero_bf16x8();
    }
    if constexpr (X3) {
#pragma unroll
      for (int u = 0; u < U; u += 2)
        if (kb + u < nkb) {  // a0*w1, a1*w0, a0*w0: the order of every kernel
#pragma unroll
          for (int j = 0; j < MT; ++j) acc1[j] = mfma16(fw[u + 1], fa[u][j], acc1[j]);
#pragma unroll
          for (int j = 0; j < MT; ++j) acc1[j] = mfma16(fw[u], fa[u + 1][j], acc1[j]);
#pragma unroll
          for (int j = 0; j < MT; ++j) acc1[j] = mfma16(fw[u], fa[u][j], acc1[j]);
        }
    } else {
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (kb + u < nkb) {
#pragma unroll
          for (int j = 0; j < MT; ++j) acc1[j] = mfma16(fw[u], fa[u][j], acc1[j]);
        }
    }
  }
#pragma unroll
  for (int j = 0; j < MT; ++j) xch[wave][j][lane] = acc1[j];
  gram_sync();
  if (wave != 0) return;
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = j < MT ? xch[i][j < MT ? j : 0][lane] : (f32x4){0.f, 0.f, 0.f, 0.f};
  if constexpr (EPI == GRAM_EPI_F32 || EPI == GRAM_EPI_F32_ADD) {
    // the row-contiguous epilogue of the 128-row kernel (its sum-of-squares butterfly is the one every default path uses);
    // the exchange buffer is free again: only this wave is left and its values are in registers
    __builtin_amdgcn_wave_barrier();
    epilogue_rows64<EPI>(acc, reinterpret_cast<char*>(&xch[0][0][0]), m0, n0, 0, 0, lane, M, ep);
  } else {
    float rs4[4];
    load_row_scales(ep, m0, r16, M, rs4, n0 == 0);
    epilogue<EPI, 4>(acc, m0, n0, 0, 0, r16, g, M, ep, rs4);
  }
}

template <int EPI, bool X3>
int launch_skinny(const void* A, const void* W, int M, int N, int K, int lda, EpiArgs ep, hipStream_t st) {
  if (M < 1 || N % 64 || K % (X3 ? 64 : 32) || (M + 63) / 64 > 65535) return GRAM_E_ARG;
  const dim3 grid(N / 64, (M + 63) / 64), block(256);
  if (M <= 16) hipLaunchKernelGGL((gemm_skinny_kernel<EPI, 1, X3>), grid, block, 0, st, (const p16*)A, (const p16*)W, M, N, K, lda, ep);
  else if (M <= 32) hipLaunchKernelGGL((gemm_skinny_kernel<EPI, 2, X3>), grid, block, 0, st, (const p16*)A, (const p16*)W, M, N, K, lda, ep);
  else hipLaunchKernelGGL((gemm_skinny_kernel<EPI, 4, X3>), grid, block, 0, st, (const p16*)A, (const p16*)W, M, N, K, lda, ep);
  GRAM_CHECK_LAUNCH();
  return 0;
}

// LDS-DMA issued through inline asm: with the builtin, hipcc treats the DMA as an LDS store that may
// alias every pending ds_read and puts s_waitcnt lgkmcnt(0) in front of it, which serialises the fragment
// prefetch this kernel is built around.  The asm form is invisible to the waitcnt pass, so the kernel waits
// for its DMA explicitly (counted s_waitcnt vmcnt) before the barrier that publishes a buffer.
// `s_nop 3`: invisible to the hazard recognizer as well.  On gfx9 a vector-memory instruction must not read an SGPR within 5 wait states
// of a VALU instruction writing it (v_readlane / v_readfirstlane: how hipcc reloads a spilled SGPR or makes an address uniform), and
// hipcc pads its own loads but cannot see the load inside an asm statement: round 4's ISA had `v_readlane_b32 s5, v255, 3` two wait
// states ahead of `global_load_lds_dword v2, s[4:5]` in four one-piece instantiations (tools/check_isa_hazards.py scans for it now).
// s_mov (1) + s_nop 3 (4) = 5 wait states between whatever precedes the statement and its load; s_nop 0 alone covers M0 -> LDS-DMA.
__device__ __forceinline__ void dma16_asm(uint32_t lds_addr /*wave-uniform*/, uint32_t voff, const char* base /*uniform*/) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 3\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_addr), "v"(voff), "s"(base));
}
__device__ __forceinline__ void dma4_asm(uint32_t lds_addr /*wave-uniform*/, uint32_t voff, const char* base /*uniform*/) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 3\n\tglobal_load_lds_dword %1, %2" ::"s"(lds_addr), "v"(voff), "s"(base));
}

// ---------------------------------------------------------------------------------------------
// M <= 64, second generation: gemm_skinny_kernel keeps 8 KiB of W in flight per wave, so an N = 768 problem (48 waves)
// streams its weights at ~0.3 TB/s and a one-user generate() -- a chain of ~600 such GEMMs -- spends 2/3 of its time
// there (rocprofv3, B = 1: 80 us for the FFN-out of the bf16x3 mode).  Here a WORKGROUP owns one 16-column n-tile and
// 16*MT rows, and its whole LDS is a ring of NS stages filled by LDS-DMA: a stage = 128 reduction columns of the W
// rows (16 x 256 B) and of the A rows (MT x 16 x 256 B), one DMA instruction per wave and tile; 50-76 KiB of W in
// flight per workgroup and as many workgroups as the launch can place (<= 256: one per CU).  Wave j < MT multiplies
// m-tile j: the SAME MFMA sequence per accumulator as every tiled kernel (k-blocks in order, chunk after chunk), so the
// results are bit-identical to theirs.  One barrier per stage publishes the landed stage and frees the consumed one.
// The sum-of-squares partials of the folded T5LayerNorm cover 16 columns here (`quarter` layout of
// gram_norm_fusion_t): the butterfly of the 64-column epilogues is (q0 + q1) + (q2 + q3) over exactly these quarters, and the
// consumer adds them in that order.
template <int MT>
struct StreamCfg {
  static constexpr int NS = MT == 1 ? 19 : MT == 2 ? 13 : 7;  // 152 / 156 / 140 KiB
  static constexpr int SB = 4096 * (1 + MT);
};

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// at most Q * later DMAs of this wave may still be in flight
template <int Q>
__device__ __forceinline__ void wait_later(int later) {
#define GRAM_WL(n) \
  case n: wait_vmcnt<(Q * n <= 63 ? Q * n : 63)>(); break;
  switch (later) {
    GRAM_WL(1) GRAM_WL(2) GRAM_WL(3) GRAM_WL(4) GRAM_WL(5) GRAM_WL(6) GRAM_WL(7) GRAM_WL(8) GRAM_WL(9) GRAM_WL(10)
    GRAM_WL(11) GRAM_WL(12) GRAM_WL(13) GRAM_WL(14) GRAM_WL(15) GRAM_WL(16) GRAM_WL(17)
    default: wait_vmcnt<0>(); break;
  }
#undef GRAM_WL
}

// 1/rms of row m of A for the streaming kernel: the partials' loads go out together (row_rscale's loop waits for each), the sums
// are row_rscale's -- pairs of 64-column blocks in order; a block of the quarter layout is (q0 + q1) + (q2 + q3)
__device__ __forceinline__ float row_rscale_stream(const EpiArgs& ep, int m, float& sum, float& mn) {
  sum = 0.f;
  mn = 0.f;
  if (ep.ss_nblk == 0) return ep.ss_in[m];  // already 1/rms
  float s = 0.f, lo = INFINITY;
  const int npair = ep.ss_nblk / 2;
  if (ep.ss_quarter) {
    const float4* p = reinterpret_cast<const float4*>(ep.ss_in + (size_t)m * ep.ss_nblk * 4);
    for (int i0 = 0; i0 < npair; i0 += 4) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = p[min(2 * i0 + u, ep.ss_nblk - 1)];
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (i0 + u < npair) {
          const float pa = (v[2 * u].x + v[2 * u].y) + (v[2 * u].z + v[2 * u].w);
          const float pb = (v[2 * u + 1].x + v[2 * u + 1].y) + (v[2 * u + 1].z + v[2 * u + 1].w);
          s += pa + pb;
          lo = fminf(lo, fminf(pa, pb));
        }
    }
  } else {
    const float2* p = reinterpret_cast<const float2*>(ep.ss_in + (size_t)m * ep.ss_nblk);
    for (int i0 = 0; i0 < npair; i0 += 8) {
      float2 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = p[min(i0 + u, npair - 1)];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (i0 + u < npair) {
          s += v[u].x + v[u].y;
          lo = fminf(lo, fminf(v[u].x, v[u].y));
        }
    }
  }
  sum = s;
  mn = lo;
  return row_rs(s, ep.inv_d, ep.eps);
}

template <int EPI, int MT, bool X3>
__global__ __launch_bounds__(256) void gemm_stream_kernel(const p16* __restrict__ A, const p16* __restrict__ W, int M, int N,
                                                          int K, int lda, int G, EpiArgs ep) {
  static_assert(EPI == GRAM_EPI_BF16 || EPI == GRAM_EPI_BF16_RELU || EPI == GRAM_EPI_F32_ADD, "stream kernel epilogues");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NS = StreamCfg<MT>::NS, SB = StreamCfg<MT>::SB, Q = 1 + MT;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, g = lane >> 4;
  // workgroup -> (n-tile, row group).  Workgroups are dealt to the 8 XCDs round-robin: the G row groups of an n-tile are 8 ids
  // apart, i.e. on the same XCD, so its W rows are pulled into one L2
  const int b = blockIdx.x;
  const int ntile = (b / (8 * G)) * 8 + (b & 7), rg = (b >> 3) % G;
  const int n0 = ntile * 16, m0 = rg * (16 * MT);
  const bool consumer = wave < MT && m0 + wave * 16 < M;

  // DMA lane map: this wave fills rows 4*wave .. +3 of every tile; slot (lane & 15) of row dr receives chunk slot ^ dr
  const int dr = wave * 4 + (lane >> 4);
  const int dch = (lane & 15) ^ dr;
  const uint32_t w_off = (uint32_t)dr * (uint32_t)K * 2u + (uint32_t)dch * 16u;
  uint32_t a_off[MT];
#pragma unroll
  for (int j = 0; j < MT; ++j) a_off[j] = (uint32_t)min(m0 + 16 * j + dr, M - 1) * (uint32_t)lda * 2u + (uint32_t)dch * 16u;
  const char* wsrc = reinterpret_cast<const char*>(W + (size_t)n0 * K);  // advances 256 B per issued stage
  const char* abase = reinterpret_cast<const char*>(A);  // advances 256 B per issued stage
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const uint32_t wave_lds = __builtin_amdgcn_readfirstlane(lds0 + wave * 1024);
  const int nst = K >> 7;  // stages of 128 physical reduction columns (X3: both pieces of two 32-column blocks of K)
  int i_slot = 0;          // ring slot of the next stage to issue
  auto issue = [&]() {
    const uint32_t dst = wave_lds + (uint32_t)i_slot * SB;
    dma16_asm(dst, w_off, wsrc);
#pragma unroll
    for (int j = 0; j < MT; ++j) dma16_asm(dst + 4096 * (1 + j), a_off[j], abase);
    wsrc += 256;
    abase += 256;
    i_slot = i_slot + 1 == NS ? 0 : i_slot + 1;
  };
  // fp32-residual epilogue: this lane's four residual values (and its row's factor of the 16-bit copy) are requested HERE, by LDS-DMA
  // into a patch behind the ring, as the wave's OLDEST vector-memory operations -- every counted wait below covers them -- instead of by
  // a load in the epilogue, where a one-user decode step (a chain of ~600 such launches) paid an exposed HBM round trip per launch
  constexpr int RES_OFF = NS * SB;  // [MT][64 lanes x 16 B] residuals, then [MT][64 x 4 B] row factors
  if constexpr (EPI == GRAM_EPI_F32_ADD) {
    if (consumer) {
      const int mr = min(m0 + wave * 16 + r16, M - 1);
      dma16_asm(lds0 + RES_OFF + wave * 1024, (uint32_t)mr * (uint32_t)ep.ldc * 4u + (uint32_t)(n0 + 4 * g) * 4u, reinterpret_cast<const char*>(ep.C));
      if (ep.xs_in && ep.xb_out) dma4_asm(lds0 + RES_OFF + MT * 1024 + wave * 256, (uint32_t)mr * 4u, reinterpret_cast<const char*>(ep.xs_in));
    }
  }
  const int npro = min(NS - 1, nst);
  for (int st = 0; st < npro; ++st) issue();

  float rs = ep.out_scale;
  if constexpr (EPI != GRAM_EPI_F32_ADD) {
    if (consumer && ep.ss_in) {
      const int mr = min(m0 + wave * 16 + r16, M - 1);
      float sum, mn;
      const float r = row_rscale_stream(ep, mr, sum, mn);
      rs = (ep.xs_in && ep.ss_nblk != 0 ? r / ep.xs_in[mr] : r) * ep.out_scale;  // (a power of two: exact)
      if (ep.xs_out && ep.ss_nblk != 0 && ntile == 0 && g == 0 && m0 + wave * 16 + r16 < M) ep.xs_out[mr] = row_xscale(sum, mn);
    }
    asm volatile("" : "+v"(rs));  // (the partials' loads are waited for here, behind the first DMAs, not in the epilogue)
  }

  // fragment reads: lane (r16, g) takes chunk 4*kb + g of tile row r16
  int foff[4];
#pragma unroll
  for (int kb = 0; kb < 4; ++kb) foff[kb] = r16 * 256 + (((kb * 4 + g) ^ r16) << 4);
  f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
  p16x8 fw0[4], fa0[4], fw1[4], fa1[4];  // fragments of the stage being read and of the one being multiplied (one stage behind)
  int c_slot = 0;
  auto read_frags = [&](p16x8 (&fw)[4], p16x8 (&fa)[4]) {
    const char* sw = smem + c_slot * SB;
    const char* sa = sw + 4096 * (1 + wave);
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      fw[kb] = *reinterpret_cast<const p16x8*>(sw + foff[kb]);
      fa[kb] = *reinterpret_cast<const p16x8*>(sa + foff[kb]);
    }
    c_slot = c_slot + 1 == NS ? 0 : c_slot + 1;
  };
  auto mfmas = [&](const p16x8 (&fw)[4], const p16x8 (&fa)[4]) {
    if constexpr (X3) {
#pragma unroll
      for (int kb = 0; kb < 4; kb += 2) {  // k-blocks (kb, kb + 1) = the two pieces of one block of K: a0*w1, a1*w0, a0*w0
        acc = mfma16(fw[kb + 1], fa[kb], acc);
        acc = mfma16(fw[kb], fa[kb + 1], acc);
        acc = mfma16(fw[kb], fa[kb], acc);
      }
    } else {
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) acc = mfma16(fw[kb], fa[kb], acc);
    }
  };
  // Stage st: wait for this wave's part of it, barrier (everyone's part has landed; stage st - 1 has been read: the barrier's
  // lgkmcnt(0)), read its fragments, refill the slot of stage st - 1, multiply stage st - 1 while the reads are in flight.
  auto step = [&](int st, bool steady, p16x8 (&cw)[4], p16x8 (&ca)[4], const p16x8 (&pw)[4], const p16x8 (&pa)[4]) {
    if (steady) wait_vmcnt<(Q * (NS - 2) <= 63 ? Q * (NS - 2) : 63)>();
    else wait_later<Q>(nst - 1 - st);
    gram_sync();
    if (consumer) read_frags(cw, ca);
    if (st + NS - 1 < nst) issue();
    if (consumer && st > 0) mfmas(pw, pa);
  };
  const int n_steady = nst - (NS - 1) > 0 ? nst - (NS - 1) : 0;  // stages with NS - 2 later stages in flight behind them
  int st = 0;
  for (; st + 1 < n_steady; st += 2) {
    step(st, true, fw0, fa0, fw1, fa1);
    step(st + 1, true, fw1, fa1, fw0, fa0);
  }
  for (; st < nst; ++st) {
    const bool steady = st < n_steady;
    if (st & 1) step(st, steady, fw1, fa1, fw0, fa0);
    else step(st, steady, fw0, fa0, fw1, fa1);
  }
  if (!consumer) return;
  if ((nst - 1) & 1) mfmas(fw1, fa1);
  else mfmas(fw0, fa0);
  const int m = m0 + wave * 16 + r16, n = n0 + 4 * g;
  const bool row_ok = m < M;
  if constexpr (EPI == GRAM_EPI_F32_ADD) {
    float ssq = 0.f;
    if (row_ok) {
      f32x4* pc = reinterpret_cast<f32x4*>(reinterpret_cast<float*>(ep.C) + (size_t)m * ep.ldc + n);
      f32x4 val = acc * ep.out_scale;
      val += *reinterpret_cast<const f32x4*>(smem + RES_OFF + wave * 1024 + lane * 16);  // (landed: the loop's last wait is vmcnt(0))
      *pc = val;
      if (ep.xb_out) {
        f32x4 rem = ep.xs_in ? val * *reinterpret_cast<const float*>(smem + RES_OFF + MT * 1024 + wave * 256 + lane * 4) : val;
        p16* xrow = ep.xb_out + (size_t)m * ep.ldc * ep.split;
        for (int p = 0; p < ep.split; ++p) {
          const uint2 pk = pack_bf16x4(rem);
          *reinterpret_cast<uint2*>(xrow + (ep.split == 2 ? inter_off(n, p) : n)) = pk;
          rem -= unpack_bf16x4(pk);
        }
        ssq = sumsq4(val);
      }
    }
    if (ep.ss_out) {  // (wave-uniform) the first two butterfly steps of the 64-column epilogues: the 4 lanes of a row in this n-tile
      ssq += __shfl_xor(ssq, 16, 64);
      ssq += __shfl_xor(ssq, 32, 64);
      if (g == 0 && row_ok) ep.ss_out[(size_t)m * ep.ss_out_nblk + ntile] = ssq;
    }
  } else {
    f32x4 v = acc * rs;
    if constexpr (EPI == GRAM_EPI_BF16_RELU) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
    }
    for (int p = 0; p < ep.split; ++p) {
      const uint2 pk = pack_bf16x4(v);
      if (row_ok)
        *reinterpret_cast<uint2*>(reinterpret_cast<p16*>(ep.C) + (size_t)m * ep.ldc + (ep.c_inter ? inter_off(n, p) : p * ep.c_pstride + n)) = pk;
      if (p + 1 < ep.split) v -= unpack_bf16x4(pk);
    }
  }
}

constexpr int kStreamMaxMLimit = GRAM_STREAM_MAX_M_LIMIT;  // (the callers size their quarter-partial buffers for this many rows)
int stream_max_m() {  // rows up to which the streaming kernel is used; 0 = off (GRAM_GEMM_STREAM_MAXM: A/B hook)
  static const int v = [] {
    const char* e = getenv("GRAM_GEMM_STREAM_MAXM");
    const int x = e ? atoi(e) : 512;  // measured: above ~500 rows the deep-ring tiles win (tests/bench_small_batch.py)
    return x < 0 ? 0 : x > kStreamMaxMLimit ? kStreamMaxMLimit : x;
  }();
  return v;
}
bool stream_enabled() { return stream_max_m() > 0; }

template <int EPI, int MT, bool X3>
int launch_stream_mt(const void* A, const void* W, int M, int N, int K, int lda, EpiArgs ep, hipStream_t st) {
  // (+ the fp32-residual epilogue's prefetch patch: 1 KiB of residuals and 256 B of row factors per consumer wave)
  constexpr int smem = StreamCfg<MT>::NS * StreamCfg<MT>::SB + (EPI == GRAM_EPI_F32_ADD ? MT * 1280 : 0);
  static_assert(smem <= 160 * 1024, "ring + residual patch must fit the LDS");
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_stream_kernel<EPI, MT, X3>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  const int G = (M + 16 * MT - 1) / (16 * MT);
  hipLaunchKernelGGL((gemm_stream_kernel<EPI, MT, X3>), dim3((N / 16) * G), dim3(256), smem, st, (const p16*)A, (const p16*)W, M, N, K, lda, G, ep);
  GRAM_CHECK_LAUNCH();
  return 0;
}

// rows per workgroup: as few as keep the launch within one workgroup per CU (more workgroups = more W in flight)
template <int EPI, bool X3>
int launch_stream(const void* A, const void* W, int M, int N, int K, int lda, EpiArgs ep, hipStream_t st) {
  if (M < 1 || M > kStreamMaxMLimit || N % 128 || K % 128) return GRAM_E_ARG;
  const int nt = N / 16;
  if (nt * ((M + 15) / 16) <= 256) return launch_stream_mt<EPI, 1, X3>(A, W, M, N, K, lda, ep, st);
  if (nt * ((M + 31) / 32) <= 256) return launch_stream_mt<EPI, 2, X3>(A, W, M, N, K, lda, ep, st);
  return launch_stream_mt<EPI, 4, X3>(A, W, M, N, K, lda, ep, st);
}

// ---------------------------------------------------------------------------------------------
template <int EPI, int WM, int NST, int TNW, bool X3>
__global__ __launch_bounds__(WM * 128, (NST >= 3 ? 1 : TNW == 8 ? 2 : (NST == 1 ? (X3 ? 3 : 4) : 2))) void gemm_dma_kernel(
    const p16* __restrict__ A, const p16* __restrict__ W, int M, int N, int K, int lda, EpiArgs ep) {
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
  const p16* a_src[AG];
  const p16* w_src[WG];
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
  if constexpr (NST >= 3) {
    // Few workgroups (at most one per CU, gridDim <= 256): nothing else is resident to hide a single stage's latency, so the
    // workgroup's LDS is a ring of NST stages filled NST - 1 k-tiles ahead (asm DMA + counted vmcnt as in gemm_stream_kernel;
    // one barrier per k-tile publishes the landed stage and frees the one just consumed).  Same MFMA order, same epilogues.
    constexpr int Q = AG + WG;
    static_assert(Q * (NST - 2) <= 63, "counted vmcnt");
    const char* abase = reinterpret_cast<const char*>(A + (size_t)m0 * lda);
    const char* wbase = reinterpret_cast<const char*>(W + (size_t)n0 * K);
    uint32_t a_off[AG], w_off[WG];
#pragma unroll
    for (int i = 0; i < AG; ++i) {
      const int row = (wave * AG + i) * 8 + (lane >> 3);
      a_off[i] = (uint32_t)(min(m0 + row, M - 1) - m0) * (uint32_t)lda * 2u + (uint32_t)(((lane & 7) ^ ((row >> 1) & 7)) * 16);
    }
#pragma unroll
    for (int i = 0; i < WG; ++i) {
      const int row = (wave * WG + i) * 8 + (lane >> 3);
      w_off[i] = (uint32_t)row * (uint32_t)K * 2u + (uint32_t)(((lane & 7) ^ ((row >> 1) & 7)) * 16);
    }
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const int uwave = __builtin_amdgcn_readfirstlane(wave);
    int i_slot = 0, i_kt = 0;  // cursor of the next k-tile to issue
    auto issue = [&]() {
      const uint32_t sa = lds0 + (uint32_t)i_slot * STAGE, sw = sa + A_BYTES;
      const char* asrc = abase + (size_t)i_kt * (BK * 2);
      const char* wsrc = wbase + (size_t)i_kt * (BK * 2);
#pragma unroll
      for (int i = 0; i < AG; ++i) dma16_asm(sa + (uwave * AG + i) * 1024, a_off[i], asrc);
#pragma unroll
      for (int i = 0; i < WG; ++i) dma16_asm(sw + (uwave * WG + i) * 1024, w_off[i], wsrc);
      ++i_kt;
      i_slot = i_slot + 1 == NST ? 0 : i_slot + 1;
    };
    const int npro = min(NST - 1, nkt);
    for (int kt = 0; kt < npro; ++kt) issue();
    load_row_scales(ep, m0 + wm * 64, r16, M, rs4, n0 == 0 && wn == 0);  // (waited for behind the first DMAs)
    int c_slot = 0;
    for (int kt = 0; kt < nkt; ++kt) {
      wait_later<Q>(min(NST - 2, nkt - 1 - kt));
      gram_sync();
      if (kt + NST - 1 < nkt) issue();
      compute_tile<TNW, X3>(smem + c_slot * STAGE, smem + c_slot * STAGE + A_BYTES, wm, wn, r16, g, acc);
      c_slot = c_slot + 1 == NST ? 0 : c_slot + 1;
    }
    gram_sync();  // (the row-contiguous epilogue re-uses the ring as its patches)
  } else {
  load_row_scales(ep, m0 + wm * 64, r16, M, rs4, n0 == 0 && wn == 0);
  if constexpr (NST == 1) {
    for (int kt = 0; kt < nkt; ++kt) {
      dma(kt, 0);
      gram_sync();  // hipcc drains the DMA (vmcnt(0)) ahead of the barrier
      compute_tile<TNW, X3>(smem, smem + A_BYTES, wm, wn, r16, g, acc);
      gram_sync();
    }
  } else {
    dma(0, 0);
    gram_sync();
    for (int kt = 0; kt < nkt; ++kt) {
      const int st = kt & 1;
      if (kt + 1 < nkt) dma(kt + 1, st ^ 1);
      compute_tile<TNW, X3>(smem + st * STAGE, smem + st * STAGE + A_BYTES, wm, wn, r16, g, acc);
      gram_sync();  // drains DMA(kt+1) and fences the reads of stage st
    }
  }
  }
  if constexpr ((EPI == GRAM_EPI_F32_ADD || EPI == GRAM_EPI_F32) && TNW == 4 && NST != 2) {
    if (ep.nt & 4) {  // (the last barrier of the k-loop has passed: the stage is free for the 4-KiB patches)
      epilogue_rows64<EPI>(acc, smem + wave * 4096, m0, n0, wm, wn, lane, M, ep);
      return;
    }
  }
  epilogue<EPI, TNW>(acc, m0, n0, wm, wn, r16, g, M, ep, rs4);
}

// ---------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------
// V_PP: persistent 256x256 "ping-pong" kernel.  8 waves = 2 groups (wr = wave >> 2) x 4 column slices
// (wc = wave & 3); a wave owns 128 rows x 64 columns = 4 quadrants of 64 x 32.  A k-tile (BK = 64) is
// computed in 4 phases, one quadrant each (16 MFMAs); a phase is a LOAD slot (ds_reads of the operand
// sub-tiles the quadrant needs + this wave's share of one 16-KiB half-tile DMA) and an MFMA slot, each
// ended by a workgroup barrier.  Group 1 runs ONE barrier behind group 0, so on every SIMD (which hosts
// one wave of each group) the MFMA slot of one wave coincides with the load slot of the other: the
// matrix pipe never waits for a DMA issue or an LDS read of its own wave.
//
// LDS: 8 half-tile buffers of 16 KiB = 2 (k-tile parity) x 4 types, in the order a k-tile consumes them:
//   H0 = A rows of the m0 quadrants (rows wr*128 + 0..63),  H1 = W rows of the n0 quadrants (cols wc*64 + 0..31),
//   H2 = W rows of the n1 quadrants (cols wc*64 + 32..63),  H3 = A rows of the m1 quadrants (rows wr*128 + 64..127).
// Phase p of stream k-tile kk:   reads                      issues the DMA of
//   p0  quadrant (m0,n0)         H0 (8 frags) + H1 (4)      (kk+1, H3)
//   p1  quadrant (m0,n1)         H2 (4)                     (kk+2, H0)
//   p2  quadrant (m1,n1)         H3 (8)                     (kk+2, H1)
//   p3  quadrant (m1,n0)         --                         (kk+2, H2)
// Every buffer is re-filled the phase after its only reading phase and needed again 6-7 phases later, so
// the counted wait after each issue is s_waitcnt vmcnt(10): all but the 5 newest half-tiles (2 DMA
// instructions per wave each) have landed, which is exactly what the NEXT phase reads.  The k-tiles of
// all the tiles a workgroup walks form one stream, so the pipeline never drains between tiles.
// The KV-bank GEMM runs on the ping-pong kernel as TWO launches over disjoint n-tiles: K tiles (PP_KV_K: normal MFMA operand
// order, rows through LDS) and V^T tiles (PP_KV_V: operands the other way round, direct 8-B stores).  One instantiation
// with both code paths spills accumulators (256 VGPRs); each half alone does not.
constexpr int PP_KV_K = 100, PP_KV_V = 101;
__device__ __forceinline__ void pp_barrier() {
  asm volatile("s_barrier" ::: "memory");
  gram_chaos_point();  // (nothing in the product build: common.h)
}

// Store the wave's output rows of m-tiles j0, j0+1 (32 rows x 64 columns) through its LDS patch as whole
// 128-B (bf16) / 256-B (fp32) row segments, 16 rows per pass.  Runs inside a LOAD slot of the ping-pong
// kernel.  All global addresses are a wave-uniform base (this wave's first row / first column of the tile, so it
// lives in SGPRs) + a 32-bit per-lane offset.  rows = number of valid rows from the wave's first row on.
// rs: this wave group's 128 row scales in LDS (folded T5LayerNorm), or nullptr.
struct PPOut {
  char* c;          // C + (m_first * ldc + n_first) * esize
  char* xb;         // xb_out likewise (bf16), or nullptr
  float* ss;        // ss_out + m_first * nblk + n_first / 64, or nullptr
  const float* xs;  // xs_in + m_first: the rows' power-of-two factors of the 16-bit copy (gram_norm_fusion_t), or nullptr
  uint32_t ldc_b;   // ldc * esize
  uint32_t ldx_b;   // ldc * 2
  uint32_t ss_nblk;
  int rows;
  int split;            // bf16 pieces written (gram_split_t)
  long c_ps_b;          // planar bf16 C: piece p is c_ps_b bytes after piece p - 1
  bool inter;           // bf16 C interleaved (c = C + (m_first * ldc + 2 * n_first) * 2, ldc_b the physical row stride); xb always is when split == 2
  bool nt;              // streaming (nt) stores for the bf16 rows (A/B hook GRAM_GEMM_NT7)
  float scale;          // EpiArgs.out_scale
  long q_ps;            // in-load-slot epilogue: bytes from piece 0 to piece 1 of a quadrant's row segment (interleaved C: 64)
  uint32_t q_step;      // ... and from the n0 quadrants' segment to the n1 quadrants' (interleaved C: 128, planar: 64)
};
template <int EPI, bool FULL>
__device__ __forceinline__ void pp_store_rows_impl(f32x4 (&acc)[4][8], int j0, char* patch, const PPOut& o, int lane_, const float* rs) {
  // opaque copy of the lane id: keeps hipcc from hoisting every store address of the tile out of the k-loop
  // (loop-invariant, 2 VGPRs each) and spilling them -- a scratch reload inside a load slot is a vmcnt(0) drain
  int lane = lane_;
  asm volatile("" : "+v"(lane));
  const int r16 = lane & 15, g = lane >> 4;
#pragma unroll
  for (int jj = 0; jj < 2; ++jj) {
    const int j = j0 + jj;
    if constexpr (EPI == GRAM_EPI_BF16 || EPI == GRAM_EPI_BF16_RELU) {
      // 16 rows x 64 cols bf16: patch[16][128 B], chunk c (16 B) at c ^ (row & 7); one pass per bf16 piece of the output
      const float sc = (rs ? rs[j * 16 + r16] : 1.f) * o.scale;
      f32x4 v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        v[i] = acc[i][j] * sc;
        if constexpr (EPI == GRAM_EPI_BF16_RELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[i][e] = fmaxf(v[i][e], 0.f);
        }
      }
      uint2 pcs[2][4];  // the values' pieces (both at once in the two-piece mode: split2x4)
      if (o.split == 2) {
#pragma unroll
        for (int i = 0; i < 4; ++i) split2x4(v[i], pcs[0][i], pcs[1][i]);
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) pcs[0][i] = pcs[1][i] = pack_bf16x4(v[i]);
      }
      for (int pc = 0; pc < o.split; ++pc) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int chunk = (i * 2 + (g >> 1)) ^ (r16 & 7);
          *reinterpret_cast<uint2*>(patch + r16 * 128 + chunk * 16 + (g & 1) * 8) = pc == 0 ? pcs[0][i] : pcs[1][i];
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          const int row = it * 8 + (lane >> 3), c = lane & 7;
          const uint4 val = *reinterpret_cast<const uint4*>(patch + row * 128 + ((c ^ (row & 7)) * 16));
          const int mr = j * 16 + row;
          // planar: the piece's 128-B row segment; interleaved: its two 64-B halves, side by side with the other piece's
          const uint32_t coff = o.inter ? (uint32_t)((c >> 2) * 128 + pc * 64 + (c & 3) * 16) : (uint32_t)(c * 16);
          if ((FULL || mr < o.rows) && (!(GRAM_PP_ABL & 16) || o.rows < -12345))  // (ablation bit 16: everything but the global stores)
            store16(o.c + (o.inter ? 0 : pc * o.c_ps_b) + ((uint32_t)mr * o.ldc_b + coff), val, o.nt);
        }
        __builtin_amdgcn_wave_barrier();
      }
    } else {
      // 16 rows x 64 cols fp32: patch[16][256 B], chunk c (16 B) at c ^ row
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int chunk = (i * 4 + g) ^ r16;
        *reinterpret_cast<f32x4*>(patch + r16 * 256 + chunk * 16) = acc[i][j];
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int row = it * 4 + (lane >> 4), c = lane & 15;
        f32x4 val = *reinterpret_cast<const f32x4*>(patch + row * 256 + ((c ^ row) * 16)) * o.scale;
        const int mr = j * 16 + row;
        float ssq = 0.f;
        if (FULL || mr < o.rows) {
          f32x4* pc = reinterpret_cast<f32x4*>(o.c + ((uint32_t)mr * o.ldc_b + c * 16));
          if constexpr (EPI == GRAM_EPI_F32_ADD) val += *pc;
          *pc = val;
          if constexpr (EPI == GRAM_EPI_F32_ADD) {
            if (o.xb) {
              *reinterpret_cast<uint2*>(o.xb + ((uint32_t)mr * o.ldx_b + c * 8)) = pack_bf16x4(o.xs ? val * o.xs[mr] : val);
              ssq = sumsq4(val);
            }
          }
        }
        if constexpr (EPI == GRAM_EPI_F32_ADD) {
          if (o.ss) {  // the 16 lanes of a row cover exactly one 64-column block
            ssq += __shfl_xor(ssq, 1, 64);
            ssq += __shfl_xor(ssq, 2, 64);
            ssq += __shfl_xor(ssq, 4, 64);
            ssq += __shfl_xor(ssq, 8, 64);
            if (c == 0 && (FULL || mr < o.rows)) o.ss[(uint32_t)mr * o.ss_nblk] = ssq;
          }
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
}

// Tile-end fp32 epilogue of the ping-pong kernel: the wave's 128 x 64 outputs in two halves of 64 rows.  The 16 residual
// loads of a half (64 registers -- fragments and address registers of the k-loop are dead here) go out together, so a
// tile exposes two HBM round trips instead of one per 16-row pass; the passes then run through the 4-KiB patch.
template <int EPI>
__device__ __forceinline__ void pp_store_tile_f32(f32x4 (&acc)[4][8], char* patch, const PPOut& o, int lane_) {
  int lane = lane_;
  asm volatile("" : "+v"(lane));
  const int r16 = lane & 15, g = lane >> 4;
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    f32x4 res[16];
    // row factors of the 16-bit copy: lane l holds the factor of the half's row l (ONE coalesced load with the residuals; a pass
    // fetches its row's by a cross-lane read -- sixteen registers of them per lane spill, the kernel sits at 256 VGPRs)
    float xsl = 1.f;
    if constexpr (EPI == GRAM_EPI_F32_ADD) {
#pragma unroll
      for (int q = 0; q < 16; ++q) res[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (o.rows > 0 && o.xs) xsl = o.xs[min(half * 64 + lane, o.rows - 1)];
      if (o.rows > 0) {  // (wave-uniform; a wave whose first row is past M has nothing to load: o.c points past the matrix)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int mr = min(half * 64 + q * 4 + (lane >> 4), o.rows - 1);  // rows past M inside the block: clamped, never stored
#if GRAM_PP_RES_NT
          res[q] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(o.c + ((uint32_t)mr * o.ldc_b + (lane & 15) * 16)));
#else
          res[q] = *reinterpret_cast<const f32x4*>(o.c + ((uint32_t)mr * o.ldc_b + (lane & 15) * 16));
#endif
        }
      }
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int j = half * 4 + p;
      // 16 rows x 64 cols fp32: patch[16][256 B], chunk c (16 B) at c ^ row
#pragma unroll
      for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(patch + r16 * 256 + (((i * 4 + g) ^ r16) * 16)) = acc[i][j];
      __builtin_amdgcn_wave_barrier();
      float ssq4[4];
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int row = it * 4 + (lane >> 4), c = lane & 15;
        f32x4 val = *reinterpret_cast<const f32x4*>(patch + row * 256 + ((c ^ row) * 16)) * o.scale;
        const int mr = j * 16 + row;
        float ssq = 0.f;
        // (fetched by ALL lanes, outside the M-tail predicate below: a cross-lane read from a lane that the predicate switched off returns garbage)
        const float xs_row = __shfl(xsl, p * 16 + it * 4 + (lane >> 4), 64);
        if (mr < o.rows && (!(GRAM_PP_ABL & 16) || o.rows < -12345)) {
          f32x4* pc = reinterpret_cast<f32x4*>(o.c + ((uint32_t)mr * o.ldc_b + c * 16));
          if constexpr (EPI == GRAM_EPI_F32_ADD) val += res[p * 4 + it];
          if (!(GRAM_PP_ABL & 32) || o.rows < -12345) *pc = val;  // (ablation bit 32: the epilogue without its fp32 store -- the
          //                                            bytes a residual stream kept as its two pieces only would move: DESIGN.md 9.1)
          if constexpr (EPI == GRAM_EPI_F32_ADD) {
            if (o.xb) {
              const f32x4 xv = val * xs_row;  // the 16-bit copy carries the row's power-of-two factor
              if (o.split == 2) {
                // interleaved copy (lanes 0..7 hold block 0 of the wave's 64 columns, lanes 8..15 block 1; a block = piece 0's 64 B, then
                // piece 1's): lanes c and c ^ 1 swap one piece each, so that the even lane stores 16 B of piece 0 (columns 4c .. 4c + 7)
                // and the odd lane 16 B of piece 1 -- one 16-B store per lane instead of two 8-B ones (the tile-end store tail is bound by
                // its instruction count: an 8-B-per-lane store costs as much as a 16-B one)
                uint2 p0, p1;
                split2x4(xv, p0, p1);
                const bool odd = c & 1;
                const uint2 send = odd ? p0 : p1;
                uint2 recv;
                recv.x = (uint32_t)__builtin_amdgcn_mov_dpp((int)send.x, 0xB1, 0xF, 0xF, true);  // quad_perm [1,0,3,2]: lane ^ 1
                recv.y = (uint32_t)__builtin_amdgcn_mov_dpp((int)send.y, 0xB1, 0xF, 0xF, true);
                const uint4 out = odd ? make_uint4(recv.x, recv.y, p1.x, p1.y) : make_uint4(p0.x, p0.y, recv.x, recv.y);
                const uint32_t xoff = (uint32_t)((c >> 3) * 128 + (odd ? 64 : 0) + ((c & 6) * 8));
                *reinterpret_cast<uint4*>(o.xb + ((uint32_t)mr * o.ldx_b + xoff)) = out;
              } else {
                *reinterpret_cast<uint2*>(o.xb + ((uint32_t)mr * o.ldx_b + (uint32_t)(c * 8))) = pack_bf16x4(xv);
              }
              ssq = sumsq4(val);
            }
          }
        }
        ssq4[it] = ssq;
      }
      if constexpr (EPI == GRAM_EPI_F32_ADD) {
        if (o.ss) {  // the 16 lanes of a row cover exactly one 64-column block; same butterfly as the other kernels
#pragma unroll
          for (int sh = 1; sh < 16; sh <<= 1)
#pragma unroll
            for (int it = 0; it < 4; ++it) ssq4[it] += __shfl_xor(ssq4[it], sh, 64);
          // after the butterfly every lane of a row holds the row's sum: lane c < 4 stores the partial of pass c's row -- one store
          // instruction with 16 active lanes instead of four with 4 each
          const int cc = lane & 15;
          const float mine = cc == 0 ? ssq4[0] : cc == 1 ? ssq4[1] : cc == 2 ? ssq4[2] : ssq4[3];
          const int mr = j * 16 + cc * 4 + (lane >> 4);
          if (cc < 4 && mr < o.rows) o.ss[(uint32_t)mr * o.ss_nblk] = mine;
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
}

// full: all 128 rows of the wave exist (no per-row predicates: the stores of a pass issue back to back and
// their number is exact, which the counted DMA waits of the following slots rely on)
template <int EPI>
__device__ __forceinline__ void pp_store_rows(f32x4 (&acc)[4][8], int j0, char* patch, const PPOut& o, int lane, const float* rs) {
  if (o.rows >= 128) pp_store_rows_impl<EPI, true>(acc, j0, patch, o, lane, rs);
  else pp_store_rows_impl<EPI, false>(acc, j0, patch, o, lane, rs);
}

template <int EPI, bool X3 = false>
__global__ __launch_bounds__(512, 1) void gemm_pp_kernel(const p16* __restrict__ A, const p16* __restrict__ W, int M, int N,
                                                         int K, int lda, EpiArgs ep, int ntiles, int stagger) {
  constexpr int TB = 256, HT = 16384;
  constexpr bool F32OUT = EPI == GRAM_EPI_F32 || EPI == GRAM_EPI_F32_ADD;
  constexpr bool KV = EPI == PP_KV_K || EPI == PP_KV_V;  // n-tiles of the K blocks only / of the V blocks only
  constexpr bool TRALL = EPI == PP_KV_V;
  // epilogue patches: fp32 outputs 8 x 4 KiB (one per wave, used in load slots); bf16 outputs 4 x 4 KiB, shared by
  // waves w and w+4 -- the two groups use them in alternate time slots (each inside its own MFMA slot)
  constexpr int PATCH = 4096;
  constexpr bool LSE = EPI == GRAM_EPI_F32_LSE;  // lm_head in sparse mode: only the (max, sum exp) partials leave the kernel
  // X3 = the two-piece kernel: interleaved operands (a k-tile's 128-B rows hold piece 0 | piece 1 of one 32-column block of K),
  // 24 MFMAs per quadrant slot instead of 16 (a0*w1, a1*w0, a0*w0 from the SAME fragment registers).
  // TEND: the epilogue runs at the end of the tile, both wave groups in step: fp32 outputs (a read-modify-write needs its residual
  // loads out of the DMA queue's way) and every X3 output (two pieces per value: a store count the in-slot jobs' counted waits do
  // not cover; with 1.5x the MFMAs per k-tile the tile-end placement costs less than it does in the plain kernel, where in-slot
  // bf16 stores measured 1 105 vs 1 035 TFLOP/s on the encoder QKV shape)
  // INSL: the two-piece 16-bit outputs (QKV, FFN-in) leave from INSIDE the pipeline, in the LOAD slots around the tile boundary -- see
  // "in-load-slot epilogue" below.  (GRAM_PP_INSL=0: A/B build with the tile-end epilogue.)
  constexpr bool INSL = GRAM_PP_INSL && X3 && (EPI == GRAM_EPI_BF16 || EPI == GRAM_EPI_BF16_RELU);
  constexpr bool TEND = (EPI == GRAM_EPI_F32 || EPI == GRAM_EPI_F32_ADD) || LSE || (X3 && !INSL);
  constexpr int PATCHB = INSL ? 6144 : 4096;   // bytes of a patch shared by waves w and w + 4 (INSL: three 2-KiB passes in flight)
  constexpr int RS_OFF = 8 * HT + 4 * PATCHB;  // bf16 epilogues: 2 x 1 KiB of row scales behind the patches
  extern __shared__ __attribute__((aligned(16))) char smem[];  // 8 half-tile buffers + epilogue patches (+ row scales)
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // tell hipcc it is wave-uniform: everything derived stays in SGPRs
  const int wr = wave >> 2, wc = wave & 3;
  const int r16 = lane & 15, g = lane >> 4;
  const int ntn = KV ? N / TB / 2 : (N + TB - 1) / TB;  // KV: every other block of inner/256 n-tiles; LSE: N may end in half a tile
  auto nt_of = [&](int ntl) {  // launch-local n-tile -> n-tile of the GEMM
    if constexpr (KV) {
      const int it = ep.inner >> 8, blk = (int)udiv_magic((uint32_t)ntl, (uint32_t)it, ep.mg_it);
      return blk * 2 * it + (TRALL ? it : 0) + (ntl - blk * it);
    } else {
      return ntl;
    }
  };
  const int G = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, local = bid >> 3, q = G >> 3, rr = G & 7;
  const int slot = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + local;
  if (slot >= ntiles) return;
  const int nkt = K / BK;  // even, >= 4 (checked by the launcher)
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const uint32_t wave_lds = __builtin_amdgcn_readfirstlane(lds0 + wave * 2048);  // this wave's 2 pieces (16 rows) of a half-tile
  const char* const Ab = reinterpret_cast<const char*>(A);
  const char* const Wb = reinterpret_cast<const char*>(W);
  char* const patch = smem + 8 * HT + (F32OUT ? wave * PATCH : TEND ? wave * 2048 : (wave & 3) * PATCHB);
  const bool has_rs = !F32OUT && ep.ss_in != nullptr;  // ss_nblk == 0 (1/rms per row), checked by the launcher
  if constexpr (INSL) {  // its passes always multiply by a row scale from the RS area: all ones for a GEMM without (published by the prologue's barrier)
    if (!has_rs) reinterpret_cast<float*>(smem + RS_OFF)[tid] = 1.f;
  }

  // tile index -> (m-tile, n-tile column).  gm <= 1: n fastest.  gm > 1: groups of gm m-tiles, m fastest inside a group, so
  // that the 32 tiles an XCD works on at a time cover gm m-tiles x 32/gm n-tiles (fewer distinct A + W panels per round).
  const int gm = (stagger >> 16) & 0xff;
  const bool clk_on = (stagger >> 30) & 1;  // (gram_prof_pp_clock_enable)
#ifdef GRAM_CHAOS
  const int entry_delay = (stagger >> 24) & 0x3f;  // (test hook of the chaos build, see the prologue; the product build has no such code)
#endif
  stagger &= 0xffff;
  auto decode = [&](int tile, int& mt, int& nr) {
    if (gm <= 1) {
      mt = tile / ntn;
      nr = tile - mt * ntn;
    } else {
      const int per = gm * ntn, grp = tile / per, first = grp * gm;
      const int gs = min(gm, (M + TB - 1) / TB - first);
      const int rem = tile - grp * per;
      nr = rem / gs;
      mt = first + (rem - nr * gs);
    }
  };
  // DMA cursor = stream k-tile kk+2 (tile, kt) + per-lane byte offsets of this wave's 2 pieces of each half-tile type
  // (relative to the tile's first A row / W row, whose addresses c_A / c_W are wave-uniform and 64-bit)
  int c_tile = slot, c_kt = 0;
  // W offsets never change (H2 = H1 + 32 rows goes into the uniform base); A offsets change only for the M-tail tile
  const char *c_A, *c_W;
  uint32_t offA[2][2], offW[2], offW2[LSE ? 2 : 1];  // LSE: W rows clamped per tile (the last n-tile may be half empty)
  if constexpr (!LSE) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int r = (wave * 2 + p) * 8 + (lane >> 3);  // row of the half-tile image
      const int chunk = (lane & 7) ^ ((r >> 1) & 7);
      offW[p] = ((uint32_t)((r >> 5) * 64 + (r & 31)) * (uint32_t)K + chunk * 8) * 2u;
    }
  }
  auto set_offsets = [&]() {
    int mt, nr;
    decode(c_tile, mt, nr);
    const int nt = nt_of(nr);
    c_A = Ab + (size_t)mt * TB * lda * 2;
    c_W = Wb + (size_t)nt * TB * K * 2;
    const int mleft = M - 1 - mt * TB;  // last valid row, tile-relative
    int ln = lane;
    asm volatile("" : "+v"(ln));  // opaque: recompute the lane constants here (once per tile) instead of keeping them live
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int r = (wave * 2 + p) * 8 + (ln >> 3);
      const int chunk = (ln & 7) ^ ((r >> 1) & 7);
      const int arow = (r >> 6) * 128 + (r & 63);  // + 64 for H3
      offA[0][p] = ((uint32_t)min(arow, mleft) * (uint32_t)lda + chunk * 8) * 2u;
      offA[1][p] = ((uint32_t)min(arow + 64, mleft) * (uint32_t)lda + chunk * 8) * 2u;
      if constexpr (LSE) {
        const int wleft = N - 1 - nt * TB, wrow = (r >> 5) * 64 + (r & 31);
        offW[p] = ((uint32_t)min(wrow, wleft) * (uint32_t)K + chunk * 8) * 2u;
        offW2[p] = ((uint32_t)min(wrow + 32, wleft) * (uint32_t)K + chunk * 8) * 2u;
      }
    }
  };
  auto advance = [&]() {
    if (++c_kt == nkt) {
      c_kt = 0;
      if (c_tile + G < ntiles) c_tile += G;  // past the end of the stream: fetch this tile again (harmless, keeps the counts uniform)
      set_offsets();
    }
  };
  bool abl_loop = false;  // (ablation builds only: the main loop has begun)
  auto issue = [&](int t, int par) {  // this wave's 2 DMA instructions of half-tile (cursor, type t) -> buffer (par, t)
    if ((GRAM_PP_ABL & 2) && abl_loop) return;
    {
      const uint32_t dst = wave_lds + (par * 4 + t) * HT;
      if (t == 0 || t == 3) {
        const char* base = c_A + c_kt * (BK * 2);
        dma16_asm(dst, offA[t == 3][0], base);
        dma16_asm(dst + 1024, offA[t == 3][1], base);
      } else {
        if constexpr (LSE) {
          const char* base = c_W + c_kt * (BK * 2);
          dma16_asm(dst, t == 2 ? offW2[0] : offW[0], base);
          dma16_asm(dst + 1024, t == 2 ? offW2[1] : offW[1], base);
        } else {
          const char* base = c_W + c_kt * (BK * 2) + (t == 2 ? (size_t)32 * K * 2 : 0);
          dma16_asm(dst, offW[0], base);
          dma16_asm(dst + 1024, offW[1], base);
        }
      }
    }
  };

  f32x4 acc[4][8];
  p16x8 fa[2][4], fw[2][2][2];
  // fragment addresses: one per-lane base per operand and k-step; the m-/n-tile (16 rows = 2 KiB: the swizzle term
  // (row >> 1) & 7 does not change) and the buffer are immediate offsets
  const char* const a_base[2] = {smem + swz(wr * 64 + r16, g), smem + swz(wr * 64 + r16, 4 + g)};
  const char* const w_base[2] = {smem + swz(wc * 32 + r16, g), smem + swz(wc * 32 + r16, 4 + g)};
  auto read_a = [&](int par, int mq) {
    if ((GRAM_PP_ABL & 4) && abl_loop) return;
    {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          fa[ks][j] = *reinterpret_cast<const p16x8*>(a_base[ks] + (par * 4 + (mq ? 3 : 0)) * HT + j * 2048);
    }
  };
  auto read_w = [&](int par, int nq) {
    if ((GRAM_PP_ABL & 4) && abl_loop) return;
    {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < 2; ++i)
          fw[nq][ks][i] = *reinterpret_cast<const p16x8*>(w_base[ks] + (par * 4 + 1 + nq) * HT + i * 2048);
    }
  };
  // TR (KV bank, V^T half): operands the other way round, so a lane ends up with 4 consecutive ROWS (bank
  // positions s) of one column d -- the V^T layout -- instead of 4 consecutive columns of one row
  constexpr int NMQ = X3 ? 24 : 16;  // MFMAs of a quadrant slot
  auto mfma_q = [&](auto TRc, int mq, int nq, int n) {
    // plain: k-steps 0, 1 of the k-tile.  X3: the "k-steps" are the two PIECES of one block of K (fa[p], fw[nq][p]) and the slot
    // issues a0*w1, a1*w0, a0*w0 -- every accumulator meets its three products 8 MFMAs apart
    const int pr = n >> 3, i = (n >> 2) & 1, j = n & 3;
    const int ka = X3 ? (pr == 1 ? 1 : 0) : pr, kw = X3 ? (pr == 0 ? 1 : 0) : pr;
    f32x4& c = acc[nq * 2 + i][mq * 4 + j];
    if constexpr (decltype(TRc)::value) c = mfma16(fa[ka][j], fw[nq][kw][i], c);
    else c = mfma16(fw[nq][kw][i], fa[ka][j], c);
  };
  auto mma = [&](int mq, int nq, auto TRc) {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int n = 0; n < NMQ; ++n) mfma_q(TRc, mq, nq, n);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    pp_barrier();
  };
  // bf16 epilogues: an MFMA slot that also stores the 32 finished rows of m-tiles J0, J0+1 (not the quadrant being
  // computed).  Scale/convert/ds_write pieces fill the issue gaps of the first 8 MFMAs, the patch reads those of
  // the next 4, and the 4 row-contiguous global stores go out behind the last MFMA, so the slot is barely longer.
  auto mma_st = [&](int mq, int nq, auto TRc, auto J0c, const PPOut& o, const float* rs, auto FULLc) {
    constexpr bool FULL = decltype(FULLc)::value;  // all 128 rows of the wave exist: stores without a predicate (no branches in the slot)
    constexpr int J0 = decltype(J0c)::value;
    int ln = lane;
    asm volatile("" : "+v"(ln));  // opaque: no hoisting of the addresses below out of the k-loop (they would be spilled)
    const int lr = ln & 15, lg = ln >> 4;
    float sc[2] = {o.scale, o.scale};
    if (rs) {
      sc[0] = rs[J0 * 16 + lr] * o.scale;
      sc[1] = rs[J0 * 16 + 16 + lr] * o.scale;
    }
    uint4 pva;
    // register-lean addressing: one per-lane base each for the patch writes, the patch reads and the stores
    //   write (m-tile jj, n-tile pi): row = jj*16 + lr, 16-B chunk (pi*2 + (lg>>1)) ^ (lr & 7), half lg & 1
    //     = wbase + jj*2048 + ((pi*32) ^ wx)   with wbase, wx per lane
    const int wx = ((lr & 7) >> 1) * 32;
    const char* const wbase = patch + lr * 128 + (((lg >> 1) ^ (lr & 1)) * 16) + (lg & 1) * 8;
    //   read it: row = it*8 + (ln>>3), chunk (ln & 7) ^ (row & 7)  = gbase + it*1024
    const char* const gbase = patch + (ln >> 3) * 128 + (((ln & 7) ^ ((ln >> 3) & 7)) * 16);
    //   store it: row J0*16 + it*8 + (ln>>3), 16 B at column (ln & 7)*8
    const int mr0 = J0 * 16 + (ln >> 3);
    const uint32_t voff0 = (uint32_t)mr0 * o.ldc_b + (ln & 7) * 16;
    auto get = [&](int it) { return *reinterpret_cast<const uint4*>(gbase + it * 1024); };
    auto put = [&](int it, const uint4& v) {  // -> its 128-B segment of C
      if (FULL || mr0 + it * 8 < o.rows) {
        uint4* dst = reinterpret_cast<uint4*>(o.c + (voff0 + (uint32_t)(it * 8) * o.ldc_b));
        // streaming (nt) store: the output is far larger than L2 and is next read by another kernel; keeping it
        // out of L2 leaves the A panels this XCD re-reads there (measured +1.6 ... 4 %)
        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
        __builtin_nontemporal_store(__builtin_bit_cast(u32x4, v), reinterpret_cast<u32x4*>(dst));
      }
    };
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int n = 0; n < 16; ++n) {
      mfma_q(TRc, mq, nq, n);
      if (n < 8 && (n & 1)) {
#pragma unroll
        for (int pc = n - 1; pc <= n; ++pc) {  // piece pc: m-tile J0 + (pc >> 2), n-tile pc & 3
          const int jj = pc >> 2, pi = pc & 3;
          f32x4 v = acc[pi][J0 + jj] * sc[jj];
          if constexpr (EPI == GRAM_EPI_BF16_RELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
          }
          *reinterpret_cast<uint2*>(const_cast<char*>(wbase) + jj * 2048 + ((pi * 32) ^ wx)) = pack_bf16x4(v);
        }
      }
      if (n == 7) __builtin_amdgcn_wave_barrier();
      if (n == 8) pva = get(0);
      if (n == 10) {
        put(0, pva);
        pva = get(1);
      }
      if (n == 12) {
        put(1, pva);
        pva = get(2);
      }
      if (n == 14) {
        put(2, pva);
        pva = get(3);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_s_setprio(0);
    put(3, pva);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the shared patch is idle before the other group's slot
    __builtin_amdgcn_sched_barrier(0);
    pp_barrier();
  };
  // ---- KV bank (gram_kv_bank_t): where the 32-row block of m-tiles J0, J0+1 of tile (tm0, tn0) goes.  A wave's 64
  // columns are one head of one layer's K or V projection; a 32-row block lies inside one passage (32 | L), so its
  // user b and first bank position s0 are wave-uniform.
  struct KVLoc {
    char* kb;  // bank_k + ((head * S + s0) * 64): the block's 32 x 128 B of K
    char* vb;  // bank_vt + (head * S/32 + s0/32) * 2048: the block's [64 d][32 positions] of V^T (4 KiB, contiguous)
    int mblk;  // first row of the block in A
  };
  auto kv_locate = [&](int tm0, int tn0, int J0) {
    KVLoc q;
    q.mblk = tm0 + wr * 128 + J0 * 16;
    // mblk, pL, S are multiples of 32: (mblk/32) / (x/32), dividend < 2^25 and divisor <= 128 -> multiply-high is exact
    // (error term < divisor, dividend * error < 2^32); flat < 2^27 with pN <= 22 likewise; n-tiles < 2^16 by inner/256
    int b_, s0;
    if (ep.pmap) {
      const int pp = (int)udiv_magic((uint32_t)q.mblk >> 5, (uint32_t)ep.pL >> 5, ep.mg_pL32), l = q.mblk - pp * ep.pL, flat = ep.pmap[pp];
      b_ = (int)udiv_magic((uint32_t)flat, (uint32_t)ep.pN, ep.mg_pN);
      s0 = (flat - b_ * ep.pN) * ep.pL + l;
    } else {
      b_ = (int)udiv_magic((uint32_t)q.mblk >> 5, (uint32_t)ep.S >> 5, ep.mg_S32);
      s0 = q.mblk - b_ * ep.S;
    }
    const int n = tn0 + wc * 64, lw = (int)udiv_magic((uint32_t)n >> 8, (uint32_t)ep.inner >> 8, ep.mg_it);
    const size_t head = ((size_t)(lw >> 1) * ep.B + b_) * ep.H + ((n - lw * ep.inner) >> 6);
    q.kb = reinterpret_cast<char*>(ep.bank_k + (head * ep.S + s0) * 64);
    q.vb = reinterpret_cast<char*>(ep.bank_vt + (head * (size_t)(ep.S >> 5) + (size_t)(s0 >> 5)) * 2048);
    return q;
  };
  // one (m-tile jj, n-tile pi) fragment of the block straight from registers (8 B per lane)
  // (wave-uniform base + 32-bit per-lane offset)
  auto kv_direct = [&](auto TRc, const KVLoc& q, int jj, int pi, const f32x4& v, int lr, int lg) {
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    const u32x2 pk = __builtin_bit_cast(u32x2, pack_bf16x4(v * ep.out_scale));
    if constexpr (decltype(TRc)::value) {  // lane: column d = pi*16 + lr, rows s0 + jj*16 + 4*lg .. +3
      const uint32_t off = ((uint32_t)(pi * 16 + lr) * 32u + jj * 16 + 4 * lg) * 2u;
      if (q.mblk + jj * 16 + 4 * lg < M) __builtin_nontemporal_store(pk, reinterpret_cast<u32x2*>(q.vb + off));
    } else {  // lane: row s0 + jj*16 + lr, columns d = pi*16 + 4*lg .. +3
      const uint32_t off = ((uint32_t)(jj * 16 + lr) * 64u + pi * 16 + 4 * lg) * 2u;
      if (q.mblk + jj * 16 + lr < M) __builtin_nontemporal_store(pk, reinterpret_cast<u32x2*>(q.kb + off));
    }
  };
  // MFMA slot + the V^T block of m-tiles J0, J0+1 (computed with swapped operands): 8 direct 8-B stores per lane,
  // 16 d-rows x 32 B each (2-byte scatter in the plain kernel), issued in the gaps of the first 8 MFMAs
  auto mma_vt = [&](int mq, int nq, auto TRc, auto J0c, const KVLoc& q) {
    // (the block exists entirely: M is a multiple of 32 and its first row is < M)
    constexpr int J0 = decltype(J0c)::value;
    int ln = lane;
    asm volatile("" : "+v"(ln));
    const int lr = ln & 15, lg = ln >> 4;
    // V^T block = 64 columns d x 32 bank positions: through the shared 4-KiB patch as [64 d][64 B] so that a lane stores
    // 16 B (8 positions) and a d-row 64 B, instead of 8-B pieces of 32-B segments straight from the registers.
    //   write (m-tile jj, n-tile pi): row d = pi*16 + lr, bytes jj*32 + lg*8;   16-B chunk c of row d sits at c ^ ((d >> 2) & 3)
    const char* const wbase = patch + lr * 64 + (lg & 1) * 8;
    const int wsw = (lr >> 2) & 3;  // (d >> 2) & 3 = (lr >> 2) & 3: pi*16 does not change it
    //   read it: row d = it*16 + (ln >> 2), chunk ln & 3
    const int rd = ln >> 2, rc = ln & 3;
    const char* const gbase = patch + rd * 64 + ((rc ^ ((rd >> 2) & 3)) * 16);
    const uint32_t voff0 = ((uint32_t)rd * 32u + rc * 8) * 2u;  // (the block is [64 d][64 B] in the bank too)
    uint4 pv;
    auto get = [&](int it) { return *reinterpret_cast<const uint4*>(gbase + it * 1024); };  // 16 rows = 1 KiB, swizzle term unchanged
    auto put = [&](int it, const uint4& v) {
      typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
      __builtin_nontemporal_store(__builtin_bit_cast(u32x4, v),
                                  reinterpret_cast<u32x4*>(q.vb + (voff0 + (uint32_t)(it * 1024))));
    };
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int n = 0; n < 16; ++n) {
      mfma_q(TRc, mq, nq, n);
      if (n < 8 && (n & 1)) {
#pragma unroll
        for (int pc = n - 1; pc <= n; ++pc) {
          const int jj = pc >> 2, pi = pc & 3;
          const int chunk = (jj * 2 + (lg >> 1)) ^ wsw;
          *reinterpret_cast<uint2*>(const_cast<char*>(wbase) + pi * 1024 + chunk * 16) = pack_bf16x4(acc[pi][J0 + jj] * ep.out_scale);
        }
      }
      if (n == 7) __builtin_amdgcn_wave_barrier();
      if (n == 8) pv = get(0);
      if (n == 10) {
        put(0, pv);
        pv = get(1);
      }
      if (n == 12) {
        put(1, pv);
        pv = get(2);
      }
      if (n == 14) {
        put(2, pv);
        pv = get(3);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_s_setprio(0);
    put(3, pv);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the shared patch is idle before the other group's slot
    __builtin_amdgcn_sched_barrier(0);
    pp_barrier();
  };
  // K half through the shared patch (whole 128-B rows): the block is 32 x 128 B contiguous in the bank
  auto kv_out_k = [&](const KVLoc& q, int tm0, int J0) {
    PPOut o{};
    o.c = q.kb - (size_t)J0 * 16 * 128;
    o.ldc_b = 128;
    o.rows = M - (tm0 + wr * 128);
    o.split = 1;
    o.inter = false;
    o.nt = true;  // (the bank is next read by another kernel, much later)
    o.scale = ep.out_scale;
    return o;
  };
  // extra = number of epilogue stores this wave has issued since the DMA that must have landed (a lower bound is
  // always safe: vmcnt counts loads, stores and DMA together, in issue order)
  auto end_load_slot = [&](int extra = 0) {
    // all but this wave's 6 newest half-tiles (2 DMA each) have landed
    if (extra == 0) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if (extra == 4) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if (extra == 8) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
    else if (extra == 12) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(28)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this slot's LDS reads are complete before their buffer can be re-filled
    __builtin_amdgcn_sched_barrier(0);
    pp_barrier();
  };
  // ---- in-load-slot epilogue (INSL).  The tile-end epilogue of a two-piece 16-bit output costs 13-14 % of these GEMMs (both wave
  // groups stop, 32 KiB per wave through a 2-KiB patch, then 256 KiB per CU through a vector-store path that moves ~32 B/clk; nothing of
  // it overlaps an MFMA: profiles/r03d, r03f).  Here a quadrant's results leave in "passes" -- one m-tile (16 rows) x the quadrant's 32
  // columns x both pieces: scale / ReLU / split, 4 ds_write_b64 into a 2-KiB patch laid out [16 rows][piece 0: 64 B | piece 1: 64 B]
  // (for an interleaved output that IS the row's 128 B), 2 ds_read_b128, 2 row-contiguous 1-KiB stores -- placed in the LOAD slots
  // between the quadrant's last MFMA slot of this tile and its first of the next one, where the wave otherwise waits for its SIMD
  // partner's MFMA slot.  With the quadrant order of the last (odd) k-tile 01 00 10 11 and of the next tile's first (even) 00 01 11 10:
  //   load slot before   | O:00   | O:10    | O:11        || E:00        | E:01        | E:11    | E:10
  //   passes (quadrant.j)| 01.0 1 | 01.2 00.0| 01.3 00.1 2 || 00.3 11.0 10.0| 11.1 2 10.1 | 11.3 10.2| 10.3
  // A quadrant is zeroed behind its last pass.  The patch is the 4 KiB waves w and w + 4 share (they are never in a load slot at the
  // same time).  Stores count in vmcnt with the DMAs, in issue order: the counted wait of a load slot adds the stores of the last six
  // load slots (exact for a wave whose 128 rows all exist; a partial wave counts none, which only makes its waits stricter).
  // Extra stores in flight at the end of each load slot (sum over the last six load slots, the slot's own included), by position:
  //   pairs of MODE 1 (first pair behind a tile of a full wave):  even k-tile 20 26 30 26, odd k-tile 22 16 10 4
  //     (previous pair's last three slots 4 + 4 + 6, then 6 6 4 0: the single pass of the fourth slot is left out of the count)
  //   pairs of MODE 2 (last pair of a tile of a full wave):       odd k-tile 0 4 8 14
  auto end_load_slot_st = [&](bool full, auto Nc) {  // N = that sum for a full wave (a partial wave counts none of its stores)
    constexpr int N = decltype(Nc)::value;
    static_assert(N >= 0 && 12 + N <= 63, "vmcnt is a 6-bit count");
    if constexpr ((GRAM_PP_ABL & 1) != 0) full = false;
    if (full) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(12 + N) : "memory");
    else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    pp_barrier();
  };
  // up to three passes as ONE straight-line group (no branch between them: a basic-block boundary makes hipcc drain lgkmcnt): all splits
  // and patch writes, then all patch reads, then the stores as their data arrives.  P = mq * 8 + nq * 4 + j-in-quadrant, -1 = none;
  // pass u uses the 2 KiB at pt + 2048 u.  rs always points at row scales (all 1.0 when the GEMM has none: see the RS area's fill).
  auto passes = [&](const PPOut& o, const float* rs, char* pt, auto P0c, auto P1c, auto P2c) {
    using U0 = std::integral_constant<int, 0>;
    using U1 = std::integral_constant<int, 1>;
    using U2 = std::integral_constant<int, 2>;
    if constexpr ((GRAM_PP_ABL & 1) != 0) {  // ablation: keep the accumulators live, store (almost) never
      auto keep = [&](auto Pc) {
        constexpr int P = decltype(Pc)::value;
        if constexpr (P >= 0) {
          constexpr int nq = (P >> 2) & 1, j = (P >> 3) * 4 + (P & 3);
          const f32x4 t4 = acc[nq * 2][j] + acc[nq * 2 + 1][j];
          const float t = (t4[0] + t4[1]) + (t4[2] + t4[3]);
          if (t == 12345.678f) reinterpret_cast<float*>(ep.C)[lane] = t;
        }
      };
      keep(P0c);
      keep(P1c);
      keep(P2c);
      return;
    }
    // The SIMD partner is in its MFMA slot at priority 1: at priority 0 this wave's ~45 vector instructions per pass would only issue
    // once the partner has run out of MFMAs (MI355X_MICROARCH.md, "Two waves per SIMD", item 2), i.e. behind the slot instead of beside it
    __builtin_amdgcn_s_setprio(GRAM_PP_PASS_PRIO);
    int ln = lane;
    asm volatile("" : "+v"(ln));  // opaque: the lane constants are recomputed here, not kept live (or spilled) across the k-loop
    const int lr = ln & 15, lg = ln >> 4;
    // patch row lr: logical 16-B chunk piece * 4 + n-tile * 2 + (lg >> 1) at chunk ^ (lr & 7), 8-B half lg & 1
    char* const wb = pt + lr * 128 + (lg & 1) * 8;
    const int sw = lr & 7, gh = lg >> 1;
    auto split_write = [&](auto Uc, auto Pc) {
      constexpr int u = decltype(Uc)::value, P = decltype(Pc)::value;
      if constexpr (P >= 0) {
        constexpr int nq = (P >> 2) & 1, j = (P >> 3) * 4 + (P & 3);
        const float sc = rs[j * 16 + lr] * o.scale;
        f32x4 v0 = acc[nq * 2][j] * sc, v1 = acc[nq * 2 + 1][j] * sc;
        // (the ROUNDED products are what is split into pieces, as in every other kernel: keep hipcc from contracting acc * sc - hi into an fma)
        asm volatile("" : "+v"(v0), "+v"(v1));
        if constexpr (EPI == GRAM_EPI_BF16_RELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v0[e] = fmaxf(v0[e], 0.f);
            v1[e] = fmaxf(v1[e], 0.f);
          }
        }
        uint2 h0, h1, l0, l1;
        split2x4(v0, h0, l0);
        split2x4(v1, h1, l1);
        *reinterpret_cast<uint2*>(wb + u * 2048 + (((0 + gh) ^ sw) << 4)) = h0;
        *reinterpret_cast<uint2*>(wb + u * 2048 + (((2 + gh) ^ sw) << 4)) = h1;
        *reinterpret_cast<uint2*>(wb + u * 2048 + (((4 + gh) ^ sw) << 4)) = l0;
        *reinterpret_cast<uint2*>(wb + u * 2048 + (((6 + gh) ^ sw) << 4)) = l1;
      }
    };
    split_write(U0{}, P0c);
    split_write(U1{}, P1c);
    split_write(U2{}, P2c);
    __builtin_amdgcn_wave_barrier();
    const int row = ln >> 3, c = ln & 7;
    const char* const gb = pt + row * 128 + ((c ^ (row & 7)) << 4);
    uint4 val[3][2];
    auto read_back = [&](auto Uc, auto Pc) {
      constexpr int u = decltype(Uc)::value;
      if constexpr (decltype(Pc)::value >= 0) {
        val[u][0] = *reinterpret_cast<const uint4*>(gb + u * 2048);
        val[u][1] = *reinterpret_cast<const uint4*>(gb + u * 2048 + 1024);
      }
    };
    read_back(U0{}, P0c);
    read_back(U1{}, P1c);
    read_back(U2{}, P2c);
    // a quadrant's 32 columns of a row: chunks 0..3 = piece 0 (64 B), 4..7 = piece 1, `o.q_ps` bytes further (interleaved C: 64, i.e. the
    // row's 128 contiguous bytes; planar C: the piece stride); the n1 quadrants `o.q_step` bytes further (128 / 64)
    char* const cb = o.c + ((c & 3) * 16 + (c >> 2) * o.q_ps);
    auto store_rows = [&](auto Uc, auto Pc) {
      constexpr int u = decltype(Uc)::value, P = decltype(Pc)::value;
      if constexpr (P >= 0) {
        constexpr int nq = (P >> 2) & 1, j = (P >> 3) * 4 + (P & 3);
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          const int mr = j * 16 + it * 8 + row;
          if (mr < o.rows && (!(GRAM_PP_ABL & 16) || o.rows < -12345))  // (ablation bit 16: everything but the global stores)
            *reinterpret_cast<uint4*>(cb + ((uint32_t)mr * o.ldc_b + (uint32_t)nq * o.q_step)) = val[u][it];
        }
      }
    };
    store_rows(U0{}, P0c);
    store_rows(U1{}, P1c);
    store_rows(U2{}, P2c);
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_setprio(0);
  };
  auto zero_q = [&](int mq, int nq) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[nq * 2 + i][mq * 4 + j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  };
  auto zero_half = [&](int mq) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][mq * 4 + j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  };

  // Consumption order of the half-tile stream (one per phase; nf = parity of the k-tile, ns = 1 - nf):
  //   ... W_nf(kk) [read in the previous k-tile's p3] | A_m0(kk) p0 | W_ns(kk) p1 | A_m1(kk) p2 | W_nf(kk+1) p3 | ...
  // Phase P issues the half-tile consumed in phase P + 7 (all of stream k-tile kk+2) into the buffer whose only
  // reading phase was P - 1, and then waits for all but its 6 newest half-tiles.
  if (stagger > 0) {  // de-synchronise the store phases: CU by CU inside an XCD, or (bit 15) XCD by XCD with each XCD's CUs in step
    const int ph = (stagger & 0x8000) ? xcd : (slot & 7);
    stagger &= 0x7fff;
    const int units = (ph * stagger * nkt) >> 3;
    for (int i = 0; i < units; ++i) __builtin_amdgcn_s_sleep(8);  // 512 cycles
  }
  // ---- prologue: everything phases -8 .. -1 would have issued = all of stream k-tiles 0 and 1
  set_offsets();
  issue(1, 0);
  issue(0, 0);
  issue(2, 0);
  issue(3, 0);
  advance();
  issue(2, 1);
  issue(0, 1);
  issue(1, 1);
  issue(3, 1);
  advance();
  asm volatile("s_waitcnt vmcnt(12)" ::: "memory");  // W_n0(0) and A_m0(0) have landed
  pp_barrier();
#ifdef GRAM_CHAOS
  // test hook (make CHAOS=1 only): group 1 late by entry_delay x 512 cycles -- the timing that exposed the missing barrier below, made
  // deterministic (gram_debug_set_gemm_variant(2000 + n); tests/test_gpu_kernels.py::test_race_screens_on_the_chaos_build)
  if (wr == 1) for (int i = 0; i < entry_delay; ++i) __builtin_amdgcn_s_sleep(8);
#endif
  read_w(0, 0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  // Every wave holds its W_n0(0) fragments before anyone goes on: group 0's first load slot re-fills that buffer (issue(1, 0) below
  // = W_n0 of stream k-tile 2).  In the steady state the buffer's readers are a barrier ahead of its re-fill; here, at the entry, group
  // 0 used to run from its own read straight into that issue, and a wave of group 1 that came late to ITS read found k-tile 2's rows
  // there: the first tile of a workgroup wrong in group 1's n0 columns by one k-tile's contribution.  The window is the DMA's latency
  // (microseconds) against a handful of ds_reads, so it never showed -- until a build whose entry block carried a compiler-made
  // `s_waitcnt vmcnt(0)` between the barrier and the read (a spill reload; round 4's "clock stamps behind a flag" build) spread the
  // waves of a workgroup over exactly that window (profiles/r04o_*).
  pp_barrier();
  if (wr == 1) pp_barrier();  // group 1 runs one barrier behind group 0

  int tile = slot;
  int m0, n0, tpar = 0;
  constexpr int ESZ = F32OUT ? 4 : 2;
  int pm0 = 0, pn0 = 0;
  // output addressing of tile (tm0, tn0), built where it is used (a handful of SALU ops) rather than kept live
  auto make_out = [&](int tm0, int tn0) {  // this wave's first row (tm0 + wr*128) and first column (tn0 + wc*64)
    PPOut o;
    const size_t mf = (size_t)tm0 + wr * 128;
    const int nf = tn0 + wc * 64;
    o.inter = !F32OUT && ep.c_inter != 0;
    o.c = reinterpret_cast<char*>(ep.C) + (mf * ep.ldc + (o.inter ? 2 * nf : nf)) * ESZ;
    // (split == 2: xb is interleaved, [M][2 * ldc])
    o.xb = ep.xb_out ? reinterpret_cast<char*>(ep.xb_out) + (mf * ep.ldc + nf) * 2 * ep.split : nullptr;
    o.ss = ep.ss_out ? ep.ss_out + mf * ep.ss_out_nblk + (nf >> 6) : nullptr;
    o.xs = ep.xs_in && ep.xb_out ? ep.xs_in + mf : nullptr;
    o.ldc_b = ep.ldc * ESZ;
    o.ldx_b = ep.ldc * 2 * ep.split;
    o.ss_nblk = ep.ss_out_nblk;
    o.rows = M - (int)mf;
    o.split = ep.split;
    o.c_ps_b = ep.c_pstride * 2;
    o.nt = (ep.nt & 8) != 0;
    o.scale = ep.out_scale;
    o.q_ps = o.inter ? 64 : o.c_ps_b;
    o.q_step = o.inter ? 128u : 64u;
    return o;
  };
  // one MFMA slot of quadrant (mq, nq), optionally with the epilogue job "store m-tiles J0, J0+1 of tile (tm0, tn0)"
  auto run_slot = [&](int mq, int nq, bool job, auto J0c, int tm0, int tn0, const float* rs) {
    constexpr int J0 = decltype(J0c)::value;
    if constexpr (KV) {
      using TR = std::integral_constant<bool, TRALL>;
      job = job && tm0 + wr * 128 + J0 * 16 < M;  // (a block past the M tail: nothing to store, and no pmap entry)
      if (!job) {
        mma(mq, nq, TR{});
      } else {
        const KVLoc q = kv_locate(tm0, tn0, J0);
        if constexpr (TRALL) mma_vt(mq, nq, TR{}, J0c, q);
        else mma_st(mq, nq, TR{}, J0c, kv_out_k(q, tm0, J0), nullptr, std::false_type{});
      }
    } else {
      if (job) {
        const PPOut o = make_out(tm0, tn0);
        if (o.rows >= 128) mma_st(mq, nq, std::false_type{}, J0c, o, rs, std::true_type{});
        else mma_st(mq, nq, std::false_type{}, J0c, o, rs, std::false_type{});
      }
      else mma(mq, nq, std::false_type{});
    }
  };
  bool pending = false;  // the m1 half of the previous tile is still in the accumulators
  zero_half(0);
  zero_half(1);
  abl_loop = true;
  // The two stamps are read unconditionally (two scalar instructions per workgroup and launch); the atomics at the end are behind
  // gram_prof_pp_clock_enable.  (A build with the reads inside `if (clk_on)` returned wrong tiles in round 4: not the stamps -- its
  // entry block differed, and that exposed the missing barrier of the prologue above.  Fixed there; with the race gone the conditional
  // form is correct but 0.4-0.5 % slower in the bench (two more SGPR spills and a scratch reload behind a vmcnt(0) in the entry block:
  // profiles/r04x_stamp_reads_ab.txt), so the two scalar reads stay unconditional.)
  const unsigned long long clk_t0 = __builtin_amdgcn_s_memtime(), clk_r0 = __builtin_amdgcn_s_memrealtime();
  if constexpr (INSL) {
    const float *rs_cur = nullptr, *rs_prev = nullptr;
    auto set_tile = [&]() {
      int mt, nr;
      decode(tile, mt, nr);
      m0 = mt * TB;
      n0 = nt_of(nr) * TB;
      rs_cur = reinterpret_cast<const float*>(smem + RS_OFF) + tpar * 256 + wr * 128;  // (all 1.0 when the GEMM has no row scales)
      rs_prev = reinterpret_cast<const float*>(smem + RS_OFF) + (tpar ^ 1) * 256 + wr * 128;
    };
    // two-piece 16-bit outputs: the passes of the in-load-slot epilogue (schedule above) around the same eight phases.  The k-tile
    // pairs of a tile come in three compile-time flavours -- the first one behind a previous tile (MODE 1: that tile's last passes),
    // plain ones (MODE 0: exactly the loop of the other kernels, no epilogue code or branches in it), the last one (MODE 2: this
    // tile's first passes) -- so the steady state is not touched.  nkt >= 4: the first and the last pair are different pairs.
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    const std::false_type NT{};
    char* const PT = patch;
    auto pair_body = [&](auto MODEc, bool first) {
      constexpr int MODE = decltype(MODEc)::value;
      // exact store counts: all 128 rows of the wave exist in the tile being stored (its output addressing is rebuilt in every slot
      // that stores -- a handful of scalar operations -- rather than kept live across the pair)
      const bool full = MODE == 1 ? M - (pm0 + wr * 128) >= 128 : MODE == 2 ? M - (m0 + wr * 128) >= 128 : false;
      // ================= even k-tile (buffers 0): quadrants 00, 01, 11, 10
      if constexpr (MODE == 1) {
        issue(1, 0);  // W_n0(kk+2)
        const PPOut o = make_out(pm0, pn0);
        passes(o, rs_prev, PT, std::integral_constant<int, 3>{}, std::integral_constant<int, 12>{}, std::integral_constant<int, 8>{});
        zero_q(0, 0);
        read_a(0, 0);
        end_load_slot_st(full, std::integral_constant<int, 20>{});
      } else {
        read_a(0, 0);
        issue(1, 0);
        end_load_slot(0);
      }
      mma(0, 0, NT);
      if constexpr (MODE == 1) {
        issue(0, 0);  // A_m0(kk+2)
        const PPOut o = make_out(pm0, pn0);
        passes(o, rs_prev, PT, std::integral_constant<int, 13>{}, std::integral_constant<int, 14>{}, std::integral_constant<int, 9>{});
        read_w(0, 1);
        end_load_slot_st(full, std::integral_constant<int, 26>{});
      } else {
        read_w(0, 1);
        issue(0, 0);
        end_load_slot(0);
      }
      mma(0, 1, NT);
      if (first && has_rs && wave < 4) {  // this tile's 256 row scales -> LDS by DMA (64 rows per wave); first read in the tile's last pair
        int ln = lane;
        asm volatile("" : "+v"(ln));
        dma4_asm(lds0 + RS_OFF + tpar * 1024 + wave * 256, (uint32_t)min(m0 + wave * 64 + ln, M - 1) * 4u,
                 reinterpret_cast<const char*>(ep.ss_in));
      }
      if constexpr (MODE == 1) {
        issue(2, 0);  // W_n1(kk+2)
        const PPOut o = make_out(pm0, pn0);
        passes(o, rs_prev, PT, std::integral_constant<int, 15>{}, std::integral_constant<int, 10>{}, std::integral_constant<int, -1>{});
        zero_q(1, 1);
        read_a(0, 1);
        end_load_slot_st(full, std::integral_constant<int, 30>{});
      } else {
        read_a(0, 1);
        issue(2, 0);
        end_load_slot(0);
      }
      mma(1, 1, NT);
      if constexpr (MODE == 1) {
        issue(3, 0);  // A_m1(kk+2)
        const PPOut o = make_out(pm0, pn0);
        passes(o, rs_prev, PT, std::integral_constant<int, 11>{}, std::integral_constant<int, -1>{}, std::integral_constant<int, -1>{});
        zero_q(1, 0);
        read_w(1, 1);  // W_n1 of the odd k-tile that follows
        end_load_slot_st(full, std::integral_constant<int, 26>{});
      } else {
        read_w(1, 1);
        issue(3, 0);
        end_load_slot(0);
      }
      mma(1, 0, NT);
      advance();
      // ================= odd k-tile (buffers 1): quadrants 01, 00, 10, 11
      read_a(1, 0);
      issue(2, 1);  // W_n1(kk+3)
      if constexpr (MODE == 1) end_load_slot_st(full, std::integral_constant<int, 22>{});
      else end_load_slot(0);
      mma(0, 1, NT);
      if constexpr (MODE == 2) {  // this tile's quadrant 01 is final
        issue(0, 1);  // A_m0(kk+3)
        const PPOut o = make_out(m0, n0);
        passes(o, rs_cur, PT, std::integral_constant<int, 4>{}, std::integral_constant<int, 5>{}, std::integral_constant<int, -1>{});
        read_w(1, 0);
        end_load_slot_st(full, std::integral_constant<int, 4>{});
      } else {
        read_w(1, 0);
        issue(0, 1);
        if constexpr (MODE == 1) end_load_slot_st(full, std::integral_constant<int, 16>{});
        else end_load_slot(0);
      }
      mma(0, 0, NT);
      if constexpr (MODE == 2) {  // ... and 00
        issue(1, 1);  // W_n0(kk+3)
        const PPOut o = make_out(m0, n0);
        passes(o, rs_cur, PT, std::integral_constant<int, 6>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, -1>{});
        read_a(1, 1);
        end_load_slot_st(full, std::integral_constant<int, 8>{});
      } else {
        read_a(1, 1);
        issue(1, 1);
        if constexpr (MODE == 1) end_load_slot_st(full, std::integral_constant<int, 10>{});
        else end_load_slot(0);
      }
      mma(1, 0, NT);
      if constexpr (MODE == 2) {
        issue(3, 1);  // A_m1(kk+3)
        const PPOut o = make_out(m0, n0);
        passes(o, rs_cur, PT, std::integral_constant<int, 7>{}, std::integral_constant<int, 1>{}, std::integral_constant<int, 2>{});
        zero_q(0, 1);
        read_w(0, 0);  // W_n0 of the even k-tile that follows
        end_load_slot_st(full, std::integral_constant<int, 14>{});
      } else {
        read_w(0, 0);
        issue(3, 1);
        if constexpr (MODE == 1) end_load_slot_st(full, std::integral_constant<int, 4>{});
        else end_load_slot(0);
      }
      mma(1, 1, NT);
      advance();
    };
    // The tile loop is rotated -- [plain pairs | last pair | next tile's first pair] -- so that no two flavours of a pair meet at a
    // join (a join of two 190-register states makes hipcc copy and spill them).
    set_tile();
    pair_body(I0{}, true);
    while (true) {
      for (int kt = 2; kt + 2 < nkt; kt += 2) pair_body(I0{}, false);
      pair_body(I2{}, false);
      const bool more = tile + G < ntiles;
      pending = true;
      pm0 = m0;
      pn0 = n0;
      tpar ^= 1;
      if (!more) break;
      tile += G;
      set_tile();
      pair_body(I1{}, true);
    }
  } else {
  while (true) {
    {
      int mt, nr;
      decode(tile, mt, nr);
      const int nt = nt_of(nr);
      m0 = mt * TB;
      n0 = nt * TB;
    }
    const float* rs_cur = has_rs ? reinterpret_cast<const float*>(smem + RS_OFF) + tpar * 256 + wr * 128 : nullptr;
    const float* rs_prev = has_rs ? reinterpret_cast<const float*>(smem + RS_OFF) + (tpar ^ 1) * 256 + wr * 128 : nullptr;
    {
    for (int kt = 0; kt < nkt; kt += 2) {
      const bool first = kt == 0, last = kt + 2 >= nkt;
      // bf16 epilogues of FULL tiles issue exactly 4 stores per store slot (the slots p2, p3 of a tile's last k-tile
      // and p0, p1 of the next tile's first): pf / lf = such stores were / are issued around this k-tile pair
      const bool pf = !TEND && !F32OUT && !KV && first && pending && M - (pm0 + wr * 128) >= 128;
      const bool lf = !TEND && !F32OUT && !KV && last && M - (m0 + wr * 128) >= 128;
      // ================= even k-tile (buffers 0): n order 0, 1
      // Epilogue of the previous tile's m1 half (m-tiles 4..7, finished by its last MFMA slot): fp32 outputs store it
      // in the load slots of p0 / p1, bf16 outputs inside the MFMA slots of p0 / p1 (which compute m0 quadrants).
      const bool st_prev = first && pending, st_cur = last;
      read_a(0, 0);
      issue(1, 0);  // W_n0(kk+2)
      end_load_slot(pf ? 8 : 0);
      run_slot(0, 0, !TEND && st_prev, std::integral_constant<int, 4>{}, pm0, pn0, rs_prev);
      read_w(0, 1);
      issue(0, 0);  // A_m0(kk+2)
      end_load_slot(pf ? 12 : 0);
      run_slot(0, 1, !TEND && st_prev, std::integral_constant<int, 6>{}, pm0, pn0, rs_prev);
      read_a(0, 1);
      if (first) {
        if constexpr (!TEND) zero_half(1);
        if (has_rs && wave < 4) {  // this tile's 256 row scales -> LDS by DMA (64 rows per wave); first read >= 6 k-tiles later
          int ln = lane;
          asm volatile("" : "+v"(ln));
          dma4_asm(lds0 + RS_OFF + tpar * 1024 + wave * 256, (uint32_t)min(m0 + wave * 64 + ln, M - 1) * 4u,
                   reinterpret_cast<const char*>(ep.ss_in));
        }
      }
      issue(2, 0);  // W_n1(kk+2)
      end_load_slot(pf ? 16 : 0);
      run_slot(1, 1, false, std::integral_constant<int, 0>{}, 0, 0, nullptr);
      read_w(1, 1);  // W_n1 of the odd k-tile that follows
      issue(3, 0);   // A_m1(kk+2)
      end_load_slot(pf ? 16 : 0);
      run_slot(1, 0, false, std::integral_constant<int, 0>{}, 0, 0, nullptr);
      advance();
      // ================= odd k-tile (buffers 1): n order 1, 0
      read_a(1, 0);
      issue(2, 1);  // W_n1(kk+3)
      end_load_slot(pf ? 16 : 0);
      run_slot(0, 1, false, std::integral_constant<int, 0>{}, 0, 0, nullptr);
      read_w(1, 0);
      issue(0, 1);  // A_m0(kk+3)
      end_load_slot(pf ? 12 : 0);
      run_slot(0, 0, false, std::integral_constant<int, 0>{}, 0, 0, nullptr);
      // this tile's m0 half (m-tiles 0..3) is final now: stored during p2 / p3, which compute the m1 quadrants
      read_a(1, 1);
      issue(1, 1);  // W_n0(kk+3)
      end_load_slot(pf ? 8 : 0);
      run_slot(1, 0, !TEND && st_cur, std::integral_constant<int, 0>{}, m0, n0, rs_cur);
      read_w(0, 0);  // W_n0 of the even k-tile that follows
      issue(3, 1);  // A_m1(kk+3)
      end_load_slot((pf ? 4 : 0) + (lf ? 4 : 0));
      run_slot(1, 1, !TEND && st_cur, std::integral_constant<int, 2>{}, m0, n0, rs_cur);
      if (last && !TEND) zero_half(0);
      advance();
    }
    }
    const bool more = tile + G < ntiles;
    if constexpr (TEND) {
      // fp32 outputs: the read-modify-write epilogue runs at the END of the tile with both wave groups in step (like the
      // 256x128-tile kernel): inside the slots its residual loads drain the in-order DMA queue four times per tile and
      // group; here once.  Group 0 waits for group 1's last MFMA slot, both store, and group 1 falls one barrier behind again.
      if (wr == 0) pp_barrier();
      if constexpr ((GRAM_PP_ABL & 1) != 0) {  // ablation: keep the accumulators live, store (almost) never
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 8; ++j) t += (acc[i][j][0] + acc[i][j][1]) + (acc[i][j][2] + acc[i][j][3]);
        if (t == 12345.678f) reinterpret_cast<float*>(ep.C ? ep.C : ep.lse_part ? (void*)ep.lse_part : (void*)ep.bank_k)[lane] = t;
      } else if constexpr (LSE) {
        // per row and 64-column block (= this wave's columns) the pair (max, sum exp(x - max)), reduced in exactly the
        // order of the 128-row kernels' epilogue, so a user's scores do not depend on which kernel its batch size selects
        const int blk = (n0 + wc * 64) >> 6;
        if (blk < ep.lse_nblk) {  // (the padding half of the last n-tile has no block)
#pragma unroll
          for (int j = 0; j < 8; ++j) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i][j] *= ep.out_scale;
            float mx = -INFINITY;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
              for (int e = 0; e < 4; ++e) mx = fmaxf(mx, acc[i][j][e]);
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            float sm = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
              for (int e = 0; e < 4; ++e) sm += __expf(acc[i][j][e] - mx);
            sm += __shfl_xor(sm, 16, 64);
            sm += __shfl_xor(sm, 32, 64);
            const int m = m0 + wr * 128 + j * 16 + r16;
            if (g == 0 && m < M) *reinterpret_cast<float2*>(ep.lse_part + ((size_t)m * ep.lse_nblk + blk) * 2) = make_float2(mx, sm);
          }
        }
      } else if constexpr (KV) {
        // bank pieces through this wave's 4-KiB patch (no row scales in this epilogue: all 32 KiB behind the half-tiles are
        // patches), so that K leaves as whole 128-B rows and V^T as 64-B d-rows instead of 8-B fragments
        char* const kvpatch = smem + 8 * HT + wave * 4096;
#pragma unroll
        for (int J0 = 0; J0 < 8; J0 += 2) {
          if (m0 + wr * 128 + J0 * 16 < M) {
            const KVLoc q = kv_locate(m0, n0, J0);
            if constexpr (!TRALL) {
              PPOut o = kv_out_k(q, m0, J0);
              o.split = ep.split;
              o.c_ps_b = ep.bank_pstride * 2;
              pp_store_rows_impl<GRAM_EPI_BF16, false>(acc, J0, kvpatch, o, lane, nullptr);
            } else {
              // V^T block = 64 columns d x 32 bank positions as [64 d][64 B] (see mma_vt): a lane stores 16 B (8 positions)
              int ln = lane;
              asm volatile("" : "+v"(ln));
              const int lr = ln & 15, lg = ln >> 4;
              const int wsw = (lr >> 2) & 3, rd = ln >> 2, rc = ln & 3;
              char* const wbase = kvpatch + lr * 64 + (lg & 1) * 8;
              const char* const gbase = kvpatch + rd * 64 + ((rc ^ ((rd >> 2) & 3)) * 16);
              const uint32_t voff0 = ((uint32_t)rd * 32u + rc * 8) * 2u;
              char* vb = q.vb;
#pragma unroll
              for (int pcs = 0; pcs < 8; ++pcs) acc[pcs & 3][J0 + (pcs >> 2)] *= ep.out_scale;
              for (int pc = 0; pc < ep.split; ++pc) {
#pragma unroll
                for (int pcs = 0; pcs < 8; ++pcs) {
                  const int jj = pcs >> 2, pi = pcs & 3;
                  f32x4& v = acc[pi][J0 + jj];
                  uint2 pk;
                  if (pc == 0) {  // both pieces at once (split2x4); the low one waits in the accumulator's first two registers
                    uint2 lo;
                    split2x4(v, pk, lo);
                    v[0] = __uint_as_float(lo.x);
                    v[1] = __uint_as_float(lo.y);
                  } else {
                    pk = make_uint2(__float_as_uint(v[0]), __float_as_uint(v[1]));
                  }
                  *reinterpret_cast<uint2*>(wbase + pi * 1024 + (((jj * 2 + (lg >> 1)) ^ wsw) * 16)) = pk;
                }
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
                  const uint4 val = *reinterpret_cast<const uint4*>(gbase + it * 1024);
                  __builtin_nontemporal_store(__builtin_bit_cast(u32x4, val),
                                              reinterpret_cast<u32x4*>(vb + (voff0 + (uint32_t)(it * 1024))));
                }
                __builtin_amdgcn_wave_barrier();
                vb += ep.bank_pstride * 2;
              }
            }
          }
        }
      } else {
      const PPOut o = make_out(m0, n0);
      if constexpr (F32OUT) {
        pp_store_tile_f32<EPI>(acc, patch, o, lane);
      } else {
        pp_store_rows<EPI>(acc, 0, patch, o, lane, rs_cur);
        pp_store_rows<EPI>(acc, 2, patch, o, lane, rs_cur);
        pp_store_rows<EPI>(acc, 4, patch, o, lane, rs_cur);
        pp_store_rows<EPI>(acc, 6, patch, o, lane, rs_cur);
      }
      }
      zero_half(0);
      zero_half(1);
      pp_barrier();
      if (more && wr == 1) pp_barrier();
    }
    pending = true;
    pm0 = m0;
    pn0 = n0;
    tpar ^= 1;
    if (!more) break;
    tile += G;
  }
  }
  if (clk_on && tid == 0) {
    atomicAdd(&g_pp_clk[0], (unsigned long long)__builtin_amdgcn_s_memtime() - clk_t0);
    atomicAdd(&g_pp_clk[1], (unsigned long long)__builtin_amdgcn_s_memrealtime() - clk_r0);
  }
  // The last tile's m1 half is still in the accumulators.  Every DMA of this workgroup must have landed before it
  // ends, and after that the half-tile buffers are dead: they serve as 8 private patches for the final stores.
  if constexpr (!TEND) {
    if (wr == 0) pp_barrier();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  pp_barrier();
  if constexpr (INSL) {  // what the in-load-slot epilogue of a next tile would have stored: 00's last m-tile, quadrants 11 and 10
    const float* rs_prev = reinterpret_cast<const float*>(smem + RS_OFF) + (tpar ^ 1) * 256 + wr * 128;
    const PPOut o = make_out(pm0, pn0);
    char* const PT = smem + wave * 6144;  // (the half-tile buffers are dead: private patches)
    passes(o, rs_prev, PT, std::integral_constant<int, 3>{}, std::integral_constant<int, 12>{}, std::integral_constant<int, 13>{});
    passes(o, rs_prev, PT, std::integral_constant<int, 14>{}, std::integral_constant<int, 15>{}, std::integral_constant<int, 8>{});
    passes(o, rs_prev, PT, std::integral_constant<int, 9>{}, std::integral_constant<int, 10>{}, std::integral_constant<int, 11>{});
  } else if constexpr (!TEND) {  // (TEND: stored at the end of every tile)
    const float* rs_prev = has_rs ? reinterpret_cast<const float*>(smem + RS_OFF) + (tpar ^ 1) * 256 + wr * 128 : nullptr;
    if constexpr (KV) {
#pragma unroll
      for (int J0 = 4; J0 < 8; J0 += 2) {
        if (pm0 + wr * 128 + J0 * 16 < M) {
          const KVLoc q = kv_locate(pm0, pn0, J0);
#pragma unroll
          for (int pc = 0; pc < 8; ++pc)
            kv_direct(std::integral_constant<bool, TRALL>{}, q, pc >> 2, pc & 3, acc[pc & 3][J0 + (pc >> 2)], r16, g);
        }
      }
    } else {
      const PPOut o = make_out(pm0, pn0);
      pp_store_rows<EPI>(acc, 4, smem + wave * 4096, o, lane, rs_prev);
      pp_store_rows<EPI>(acc, 6, smem + wave * 4096, o, lane, rs_prev);
    }
  }
}

template <int EPI, bool X3 = false>
int launch_pp(const void* A, const void* W, int M, int N, int K, int lda, EpiArgs ep, hipStream_t st) {
  if constexpr (EPI == GRAM_EPI_KV_BANK) {  // the public id: both halves, one launch each (no kernel of its own)
    const int r = launch_pp<PP_KV_K, X3>(A, W, M, N, K, lda, ep, st);
    return r ? r : launch_pp<PP_KV_V, X3>(A, W, M, N, K, lda, ep, st);
  } else {
  if ((ep.split > 1) != X3) return GRAM_E_ARG;
  constexpr int smem = 8 * 16384 + 8 * 4096;  // 160 KiB
  if ((EPI == GRAM_EPI_F32_LSE ? N % 128 : N % 256) || (K / BK) % 2 || K / BK < 4 || (ep.ss_in && ep.ss_nblk != 0)) return GRAM_E_ARG;
  if (EPI == GRAM_EPI_F32_LSE && ep.C) return GRAM_E_ARG;  // dense logits: the 128-row kernels
  if ((size_t)256 * lda * 2 >= (1ull << 31) || (size_t)256 * K * 2 >= (1ull << 31) || (size_t)256 * ep.ldc * 4 >= (1ull << 31))
    return GRAM_E_ARG;  // per-tile 32-bit offsets
  {
    if constexpr (EPI == PP_KV_K || EPI == PP_KV_V) {  // a tile inside one layer's K or V block, 32-row blocks inside one passage
      if (ep.inner % 256 || ((ep.pmap ? ep.pL : ep.S) % 32) || M % 32) return GRAM_E_ARG;
      if (ep.S > 4096 || (ep.pmap && (ep.pL > 4096 || ep.pN > 64 || (long)ep.B * ep.pN >= (1l << 26))) || M >= (1 << 30)) return GRAM_E_ARG;
      ep.mg_pL32 = magic_u32(ep.pmap ? ep.pL >> 5 : 1);
      ep.mg_S32 = magic_u32(ep.S >> 5);
      ep.mg_pN = magic_u32(ep.pmap ? ep.pN : 1);
      ep.mg_it = magic_u32(ep.inner >> 8);
    }
    const int ntiles = ((EPI == PP_KV_K || EPI == PP_KV_V) ? N / 512 : (N + 255) / 256) * ((M + 255) / 256);
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
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_pp_kernel<EPI, X3>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, smem);
      if (e != hipSuccess) return (int)e;
      attr_set = true;
    }
    // tile order (see decode() in the kernel): m-grouped where there are enough n-tiles for it to shrink an XCD's panel set
    // measured (same box, whole path): n-fastest 6 684, groups of 4: 6 752, 6: 6 774, 8: 6 762 users/s
    // split operands (K' = 3K / 6K: panels three / six times the size), same box, GEMM ms per step: groups of 2: 1 097, 3: 1 100, 4: 1 088,
    // 5: 1 098, 6: 1 097, 8: 1 091, 12: 1 101 -- 4 x 8 = the 32 tiles an XCD runs at a time
    static const int gm_set = getenv("GRAM_GEMM_GROUPM") ? atoi(getenv("GRAM_GEMM_GROUPM")) : -1;  // A/B hook (0/1: n fastest)
    const int gm_env = gm_set >= 0 ? gm_set : (ep.split > 1 ? 4 : 6);
    const int ntn_ = ntiles / ((M + 255) / 256);
    const int gm = ntn_ >= 8 && gm_env > 1 && gm_env < 256 ? gm_env : 0;
    // Tile-end epilogues (the two-piece kernel) leave every CU storing at the same moment when all workgroups run in step:
    // a start stagger spreads the bursts over the tile period (GRAM_GEMM_STAGGER, A/B hook; see gemm_pp_kernel)
    static const int nt7 = getenv("GRAM_GEMM_NT7") ? atoi(getenv("GRAM_GEMM_NT7")) : 0;  // A/B hook: streaming stores of the tile-end bf16 rows (measured: no gain)
    if (nt7) ep.nt |= 8;
    static const int stagger_env = getenv("GRAM_GEMM_STAGGER") ? atoi(getenv("GRAM_GEMM_STAGGER")) : -1;
    static const int stagger_xcd = getenv("GRAM_GEMM_STAGGER_XCD") ? atoi(getenv("GRAM_GEMM_STAGGER_XCD")) : 0;  // phase = XCD instead of slot & 7
    const int stagger = (stagger_env >= 0 ? stagger_env : g_stagger) | (stagger_xcd ? 0x8000 : 0);
    hipLaunchKernelGGL((gemm_pp_kernel<EPI, X3>), dim3(nblocks), dim3(512), smem, st, (const p16*)A, (const p16*)W, M, N, K, lda,
                       ep, ntiles, (stagger & 0xffff) | (gm << 16) | ((g_pp_entry_delay & 0x3f) << 24) | (g_pp_clk_on ? 1 << 30 : 0));
    GRAM_CHECK_LAUNCH();
    return 0;
  }
  }
}

// Measured on MI355X (tests/bench_gemm.py; TFLOP/s):
//   M >= 32768 (encoder, bank):  ping-pong 256x256 persistent kernel 1 100-1 190 > 256x128 tiles 780-815 > 128x128
//   lm_head with logits stored (N = 32128, not a multiple of 256):  256x128 783 > 128x128 614
//   M ~ 10240 (decoder):         128x128 475-832 > 256x128 418-801 > persistent 299-696 (too few tiles per CU)
int pp_min_m() {  // rows from which the persistent ping-pong kernel is considered (GRAM_GEMM_PP_MINM: A/B hook)
  static const int v = getenv("GRAM_GEMM_PP_MINM") ? atoi(getenv("GRAM_GEMM_PP_MINM")) : 32768;
  return v;
}
int pick_variant(int M, int N, int K) {
  if (g_force_variant >= 0) return g_force_variant;
  const long tiles256 = (long)((M + 255) / 256) * (N / BN);
  // persistent 256x256: needs >= ~2 tiles per CU to amortise its fill/drain; with a long K (FFN-wo, K = 4*d) it
  // already wins at 480 tiles (M = 40960, N = 768: 230 us vs 277 us for the 128x128 variant)
  // (1 536 tiles of 256 x 128 = 3 ping-pong tiles per CU; measured on the decoder's cross-q GEMM, M = 81 920, N = 768: 1 110 -> ~1 250
  // TFLOP/s executed, GEMM time per step 1 032.7 -> 1 027.5 ms: profiles/r03s_pp_mintiles_ab.txt)
  static const long pp_min_tiles = getenv("GRAM_GEMM_PP_MINTILES") ? atol(getenv("GRAM_GEMM_PP_MINTILES")) : 1536;  // A/B hook
  if (N % 256 == 0 && M >= pp_min_m() && (tiles256 >= pp_min_tiles * (long)pp_min_m() / 32768 || K >= 2048)) return V_PP;
  if (tiles256 >= 2048) return V_DMA_M256;
  // at most one workgroup per CU: nothing else is resident to hide a DMA's latency -> the deep-ring instantiations (64- or 128-row tiles)
  static const int use_ring = getenv("GRAM_GEMM_RING") ? atoi(getenv("GRAM_GEMM_RING")) : 1;  // A/B hook
  if (use_ring) {
    if ((long)((M + 63) / 64) * (N / BN) <= 256) return V_RING_M64;
    if ((long)((M + 127) / 128) * (N / BN) <= 256) return V_RING_M128;
  }
  // few 128 x 128 tiles (a batch of 4 .. ~100 users in the decoder): 64-row tiles double the workgroups that pull the weights
  static const int m64_max = getenv("GRAM_GEMM_M64_MAXTILES") ? atoi(getenv("GRAM_GEMM_M64_MAXTILES")) : 256;  // A/B hook; measured +1-2 % at B = 8 .. 256
  if ((long)((M + 127) / 128) * (N / BN) < m64_max) return V_DMA_M64;
  return V_DMA;
}

template <int EPI, int WM, int NST, bool X3, int TNW = 4>
int launch_dma(const void* A, const void* W, int M, int N, int K, int lda, EpiArgs ep, hipStream_t st) {
  constexpr int TBM = 64 * WM, TBN = 32 * TNW;
  constexpr int smem = NST * (TBM + TBN) * BK * 2;
  if (N % TBN) return GRAM_E_ARG;
  const int nblocks = (N / TBN) * ((M + TBM - 1) / TBM);
  static bool attr_set = false;
  if (!attr_set && smem > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_dma_kernel<EPI, WM, NST, TNW, X3>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  static const int rows_env = getenv("GRAM_GEMM_DMAROWS") ? atoi(getenv("GRAM_GEMM_DMAROWS")) : 1;  // A/B hook; measured +1.2 % end to end
  if (rows_env) ep.nt |= 4;
  hipLaunchKernelGGL((gemm_dma_kernel<EPI, WM, NST, TNW, X3>), dim3(nblocks), dim3(WM * 128), smem, st, (const p16*)A, (const p16*)W,
                     M, N, K, lda, ep);
  GRAM_CHECK_LAUNCH();
  return 0;
}

constexpr int V_SKINNY = 30, V_STREAM = 32;

template <int EPI, bool X3>
int launch(const void* A, const void* W, int M, int N, int K, int lda, EpiArgs ep, hipStream_t st) {
  gram_prof::Scope prof(GRAM_K_GEMM, st, (X3 ? 3.0 : 2.0) * M * N * K);  // EXECUTED MFMA flops: K = physical columns; X3: 3 products per 2 of them
  if constexpr (EPI == GRAM_EPI_BF16 || EPI == GRAM_EPI_BF16_RELU || EPI == GRAM_EPI_F32_ADD) {
    // one user (or a few): the streaming kernel.  A producer of 64-column sum-of-squares partials stays on the older skinny kernel.
    const bool fits = M <= stream_max_m() && N % 128 == 0 && K % 128 == 0 && (EPI != GRAM_EPI_F32_ADD || !ep.ss_out || ep.ss_quarter);
    if (g_force_variant == V_STREAM || (g_force_variant < 0 && stream_enabled() && fits)) return launch_stream<EPI, X3>(A, W, M, N, K, lda, ep, st);
  }
  if (ep.ss_quarter) return GRAM_E_ARG;  // (only the streaming kernel reads / writes that layout)
  if constexpr (EPI != GRAM_EPI_KV_BANK) {
    static const int use_skinny = getenv("GRAM_GEMM_SKINNY") ? atoi(getenv("GRAM_GEMM_SKINNY")) : 1;  // A/B hook
    static const int skinny_max_m = getenv("GRAM_GEMM_SKINNY_MAXM") ? atoi(getenv("GRAM_GEMM_SKINNY_MAXM")) : 64;  // A/B hook
    if (g_force_variant == V_SKINNY || (g_force_variant < 0 && use_skinny && M <= skinny_max_m && N % 64 == 0 && K % 64 == 0))
      return launch_skinny<EPI, X3>(A, W, M, N, K, lda, ep, st);
  }
  // The ping-pong kernel declines shapes it does not cover (GRAM_E_ARG); those run on the 256x128 tiles.
  // GRAM_GEMM_PP=0 (A/B hook) keeps it out altogether.
  static const int use_pp = getenv("GRAM_GEMM_PP") ? atoi(getenv("GRAM_GEMM_PP")) : 1;
  const bool pp_ok = use_pp != 0;
  // (two-piece operands: the ping-pong kernel's X3 instantiation; its results leave at the tile end)
  auto pp = [&]() { return launch_pp<EPI, X3>(A, W, M, N, K, lda, ep, st); };
  if constexpr (EPI == GRAM_EPI_F32_LSE) {  // 64-column wave tiles only: the ping-pong kernel (partials only) or the 128-row kernels
    if ((g_force_variant == V_PP || (g_force_variant < 0 && pp_ok && M >= pp_min_m())) && !ep.C) {
      const int r = pp();
      if (r != GRAM_E_ARG) return r;
    }
    const int pv = pick_variant(M, N, K);
    return pv == V_RING_M64    ? launch_dma<EPI, 1, 6, X3>(A, W, M, N, K, lda, ep, st)
           : pv == V_RING_M128 ? launch_dma<EPI, 2, 4, X3>(A, W, M, N, K, lda, ep, st)
           : pv == V_DMA_M64   ? launch_dma<EPI, 1, 1, X3>(A, W, M, N, K, lda, ep, st)
           : pv == V_DMA   ? launch_dma<EPI, 2, 1, X3>(A, W, M, N, K, lda, ep, st)
                           : launch_dma<EPI, 4, 1, X3>(A, W, M, N, K, lda, ep, st);
  } else {
  if constexpr (EPI == GRAM_EPI_F32_ADD) {  // big-M residual GEMMs: the ping-pong kernel with its tile-end epilogue, whatever N and K
    if (g_force_variant < 0 && pp_ok && M >= pp_min_m()) {
      const int r = pp();
      if (r != GRAM_E_ARG) return r;
    }
  }
  switch (pick_variant(M, N, K)) {
    case V_DMA_M64: return launch_dma<EPI, 1, 1, X3>(A, W, M, N, K, lda, ep, st);
    case V_RING_M64: return launch_dma<EPI, 1, 6, X3>(A, W, M, N, K, lda, ep, st);
    case V_RING_M128: return launch_dma<EPI, 2, 4, X3>(A, W, M, N, K, lda, ep, st);
    case V_PP:
      if (g_force_variant == V_PP || pp_ok) {
        const int r = pp();
        if (r != GRAM_E_ARG || g_force_variant == V_PP) return r;
      }
      [[fallthrough]];
    case V_DMA_M256: return launch_dma<EPI, 4, 1, X3>(A, W, M, N, K, lda, ep, st);
    default: return launch_dma<EPI, 2, 1, X3>(A, W, M, N, K, lda, ep, st);  // V_DMA: 128 x 128 tiles
  }
  }
}

}  // namespace

extern "C" int gram_prof_pp_clock_enable(int on) {
  g_pp_clk_on = on != 0;
  return 0;
}

extern "C" int gram_prof_pp_clock(double* ghz, int reset) {
  unsigned long long h[2] = {0ull, 0ull};
  if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_pp_clk), sizeof(h)) != hipSuccess) return GRAM_E_ARG;  // (synchronises: not for a timed region)
  if (ghz) *ghz = h[1] ? (double)h[0] / (double)h[1] * 0.1 : 0.0;
  if (reset) {
    const unsigned long long z[2] = {0ull, 0ull};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_pp_clk), z, sizeof(z)) != hipSuccess) return GRAM_E_ARG;
  }
  return 0;
}

extern "C" int gram_gemm_stream_max_m(void) { return g_force_variant < 0 ? stream_max_m() : 0; }

extern "C" int gram_debug_set_gemm_variant(int v) {
  if (v >= 2000) {  // 2000 + n: wave group 1 of the ping-pong kernel enters n x 512 cycles late (test hook of the chaos build, n <= 63; 2000 = off)
    g_pp_entry_delay = (v - 2000) & 0x3f;
    return 0;
  }
  if (v >= 1000) {  // 1000 + s: set the persistent kernel's start stagger instead
    g_stagger = v - 1000;
    return 0;
  }
  g_force_variant = v;
  return 0;
}

namespace {
// host side of gram_split_t
int make_split(const gram_split_t* sp, EpiArgs& ep) {
  const int pieces = sp ? sp->pieces : 1;
  if (pieces < 1 || pieces > GRAM_MAX_PIECES) return GRAM_E_ARG;
  ep.split = pieces;
  ep.c_pstride = sp ? sp->c_pstride : 0;
  ep.bank_pstride = sp ? sp->bank_pstride : 0;
  ep.c_inter = sp && pieces == 2 && sp->c_interleaved != 0;
  ep.out_scale = sp && sp->out_scale != 0.f ? sp->out_scale : 1.f;
  int ex = 0;
  if (!(ep.out_scale > 0.f) || frexpf(ep.out_scale, &ex) != 0.5f) return GRAM_E_ARG;  // a power of two: exact wherever it is applied
  return 0;
}
template <int EPI>
int launch_p(const void* A, const void* W, int M, int N, int K, int lda, const EpiArgs& ep, hipStream_t st) {
  return ep.split == 2 ? launch<EPI, true>(A, W, M, N, K, lda, ep, st) : launch<EPI, false>(A, W, M, N, K, lda, ep, st);
}
}  // namespace

extern "C" int gram_gemm_bf16_lse_split(const void* A, const void* W, float* logits, float* lse_part, int M, int N, int kc, int lda,
                                        int ldc, const gram_split_t* split, void* stream) {
  EpiArgs ep{};
  const int r = make_split(split, ep);
  if (r) return r;
  const int K = ep.split * kc;  // physical reduction length: W is [N][K], A [M][lda >= K]; two pieces: interleaved by 32-column blocks
  if (M < 1 || N % BN != 0 || kc % BK != 0 || lda < K || (lda & 7) || !lse_part || (logits && (ldc & 3))) return GRAM_E_ARG;
  ep.C = logits;
  ep.ldc = ldc;
  ep.lse_part = lse_part;
  ep.lse_nblk = N / 64;
  return launch_p<GRAM_EPI_F32_LSE>(A, W, M, N, K, lda, ep, (hipStream_t)stream);
}

extern "C" int gram_gemm_bf16_lse(const void* A, const void* W, float* logits, float* lse_part, int M, int N, int K, int lda,
                                  int ldc, void* stream) {
  return gram_gemm_bf16_lse_split(A, W, logits, lse_part, M, N, K, lda, ldc, nullptr, stream);
}

extern "C" int gram_gemm_bf16(const void* A, const void* W, void* C, int M, int N, int K, int lda, int ldc, int epilogue,
                              const gram_kv_bank_t* bank, void* stream) {
  return gram_gemm_bf16_split(A, W, C, M, N, K, lda, ldc, epilogue, bank, nullptr, nullptr, stream);
}

extern "C" int gram_gemm_bf16_ex(const void* A, const void* W, void* C, int M, int N, int K, int lda, int ldc, int epilogue,
                                 const gram_kv_bank_t* bank, const gram_norm_fusion_t* nf, void* stream) {
  return gram_gemm_bf16_split(A, W, C, M, N, K, lda, ldc, epilogue, bank, nf, nullptr, stream);
}

extern "C" int gram_gemm_bf16_split(const void* A, const void* W, void* C, int M, int N, int kc, int lda, int ldc, int epilogue,
                                    const gram_kv_bank_t* bank, const gram_norm_fusion_t* nf, const gram_split_t* split,
                                    void* stream) {
  hipStream_t st = (hipStream_t)stream;
  EpiArgs ep{};
  {
    const int r = make_split(split, ep);
    if (r) return r;
  }
  const int K = ep.split * kc;  // physical reduction length (see gram_gemm_bf16_lse_split)
  if (M < 1 || N % BN != 0 || kc % BK != 0 || lda < K || (lda & 7)) return GRAM_E_ARG;
  if (ep.split > 1) {
    const bool bf16_out = epilogue == GRAM_EPI_BF16 || epilogue == GRAM_EPI_BF16_RELU;
    if ((bf16_out && !ep.c_inter && ep.c_pstride < 1) || (bf16_out && ep.c_inter && ldc < 2 * N) ||
        (epilogue == GRAM_EPI_KV_BANK && ep.bank_pstride < 1))
      return GRAM_E_ARG;
  }
  ep.C = C;
  ep.ldc = ldc;
  if (nf) {
    if (epilogue == GRAM_EPI_F32_ADD) {
      if ((nf->xb_out == nullptr) != (nf->ss_out == nullptr)) return GRAM_E_ARG;
      ep.xb_out = (p16*)nf->xb_out;
      ep.ss_out = nf->ss_out;
      ep.ss_quarter = nf->quarter != 0 && nf->ss_out;
      ep.ss_out_nblk = ep.ss_quarter ? N / 16 : N / 64;
      ep.xs_in = nf->xb_out ? nf->xs_in : nullptr;
    } else if (epilogue == GRAM_EPI_BF16 || epilogue == GRAM_EPI_BF16_RELU) {
      if (nf->ss_in && nf->nblk_in != 0 && (nf->nblk_in < 2 || (nf->nblk_in & 1) || nf->d < 64)) return GRAM_E_ARG;
      ep.ss_in = nf->ss_in;
      ep.ss_nblk = nf->nblk_in;
      ep.ss_quarter = nf->quarter != 0 && nf->ss_in && nf->nblk_in != 0;
      ep.inv_d = 1.0f / (float)nf->d;
      ep.eps = nf->eps;
      if (nf->xs_out && (nf->xs_out == nf->xs_in || !nf->ss_in || nf->nblk_in == 0)) return GRAM_E_ARG;
      ep.xs_in = nf->ss_in && nf->nblk_in != 0 ? nf->xs_in : nullptr;
      ep.xs_out = nf->xs_out;
    } else if (nf->xb_out || nf->ss_out || nf->ss_in) {
      return GRAM_E_ARG;
    }
  }
  switch (epilogue) {
    case GRAM_EPI_BF16:
      if (!C || (ldc & 7)) return GRAM_E_ARG;  // 16-byte row-aligned bf16 stores
      return launch_p<GRAM_EPI_BF16>(A, W, M, N, K, lda, ep, st);
    case GRAM_EPI_BF16_RELU:
      if (!C || (ldc & 7)) return GRAM_E_ARG;
      return launch_p<GRAM_EPI_BF16_RELU>(A, W, M, N, K, lda, ep, st);
    case GRAM_EPI_F32_ADD:
      if (!C || (ldc & 3)) return GRAM_E_ARG;
      return launch_p<GRAM_EPI_F32_ADD>(A, W, M, N, K, lda, ep, st);
    case GRAM_EPI_F32:
      if (!C || (ldc & 3)) return GRAM_E_ARG;
      return launch_p<GRAM_EPI_F32>(A, W, M, N, K, lda, ep, st);
    case GRAM_EPI_KV_BANK: {
      if (!bank || !bank->k || !bank->vt) return GRAM_E_ARG;
      const int inner = bank->H * 64;
      if (inner % BN != 0 || N != bank->n_layers * 2 * inner || bank->S % 32 != 0) return GRAM_E_ARG;  // (V^T is blocked by 32 keys)
      if (bank->passage_map) {
        if (bank->L < 16 || bank->N < 1 || bank->N * bank->L != bank->S || M % bank->L != 0 || M > bank->B * bank->S) return GRAM_E_ARG;
      } else if (M != bank->B * bank->S) {
        return GRAM_E_ARG;
      }
      ep.bank_k = (p16*)bank->k;
      ep.bank_vt = (p16*)bank->vt;
      ep.S = bank->S;
      ep.H = bank->H;
      ep.B = bank->B;
      ep.inner = inner;
      ep.pmap = bank->passage_map;
      ep.pL = bank->L;
      ep.pN = bank->N;
      return launch_p<GRAM_EPI_KV_BANK>(A, W, M, N, K, lda, ep, st);
    }
    default:
      return GRAM_E_ARG;
  }
}
