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
#include "common.h"
#include "prof.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;  // 16 KiB per operand per stage

enum { V_REG2 = 0, V_DMA = 1 };
int g_force_variant = -1;  // tuning hook (gram_debug_set_gemm_variant)

__device__ __forceinline__ int swz(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

struct EpiArgs {
  void* C;
  int ldc;
  // KV bank
  bf16* bank_k;
  bf16* bank_vt;
  int S, H, B, inner;
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
__device__ __forceinline__ void compute_tile(const char* sa, const char* sw, int wm, int wn, int r16, int g, f32x4 (&acc)[4][4]) {
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    bf16x8 fw[4], fa[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      fw[i] = *reinterpret_cast<const bf16x8*>(sw + swz(wn * 64 + i * 16 + r16, ks * 4 + g));
      fa[i] = *reinterpret_cast<const bf16x8*>(sa + swz(wm * 64 + i * 16 + r16, ks * 4 + g));
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = mfma16(fw[i], fa[j], acc[i][j]);
  }
}

// acc[i][j] lane (r16,g) element e = C[m = m0+wm*64+j*16+r16][n = n0+wn*64+i*16+4g+e]
template <int EPI>
__device__ __forceinline__ void epilogue(f32x4 (&acc)[4][4], int m0, int n0, int wm, int wn, int r16, int g, int M,
                                         const EpiArgs& ep) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int m = m0 + wm * 64 + j * 16 + r16;
    if (m >= M) continue;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int n = n0 + wn * 64 + i * 16 + 4 * g;
      f32x4 v = acc[i][j];
      if constexpr (EPI == GRAM_EPI_BF16 || EPI == GRAM_EPI_BF16_RELU) {
        if constexpr (EPI == GRAM_EPI_BF16_RELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (bf16)v[e];
        *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16*>(ep.C) + (size_t)m * ep.ldc + n) = o;
      } else if constexpr (EPI == GRAM_EPI_F32_ADD) {
        f32x4* p = reinterpret_cast<f32x4*>(reinterpret_cast<float*>(ep.C) + (size_t)m * ep.ldc + n);
        f32x4 old = *p;
        *p = old + v;
      } else if constexpr (EPI == GRAM_EPI_F32) {
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(ep.C) + (size_t)m * ep.ldc + n) = v;
      } else {  // GRAM_EPI_KV_BANK
        const int b = m / ep.S, s = m - b * ep.S;
        const int lw = n / ep.inner;  // layer*2 + which   (uniform per block: inner % 128 == 0)
        const int layer = lw >> 1, which = lw & 1;
        const int rem = n - lw * ep.inner;
        const int h = rem >> 6, d = rem & 63;
        const size_t head = ((size_t)layer * ep.B + b) * ep.H + h;
        if (which == 0) {
          bf16x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (bf16)v[e];
          *reinterpret_cast<bf16x4*>(ep.bank_k + (head * ep.S + s) * 64 + d) = o;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) ep.bank_vt[(head * 64 + d + e) * ep.S + s] = (bf16)v[e];
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
  load_tile(0);
  store_tile(0);
  __syncthreads();
  for (int kt = 0; kt < nkt; ++kt) {
    const int stage = kt & 1;
    if (kt + 1 < nkt) load_tile(kt + 1);
    const char* sa = smem + stage * 2 * TILE_BYTES;
    compute_tile(sa, sa + TILE_BYTES, wm, wn, r16, g, acc);
    if (kt + 1 < nkt) store_tile(stage ^ 1);
    __syncthreads();
  }
  epilogue<EPI>(acc, m0, n0, wm, wn, r16, g, M, ep);
}

// ---------------------------------------------------------------------------------------------
template <int EPI>
__global__ __launch_bounds__(256, 4) void gemm_dma_kernel(const bf16* __restrict__ A, const bf16* __restrict__ W, int M,
                                                          int N, int K, int lda, EpiArgs ep) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  int mt, nt;
  tile_of_block(N / BN, mt, nt);
  const int m0 = mt * BM, n0 = nt * BN;
  const int r16 = lane & 15, g = lane >> 4;

  // LDS-DMA map: a wave instruction fills one 8-row x 128-B group (1 KiB, lane-linear);
  // wave w owns groups 4w..4w+3 of both operands.  Lane l lands at (row 8*grp + l>>3, slot l&7),
  // so it must FETCH chunk (l&7) ^ ((row>>1)&7) for the read-side swizzle to find it.
  const bf16* a_src[4];
  const bf16* w_src[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int grp = wave * 4 + i;
    const int row = grp * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    const int am = min(m0 + row, M - 1);  // rows past M: any valid address (never stored)
    a_src[i] = A + (size_t)am * lda + chunk * 8;
    w_src[i] = W + (size_t)(n0 + row) * K + chunk * 8;
  }
  char* sa = smem;
  char* sw = smem + TILE_BYTES;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nkt = K / BK;
  for (int kt = 0; kt < nkt; ++kt) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int grp = wave * 4 + i;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_src[i] + kt * BK),
                                       (__attribute__((address_space(3))) void*)(sa + grp * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(w_src[i] + kt * BK),
                                       (__attribute__((address_space(3))) void*)(sw + grp * 1024), 16, 0, 0);
    }
    __syncthreads();  // hipcc drains the DMA (vmcnt(0)) ahead of the barrier
    compute_tile(sa, sw, wm, wn, r16, g, acc);
    __syncthreads();
  }
  epilogue<EPI>(acc, m0, n0, wm, wn, r16, g, M, ep);
}

int pick_variant(int M, int N) {
  if (g_force_variant >= 0) return g_force_variant;
  (void)M;
  (void)N;
  return V_DMA;  // measured faster on every shape of the path, small grids included (tests/bench_gemm.py)
}

template <int EPI>
int launch(const void* A, const void* W, int M, int N, int K, int lda, EpiArgs ep, hipStream_t st) {
  const int nblocks = (N / BN) * ((M + BM - 1) / BM);
  gram_prof::Scope prof(GRAM_K_GEMM, st, 2.0 * M * N * K);
  if (pick_variant(M, N) == V_DMA) {
    hipLaunchKernelGGL(gemm_dma_kernel<EPI>, dim3(nblocks), dim3(256), 2 * TILE_BYTES, st, (const bf16*)A, (const bf16*)W, M, N,
                       K, lda, ep);
  } else {
    static bool attr_set = false;
    if (!attr_set) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_reg2_kernel<EPI>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 4 * TILE_BYTES);
      if (e != hipSuccess) return (int)e;
      attr_set = true;
    }
    hipLaunchKernelGGL(gemm_reg2_kernel<EPI>, dim3(nblocks), dim3(256), 4 * TILE_BYTES, st, (const bf16*)A, (const bf16*)W, M,
                       N, K, lda, ep);
  }
  GRAM_CHECK_LAUNCH();
  return 0;
}

}  // namespace

extern "C" int gram_debug_set_gemm_variant(int v) {
  g_force_variant = v;
  return 0;
}

extern "C" int gram_gemm_bf16(const void* A, const void* W, void* C, int M, int N, int K, int lda, int ldc, int epilogue,
                              const gram_kv_bank_t* bank, void* stream) {
  if (M < 1 || N % BN != 0 || K % BK != 0 || lda < K || (lda & 7)) return GRAM_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  EpiArgs ep{};
  ep.C = C;
  ep.ldc = ldc;
  switch (epilogue) {
    case GRAM_EPI_BF16:
      if (!C || (ldc & 3)) return GRAM_E_ARG;
      return launch<GRAM_EPI_BF16>(A, W, M, N, K, lda, ep, st);
    case GRAM_EPI_BF16_RELU:
      if (!C || (ldc & 3)) return GRAM_E_ARG;
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
      if (inner % BN != 0 || N != bank->n_layers * 2 * inner || M != bank->B * bank->S || bank->S % 16 != 0)
        return GRAM_E_ARG;
      ep.bank_k = (bf16*)bank->k;
      ep.bank_vt = (bf16*)bank->vt;
      ep.S = bank->S;
      ep.H = bank->H;
      ep.B = bank->B;
      ep.inner = inner;
      return launch<GRAM_EPI_KV_BANK>(A, W, M, N, K, lda, ep, st);
    }
    default:
      return GRAM_E_ARG;
  }
}
