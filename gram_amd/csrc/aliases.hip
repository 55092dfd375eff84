// aliases.hip -- the `_f16` names of the C-ABI entry points whose historical names say `_bf16` (include/gram_hip.h): the default build
// of the library computes on IEEE half, and a maintainer binding `gram_gemm_bf16` from its name alone would hand it the wrong 16-bit
// type.  Each alias forwards to the historical name in a library built on IEEE half (gram_piece_format() == 1) and returns
// GRAM_E_ARG in the bfloat16 A/B build (`make PIECE=bf16`), where only the `_bf16` names describe the data.
#include "common.h"

#if GRAM_PIECE_FORMAT == 1
#define GRAM_F16_ALIAS(call) return call
#else
#define GRAM_F16_ALIAS(call) return GRAM_E_ARG
#endif

extern "C" int gram_gemm_f16(const void* A, const void* W, void* C, int M, int N, int K, int lda, int ldc, int epilogue,
                             const gram_kv_bank_t* bank_host, void* stream) {
  GRAM_F16_ALIAS(gram_gemm_bf16(A, W, C, M, N, K, lda, ldc, epilogue, bank_host, stream));
}
extern "C" int gram_gemm_f16_ex(const void* A, const void* W, void* C, int M, int N, int K, int lda, int ldc, int epilogue,
                                const gram_kv_bank_t* bank_host, const gram_norm_fusion_t* nf_host, void* stream) {
  GRAM_F16_ALIAS(gram_gemm_bf16_ex(A, W, C, M, N, K, lda, ldc, epilogue, bank_host, nf_host, stream));
}
extern "C" int gram_gemm_f16_split(const void* A, const void* W, void* C, int M, int N, int kc, int lda, int ldc, int epilogue,
                                   const gram_kv_bank_t* bank_host, const gram_norm_fusion_t* nf_host, const gram_split_t* split_host,
                                   void* stream) {
  GRAM_F16_ALIAS(gram_gemm_bf16_split(A, W, C, M, N, kc, lda, ldc, epilogue, bank_host, nf_host, split_host, stream));
}
extern "C" int gram_gemm_f16_lse(const void* A, const void* W, float* logits, float* lse_part, int M, int N, int K, int lda, int ldc,
                                 void* stream) {
  GRAM_F16_ALIAS(gram_gemm_bf16_lse(A, W, logits, lse_part, M, N, K, lda, ldc, stream));
}
extern "C" int gram_gemm_f16_lse_split(const void* A, const void* W, float* logits, float* lse_part, int M, int N, int kc, int lda,
                                       int ldc, const gram_split_t* split_host, void* stream) {
  GRAM_F16_ALIAS(gram_gemm_bf16_lse_split(A, W, logits, lse_part, M, N, kc, lda, ldc, split_host, stream));
}
extern "C" int gram_rmsnorm_f16(const float* x, const float* w, void* out_f16, int rows, int d, float eps, float scale, const float* pos,
                                int N, int L, void* stream) {
  GRAM_F16_ALIAS(gram_rmsnorm_bf16(x, w, out_f16, rows, d, eps, scale, pos, N, L, stream));
}
extern "C" int gram_rmsnorm_f16_map(const float* x, const float* w, void* out_f16, int rows, int d, float eps, float scale,
                                    const float* pos, int N, int L, const int32_t* passage_map, void* stream) {
  GRAM_F16_ALIAS(gram_rmsnorm_bf16_map(x, w, out_f16, rows, d, eps, scale, pos, N, L, passage_map, stream));
}
extern "C" int gram_rmsnorm_f16_split(const float* x, const float* w, void* out_f16, int rows, int d, float eps, float scale,
                                      const float* pos, int N, int L, const int32_t* passage_map, int pieces, void* stream) {
  GRAM_F16_ALIAS(gram_rmsnorm_bf16_split(x, w, out_f16, rows, d, eps, scale, pos, N, L, passage_map, pieces, stream));
}
