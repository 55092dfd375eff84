// Probe (gfx950): does v_mfma_f32_16x16x32_f16 keep f16 SUBNORMAL inputs, or flush them to zero?
// The split-f16 arithmetic (two f16 pieces per value) puts the low piece of every |v| < 2^-3 into the subnormal range, so a
// flushing matrix pipe would cost 10 bits of those values.  Prints one JSON line.
//   hipcc --offload-arch=gfx950 -O2 probe.hip -o probe && ./probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k(float a, float b, float* out) {
  h8 va, vb;
  for (int i = 0; i < 8; ++i) { va[i] = (_Float16)a; vb[i] = (_Float16)b; }
  f4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(va, vb, c, 0, 0, 0);
  if (threadIdx.x == 0) out[0] = c[0];
}
int main() {
  float* d;
  hipMalloc(&d, 4);
  const float cases[4][2] = {{1.0f, 1.0f}, {ldexpf(1.f, -20), 1.0f}, {1.0f, ldexpf(1.f, -20)}, {ldexpf(1.f, -20), 1024.f}};
  printf("{\"probe\": \"mfma_f32_16x16x32_f16 subnormal inputs\", \"cases\": [");
  int ok = 1;
  for (int i = 0; i < 4; ++i) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, cases[i][0], cases[i][1], d);
    float h = 0.f;
    hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
    const float want = 32.f * cases[i][0] * cases[i][1];
    if (h != want) ok = 0;
    printf("%s{\"a\": %g, \"b\": %g, \"got\": %g, \"want\": %g}", i ? ", " : "", cases[i][0], cases[i][1], h, want);
  }
  printf("], \"subnormals_kept\": %s}\n", ok ? "true" : "false");
  return 0;
}
