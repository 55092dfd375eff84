// What does an f32 -> f16 conversion of an out-of-range value give on gfx950 (IEEE: +-inf; with the FP16_OVFL mode bit: +-65504)?
//   hipcc --offload-arch=gfx950 -O3 probe.hip -o probe && ./probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
__global__ void k(const float* in, unsigned* out) {
  const float a = in[0], b = in[1];
  const _Float16 s = (_Float16)a;
  const f32x2 ab = {a, b};
  const h16x2 p = __builtin_convertvector(ab, h16x2);
  out[0] = (unsigned)__builtin_bit_cast(unsigned short, s);
  out[1] = __builtin_bit_cast(unsigned, p);
  float r;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(out[1]), "v"(a));
  out[2] = __builtin_bit_cast(unsigned, r);
}
int main() {
  float h[2] = {1.0e7f, -3.0e5f}, *d;
  unsigned *o, ho[3];
  hipMalloc(&d, 8); hipMalloc(&o, 12);
  hipMemcpy(d, h, 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(1), 0, 0, d, o);
  hipMemcpy(ho, o, 12, hipMemcpyDeviceToHost);
  printf("{\"scalar_cvt_1e7\": \"0x%04x\", \"packed_cvt_1e7_m3e5\": \"0x%08x\", \"fma_mix_remainder_bits\": \"0x%08x\"}\n", ho[0], ho[1], ho[2]);
  return 0;
}
