// Probe: does v_pk_mul_f32 vD, vA, vD op_sel_hi:[1,0] (dst pair == broadcast source pair) give
// D.hi = A.hi * D_old.lo (architecturally expected) or A.hi * D_new.lo (low lane written first)?
// Build: hipcc --offload-arch=gfx950 -O2 pk_mul_self.hip -o pk_mul_self ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>

__global__ void probe(const float* a, const float* s, float* out) {
  const int t = threadIdx.x;
  float2 av = reinterpret_cast<const float2*>(a)[t];
  float2 sv = make_float2(s[t], 12345.f);
  uint64_t A, D;
  memcpy(&A, &av, 8);
  memcpy(&D, &sv, 8);
  asm volatile("v_pk_mul_f32 %0, %1, %0 op_sel_hi:[1,0]" : "+v"(D) : "v"(A));
  float2 r;
  memcpy(&r, &D, 8);
  out[2 * t] = r.x;
  out[2 * t + 1] = r.y;
}

int main() {
  float ha[128], hs[64], ho[128];
  for (int i = 0; i < 64; ++i) {
    ha[2 * i] = 2.f + i;
    ha[2 * i + 1] = 3.f + i;
    hs[i] = 0.5f;
  }
  float *a, *s, *o;
  hipMalloc(&a, sizeof ha);
  hipMalloc(&s, sizeof hs);
  hipMalloc(&o, sizeof ho);
  hipMemcpy(a, ha, sizeof ha, hipMemcpyHostToDevice);
  hipMemcpy(s, hs, sizeof hs, hipMemcpyHostToDevice);
  probe<<<1, 64>>>(a, s, o);
  hipMemcpy(ho, o, sizeof ho, hipMemcpyDeviceToHost);
  int ok = 0, self = 0;
  for (int i = 0; i < 64; ++i) {
    const float lo = ha[2 * i] * hs[i], hi_ok = ha[2 * i + 1] * hs[i], hi_self = ha[2 * i + 1] * lo;
    ok += (ho[2 * i] == lo && ho[2 * i + 1] == hi_ok);
    self += (ho[2 * i] == lo && ho[2 * i + 1] == hi_self);
  }
  printf("lane0: lo=%g hi=%g (expected lo=%g hi=%g; self-overwrite would give hi=%g)\n", ho[0], ho[1], ha[0] * hs[0], ha[1] * hs[0],
         ha[1] * ha[0] * hs[0]);
  printf("RESULT architecturally-correct lanes: %d/64, self-overwrite lanes: %d/64\n", ok, self);
  return 0;
}
