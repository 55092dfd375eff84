// Probe (not product code): what does a grid-wide barrier cost on MI355X (256 CUs in 8 XCDs, one L2 per XCD)?  The persistent
// decoder-step kernel sketched in DESIGN.md section 9.4 replaces ~100 dependent launches per step (4.8 us apart) by grid barriers;
// it only pays if a barrier WITH the device-scope release / acquire that makes one phase's stores visible to the next phase's loads on
// other XCDs is well below that.  One workgroup per CU, sense-reversing counter barrier, bounded spins (a wave that waits too long
// sets an abort flag and everybody leaves: no hang).
//   hipcc --offload-arch=gfx950 -O3 probe.hip -o probe && ./probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

struct Bar { unsigned int count, gen, abort_flag, pad; };

template <int MODE>  // 0: barrier only (relaxed), 1: + __threadfence() on both sides, 2: + every workgroup writes a line before and reads another workgroup's line after
__global__ __launch_bounds__(256) void barrier_kernel(Bar* bar, int nbar, float* data, unsigned int* bad) {
  const int nwg = gridDim.x, wg = blockIdx.x, tid = threadIdx.x;
  unsigned int gen = 0;
  for (int it = 0; it < nbar; ++it) {
    if (MODE == 2) data[(size_t)wg * 256 + tid] = (float)(it + wg);
    __syncthreads();
    if (tid == 0) {
      if (MODE >= 1) __threadfence();
      const unsigned int old = atomicAdd(&bar->count, 1u);
      if (old == (unsigned)nwg - 1) {
        atomicExch(&bar->count, 0u);
        __threadfence();
        atomicAdd(&bar->gen, 1u);
      } else {
        long spins = 0;
        while (__hip_atomic_load(&bar->gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gen) {
          if (++spins > 20000000L || __hip_atomic_load(&bar->abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
            atomicExch(&bar->abort_flag, 1u);
            break;
          }
          __builtin_amdgcn_s_sleep(1);
        }
      }
      if (MODE >= 1) __threadfence();
    }
    ++gen;
    __syncthreads();
    if (__hip_atomic_load(&bar->abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
    if (MODE == 2) {
      const int src = (wg + 97) % nwg;  // another XCD's workgroup (ids are dealt round-robin to the XCDs)
      const float v = __builtin_nontemporal_load(&data[(size_t)src * 256 + tid]);
      if (v != (float)(it + src)) atomicAdd(bad, 1u);
    }
  }
}

int main() {
  int dev = 0, cus = 0;
  hipGetDevice(&dev);
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  Bar* bar;
  float* data;
  unsigned int* bad;
  hipMalloc(&bar, sizeof(Bar));
  hipMalloc(&data, (size_t)cus * 256 * 4);
  hipMalloc(&bad, 4);
  const int nbar = 2000;
  for (int mode = 0; mode < 3; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      hipMemset(bar, 0, sizeof(Bar));
      hipMemset(bad, 0, 4);
      hipEvent_t e0, e1;
      hipEventCreate(&e0);
      hipEventCreate(&e1);
      hipEventRecord(e0, 0);
      if (mode == 0) hipLaunchKernelGGL(barrier_kernel<0>, dim3(cus), dim3(256), 0, 0, bar, nbar, data, bad);
      if (mode == 1) hipLaunchKernelGGL(barrier_kernel<1>, dim3(cus), dim3(256), 0, 0, bar, nbar, data, bad);
      if (mode == 2) hipLaunchKernelGGL(barrier_kernel<2>, dim3(cus), dim3(256), 0, 0, bar, nbar, data, bad);
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float ms = 0.f;
      hipEventElapsedTime(&ms, e0, e1);
      Bar h;
      unsigned int hb = 0;
      hipMemcpy(&h, bar, sizeof(Bar), hipMemcpyDeviceToHost);
      hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
      printf("{\"mode\": %d, \"workgroups\": %d, \"barriers\": %d, \"us_per_barrier\": %.3f, \"aborted\": %u, \"stale_reads\": %u}\n", mode, cus, nbar,
             1e3 * ms / nbar, h.abort_flag, hb);
    }
  }
  // for comparison: dependent empty launches
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipMemset(bar, 0, sizeof(Bar));
  hipLaunchKernelGGL(barrier_kernel<0>, dim3(cus), dim3(256), 0, 0, bar, 0, data, bad);
  hipEventRecord(e0, 0);
  for (int i = 0; i < 2000; ++i) hipLaunchKernelGGL(barrier_kernel<0>, dim3(cus), dim3(256), 0, 0, bar, 0, data, bad);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  printf("{\"dependent_empty_launches\": 2000, \"us_per_launch\": %.3f}\n", 1e3 * ms / 2000);
  return 0;
}
