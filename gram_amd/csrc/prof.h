// prof.h -- optional per-kernel-kind HIP-event timing on the launch stream (used by bench.py to
// measure the dominant kernels live, inside the timed region).  Disabled = one relaxed load.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gram_prof {
extern uint32_t g_mask;
void begin(int kind, hipStream_t st);
void end(int kind, hipStream_t st, double work);
struct Scope {
  int kind;
  hipStream_t st;
  double work;
  bool on;
  Scope(int k, hipStream_t s, double w) : kind(k), st(s), work(w), on((g_mask >> k) & 1u) {
    if (on) begin(kind, st);
  }
  ~Scope() {
    if (on) end(kind, st, work);
  }
};
}  // namespace gram_prof
