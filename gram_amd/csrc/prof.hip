// prof.hip -- event pool behind prof.h / gram_prof_* (include/gram_hip.h).
#include <vector>

#include "common.h"
#include "prof.h"

namespace gram_prof {
uint32_t g_mask = 0;
namespace {
struct Pair {
  hipEvent_t a, b;
  int kind;
  double work;
};
std::vector<Pair> g_pool;   // pre-created event pairs
size_t g_used = 0;
int64_t g_dropped = 0;
}  // namespace

void begin(int kind, hipStream_t st) {
  if (g_used >= g_pool.size()) {
    ++g_dropped;
    return;
  }
  g_pool[g_used].kind = kind;
  (void)hipEventRecord(g_pool[g_used].a, st);
}
void end(int kind, hipStream_t st, double work) {
  if (g_used >= g_pool.size()) return;
  g_pool[g_used].work = work;
  (void)hipEventRecord(g_pool[g_used].b, st);
  ++g_used;
}
}  // namespace gram_prof

using namespace gram_prof;

extern "C" int gram_prof_enable(uint32_t kind_mask, int max_events) {
  g_mask = 0;
  for (auto& p : g_pool) {
    (void)hipEventDestroy(p.a);
    (void)hipEventDestroy(p.b);
  }
  g_pool.clear();
  g_used = 0;
  g_dropped = 0;
  if (kind_mask == 0) return 0;
  if (max_events < 1) return GRAM_E_ARG;
  g_pool.resize(max_events);
  for (auto& p : g_pool) {
    hipError_t e = hipEventCreate(&p.a);
    if (e == hipSuccess) e = hipEventCreate(&p.b);
    if (e != hipSuccess) return (int)e;
  }
  g_mask = kind_mask;
  return 0;
}

extern "C" int gram_prof_reset(void) {
  g_used = 0;
  g_dropped = 0;
  return 0;
}

extern "C" int gram_prof_collect(int kind, double* total_ms, int64_t* launches, double* work, int64_t* dropped) {
  if (kind < 0 || kind >= GRAM_K_COUNT) return GRAM_E_ARG;
  double ms = 0, w = 0;
  int64_t n = 0;
  for (size_t i = 0; i < g_used; ++i) {
    if (g_pool[i].kind != kind) continue;
    hipError_t e = hipEventSynchronize(g_pool[i].b);
    if (e != hipSuccess) return (int)e;
    float t = 0.f;
    e = hipEventElapsedTime(&t, g_pool[i].a, g_pool[i].b);
    if (e != hipSuccess) return (int)e;
    ms += t;
    w += g_pool[i].work;
    ++n;
  }
  if (total_ms) *total_ms = ms;
  if (launches) *launches = n;
  if (work) *work = w;
  if (dropped) *dropped = g_dropped;
  return 0;
}
