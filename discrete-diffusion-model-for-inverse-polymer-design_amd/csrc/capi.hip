// Error reporting, diagnostics, host-side RNG helpers and the optional event profiler of
// libbidiff_hip.so.
#include "bd_common.h"
#include "bd_rng.h"
#include "bd_drop.h"
#include "bd_prof.h"
#include "bidiff.h"

#include <stdarg.h>
#include <stdio.h>
#include <mutex>
#include <vector>

namespace {
thread_local char g_err[512] = "";
}

int bd_fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

int bd_check_launch(const char* what) {
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return bd_fail(BD_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
  return BD_OK;
}

extern "C" const char* bd_version(void) { return "bidiff-hip 0.1 (gfx950)"; }
extern "C" const char* bd_last_error(void) { return g_err; }

extern "C" int bd_device_available(void) {
  int n = 0;
  const hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) { (void)hipGetLastError(); return 0; }
  return n > 0 ? 1 : 0;
}

extern "C" void bd_host_gumbel(const float* u, float* out, long long n) {
  for (long long i = 0; i < n; ++i) out[i] = bd_gumbel(u[i]);
}
extern "C" void bd_host_philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                               uint32_t k1, uint32_t out[4]) {
  const bd_philox_out o = bd_philox4x32_10(c0, c1, c2, c3, k0, k1);
  for (int i = 0; i < 4; ++i) out[i] = o.v[i];
}
extern "C" float bd_host_uniform_select(unsigned long long seed, uint32_t seq, uint32_t t, uint32_t pos) {
  return bd_uniform_select(seed, seq, t, pos);
}
extern "C" float bd_host_uniform_token(unsigned long long seed, uint32_t seq, uint32_t t, uint32_t pos,
                                       uint32_t v) {
  return bd_uniform_token(seed, seq, t, pos, v);
}
// Host evaluation of the training-dropout decision stream (the code the kernels run): out[r, c] =
// keep / (1 - p) of element (r, c) of a site. Returns 0, or -1 for p outside [0, 1).
extern "C" int bd_host_dropout_factors(const bd_dropout* drop, unsigned site, long long rows, long long cols,
                                       float* out) {
  bd_drop_args da;
  if (!bd_drop_make(drop, site, &da)) return -1;
  for (long long r = 0; r < rows; ++r)
    for (long long c8 = 0; 8 * c8 < cols; ++c8) {
      const bd_philox_out w = bd_drop_block(da, (uint32_t)r, (uint32_t)c8);
      for (int i = 0; i < 8 && 8 * c8 + i < cols; ++i) out[r * cols + 8 * c8 + i] = bd_drop_factor(da, w, i);
    }
  return 0;
}

// ---- event profiler: pairs of events around selected launches, resolved lazily
namespace {
struct Pair { hipEvent_t a, b; double flops; int kind; };
std::mutex g_mu;
bool g_on = false;
std::vector<Pair> g_pending;
std::vector<hipEvent_t> g_free;
long long g_launches[BD_PROF_KINDS] = {};
double g_ms[BD_PROF_KINDS] = {}, g_flops[BD_PROF_KINDS] = {};

hipEvent_t get_event() {
  if (!g_free.empty()) { hipEvent_t e = g_free.back(); g_free.pop_back(); return e; }
  hipEvent_t e = nullptr;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  return e;
}

void drain_locked() {
  for (Pair& pr : g_pending) {
    float ms = 0.f;
    if (hipEventSynchronize(pr.b) == hipSuccess && hipEventElapsedTime(&ms, pr.a, pr.b) == hipSuccess) {
      g_launches[pr.kind]++; g_ms[pr.kind] += ms; g_flops[pr.kind] += pr.flops;
    }
    g_free.push_back(pr.a); g_free.push_back(pr.b);
  }
  g_pending.clear();
}
}  // namespace

bool bd_prof_on() { return g_on; }

void* bd_prof_begin(hipStream_t s) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (!g_on) return nullptr;
  hipEvent_t a = get_event();
  if (!a) return nullptr;
  hipEventRecord(a, s);
  return (void*)a;
}

void bd_prof_end(void* tok, hipStream_t s, double flops, int kind) {
  if (!tok) return;
  std::lock_guard<std::mutex> lk(g_mu);
  hipEvent_t b = get_event();
  if (!b) { g_free.push_back((hipEvent_t)tok); return; }
  hipEventRecord(b, s);
  g_pending.push_back(Pair{(hipEvent_t)tok, b, flops, kind >= 0 && kind < BD_PROF_KINDS ? kind : 0});
  if (g_pending.size() >= 4096) drain_locked();
}

extern "C" int bd_prof_enable(int on) {
  std::lock_guard<std::mutex> lk(g_mu);
  g_on = on != 0;
  return BD_OK;
}

extern "C" int bd_prof_read_kind(int kind, long long* launches, double* total_ms, double* total_flops, int reset) {
  std::lock_guard<std::mutex> lk(g_mu);
  drain_locked();
  long long n = 0;
  double ms = 0.0, fl = 0.0;
  for (int k = 0; k < BD_PROF_KINDS; ++k) {
    if (kind >= 0 && k != kind) continue;
    n += g_launches[k]; ms += g_ms[k]; fl += g_flops[k];
    if (reset) { g_launches[k] = 0; g_ms[k] = 0.0; g_flops[k] = 0.0; }
  }
  if (launches) *launches = n;
  if (total_ms) *total_ms = ms;
  if (total_flops) *total_flops = fl;
  return BD_OK;
}

extern "C" int bd_prof_read(long long* launches, double* total_ms, double* total_flops, int reset) {
  return bd_prof_read_kind(-1, launches, total_ms, total_flops, reset);
}
