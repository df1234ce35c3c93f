// Version / error plumbing of the C ABI.
#include "sfm_common.h"

#include <atomic>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace sfm {

char* error_buffer() {
  static thread_local char buf[512] = {0};
  return buf;
}

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(error_buffer(), 512, fmt, ap);
  va_end(ap);
  return code;
}

namespace {
// Option table: explicit settings win over the process environment.  A value
// string is interned once and never freed or modified afterwards (the set of
// distinct values a process ever uses is tiny), so a pointer handed out by
// option() stays valid while another thread sets the option again.
std::mutex g_opt_mu;
std::map<std::string, const std::string*> g_opts;
std::map<std::string, std::string> g_opt_values;   // node addresses are stable

const std::string* intern(const char* value) {
  return &g_opt_values.emplace(value, value).first->second;
}
}  // namespace

const char* option(const char* name) {
  {
    std::lock_guard<std::mutex> lk(g_opt_mu);
    auto it = g_opts.find(name);
    if (it != g_opts.end()) return it->second->c_str();
  }
  return std::getenv(name);
}

std::string option_str(const char* name) {
  const char* v = option(name);
  return v ? std::string(v) : std::string();
}

namespace {
std::mutex g_prof_mu;
std::atomic<bool> g_prof_on{false};
struct Span {
  hipEvent_t a, b;
};
std::vector<Span> g_spans[2];
std::vector<Span> g_free;
Span g_open[2];
struct ClockSample {
  long long v[5];  // cycles, 10 ns ticks[, row tiles drawn << 32 | skipped, column tiles skipped, MFMAs issued]
};
// Targets of the probes' device-to-host copies: PINNED blocks (a copy into pageable
// memory holds the host until the stream reaches it -- the launches behind the
// correlation kernel were not enqueued before it had finished: a 20 us hole in
// front of the peak kernels of every timed flow).  Blocks are never freed.
constexpr int kClockBlock = 1024;
struct ClockPool {
  std::vector<ClockSample*> blocks;
  size_t used = 0;
  ClockSample* next() {
    if (used == blocks.size() * kClockBlock) {
      void* p = nullptr;
      if (hipHostMalloc(&p, sizeof(ClockSample) * kClockBlock, hipHostMallocDefault) != hipSuccess)
        return nullptr;
      blocks.push_back(static_cast<ClockSample*>(p));
    }
    ClockSample* c = &blocks[used / kClockBlock][used % kClockBlock];
    ++used;
    *c = ClockSample{{0, 0, 0, 0, 0}};
    return c;
  }
  ClockSample& at(size_t i) { return blocks[i / kClockBlock][i % kClockBlock]; }
};
ClockPool g_clock[2];
}  // namespace

bool profiling() { return g_prof_on.load(std::memory_order_relaxed); }

void prof_begin(int kind, hipStream_t st) {
  if (!profiling()) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  Span s;
  if (!g_free.empty()) {
    s = g_free.back();
    g_free.pop_back();
  } else {
    if (hipEventCreate(&s.a) != hipSuccess || hipEventCreate(&s.b) != hipSuccess)
      return;
  }
  (void)hipEventRecord(s.a, st);
  g_open[kind] = s;
}

void prof_end(int kind, hipStream_t st) {
  if (!profiling()) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  (void)hipEventRecord(g_open[kind].b, st);
  g_spans[kind].push_back(g_open[kind]);
}

void prof_clock(int kind, const long long* dev_pair, hipStream_t st, int n) {
  if (!profiling() || !dev_pair) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  ClockSample* c = g_clock[kind].next();
  if (!c) return;
  (void)hipMemcpyAsync(c->v, dev_pair, sizeof(long long) * (n >= 2 && n <= 5 ? n : 2),
                       hipMemcpyDeviceToHost, st);
}

}  // namespace sfm

extern "C" {

int sfm_profile_enable(int on) {
  sfm::g_prof_on.store(on != 0);
  return SFM_OK;
}

int sfm_profile_read(SfmProfile* out) {
  if (!out) return sfm::fail(SFM_ERR_INVALID, "out is NULL");
  std::lock_guard<std::mutex> lk(sfm::g_prof_mu);
  if (sfm::g_clock[0].used || sfm::g_clock[1].used)
    SFM_HIP_CHECK(hipDeviceSynchronize());  // the probe copies trail the events
  for (int k = 0; k < 2; ++k) {
    double ms = 0.0;
    for (auto& s : sfm::g_spans[k]) {
      SFM_HIP_CHECK(hipEventSynchronize(s.b));
      float t = 0.f;
      SFM_HIP_CHECK(hipEventElapsedTime(&t, s.a, s.b));
      ms += t;
      sfm::g_free.push_back(s);
    }
    out->kernel_ms[k] = ms;
    out->launches[k] = static_cast<int64_t>(sfm::g_spans[k].size());
    sfm::g_spans[k].clear();
    // the events above are later in stream order than the probe copies
    long long cyc = 0, ticks = 0, skipped = 0, drawn = 0, cols = 0, issued = 0, early = 0;
    for (size_t ci = 0; ci < sfm::g_clock[k].used; ++ci) {
      const sfm::ClockSample& c = sfm::g_clock[k].at(ci);
      cyc += c.v[0];
      ticks += c.v[1];
      skipped += c.v[2] & 0xffffffffLL;
      drawn += (c.v[2] >> 32) & 0xffffffffLL;
      cols += c.v[3] & 0xffffffffLL;
      early += (c.v[3] >> 32) & 0xffffffffLL;
      issued += c.v[4];
    }
    out->clock_mhz[k] = ticks > 0 ? static_cast<double>(cyc) * 100.0 / ticks : 0.0;
    out->tiles_skipped[k] = skipped;
    out->tiles_drawn[k] = drawn;
    out->col_tiles_skipped[k] = cols;
    out->mfma_issued[k] = issued;
    out->tiles_abandoned[k] = early;
    sfm::g_clock[k].used = 0;
  }
  return SFM_OK;
}

int sfm_set_option(const char* name, const char* value) {
  if (!name || std::strncmp(name, "SFM_", 4) != 0)
    return sfm::fail(SFM_ERR_INVALID, "option names start with SFM_");
  std::lock_guard<std::mutex> lk(sfm::g_opt_mu);
  if (value)
    sfm::g_opts[name] = sfm::intern(value);
  else
    sfm::g_opts.erase(name);   // back to the default: the environment variable, if any
  return SFM_OK;
}

int sfm_get_option(const char* name, char* value, size_t capacity) {
  if (!name || !value || capacity == 0) return sfm::fail(SFM_ERR_INVALID, "option: NULL argument");
  const char* v = sfm::option(name);
  if (std::strcmp(name, "SFM_BUILD_MEASUREMENT_SWITCHES") == 0) {
#ifdef SFM_MEASUREMENT_SWITCHES
    v = "1";
#else
    v = "0";
#endif
  }
  if (!v) {
    value[0] = 0;
    return 1;
  }
  std::strncpy(value, v, capacity - 1);
  value[capacity - 1] = 0;
  return SFM_OK;
}

int sfm_version(void) { return SFM_ABI_VERSION; }

const char* sfm_last_error(void) { return sfm::error_buffer(); }

int sfm_device_count(int* count) {
  if (!count) return sfm::fail(SFM_ERR_INVALID, "count is NULL");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    n = 0;
  }
  *count = n;
  return SFM_OK;
}

}  // extern "C"
