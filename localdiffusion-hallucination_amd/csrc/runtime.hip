// Runtime plumbing of the C-ABI: error string, device info, HIP-graph capture, events.
#include "common.hip.h"
#include <dlfcn.h>
#include <string.h>
#include <atomic>
#include <map>
#include <mutex>

static thread_local char g_err[512] = "";

int ld_fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

extern "C" const char* ld_last_error(void) { return g_err; }
extern "C" int ld_version(void) { return 100; }

extern "C" int ld_device_info(char* name, int name_len, int* cus, int64_t* mem) {
  int dev = 0;
  LD_HIP(hipGetDevice(&dev));
  hipDeviceProp_t p;
  LD_HIP(hipGetDeviceProperties(&p, dev));
  if (name && name_len > 0) snprintf(name, name_len, "%s (%s)", p.name, p.gcnArchName);
  if (cus) *cus = p.multiProcessorCount;
  if (mem) *mem = (int64_t)p.totalGlobalMem;
  return LD_OK;
}

extern "C" int ld_graph_begin(void* stream) {
  LD_HIP(hipStreamBeginCapture(reinterpret_cast<hipStream_t>(stream), hipStreamCaptureModeThreadLocal));
  return LD_OK;
}
extern "C" int ld_graph_end(void* stream, void** exec_out) {
  LD_REQUIRE(exec_out, "ld_graph_end: null out");
  hipGraph_t g = nullptr;
  LD_HIP(hipStreamEndCapture(reinterpret_cast<hipStream_t>(stream), &g));
  hipGraphExec_t ex = nullptr;
  hipError_t e = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
  (void)hipGraphDestroy(g);
  if (e != hipSuccess) return ld_fail(LD_EHIP, "hipGraphInstantiate: %s", hipGetErrorString(e));
  *exec_out = ex;
  return LD_OK;
}
extern "C" int ld_graph_launch(void* exec, void* stream) {
  LD_HIP(hipGraphLaunch(reinterpret_cast<hipGraphExec_t>(exec), reinterpret_cast<hipStream_t>(stream)));
  return LD_OK;
}
extern "C" int ld_graph_destroy(void* exec) {
  if (exec) LD_HIP(hipGraphExecDestroy(reinterpret_cast<hipGraphExec_t>(exec)));
  return LD_OK;
}

extern "C" int ld_event_create(void** ev) {
  LD_REQUIRE(ev, "ld_event_create: null");
  hipEvent_t e;
  LD_HIP(hipEventCreate(&e));
  *ev = e;
  return LD_OK;
}
extern "C" int ld_event_record(void* ev, void* stream) {
  LD_HIP(hipEventRecord(reinterpret_cast<hipEvent_t>(ev), reinterpret_cast<hipStream_t>(stream)));
  return LD_OK;
}
extern "C" int ld_event_elapsed_ms(void* a, void* b, float* ms) {
  LD_REQUIRE(ms, "ld_event_elapsed_ms: null");
  LD_HIP(hipEventSynchronize(reinterpret_cast<hipEvent_t>(b)));
  LD_HIP(hipEventElapsedTime(ms, reinterpret_cast<hipEvent_t>(a), reinterpret_cast<hipEvent_t>(b)));
  return LD_OK;
}
extern "C" int ld_event_destroy(void* ev) {
  if (ev) LD_HIP(hipEventDestroy(reinterpret_cast<hipEvent_t>(ev)));
  return LD_OK;
}

// ---- per-launch timing session (bench.py's per-kernel leg) --------------------------------------------------
namespace {
hipEvent_t* g_tev = nullptr;      // 2 events per launch
int g_tcap = 0, g_tn = -1;        // g_tn < 0: no session
}
bool ld_timing_next(hipEvent_t* start, hipEvent_t* stop) {
  if (g_tn < 0 || g_tn >= g_tcap) return false;
  *start = g_tev[2 * g_tn];
  *stop = g_tev[2 * g_tn + 1];
  ++g_tn;
  return true;
}
extern "C" int ld_timing_begin(int max_launches) {
  LD_REQUIRE(max_launches > 0 && max_launches <= (1 << 20), "ld_timing_begin: max_launches %d", max_launches);
  if (max_launches > g_tcap) {
    hipEvent_t* ev = new hipEvent_t[2 * (size_t)max_launches];
    for (int i = 0; i < 2 * g_tcap; ++i) ev[i] = g_tev[i];
    for (int i = 2 * g_tcap; i < 2 * max_launches; ++i) LD_HIP(hipEventCreate(&ev[i]));
    delete[] g_tev;
    g_tev = ev;
    g_tcap = max_launches;
  }
  g_tn = 0;
  return LD_OK;
}
extern "C" int ld_timing_count(void) { return g_tn < 0 ? 0 : g_tn; }
extern "C" int ld_timing_end(float* ms, int cap, int* count) {
  LD_REQUIRE(g_tn >= 0, "ld_timing_end: no session");
  const int n = g_tn;
  g_tn = -1;
  if (count) *count = n;
  for (int i = 0; i < n && i < cap && ms; ++i) {
    LD_HIP(hipEventSynchronize(g_tev[2 * i + 1]));
    LD_HIP(hipEventElapsedTime(&ms[i], g_tev[2 * i], g_tev[2 * i + 1]));
  }
  return LD_OK;
}

extern "C" int ld_timing_end_abs(float* start_ms, float* stop_ms, int cap, int* count) {
  LD_REQUIRE(g_tn >= 0, "ld_timing_end_abs: no session");
  LD_REQUIRE(start_ms && stop_ms, "ld_timing_end_abs: null output");
  const int n = g_tn;
  g_tn = -1;
  if (count) *count = n;
  for (int i = 0; i < n && i < cap; ++i) {
    LD_HIP(hipEventSynchronize(g_tev[2 * i + 1]));
    LD_HIP(hipEventElapsedTime(&start_ms[i], g_tev[0], g_tev[2 * i]));
    LD_HIP(hipEventElapsedTime(&stop_ms[i], g_tev[0], g_tev[2 * i + 1]));
  }
  return LD_OK;
}

extern "C" int ld_memset_zero(void* ptr, size_t bytes, void* stream) {
  LD_REQUIRE(ptr || bytes == 0, "ld_memset_zero: null");
  if (bytes) LD_HIP(hipMemsetAsync(ptr, 0, bytes, reinterpret_cast<hipStream_t>(stream)));
  return LD_OK;
}

extern "C" int ld_memset_bytes(void* ptr, int value, size_t bytes, void* stream) {
  LD_REQUIRE(ptr || bytes == 0, "ld_memset_bytes: null");
  if (bytes) LD_HIP(hipMemsetAsync(ptr, value & 0xff, bytes, reinterpret_cast<hipStream_t>(stream)));
  return LD_OK;
}

extern "C" int ld_stream_wait_event(void* stream, void* ev) {
  LD_REQUIRE(ev, "ld_stream_wait_event: null event");
  LD_HIP(hipStreamWaitEvent(reinterpret_cast<hipStream_t>(stream), reinterpret_cast<hipEvent_t>(ev), 0));
  return LD_OK;
}

// ---- profiler-visible phase markers (roctx ranges) -----------------------------------------------------------
// The reference's only hook is a wall-clock timer around sample() (test.py:392-415).  These ranges show up in
// `rocprofv3 --marker-trace` (rocprofiler-sdk's roctx) or any tool that interposes libroctx64.  Only a roctx library
// the profiler has ALREADY mapped is used (RTLD_NOLOAD): an unprofiled process loads nothing and a range costs one
// predictable branch.  LD_ROCTX=1 loads the library on demand (a tool that attaches later).
namespace {
typedef int (*RangePushFn)(const char*);
typedef int (*RangePopFn)(void);
struct Roctx {
  RangePushFn push = nullptr;
  RangePopFn pop = nullptr;
};
Roctx* roctx() {
  static Roctx r;
  static std::once_flag once;
  std::call_once(once, [] {
    if (getenv("LD_NO_ROCTX")) return;
    const char* names[3] = {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so"};
    const int passes = getenv("LD_ROCTX") ? 2 : 1;
    for (int pass = 0; pass < passes && !r.push; ++pass)                // pass 0: whatever the profiler already mapped
      for (const char* n : names) {
        void* h = dlopen(n, RTLD_NOW | RTLD_LOCAL | (pass == 0 ? RTLD_NOLOAD : 0));
        if (!h) continue;
        r.push = (RangePushFn)dlsym(h, "roctxRangePushA");
        r.pop = (RangePopFn)dlsym(h, "roctxRangePop");
        if (r.push && r.pop) break;
        r.push = nullptr;
        r.pop = nullptr;
      }
  });
  return &r;
}
}  // namespace
extern "C" int ld_range_push(const char* name) {
  Roctx* r = roctx();
  if (r->push && name) r->push(name);
  return LD_OK;
}
extern "C" int ld_range_pop(void) {
  Roctx* r = roctx();
  if (r->pop) r->pop();
  return LD_OK;
}

// ---- tuning table ---------------------------------------------------------------------------------------------------
namespace {
LdTuning g_tuning = {/*c1_group*/ 1, /*c1_group_max_px*/ 32768, /*c1_group_min_ch*/ 4, /*c1_pair_max_px*/ 1LL << 40, /*c1_small_min*/ 256,
                     /*conv_raw*/ 1, /*conv_mt4_min_wgs*/ 256, /*conv_big_min*/ 512, /*conv_sk*/ 0, /*conv_sk_max_wgs*/ 256,
                     /*conv_c32*/ 0, /*conv_c32_min_tiles*/ 2048, /*conv_s32*/ 3, /*conv_s32_min_tiles*/ 1024, /*conv_big4_min*/ 256,
                     /*gn_frags_per_block*/ 512, /*fold_split_min*/ 32,
                     /*attn_split_max_wgs*/ 256, /*attn_split_min_n*/ 2048, /*lead_args*/ 1, /*attn_xcd_map*/ 1};
struct TuningEntry { const char* name; long long LdTuning::*field; };
const TuningEntry g_tuning_entries[] = {
    {"c1_group", &LdTuning::c1_group}, {"c1_group_max_px", &LdTuning::c1_group_max_px}, {"c1_group_min_ch", &LdTuning::c1_group_min_ch},
    {"c1_pair_max_px", &LdTuning::c1_pair_max_px}, {"c1_small_min", &LdTuning::c1_small_min}, {"conv_raw", &LdTuning::conv_raw},
    {"conv_mt4_min_wgs", &LdTuning::conv_mt4_min_wgs}, {"conv_big_min", &LdTuning::conv_big_min}, {"conv_sk", &LdTuning::conv_sk},
    {"conv_sk_max_wgs", &LdTuning::conv_sk_max_wgs}, {"conv_c32", &LdTuning::conv_c32}, {"conv_c32_min_tiles", &LdTuning::conv_c32_min_tiles},
    {"conv_s32", &LdTuning::conv_s32}, {"conv_s32_min_tiles", &LdTuning::conv_s32_min_tiles}, {"conv_big4_min", &LdTuning::conv_big4_min},
    {"gn_frags_per_block", &LdTuning::gn_frags_per_block}, {"fold_split_min", &LdTuning::fold_split_min},
    {"attn_split_max_wgs", &LdTuning::attn_split_max_wgs}, {"attn_split_min_n", &LdTuning::attn_split_min_n}, {"lead_args", &LdTuning::lead_args}, {"attn_xcd_map", &LdTuning::attn_xcd_map}};
constexpr int g_tuning_n = sizeof(g_tuning_entries) / sizeof(g_tuning_entries[0]);
void tuning_env_once() {
  // the ONE place the library reads tuning from the environment: LD_<NAME IN CAPITALS>, once per process
  static std::once_flag once;
  std::call_once(once, [] {
    for (const TuningEntry& e : g_tuning_entries) {
      char var[64] = "LD_";
      size_t k = 3;
      for (const char* c = e.name; *c && k + 1 < sizeof(var); ++c) var[k++] = (*c >= 'a' && *c <= 'z') ? (char)(*c - 32) : *c;
      var[k] = 0;
      if (const char* v = getenv(var)) g_tuning.*(e.field) = atoll(v);
    }
  });
}
}  // namespace
const LdTuning& ld_tuning() {
  tuning_env_once();
  return g_tuning;
}
extern "C" int ld_tuning_set(const char* name, long long value) {
  LD_REQUIRE(name, "ld_tuning_set: null name");
  tuning_env_once();
  for (const TuningEntry& e : g_tuning_entries)
    if (strcmp(e.name, name) == 0) {
      g_tuning.*(e.field) = value;
      return LD_OK;
    }
  return ld_fail(LD_EINVAL, "ld_tuning_set: unknown entry '%s'", name);
}
extern "C" int ld_tuning_get(const char* name, long long* value_out) {
  LD_REQUIRE(name && value_out, "ld_tuning_get: null argument");
  tuning_env_once();
  for (const TuningEntry& e : g_tuning_entries)
    if (strcmp(e.name, name) == 0) {
      *value_out = g_tuning.*(e.field);
      return LD_OK;
    }
  return ld_fail(LD_EINVAL, "ld_tuning_get: unknown entry '%s'", name);
}
extern "C" int ld_tuning_count(void) { return g_tuning_n; }
extern "C" const char* ld_tuning_name(int index) { return index >= 0 && index < g_tuning_n ? g_tuning_entries[index].name : nullptr; }

// ---- host-side launch counters: which kernel a dispatcher routed a call to ------------------------------------
namespace {
std::atomic<long long> g_counters[LD_COUNTER_MAX];
}
void ld_count(int which) {
  if (which >= 0 && which < LD_COUNTER_MAX) g_counters[which].fetch_add(1, std::memory_order_relaxed);
}
extern "C" long long ld_counter(int which) {
  return (which >= 0 && which < LD_COUNTER_MAX) ? g_counters[which].load(std::memory_order_relaxed) : -1;
}

hipError_t ld_allow_lds_ptr(const void* kernel, size_t bytes) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  // per-thread fast path: the last grants this thread saw (launching threads do not contend on the mutex once a
  // kernel's limit has been raised on their device)
  thread_local std::map<std::pair<const void*, int>, size_t> seen;
  const auto key = std::make_pair(kernel, dev);
  const auto it = seen.find(key);
  if (it != seen.end() && bytes <= it->second) return hipSuccess;
  static std::mutex mu;
  static std::map<std::pair<const void*, int>, size_t> granted;
  std::lock_guard<std::mutex> lock(mu);
  size_t& have = granted[std::make_pair(kernel, dev)];
  if (bytes > have) {
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return e;
    have = bytes;
  }
  seen[key] = have;
  return hipSuccess;
}
