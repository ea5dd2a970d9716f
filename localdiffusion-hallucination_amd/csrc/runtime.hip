// Runtime plumbing of the C-ABI: error string, device info, HIP-graph capture, events.
#include "common.cuh"

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

extern "C" int ld_memset_zero(void* ptr, size_t bytes, void* stream) {
  LD_REQUIRE(ptr || bytes == 0, "ld_memset_zero: null");
  if (bytes) LD_HIP(hipMemsetAsync(ptr, 0, bytes, reinterpret_cast<hipStream_t>(stream)));
  return LD_OK;
}

extern "C" int ld_stream_wait_event(void* stream, void* ev) {
  LD_REQUIRE(ev, "ld_stream_wait_event: null event");
  LD_HIP(hipStreamWaitEvent(reinterpret_cast<hipStream_t>(stream), reinterpret_cast<hipEvent_t>(ev), 0));
  return LD_OK;
}
