// GPU box: how many streams really run at the same time (DESIGN finding 40)?  S streams each replay a graph of N
// one-workgroup kernels of ~34 us; if all S run concurrently the time per kernel per stream stays ~34 us.
// usage: queue_concurrency [prio]   prio=1: streams cycle through the priority levels the device offers
// build: hipcc --offload-arch=gfx950 -O2 -o tools/probes/queue_concurrency tools/probes/queue_concurrency.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void tiny(float* p, int spin) {
  float v = p[threadIdx.x];
  for (int i = 0; i < spin; ++i) v = v * 1.0001f + 0.5f;
  if (v == 12345.678f) p[threadIdx.x] = v;
}

int main(int argc, char** argv) {
  const int N = 106, REP = 60, spin = 2000;
  int prio = argc > 1 ? atoi(argv[1]) : 0;
  int lo = 0, hi = 0;
  CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  printf("priority range: least %d greatest %d; mode %s\n", lo, hi, prio ? "cycling priorities" : "default priority");
  float* buf; CK(hipMalloc(&buf, 1 << 20));
  for (int S : {1, 2, 3, 4, 6, 8, 12, 16}) {
    std::vector<hipStream_t> st(S);
    std::vector<hipGraphExec_t> ex(S);
    for (int s = 0; s < S; ++s) {
      int span = lo - hi + 1;
      if (prio) CK(hipStreamCreateWithPriority(&st[s], hipStreamNonBlocking, hi + (s % span)));
      else CK(hipStreamCreateWithFlags(&st[s], hipStreamNonBlocking));
      hipGraph_t g;
      CK(hipStreamBeginCapture(st[s], hipStreamCaptureModeThreadLocal));
      for (int i = 0; i < N; ++i) tiny<<<1, 256, 0, st[s]>>>(buf + s * 4096, spin);
      CK(hipStreamEndCapture(st[s], &g));
      CK(hipGraphInstantiate(&ex[s], g, nullptr, nullptr, 0));
      CK(hipGraphDestroy(g));
    }
    auto run = [&](int rep) { for (int r = 0; r < rep; ++r) for (int s = 0; s < S; ++s) (void)hipGraphLaunch(ex[s], st[s]); };
    run(5); CK(hipDeviceSynchronize());
    auto t0 = std::chrono::steady_clock::now();
    run(REP); CK(hipDeviceSynchronize());
    double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    printf("S=%2d: %.1f us per kernel per stream -> %.2f streams running at a time\n", S, us / (REP * N), 34.2 * S / (us / (REP * N)));
    for (int s = 0; s < S; ++s) { (void)hipGraphExecDestroy(ex[s]); (void)hipStreamDestroy(st[s]); }
  }
  return 0;
}
