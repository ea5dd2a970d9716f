// GPU box: how many dependent kernel dispatches per second does the front end sustain, as a function of the number
// of streams (DESIGN finding 40)?  Chains of N tiny kernels, replayed as HIP graphs or launched eagerly, on S streams.
// build: hipcc --offload-arch=gfx950 -O2 -o /tmp/dispatch_rate tools/probes/dispatch_rate.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void tiny(float* p, int spin) {
  float v = p[threadIdx.x];
  for (int i = 0; i < spin; ++i) v = v * 1.0001f + 0.5f;
  if (v == 12345.678f) p[threadIdx.x] = v;
}

int main(int argc, char** argv) {
  const int N = 106, REP = 300;
  float* buf; CK(hipMalloc(&buf, 1 << 20));
  for (int wgs : {1, 256, 2048}) for (int spin : {0, 2000}) for (int mode = 0; mode < 2; ++mode) for (int S : {1, 2, 4, 8}) {
    std::vector<hipStream_t> st(S);
    std::vector<hipGraphExec_t> ex(S);
    for (int s = 0; s < S; ++s) {
      CK(hipStreamCreateWithFlags(&st[s], hipStreamNonBlocking));
      if (mode == 0) {
        hipGraph_t g;
        CK(hipStreamBeginCapture(st[s], hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < N; ++i) tiny<<<wgs, 256, 0, st[s]>>>(buf + s * 4096, spin);
        CK(hipStreamEndCapture(st[s], &g));
        CK(hipGraphInstantiate(&ex[s], g, nullptr, nullptr, 0));
        CK(hipGraphDestroy(g));
      }
    }
    auto run = [&](int rep) {
      for (int r = 0; r < rep; ++r) for (int s = 0; s < S; ++s) {
        if (mode == 0) (void)hipGraphLaunch(ex[s], st[s]);
        else for (int i = 0; i < N; ++i) tiny<<<wgs, 256, 0, st[s]>>>(buf + s * 4096, spin);
      }
    };
    run(20); CK(hipDeviceSynchronize());
    auto t0 = std::chrono::steady_clock::now();
    run(REP); CK(hipDeviceSynchronize());
    double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    printf("wgs=%4d spin=%4d %s S=%d: %.2f us per kernel per stream, %.2f us per kernel overall (%.0f k dispatches/s)\n", wgs, spin,
           mode == 0 ? "graph" : "eager", S, us / (REP * N), us / (REP * N * S), 1e3 * REP * N * S / us);
    for (int s = 0; s < S; ++s) { if (mode == 0) (void)hipGraphExecDestroy(ex[s]); (void)hipStreamDestroy(st[s]); }
  }
  return 0;
}
