// GPU box: what can the CUs pull out of L2 (DESIGN finding 41)?  Every workgroup streams the same R-byte region
// (L2-resident, far larger than the 32 KB vL1D) with 16-byte loads, U loads in flight per thread.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/l2_bw tools/probes/l2_bw.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int U>
__global__ __launch_bounds__(256) void stream_kernel(const uint4* __restrict__ p, long n16, int reps, uint4* sink) {
  uint4 acc = make_uint4(0, 0, 0, 0);
  const long stride = 256L * U;
  for (int r = 0; r < reps; ++r) {
    // workgroups start at different offsets so that they do not march through the channels in lockstep
    long base = ((long)blockIdx.x * 4099 * 256) % n16;
    for (long i = 0; i < n16; i += stride) {
      uint4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        long j = base + i + u * 256 + threadIdx.x;
        if (j >= n16) j -= n16;
        v[u] = p[j];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) { acc.x ^= v[u].x; acc.y += v[u].y; acc.z ^= v[u].z; acc.w += v[u].w; }
    }
  }
  if (acc.x == 0x12345678u && acc.y == 0x9abcdefu) sink[0] = acc;
}

template <int U>
int run(const uint4* buf, long bytes, int wgs, uint4* sink) {
  const long n16 = bytes / 16;
  const int reps = (int)((64L << 20) / bytes) + 1;
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  stream_kernel<U><<<wgs, 256>>>(buf, n16, 1, sink);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  stream_kernel<U><<<wgs, 256>>>(buf, n16, reps, sink);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  const double tb = (double)wgs * reps * bytes / (ms * 1e-3) / 1e12;
  printf("region %6ld KB  %4d workgroups x 256 thr, %d loads in flight per thread: %6.2f TB/s  (%.1f B/clk/CU at 2.4 GHz, 256 CUs)\n",
         bytes >> 10, wgs, U, tb, tb * 1e12 / 256 / 2.4e9);
  return 0;
}

int main() {
  uint4* buf; uint4* sink;
  CK(hipMalloc(&buf, 64 << 20)); CK(hipMalloc(&sink, 64));
  CK(hipMemset(buf, 1, 64 << 20));
  for (long kb : {256L, 1024L, 2048L, 16384L}) for (int wgs : {256, 512, 1024, 2048}) {
    if (run<4>(buf, kb << 10, wgs, sink)) return 1;
    if (run<8>(buf, kb << 10, wgs, sink)) return 1;
  }
  return 0;
}
