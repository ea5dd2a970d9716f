// Probe (gfx950): cycles per workgroup barrier for 4- and 8-wave workgroups, one workgroup per CU,
// with and without a little scalar/vector work and an LDS write between barriers.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(512) void k(int mode, int iters, unsigned long long* out, int* sink) {
  __shared__ int lds[1024];
  const int tid = threadIdx.x;
  int acc = tid;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
    if (mode >= 1) { lds[(tid + i) & 1023] = acc; acc = acc * 3 + i; }
    if (mode >= 2) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (mode >= 3) acc += lds[(tid * 7 + i) & 1023];
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  sink[blockIdx.x * blockDim.x + tid] = acc;
  if (blockIdx.x == 0 && tid == 0) out[0] = t1 - t0;
}
int main() {
  unsigned long long* d; int* sink;
  (void)hipMalloc(&d, 8); (void)hipMalloc(&sink, 256 * 512 * 4);
  const int iters = 1000;
  for (int threads = 256; threads <= 512; threads += 256)
    for (int mode = 0; mode < 4; ++mode) {
      for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k, dim3(256), dim3(threads), 0, 0, mode, iters, d, sink); (void)hipDeviceSynchronize(); }
      unsigned long long h; (void)hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
      printf("threads %d mode %d: %.1f cycles per iteration\n", threads, mode, (double)h / iters);
    }
  return 0;
}
