// Probe (gfx950): duration floor of a 256-workgroup launch as a function of block size, dynamic LDS and a
// minimal body (mode 0: empty; 1: 18 KB global->LDS copy + barrier; 2: + 8 barriers with a little ALU).
// Run under `rocprofv3 --kernel-trace --output-format csv`; the kernel name encodes nothing, read the order.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(512) void k(int mode, const uint4* src, int* sink) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x;
  int acc = tid;
  if (mode >= 1) {
    for (int u = tid; u < 18 * 64; u += blockDim.x) ((uint4*)lds)[u] = src[u];
    __syncthreads();
    acc += ((int*)lds)[tid];
  }
  if (mode >= 2) {
    for (int i = 0; i < 8; ++i) {
      acc = acc * 3 + i;
      __builtin_amdgcn_s_barrier();
    }
  }
  if (acc == 0x7fffffff) sink[blockIdx.x] = acc;
}
int main() {
  uint4* src; int* sink;
  (void)hipMalloc(&src, 1 << 20); (void)hipMalloc(&sink, 4096);
  (void)hipMemset(src, 1, 1 << 20);
  (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const int ldss[3] = {1024, 40 * 1024, 147 * 1024};
  for (int threads = 256; threads <= 512; threads += 256)
    for (int li = 0; li < 3; ++li)
      for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < 20; ++rep) hipLaunchKernelGGL(k, dim3(32, 8), dim3(threads), ldss[li], 0, mode, src, sink);
        (void)hipDeviceSynchronize();
        printf("cfg threads=%d lds=%d mode=%d\n", threads, ldss[li], mode);
      }
  return 0;
}
