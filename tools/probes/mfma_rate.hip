// Probe (gfx950): cycles per v_mfma_f32_16x16x32_bf16 in the conv inner-loop shape.
//   mode 0: 72 MFMAs on 8 accumulators, operands in registers, no LDS
//   mode 1: + 48 ds_read_b128 per 72 MFMAs, reads of the next 24-MFMA group issued before the group (conv3x3 loop)
//   mode 2: as 1 with 4 extra idle waves in the workgroup (wave-specialised layout)
// One workgroup per CU on every CU; prints cycles per 72-MFMA item of workgroup 0, wave 0.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
__global__ __launch_bounds__(512) void k(int mode, int items, unsigned long long* out, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < 48 * 1024 / 16; i += blockDim.x) ((uint4*)lds)[i] = make_uint4(i, i * 3, i * 5, i * 7);
  __syncthreads();
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
  unsigned long long t0 = 0, t1 = 0;
  if (wid < 4) {
    uint4 A[2][12], B[2][4];
    for (int i = 0; i < 12; ++i) A[0][i] = A[1][i] = make_uint4(lane, i, 3, 4);
    for (int i = 0; i < 4; ++i) B[0][i] = B[1][i] = make_uint4(lane * 7, i, 5, 6);
    const char* wb = lds + lane * 16;
    t0 = __builtin_readcyclecounter();
    for (int it = 0; it < items; ++it) {
      if (mode >= 1) {
        for (int i = 0; i < 12; ++i) A[0][i] = *(const uint4*)(wb + i * 1024);
        for (int i = 0; i < 4; ++i) B[0][i] = *(const uint4*)(wb + 36864 + i * 1024);
      }
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        if (mode >= 1 && dx < 2) {
          for (int i = 0; i < 12; ++i) A[(dx + 1) & 1][i] = *(const uint4*)(wb + ((dx + 1) * 12 + i) * 1024);
          for (int i = 0; i < 4; ++i) B[(dx + 1) & 1][i] = *(const uint4*)(wb + 36864 + ((dx + 1) * 4 + i) * 1024);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int rr = 0; rr < 4; ++rr)
#pragma unroll
          for (int dy = 0; dy < 3; ++dy) {
            const int j = rr - dy;
            if (j >= 0 && j < 2)
#pragma unroll
              for (int m = 0; m < 4; ++m)
                acc[m * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, A[dx & 1][dy * 4 + m]),
                                                                         __builtin_bit_cast(bf16x8, B[dx & 1][rr]), acc[m * 2 + j], 0, 0, 0);
          }
      }
    }
    t1 = __builtin_readcyclecounter();
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  sink[blockIdx.x * blockDim.x + tid] = s;
  if (blockIdx.x == 0 && tid == 0) out[0] = t1 - t0;
}
int main() {
  unsigned long long* d; float* sink;
  hipMalloc(&d, 8); hipMalloc(&sink, 256 * 512 * 4);
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
  const int items = 64;
  for (int mode = 0; mode < 3; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      hipLaunchKernelGGL(k, dim3(256), dim3(mode == 2 ? 512 : 256), 64 * 1024, 0, mode, items, d, sink);
      hipDeviceSynchronize();
    }
    unsigned long long h; hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    printf("mode %d: %.1f cycles per 72-MFMA item = %.2f cycles per MFMA\n", mode, (double)h / items, (double)h / items / 72);
  }
  return 0;
}
