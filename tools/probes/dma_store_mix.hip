// Probe (gfx950): do LDS-DMA reads and global stores of the same workgroup overlap?  256 workgroups x 640
// threads, 8 "tiles" each (workgroup-strided): per tile, mode bit 0: waves 8-9 stream a 21-KiB input chunk into
// a 6-slot LDS ring with LDS-DMA (3 tiles in flight, counted vmcnt); mode bit 1: waves 0-7 write a 16-KiB
// output chunk (2 x 16-B-per-lane stores each); mode bit 2: the reads are plain global_load_dwordx4 into VGPRs
// by waves 0-7 instead (5.25 per lane ~ 6).  One barrier per tile in every mode.
#include <hip/hip_runtime.h>
#include <stdio.h>
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int MODE>
__global__ __launch_bounds__(640) void k(const uint4* in, uint4* out, int* sink) {
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntile = 8, R = 6;
  const unsigned base = (unsigned)(size_t)lds;
  unsigned acc = 0;
  auto dma = [&](int t) {   // tile t of this workgroup: 21 blocks of 1 KiB, wave 8 takes even, wave 9 odd blocks
    const uint4* src = in + ((size_t)(blockIdx.x + t * 256) * 21 * 64);
    for (int r = 0; r < 11; ++r) {
      const int blk = r * 2 + (wv - 8);
      if (blk < 21) glds16(src + blk * 64 + lane, __builtin_amdgcn_readfirstlane(base + (t % R) * 21 * 1024 + blk * 1024));
    }
  };
  if ((MODE & 1) && wv >= 8) { for (int t = 0; t < R - 1; ++t) dma(t); }
  for (int t = 0; t < ntile; ++t) {
    if (wv >= 8) {
      if (MODE & 1) {
        if (t + R - 1 < ntile) dma(t + R - 1);
        // wait until tile t+2 landed: younger tiles t+3..min(t+R-1, 7)
        const int last = t + R - 1 < ntile - 1 ? t + R - 1 : ntile - 1;
        const int keep = last - (t + 2) > 0 ? (last - (t + 2)) * (wv == 8 ? 11 : 10) : 0;
        if (keep >= 30) asm volatile("s_waitcnt vmcnt(30)" ::: "memory");
        else if (keep >= 20) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
        else if (keep >= 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    } else {
      if (MODE & 4) {
        const uint4* src = in + ((size_t)(blockIdx.x + t * 256) * 21 * 64);
        for (int r = 0; r < 3; ++r) { const int blk = r * 8 + wv; if (blk < 21) { uint4 v = src[blk * 64 + lane]; acc += v.x ^ v.y ^ v.z ^ v.w; } }
      }
      if (MODE & 2) {
        uint4* dst = out + ((size_t)(blockIdx.x + t * 256) * 16 * 64);
        const uint4 v = make_uint4(tid, t, acc, 3);
        dst[(wv * 2 + 0) * 64 + lane] = v;
        dst[(wv * 2 + 1) * 64 + lane] = v;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  if (acc == 0x12345678u) sink[tid] = acc;
}
template <int MODE> void run(const uint4* in, uint4* out, int* sink, const char* what) {
  hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 6 * 21 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(640), 6 * 21 * 1024, 0, in, out, sink);
  hipEventRecord(e0);
  for (int r = 0; r < 50; ++r) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(640), 6 * 21 * 1024, 0, in, out, sink);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-44s %6.1f us\n", what, ms * 1e3 / 50);
}
int main() {
  uint4 *in, *out; int* sink;
  (void)hipMalloc(&in, 256ull * 8 * 21 * 1024); (void)hipMalloc(&out, 256ull * 8 * 16 * 1024); (void)hipMalloc(&sink, 4096);
  (void)hipMemset(in, 1, 256ull * 8 * 21 * 1024);
  run<0>(in, out, sink, "barriers only");
  run<1>(in, out, sink, "LDS-DMA reads (44 MB)");
  run<2>(in, out, sink, "stores (33.5 MB)");
  run<3>(in, out, sink, "LDS-DMA reads + stores");
  run<4>(in, out, sink, "register loads (44 MB)");
  run<6>(in, out, sink, "register loads + stores");
  return 0;
}
