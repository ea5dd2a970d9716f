// Probe: does a tensor written by one launch stay readable from the SAME XCD's L2 in the next launch?
// Writer: block b writes segment b.  Reader: block b reads segment (b + shift) % nblk.  Blocks b and b+8 share an XCD
// (round-robin placement), so shift 8 = same XCD / other CU, shift 1 = other XCD.  If kernel boundaries invalidate
// L2 (or lines are not kept), both shifts read at the same rate.
// build: hipcc --offload-arch=gfx950 -O3 xcd_reuse.hip -o xcd_reuse
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void writer(uint4* buf, int seg16) {
  uint4* p = buf + (size_t)blockIdx.x * seg16;
  for (int i = threadIdx.x; i < seg16; i += blockDim.x) p[i] = make_uint4(i, blockIdx.x, 3, 4);
}
__global__ void reader(const uint4* buf, int seg16, int shift, unsigned* sink) {
  const int s = (blockIdx.x + shift) % gridDim.x;
  const uint4* p = buf + (size_t)s * seg16;
  unsigned acc = 0;
  for (int i = threadIdx.x; i < seg16; i += blockDim.x) { uint4 v = p[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
  if (acc == 0x12345678u) sink[0] = acc;
}
__global__ void dummy(unsigned* sink) { if (sink[0] == 0x7777u) sink[0] = 1; }
__global__ void xcc_of_block0(unsigned* out) {
  if (threadIdx.x == 0) { unsigned x; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x)); out[blockIdx.x] = x & 15; }
}
int main() {
  const int nblk = 1024;
  {  // which XCD do blocks 0..15 of consecutive launches land on, with grids of 16, 16, 3, 16, 1, 16, 8, 16 blocks?
    unsigned* d; hipMalloc(&d, 64 * 4); unsigned h[16];
    const int grids[8] = {16, 16, 3, 16, 1, 16, 8, 16};
    for (int k = 0; k < 8; ++k) {
      hipMemset(d, 0xff, 64 * 4);
      hipLaunchKernelGGL(xcc_of_block0, dim3(grids[k]), dim3(64), 0, 0, d);
      hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
      printf("launch %d grid %2d: xcc of blocks:", k, grids[k]);
      for (int i = 0; i < grids[k] && i < 16; ++i) printf(" %u", h[i]);
      printf("\n");
    }
  }
  for (int segkb : {32}) {
    const int seg16 = segkb * 1024 / 16;
    uint4* buf; unsigned* sink;
    hipMalloc(&buf, (size_t)nblk * seg16 * 16); hipMalloc(&sink, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int nd : {0, 1, 3, 8})
    for (int shift : {0, 1, 5, 7}) {
      float tot = 0;
      for (int rep = 0; rep < 20; ++rep) {
        hipLaunchKernelGGL(writer, dim3(nblk), dim3(256), 0, 0, buf, seg16);
        if (nd) hipLaunchKernelGGL(dummy, dim3(nd), dim3(64), 0, 0, sink);
        hipEventRecord(e0);
        hipLaunchKernelGGL(reader, dim3(nblk), dim3(256), 0, 0, buf, seg16, shift, sink);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 4) tot += ms;
      }
      const double us = tot / 16 * 1e3, mb = (double)nblk * segkb / 1024;
      printf("dummy grid %d  total %6.1f MB (seg %3d KB x %d blocks) shift %3d: read %7.2f us  %7.2f TB/s\n", nd, mb, segkb, nblk, shift, us, mb / us);
    }
    hipFree(buf); hipFree(sink);
  }
  return 0;
}
