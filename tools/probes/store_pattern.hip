// Probe (gfx950): write bandwidth of the conv epilogue's store pattern.  256 workgroups x 512 threads write a
// [8][256][256][32] bf16 tensor (33.5 MB) in tiles of ROWS x COLS pixels (one 16-B store per lane: lane = pixel
// x 4 channel-quads, a wave covers 16 consecutive pixels of one row), tiles assigned like the persistent conv
// (workgroup g of image b walks tiles g, g+32, ...).  Shapes: 16x16, 8x32, 4x64, 2x128, 1x256 (= linear rows).
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int ROWS, int COLS>
__global__ __launch_bounds__(512) void k(uint4* out, int iters_pad) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, px = lane & 15, kq = lane >> 4;
  const int b = blockIdx.y, G = gridDim.x;
  constexpr int TX = 256 / COLS, NT = (256 / ROWS) * TX, SEG = COLS / 16;   // 16-pixel segments per tile row
  const uint4 v = make_uint4(tid, wv, px, kq);
  for (int t = blockIdx.x; t < NT; t += G) {
    const int ty0 = (t / TX) * ROWS, tx0 = (t % TX) * COLS;
    // 256 pixels per tile = 16 wave-stores; wave wv takes stores wv and wv + 8
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int u = wv + 8 * s, row = u / SEG, seg = u % SEG;
      const int gy = ty0 + row, gx = tx0 + seg * 16 + px;
      out[((size_t)(b * 256 + gy) * 256 + gx) * 4 + kq] = v;
    }
    if (iters_pad) __builtin_amdgcn_s_barrier();
  }
}
template <int ROWS, int COLS>
void run(uint4* d) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int pad = 0; pad < 2; ++pad) {
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k<ROWS, COLS>), dim3(32, 8), dim3(512), 0, 0, d, pad);
    hipEventRecord(e0);
    for (int r = 0; r < 50; ++r) hipLaunchKernelGGL((k<ROWS, COLS>), dim3(32, 8), dim3(512), 0, 0, d, pad);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("tile %3dx%-3d barrier=%d: %6.1f us  %5.2f TB/s\n", ROWS, COLS, pad, ms * 1e3 / 50, 33.55e6 / (ms * 1e-3 / 50) / 1e12);
  }
}
int main() {
  uint4* d; (void)hipMalloc(&d, 8ull * 256 * 256 * 64);
  run<16, 16>(d); run<8, 32>(d); run<4, 64>(d); run<2, 128>(d); run<1, 256>(d);
  return 0;
}
