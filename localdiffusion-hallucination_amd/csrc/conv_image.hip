// Direct KxK convolution of an NCHW fp32 image with 1 or 3 channels into 32 NHWC feature
// channels: init_conv 7x7 pad 3 (ddpm.py:319,413) and the Cin=1/3 convs of the first BasicBlock
// of the conditioning encoder (unet_model.py:20,30; pad 1).  K*K*Cin <= 147 is too shallow and
// too ragged for the MFMA K dimension, so this is a VALU kernel: two output pixels per thread,
// 2 x 32 accumulators, the input halo and the transposed weights [tap][32] in LDS (broadcast reads).
// Optional epilogue: GroupNorm statistics of the result (conditioning encoder).
#include "common.cuh"

namespace {
constexpr int TS = 16;       // tile = 16 columns x 32 rows, 256 threads, TWO output rows per thread
constexpr int TH = 32;       // (one weight read from LDS feeds 2 x 32 FMAs: VALU-bound instead of LDS-bound)
constexpr int CO = 32;

template <typename T, int KS>
__global__ __launch_bounds__(256) void conv_image_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ bias, T* out, double* ostats,
                                                         int ogroups, int B, int Cin, int H, int W, int tiles_x) {
  constexpr int PAD = KS / 2, HSX = TS + KS - 1, HSY = TH + KS - 1;
  __shared__ float s_in[3 * HSY * HSX];
  __shared__ __attribute__((aligned(16))) float s_wt[3 * KS * KS * CO];   // [c*KS*KS + tap][o]
  __shared__ double s_red[2 * CO];
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const int b = blockIdx.y;
  const int y0 = (blockIdx.x / tiles_x) * TH, x0 = (blockIdx.x % tiles_x) * TS;
  for (int i = tid; i < Cin * HSY * HSX; i += 256) {
    const int c = i / (HSY * HSX), r = i - c * HSY * HSX, hy = r / HSX, hx = r - hy * HSX;
    const int gy = y0 - PAD + hy, gx = x0 - PAD + hx;
    float v = 0.f;
    if (gy >= 0 && gy < H && gx >= 0 && gx < W) v = x[(((size_t)b * Cin + c) * H + gy) * W + gx];
    s_in[i] = v;
  }
  const int ktot = Cin * KS * KS;
  for (int i = tid; i < ktot * CO; i += 256) {          // OIHW -> [k][o]
    const int o = i / ktot, k = i - o * ktot;
    s_wt[k * CO + o] = w[i];
  }
  if (tid < 2 * CO) s_red[tid] = 0.0;
  __syncthreads();
  float acc0[CO], acc1[CO];
#pragma unroll
  for (int o = 0; o < CO; ++o) acc0[o] = acc1[o] = bias[o];
  for (int c = 0; c < Cin; ++c)
    for (int ky = 0; ky < KS; ++ky)
#pragma unroll
      for (int kx = 0; kx < KS; ++kx) {
        const float v0 = s_in[(c * HSY + ty + ky) * HSX + tx + kx];
        const float v1 = s_in[(c * HSY + ty + 16 + ky) * HSX + tx + kx];
        const float4* wp = reinterpret_cast<const float4*>(s_wt + ((c * KS + ky) * KS + kx) * CO);
#pragma unroll
        for (int o4 = 0; o4 < CO / 4; ++o4) {
          const float4 wv = wp[o4];                      // same address in every lane: LDS broadcast
          acc0[4 * o4 + 0] = fmaf(v0, wv.x, acc0[4 * o4 + 0]); acc1[4 * o4 + 0] = fmaf(v1, wv.x, acc1[4 * o4 + 0]);
          acc0[4 * o4 + 1] = fmaf(v0, wv.y, acc0[4 * o4 + 1]); acc1[4 * o4 + 1] = fmaf(v1, wv.y, acc1[4 * o4 + 1]);
          acc0[4 * o4 + 2] = fmaf(v0, wv.z, acc0[4 * o4 + 2]); acc1[4 * o4 + 2] = fmaf(v1, wv.z, acc1[4 * o4 + 2]);
          acc0[4 * o4 + 3] = fmaf(v0, wv.w, acc0[4 * o4 + 3]); acc1[4 * o4 + 3] = fmaf(v1, wv.w, acc1[4 * o4 + 3]);
        }
      }
  const int gx = x0 + tx;
  const bool valid0 = (y0 + ty) < H && gx < W, valid1 = (y0 + ty + 16) < H && gx < W;
  if (valid0) {
    T* op = out + (((size_t)b * H + y0 + ty) * W + gx) * CO;
#pragma unroll
    for (int o = 0; o < CO; o += 4) store4<T>(op + o, acc0 + o);
  }
  if (valid1) {
    T* op = out + (((size_t)b * H + y0 + ty + 16) * W + gx) * CO;
#pragma unroll
    for (int o = 0; o < CO; o += 4) store4<T>(op + o, acc1 + o);
  }
  if (ostats) {
    // per-channel sums over the tile: wave shuffle tree, then LDS, then one fp64 atomic per group
#pragma unroll
    for (int o = 0; o < CO; ++o) {
      double s1 = (valid0 ? (double)acc0[o] : 0.0) + (valid1 ? (double)acc1[o] : 0.0);
      double s2 = (valid0 ? (double)acc0[o] * (double)acc0[o] : 0.0) + (valid1 ? (double)acc1[o] * (double)acc1[o] : 0.0);
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) { s1 += __shfl_xor(s1, d); s2 += __shfl_xor(s2, d); }
      if ((tid & 63) == 0) { atomicAdd(&s_red[o], s1); atomicAdd(&s_red[CO + o], s2); }
    }
    __syncthreads();
    const int gs = CO / ogroups;
    if (tid < ogroups) {
      double s1 = 0.0, s2 = 0.0;
      for (int c = 0; c < gs; ++c) { s1 += s_red[tid * gs + c]; s2 += s_red[CO + tid * gs + c]; }
      const int stripe = blockIdx.x % LD_STAT_STRIPES;
      atomicAdd(&ostats[(((size_t)b * LD_STAT_STRIPES + stripe) * ogroups + tid) * 2 + 0], s1);
      atomicAdd(&ostats[(((size_t)b * LD_STAT_STRIPES + stripe) * ogroups + tid) * 2 + 1], s2);
    }
  }
}

template <typename T>
int run(const float* x, const float* w, const float* bias, void* out, double* ostats, int ogroups, int B,
        int Cin, int H, int W, int ks, hipStream_t st) {
  const int tiles_x = (W + TS - 1) / TS, tiles_y = (H + TH - 1) / TH;
  dim3 grid(tiles_x * tiles_y, B);
  if (ks == 7)
    hipLaunchKernelGGL((conv_image_kernel<T, 7>), grid, dim3(256), 0, st, x, w, bias, (T*)out, ostats, ogroups, B, Cin, H, W, tiles_x);
  else
    hipLaunchKernelGGL((conv_image_kernel<T, 3>), grid, dim3(256), 0, st, x, w, bias, (T*)out, ostats, ogroups, B, Cin, H, W, tiles_x);
  LD_LAUNCH_CHECK("conv_image");
  return LD_OK;
}
}  // namespace

extern "C" int ld_conv_image(const float* x, const float* w, const float* bias, void* out, double* out_stats,
                             int out_groups, int B, int Cin, int H, int W, int ksize, int dtype, void* stream) {
  LD_REQUIRE(x && w && bias && out, "ld_conv_image: null pointer");
  LD_REQUIRE(Cin >= 1 && Cin <= 3, "ld_conv_image: Cin %d (1..3)", Cin);
  LD_REQUIRE(ksize == 3 || ksize == 7, "ld_conv_image: ksize %d (3 or 7)", ksize);
  LD_REQUIRE(!out_stats || (out_groups > 0 && 32 % out_groups == 0), "ld_conv_image: out_groups %d", out_groups);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == LD_F32) return run<float>(x, w, bias, out, out_stats, out_groups, B, Cin, H, W, ksize, st);
  if (dtype == LD_BF16) return run<bf16>(x, w, bias, out, out_stats, out_groups, B, Cin, H, W, ksize, st);
  return ld_fail(LD_EINVAL, "ld_conv_image: bad dtype %d", dtype);
}
