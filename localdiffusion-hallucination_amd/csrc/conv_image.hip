// Direct KxK convolution of an NCHW fp32 image with 1 or 3 channels into 32 NHWC feature
// channels: init_conv 7x7 pad 3 (ddpm.py:319,413) and the Cin=1/3 convs of the first BasicBlock
// of the conditioning encoder (unet_model.py:20,30; pad 1).  K*K*Cin <= 147 is too shallow and
// too ragged for the MFMA K dimension, so this is a VALU kernel: two output pixels per thread,
// 2 x 32 accumulators, the input halo and the transposed weights [tap][32] in LDS (broadcast reads).
// Optional epilogue: GroupNorm statistics of the result (conditioning encoder).
#include "common.hip.h"

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

// ------------------------------------------------------------------------------------------------
// init_conv 7x7 for bf16 storage on MFMA (ld_conv_stem).  The direct kernel above runs at the fp32 VALU roofline
// (147 x 32 MACs per pixel: 72-84 us at 256^2, B=8); as an implicit GEMM the same convolution is
// K = 21 kernel rows x 8 (kx padded from 7, zero weight) = 168 -> 6 K-chunks x 32 output channels.
//   * K order k = (c*7 + ky)*8 + kx: the im2col B fragment of a lane (pixel px, k-quad kq) is ONE kernel row,
//     8 consecutive words of the halo tile at a per-lane base (row 4*chunk + kq) + compile-time offsets;
//   * precision: the fp32 image and the fp32 weights are each split into bf16 hi + lo parts and three MFMAs
//     (w_hi x_hi + w_hi x_lo + w_lo x_hi) recover the product to ~2^-16 relative, fp32 accumulate -- the result
//     matches the fp32-FMA kernel far below the bf16 rounding of the stored output.  The image is split once per
//     halo pixel while it is staged (LDS word = hi | lo << 16; one v_perm_b32 per pair assembles a fragment), the
//     weights once per model by ld_pack_stem_weight;
//   * persistent workgroups (~2 per CU) keep the 24 weight fragments in registers and register-prefetch the next
//     tile's halo before the MFMAs of the current one.
constexpr int STEM_NCHK = 6;
constexpr int STEM_PACKED_U16 = 2 * STEM_NCHK * 2 * 64 * 8;      // [hi|lo][chunk][m-tile][lane][8 bf16]

__global__ void stem_pack_kernel(const float* __restrict__ w, unsigned short* __restrict__ out, int Cin) {
  constexpr int KS = 7;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 32 * STEM_NCHK * 32) return;
  const int co = i / (STEM_NCHK * 32), k = i - co * (STEM_NCHK * 32);
  const int kr = k >> 3, kx = k & 7, c = kr / KS, ky = kr - c * KS;
  const float v = (c < Cin && kx < KS) ? w[(((size_t)co * Cin + c) * KS + ky) * KS + kx] : 0.f;
  const bf16 hi = (bf16)v;
  const bf16 lo = (bf16)(v - (float)hi);
  const int ch = k >> 5, l = ((k >> 3) & 3) * 16 + (co & 15), e = k & 7, m = co >> 4;
  const int slot = ((ch * 2 + m) * 64 + l) * 8 + e;
  out[slot] = __builtin_bit_cast(unsigned short, hi);
  out[STEM_NCHK * 2 * 64 * 8 + slot] = __builtin_bit_cast(unsigned short, lo);
}

// BEGIN (ld_conv_stem_begin): the first `nbeg` workgroups in x are not convolution workgroups -- they do the head-of-step
// work (step_begin_work: zero the statistics arenas, move the step counter, copy the timestep's FiLM row) beside the
// convolution, which reads none of it: the evaluation's first launch disappears into its second.
template <typename T, bool BEGIN>
__global__ __launch_bounds__(256) void conv_stem_mfma_kernel(const float* __restrict__ x, const uint4* __restrict__ wp,
                                                             const float* __restrict__ bias, T* out, int B, int Cin,
                                                             int H, int W, int tiles_x, int ntiles, StepBeginDev sb, int nbeg) {
  constexpr int KS = 7, PAD = 3, TSX = 16, TSY = 32, HSX = TSX + KS - 1 + 1, HSY = TSY + KS - 1, NCHK = STEM_NCHK;
  constexpr int NIN = 3 * HSY * HSX, NLD = (NIN + 255) / 256;
  __shared__ unsigned s_in[NIN + 8];                     // per input pixel: bf16 hi part | bf16 lo part << 16
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, px = lane & 15, kq = lane >> 4;
  const int b = blockIdx.y;
  if constexpr (BEGIN) {
    if ((int)blockIdx.x < nbeg) {                        // (uniform per workgroup)
      __shared__ int s_t;
      step_begin_work(sb, (long)blockIdx.x * gridDim.y + blockIdx.y, (long)nbeg * gridDim.y, &s_t);
      return;
    }
  }
  const int bx = BEGIN ? (int)blockIdx.x - nbeg : (int)blockIdx.x, gdx = BEGIN ? (int)gridDim.x - nbeg : (int)gridDim.x;
  uint4 Ahi[NCHK][2], Alo[NCHK][2];                      // this lane's weight fragments, straight from the packed image
#pragma unroll
  for (int ch = 0; ch < NCHK; ++ch)
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      Ahi[ch][m] = wp[(ch * 2 + m) * 64 + lane];
      Alo[ch][m] = wp[(NCHK * 2 + ch * 2 + m) * 64 + lane];
    }
  int offb[NCHK];                                        // halo offset of this lane's kernel row per chunk
#pragma unroll
  for (int ch = 0; ch < NCHK; ++ch) {
    const int kr = ch * 4 + kq, c = kr / KS, ky = kr - c * KS;
    offb[ch] = c < 3 ? (c * HSY + ky) * HSX : 0;         // rows past the last channel carry zero weights
  }
  float4 bv[2];
#pragma unroll
  for (int m = 0; m < 2; ++m) bv[m] = *reinterpret_cast<const float4*>(bias + m * 16 + kq * 4);
  if (tid < 8) s_in[NIN + tid] = 0u;
  // halo staging is register-prefetched: tile t+1's pixels are requested before tile t's MFMAs
  float stage[NLD];
  auto request = [&](int t) {
    const int y0 = (t / tiles_x) * TSY, x0 = (t % tiles_x) * TSX;
#pragma unroll
    for (int r = 0; r < NLD; ++r) {
      const int i = r * 256 + tid;
      float v = 0.f;
      if (i < NIN) {
        const int c = i / (HSY * HSX), q = i - c * HSY * HSX, hy = q / HSX, hx = q - hy * HSX;
        const int gy = y0 - PAD + hy, gx = x0 - PAD + hx;
        if (c < Cin && gy >= 0 && gy < H && gx >= 0 && gx < W) v = x[(((size_t)b * Cin + c) * H + gy) * W + gx];
      }
      stage[r] = v;
    }
  };
  if (bx < ntiles) request(bx);
  for (int t = bx; t < ntiles; t += gdx) {
    const int y0 = (t / tiles_x) * TSY, x0 = (t % tiles_x) * TSX;
    __syncthreads();                                     // previous tile's gathers are done
#pragma unroll
    for (int r = 0; r < NLD; ++r) {
      const int i = r * 256 + tid;
      if (i < NIN) {
        const float v = stage[r];
        const bf16 hi = (bf16)v;                         // split ONCE per halo pixel, not once per im2col use
        const bf16 lo = (bf16)(v - (float)hi);
        s_in[i] = (unsigned)__builtin_bit_cast(unsigned short, hi) | ((unsigned)__builtin_bit_cast(unsigned short, lo) << 16);
      }
    }
    __syncthreads();
    if (t + gdx < ntiles) request(t + gdx);
    const int gx = x0 + px;
#pragma unroll 2
    for (int j = 0; j < TSY / 4; ++j) {
      const int row = wv * (TSY / 4) + j;
      const unsigned* pin = s_in + row * HSX + px;
      f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int ch = 0; ch < NCHK; ++ch) {
        const unsigned* pr = pin + offb[ch];
        unsigned v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = pr[e];
        // one byte-permute per pair assembles the packed hi (low halves) and lo (high halves) fragments
        const uint4 Bhi = make_uint4(__builtin_amdgcn_perm(v[1], v[0], 0x05040100u), __builtin_amdgcn_perm(v[3], v[2], 0x05040100u),
                                     __builtin_amdgcn_perm(v[5], v[4], 0x05040100u), __builtin_amdgcn_perm(v[7], v[6], 0x05040100u));
        const uint4 Blo = make_uint4(__builtin_amdgcn_perm(v[1], v[0], 0x07060302u), __builtin_amdgcn_perm(v[3], v[2], 0x07060302u),
                                     __builtin_amdgcn_perm(v[5], v[4], 0x07060302u), __builtin_amdgcn_perm(v[7], v[6], 0x07060302u));
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          mma16<bf16>(acc[m], Ahi[ch][m], Bhi);
          mma16<bf16>(acc[m], Ahi[ch][m], Blo);
          mma16<bf16>(acc[m], Alo[ch][m], Bhi);
        }
      }
      const int gy = y0 + row;
      {
        // the two m-tiles leave as ONE 16-byte store per lane (pair_frag16; only the store is predicated)
        const bool in = gy < H && gx < W;
        char* op = reinterpret_cast<char*>(out + (((size_t)b * H + (in ? gy : 0)) * W + (in ? gx : 0)) * CO);
        float r4[2][4];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          r4[m][0] = acc[m][0] + bv[m].x; r4[m][1] = acc[m][1] + bv[m].y; r4[m][2] = acc[m][2] + bv[m].z; r4[m][3] = acc[m][3] + bv[m].w;
        }
        const uint4 w16 = pair_frag16<T>(r4[0], r4[1]);
        if (in) store16_out(op + pair_frag16_off(kq), w16);
      }
    }
  }
}

template <typename T>
int run(const float* x, const float* w, const float* bias, void* out, double* ostats, int ogroups, int B,
        int Cin, int H, int W, int ks, hipStream_t st) {
  const int tiles_x = (W + TS - 1) / TS, tiles_y = (H + TH - 1) / TH;
  dim3 grid(tiles_x * tiles_y, B);
  if (ks == 7)
    LD_LAUNCH((conv_image_kernel<T, 7>), grid, dim3(256), 0, st, x, w, bias, (T*)out, ostats, ogroups, B, Cin, H, W, tiles_x);
  else
    LD_LAUNCH((conv_image_kernel<T, 3>), grid, dim3(256), 0, st, x, w, bias, (T*)out, ostats, ogroups, B, Cin, H, W, tiles_x);
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
  LD_REQUIRE(ld_dtype_ok(dtype), "ld_conv_image: bad dtype %d", dtype);
  return LD_DISPATCH(dtype, run<T>(x, w, bias, out, out_stats, out_groups, B, Cin, H, W, ksize, st));
}

extern "C" size_t ld_stem_packed_bytes(void) { return (size_t)STEM_PACKED_U16 * sizeof(unsigned short); }

extern "C" int ld_pack_stem_weight(const float* w_oihw, void* out_packed, int Cin, void* stream) {
  LD_REQUIRE(w_oihw && out_packed, "ld_pack_stem_weight: null pointer");
  LD_REQUIRE(Cin >= 1 && Cin <= 3, "ld_pack_stem_weight: Cin %d (1..3)", Cin);
  LD_LAUNCH(stem_pack_kernel, dim3((32 * STEM_NCHK * 32 + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     w_oihw, (unsigned short*)out_packed, Cin);
  LD_LAUNCH_CHECK("pack_stem_weight");
  return LD_OK;
}

namespace {
int stem_launch(const float* x, const void* w_packed, const float* bias, void* out, int B, int Cin, int H, int W, int dtype,
                const StepBeginDev* sb, void* stream) {
  LD_REQUIRE(x && w_packed && bias && out, "ld_conv_stem: null pointer");
  LD_REQUIRE(ld_dtype_16(dtype), "ld_conv_stem: 16-bit storage only (fp32 uses ld_conv_image), got dtype %d", dtype);
  LD_REQUIRE(Cin >= 1 && Cin <= 3 && B > 0 && H > 0 && W > 0, "ld_conv_stem: bad shape (Cin %d)", Cin);
  const int tiles_x = (W + 15) / 16, tiles_y = (H + 31) / 32, ntiles = tiles_x * tiles_y;
  int G = (512 + B - 1) / B;                             // ~2 persistent workgroups per CU
  if (G > ntiles) G = ntiles;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (sb) {
    // head-of-step workgroups: 8 stores of 16 bytes per thread, at most 8 workgroups per image
    long nbeg = ((sb->na + sb->nb) / (256L * 8) + B) / B;
    if (nbeg < 1) nbeg = 1;
    if (nbeg > 8) nbeg = 8;
    LD_DISPATCH16(dtype, [&] {
      LD_LAUNCH((conv_stem_mfma_kernel<T, true>), dim3(G + (int)nbeg, B), dim3(256), 0, st, x, (const uint4*)w_packed, bias, (T*)out,
                B, Cin, H, W, tiles_x, ntiles, *sb, (int)nbeg);
      return 0;
    }());
  } else {
    const StepBeginDev none{};
    LD_DISPATCH16(dtype, [&] {
      LD_LAUNCH((conv_stem_mfma_kernel<T, false>), dim3(G, B), dim3(256), 0, st, x, (const uint4*)w_packed, bias, (T*)out,
                B, Cin, H, W, tiles_x, ntiles, none, 0);
      return 0;
    }());
  }
  LD_LAUNCH_CHECK("conv_stem");
  return LD_OK;
}
}  // namespace

extern "C" int ld_conv_stem(const float* x, const void* w_packed, const float* bias, void* out, int B, int Cin, int H,
                            int W, int dtype, void* stream) {
  return stem_launch(x, w_packed, bias, out, B, Cin, H, W, dtype, nullptr, stream);
}

extern "C" int ld_conv_stem_begin(const float* x, const void* w_packed, const float* bias, void* out, int B, int Cin, int H,
                                  int W, int dtype, const ld_step_begin_args* g, void* stream) {
  LD_REQUIRE(g != nullptr, "ld_conv_stem_begin: null step-begin arguments");
  if (int rc = ld_step_begin_check(g->zero_a, g->bytes_a, g->zero_b, g->bytes_b, g->t_ptr, g->idx_ptr, g->t_table, g->film_rows,
                                   g->row_floats, g->film_cur)) return rc;
  const StepBeginDev sb{(uint4*)g->zero_a, (long)(g->bytes_a / 16), (uint4*)g->zero_b, (long)(g->bytes_b / 16), g->t_ptr, g->delta,
                        g->idx_ptr, g->t_table, g->film_rows, g->row_floats, g->film_cur};
  return stem_launch(x, w_packed, bias, out, B, Cin, H, W, dtype, &sb, stream);
}
