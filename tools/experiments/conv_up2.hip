// Upsample (nn.Upsample(scale_factor = 2, mode = 'nearest') + Conv2d(3x3, padding 1), ddpm.py:113-118) as FOUR 2x2
// convolutions of the LOW-resolution map, 16-bit storage, gfx950.
//
// A 3x3 convolution over a nearest-neighbour x2 upsampled map reads, for output pixel (2i + a, 2j + b), hi-res rows 2i + a - 1 .. 2i + a + 1:
//   a = 0: low rows i-1 (tap dy = 0), i (dy = 1 and 2);     a = 1: low rows i (dy = 0 and 1), i+1 (dy = 2)
// (and the same for columns; the zero padding of the hi-res map is exactly the out-of-range low rows / columns).  So each of the four
// output PHASES (a, b) is a 2x2 convolution of the low-res map whose weights are sums of the 3x3 kernel's taps,
//   W_ab[u][v] = sum_{dy in R_a[u]} sum_{dx in R_b[v]} W[dy][dx],   R_0 = {0}, {1, 2};  R_1 = {0, 1}, {2},
// i.e. 16 tap-products per low-res pixel instead of the 4 x 9 = 36 the generic kernel (conv3x3.hip, `upsample` source flag) issues
// for the same four outputs, and ONE halo tile of the low-res map in LDS instead of the four times larger hi-res halo with every
// low-res pixel duplicated.  The phase weights are summed in fp32 at load time (unet.py) and packed by ld_pack_conv_weight with
// ksize = 4 ("taps" = phase * 4 + u * 2 + v): the same fragment order as every other convolution weight.
//
// Workgroup = 256 threads = 4 waves; low-res tile 4 rows x 16 columns (8 x 32 outputs) x 32 output channels; wave w owns low row w.
// Per 64-byte channel chunk a wave reads 9 activation fragments (rows w .. w+2 of the halo, column shifts 0 .. 2) and 32 weight
// fragments for 32 MFMAs (the generic kernel: 36 MFMAs for a QUARTER of these outputs).  Register-staged prefetch of the next chunk,
// weights requested before the halo, write-through output stores -- the house rules of conv3x3_body.hip.h.  No prologue, no
// statistics: the up path's resampling convolutions read the attention block's raw output and feed raw block inputs (ddpm.py:446).
#include "common.hip.h"

namespace {

struct Up2Dev {                   // behind the preloaded head
  const float* bias;
  void* out;
  int ld;                         // elements between low-res pixels of the source
};
typedef const Up2Dev __attribute__((address_space(4)))* Up2KernargPtr;

constexpr int U_TR = 4, U_TC = 16, U_HR = U_TR + 2, U_HC = U_TC + 2, U_NPIX = U_HR * U_HC, U_NPIXP = 112, U_PLANE = U_NPIXP * 16;
constexpr int U_MT = 2, U_TAPS = 16, U_UNITS = U_TAPS * U_MT * 64, U_WU = U_UNITS / 256, U_ITER = 2;
constexpr int U_LDS = 4 * U_PLANE + U_TAPS * U_MT * 1024;

template <typename T>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3)))      // (at 128 registers it spills 72 bytes: three workgroups per CU)
void conv_up2_kernel(const void* data, const void* wts, int hw, int cin, int cout, Up2Dev rest_) {
  static_assert(sizeof(T) == 2, "16-bit storage only");
  constexpr int E = 8, CK = 32;
  __shared__ __attribute__((aligned(256))) char smem[U_LDS];
  char* s_x = smem;
  char* s_w = smem + 4 * U_PLANE;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, px = lane & 15, kq = lane >> 4;
  const int h = hw & 0xffff, w = hw >> 16;
  const int tiles_x = w >> 4;
  const int b = blockIdx.z, m0 = blockIdx.y * U_MT, mt_total = cout >> 4;
  const int ty0 = (blockIdx.x / tiles_x) * U_TR, tx0 = (blockIdx.x % tiles_x) * U_TC;
  const int nch = cin / CK;

  // ---- requests of chunk 0: weights (no per-lane geometry) first, then the low-res halo fragments
  unsigned hvalid = 0, hoff[U_ITER];
#pragma unroll
  for (int it = 0; it < U_ITER; ++it) {
    const int q = (it * 4 + wv) * 16 + px;
    const int hy = (q * 3641) >> 16, hx_ = q - hy * U_HC;            // q / 18 for q < 400
    const int gy = ty0 - 1 + hy, gx = tx0 - 1 + hx_;
    const bool in = q < U_NPIX && gy >= 0 && gy < h && gx >= 0 && gx < w;
    if (in) hvalid |= 1u << it;
    hoff[it] = in ? (unsigned)(__umul24(gy, w) + gx) : 0u;
  }
  u32x4 hx[U_ITER], wx[U_WU];
  const char* wbase = reinterpret_cast<const char*>(wts) + (long)m0 * 1024;
  auto issue_w = [&](int ch) {
    const char* wc = wbase + (long)ch * U_TAPS * mt_total * 1024;
#pragma unroll
    for (int k = 0; k < U_WU; ++k) {
      const int u = k * 256 + tid, tap = u >> 7, r = u & 127;         // U_MT * 64 = 128 units per tap
      wx[k] = *reinterpret_cast<const u32x4*>(wc + (long)tap * mt_total * 1024 + r * 16);
    }
  };
  // (the source's pixel stride arrives with the rest of the block; the head uses the dense stride, which is what the plan passes:
  //  checked on the host)
  const long img_px = (long)b * h * w;
  auto issue_h = [&](int ch, int ld) {
    const char* sp = reinterpret_cast<const char*>(data) + ((img_px * ld) + ch * CK + kq * E) * (long)sizeof(T);
    const unsigned ldb = (unsigned)ld * (unsigned)sizeof(T);
#pragma unroll
    for (int it = 0; it < U_ITER; ++it) {
      hx[it] = u32x4{0u, 0u, 0u, 0u};
      if ((hvalid >> it) & 1u) hx[it] = load16_act(sp + (size_t)hoff[it] * ldb);
    }
  };
  issue_w(0);
  issue_h(0, cin);

  // ---- the rest of the argument block: one scalar batch behind the requests above
  constexpr unsigned REST_OFF = ld_kernarg_offset<const void*, const void*, int, int, int>(alignof(Up2Dev));
  typedef const char __attribute__((address_space(4)))* KChar;
  Up2KernargPtr pr = (Up2KernargPtr)((KChar)__builtin_amdgcn_kernarg_segment_ptr() + REST_OFF);
  asm volatile("" : "+s"(pr));
  Up2Dev a;
  a.bias = pr->bias; a.out = pr->out; a.ld = pr->ld;
  (void)rest_;
  f32x4 bias[U_MT];
#pragma unroll
  for (int m = 0; m < U_MT; ++m) bias[m] = *reinterpret_cast<const f32x4*>(a.bias + (m0 + m) * 16 + kq * 4);

  f32x4 acc[4][U_MT];
#pragma unroll
  for (int ph = 0; ph < 4; ++ph)
#pragma unroll
    for (int m = 0; m < U_MT; ++m) acc[ph][m] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int ch = 0; ch < nch; ++ch) {
    __syncthreads();                                       // the previous chunk's fragments have been read
#pragma unroll
    for (int k = 0; k < U_WU; ++k) *reinterpret_cast<u32x4*>(s_w + (k * 256 + tid) * 16) = wx[k];
#pragma unroll
    for (int it = 0; it < U_ITER; ++it) {
      const int q = (it * 4 + wv) * 16 + px;
      if (q < U_NPIXP) *reinterpret_cast<u32x4*>(s_x + kq * U_PLANE + q * 16) = hx[it];
    }
    __syncthreads();
    if (ch + 1 < nch) {                                    // the next chunk under this chunk's MFMAs
      issue_w(ch + 1);
      issue_h(ch + 1, a.ld);
    }
    // 9 activation fragments: halo rows wv .. wv + 2, column shifts 0 .. 2
    uint4 Bq[3][3];
#pragma unroll
    for (int rr = 0; rr < 3; ++rr)
#pragma unroll
      for (int cc = 0; cc < 3; ++cc)
        Bq[rr][cc] = *reinterpret_cast<const uint4*>(s_x + kq * U_PLANE + (((wv + rr) * U_HC + cc + px) * 16));
#pragma unroll
    for (int ph = 0; ph < 4; ++ph) {
      const int pa = ph >> 1, pb = ph & 1;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int u = t >> 1, v = t & 1;
#pragma unroll
        for (int m = 0; m < U_MT; ++m) {
          const uint4 A = *reinterpret_cast<const uint4*>(s_w + ((ph * 4 + t) * U_MT + m) * 1024 + lane * 16);
          mma16<T>(acc[ph][m], A, Bq[pa + u][pb + v]);
        }
      }
    }
  }

  // ---- epilogue: bias, NHWC store of the four phases.  Lane holds channels 16m + 4kq .. + 3 of low-res pixel (ty0 + wv, tx0 + px);
  // two m-tiles leave as ONE 16-byte write-through store per lane and phase (pair_frag16).
  asm volatile("" ::"v"(bias[0]), "v"(bias[1]));         // retired before the first store (in-order counter: finding 59)
  const int H2 = 2 * h, W2 = 2 * w;
  char* outb = reinterpret_cast<char*>(a.out) + (((long)b * H2 + 2 * (ty0 + wv)) * W2 + 2 * (tx0 + px)) * cout * (long)sizeof(T)
               + (long)m0 * 16 * sizeof(T) + pair_frag16_off(kq);
  const unsigned row_b = (unsigned)W2 * cout * (unsigned)sizeof(T), px_b = (unsigned)cout * (unsigned)sizeof(T);
#pragma unroll
  for (int ph = 0; ph < 4; ++ph) {
    float v[U_MT][4];
#pragma unroll
    for (int m = 0; m < U_MT; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) v[m][r] = acc[ph][m][r] + bias[m][r];
    store16_out(outb + (ph >> 1) * row_b + (ph & 1) * px_b, pair_frag16<T>(v[0], v[1]));
  }
}

}  // namespace

extern "C" int ld_conv_up2(const ld_conv_up2_args* p, void* stream) {
  LD_REQUIRE(p && p->x && p->weight && p->bias && p->out, "ld_conv_up2: null pointer");
  LD_REQUIRE(ld_dtype_16(p->dtype), "ld_conv_up2: 16-bit storage only (fp32 keeps ld_conv3x3 with an upsampled source)");
  LD_REQUIRE(p->B > 0 && p->h > 0 && p->w > 0 && p->h % U_TR == 0 && p->w % U_TC == 0 && p->h < 65536 && p->w < 32768,
             "ld_conv_up2: low-res map %d x %d must be a multiple of %d x %d", p->h, p->w, U_TR, U_TC);
  LD_REQUIRE(p->Cin > 0 && p->Cin % 32 == 0 && p->Cout > 0 && p->Cout % 32 == 0, "ld_conv_up2: Cin %d / Cout %d must be multiples of 32", p->Cin, p->Cout);
  const int ld = p->pix_stride > 0 ? p->pix_stride : p->Cin;
  LD_REQUIRE(ld == p->Cin, "ld_conv_up2: the source must be dense");
  LD_REQUIRE((long)p->h * p->w * ld * 2 < (1L << 32) && (long)p->B * p->h * p->w < (1L << 31), "ld_conv_up2: map too large for 32-bit offsets");
  Up2Dev d;
  d.bias = p->bias; d.out = p->out; d.ld = ld;
  const dim3 grid((p->w / U_TC) * (p->h / U_TR), p->Cout / (16 * U_MT), p->B);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (p->dtype == LD_BF16) LD_LAUNCH((conv_up2_kernel<bf16>), grid, dim3(256), 0, st, p->x, p->weight, p->h | (p->w << 16), p->Cin, p->Cout, d);
  else LD_LAUNCH((conv_up2_kernel<f16>), grid, dim3(256), 0, st, p->x, p->weight, p->h | (p->w << 16), p->Cin, p->Cout, d);
  LD_LAUNCH_CHECK("conv_up2");
  return LD_OK;
}
