// Small-map 3x3 convolution, "weights in registers, whole K in LDS, K split over waves" (16-bit storage).
//
// Same op as conv3x3.hip (nn.Conv2d(k=3,p=1) of Block.proj ddpm.py:173, Upsample :117, the last-stage convs :372,:391),
// same fragment layouts, different schedule, for the launches the generic kernel runs as a latency chain: the 32^2 and
// 64^2 maps with 64-256 input channels (a 128-pixel x 32-channel workgroup of the generic kernel walks
// 8 K-chunks with two barriers and one dependent global round trip each -- 2,300 cycles per chunk around 576 cycles
// of MFMA issue, one wave per SIMD, no unit more than a quarter busy).
//
// Here a workgroup (512 threads = 8 waves, two per SIMD) owns ONE 8 x 16-pixel x 32-channel output tile and
//   * issues EVERY operand load of the tile up front: the halo tile for ALL K-chunks by LDS-DMA
//     (global_load_lds_dwordx4, blocks of 16 pixels x 64 B = 1 KiB per instruction, layout [chunk][block][kq][px][16 B]
//     as in conv3x3_c32.hip: conflict-free fragment reads for every tap), and the weights straight into REGISTERS --
//     they are packed in fragment order (ld_pack_conv_weight), so a wave's A operand of (chunk, tap, m) is one
//     coalesced 1-KiB load.  One wait, one barrier, and the whole K extent is resident;
//   * splits K over the waves: wave (ks, ps) owns NCW consecutive chunks (its 18*NCW weight fragments never touch
//     LDS) and the pixel rows [ps*RW, (ps+1)*RW) of the tile; it runs its 9*2*RW*NCW MFMAs with no barrier at all, two
//     waves per SIMD covering each other's LDS fragment reads;
//   * joins the KS partial sums through LDS (the halo region is dead by then) in a FIXED order -- results do not depend
//     on timing, replays are bitwise equal -- with the epilogue (bias, optional addend, GroupNorm statistics, NHWC
//     store) spread over all eight waves, two output fragments each.
// Three barriers per workgroup instead of 2 * nch, and a memory phase whose length is bytes / bandwidth instead of
// nch dependent round trips.
//
// Scope (ld_conv3x3_ws_try returns 0 for anything else and the generic kernel takes the launch): bf16 / fp16 storage,
// one source, Cin = 64 / 128 / 256 (2 / 4 / 8 chunks), Cout % 32 == 0, H % 8 == 0, W % 16 == 0, maps of at most
// LD_CONV_WS_MAX_PX pixels; nearest-x2 upsample of the source, addend and output statistics are supported; a GroupNorm
// prologue on the source is applied in place in LDS after the tile has landed.
#include "common.hip.h"
#include <stdlib.h>

namespace {

struct WsDev {
  SrcDev s;
  const void* w;
  const float* bias;
  const void* addend;
  void* out;
  double* ostats;
  int ogroups;
  int B, H, W, Cout;
  const int* t_ptr;
  int tiles_x, ntile, ncout, nwg;     // tiles per image, cout tiles, total workgroups
};

template <typename T, int NCW, int KS>
__global__ __launch_bounds__(512) void conv3x3_ws_kernel(WsDev a) {
  constexpr int E = DT<T>::E, CK = DT<T>::CK;
  constexpr int MT = 2, NWAVE = 8, PS = NWAVE / KS, TR = 8, TC = 16, RW = TR / PS, HR = TR + 2, HC = TC + 2;
  constexpr int NPIX = HR * HC, NBLK = (NPIX + 15) / 16;          // 180 halo pixels, 12 blocks of 16
  constexpr int NCH = NCW * KS;                                    // K-chunks of the launch
  constexpr int XCH = NBLK * 1024;                                 // bytes of halo per chunk
  constexpr int NDMA = (NCH * NBLK + NWAVE - 1) / NWAVE;           // DMA instructions per wave
  constexpr bool P = DT<T>::precise;
  static_assert(sizeof(T) == 2, "16-bit storage only");

  extern __shared__ __attribute__((aligned(1024))) char smem[];
  char* s_x = smem;                                                // [NCH][NBLK][kq][16 px][16 B]; later the partial sums
  float* s_coef = reinterpret_cast<float*>(smem + (NCH * XCH > KS * 16384 ? NCH * XCH : KS * 16384));
  double* s_stat = reinterpret_cast<double*>(s_coef + 2 * NCH * CK);

  const int tid = threadIdx.x, lane = tid & 63, px = lane & 15, kq = lane >> 4;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ks = wv % KS, ps = wv / KS;

  // Workgroup -> (image, tile, cout tile).  Workgroups are dealt round-robin over the 8 XCDs; when the grid divides
  // by 8 the logical index is permuted so that the workgroups of ONE XCD cover a contiguous range of (image, tile)
  // with all their cout tiles: each XCD's L2 then pulls an eighth of the activations (speed only, never correctness).
  int L = blockIdx.x;
  if ((a.nwg & 7) == 0) L = (L & 7) * (a.nwg >> 3) + (L >> 3);
  const int ct = L % a.ncout;
  const int til = (L / a.ncout) % a.ntile;
  const int b = L / (a.ncout * a.ntile);
  const int ty0 = (til / a.tiles_x) * TR, tx0 = (til % a.tiles_x) * TC;
  const int H = a.H, W = a.W;
  const int m0 = ct * MT, mt_total = a.Cout / 16;

  // ---- (1) weights of this wave's chunks -> registers (fragment order in HBM: one coalesced KiB per load)
  uint4 Areg[NCW][9][MT];
  {
    const uint4* wg = reinterpret_cast<const uint4*>(a.w);
#pragma unroll
    for (int i = 0; i < NCW; ++i)
#pragma unroll
      for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int m = 0; m < MT; ++m)
          Areg[i][tap][m] = wg[(size_t)(((ks * NCW + i) * 9 + tap) * mt_total + m0 + m) * 64 + lane];
  }
  // ---- (2) the halo tile, every chunk, by LDS-DMA: block g = r*8 + wave -> (chunk g / NBLK, block g % NBLK)
  const SrcDev S = a.s;
  const int Hs = S.ups ? H / 2 : H, Ws = S.ups ? W / 2 : W;
  const unsigned x_a = lds_addr(s_x);
  {
    const T* sdata = reinterpret_cast<const T*>(S.data) + kq * E;
#pragma unroll
    for (int r = 0; r < NDMA; ++r) {
      const int g = r * NWAVE + wv;
      if (g < NCH * NBLK) {                                         // wave-uniform
        const int ch = g / NBLK, blk = g - ch * NBLK;
        const int q = blk * 16 + px;
        const int hy = (q * 3641) >> 16, hx = q - hy * HC;          // q / 18
        int gy = ty0 - 1 + hy, gx = tx0 - 1 + hx;
        gy = gy < 0 ? 0 : (gy > H - 1 ? H - 1 : gy);                // out-of-image (and padding) lanes read an in-image
        gx = gx < 0 ? 0 : (gx > W - 1 ? W - 1 : gx);                // pixel; the fix-up below zeroes their slots
        const int sy = S.ups ? gy >> 1 : gy, sx = S.ups ? gx >> 1 : gx;
        glds16(sdata + ((size_t)(b * Hs + sy) * Ws + sx) * S.ld + ch * CK, __builtin_amdgcn_readfirstlane(x_a + g * 1024));
      }
    }
  }
  // ---- (3) GroupNorm + FiLM + activation coefficients of the source (overlaps the loads in flight)
  const bool has_coef = S.stats != nullptr;
  if (has_coef) {
    const int trow = a.t_ptr ? *a.t_ptr : 0;
    const long npix = S.ups ? (long)(H / 2) * (W / 2) : (long)H * W;
    build_gn_coef(S, b, trow, npix, s_coef, s_stat, tid, 512);
  }
  // bias of the output fragments this wave finalises in (7): fragments 2w, 2w+1 -> m-tile w / 4
  const float4 bv = *reinterpret_cast<const float4*>(a.bias + (m0 + ((2 * wv) >> 3)) * 16 + kq * 4);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // weights in registers, this wave's DMAs landed
  // ---- (4) in place: zero padding of out-of-image pixels; normalise + activate when the source carries a prologue
  {
    const bool border = ty0 == 0 || tx0 == 0 || ty0 + TR >= H || tx0 + TC >= W;
    if (border || has_coef) {
#pragma unroll
      for (int r = 0; r < NDMA; ++r) {
        const int g = r * NWAVE + wv;
        if (g < NCH * NBLK) {
          const int ch = g / NBLK, blk = g - ch * NBLK;
          const int q = blk * 16 + px;
          const int hy = (q * 3641) >> 16, hx = q - hy * HC;
          const int gy = ty0 - 1 + hy, gx = tx0 - 1 + hx;
          const bool valid = q < NPIX && gy >= 0 && gy < H && gx >= 0 && gx < W;
          uint4* ptr = reinterpret_cast<uint4*>(s_x + g * 1024 + lane * 16);
          if (!valid) {
            *ptr = make_uint4(0u, 0u, 0u, 0u);
          } else if (has_coef) {
            float v[E], ca[E], cs[E];
            const float* cap = s_coef + ch * CK + kq * E;
#pragma unroll
            for (int e = 0; e < E; ++e) { ca[e] = cap[e]; cs[e] = cap[S.C + e]; }
            unpack16<T>(*ptr, v);
            affine_act_n<P, E>(v, ca, cs, S.act);
            *ptr = pack16<T>(v);
          }
        }
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  // ---- (5) MFMAs of this wave's K-slice and pixel rows: no barrier, fragment reads of tap column dx+1 in flight
  //          during the MFMAs of column dx
  f32x4 acc[MT][RW];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int j = 0; j < RW; ++j) acc[m][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < NCW; ++i) {
    const char* xb = s_x + (ks * NCW + i) * XCH + kq * 256;
    uint4 Bq[2][RW + 2];
    auto load_frags = [&](int dx, int set) {
#pragma unroll
      for (int rr = 0; rr < RW + 2; ++rr) {
        const int q = (ps * RW + rr) * HC + dx + px;
        Bq[set][rr] = *reinterpret_cast<const uint4*>(xb + ((q >> 4) << 10) + ((q & 15) << 4));
      }
    };
    load_frags(0, 0);
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      if (dx + 1 < 3) load_frags(dx + 1, (dx + 1) & 1);
#pragma unroll
      for (int rr = 0; rr < RW + 2; ++rr) {
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
          const int j = rr - dy;
          if (j >= 0 && j < RW) {
#pragma unroll
            for (int m = 0; m < MT; ++m) mma16<T>(acc[m][j], Areg[i][dy * 3 + dx][m], Bq[dx & 1][rr]);
          }
        }
      }
    }
  }

  // ---- (6) join the K-slices: partial sums -> LDS [ks][fragment f = m*8 + row][lane] (16 KiB per slice)
  __syncthreads();                                                  // every wave is done reading the halo tile
  float4* s_red = reinterpret_cast<float4*>(smem);
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int j = 0; j < RW; ++j)
      s_red[(ks * 16 + m * 8 + ps * RW + j) * 64 + lane] = make_float4(acc[m][j][0], acc[m][j][1], acc[m][j][2], acc[m][j][3]);
  __syncthreads();

  // ---- (7) epilogue: wave w finalises fragments 2w, 2w+1 (m = w / 4, rows 2 (w % 4), +1): fixed summation order over ks
  const int mf = (2 * wv) >> 3, row0 = (2 * wv) & 7;
  const int gx = tx0 + px;
  T* out = reinterpret_cast<T*>(a.out);
  const T* add = reinterpret_cast<const T*>(a.addend);
  float ssum[4] = {0.f, 0.f, 0.f, 0.f}, ssq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int jj = 0; jj < 2; ++jj) {
    const int f = 2 * wv + jj;
    float4 v4 = s_red[f * 64 + lane];
#pragma unroll
    for (int k = 1; k < KS; ++k) {
      const float4 o = s_red[(k * 16 + f) * 64 + lane];
      v4.x += o.x; v4.y += o.y; v4.z += o.z; v4.w += o.w;
    }
    float v[4] = {v4.x + bv.x, v4.y + bv.y, v4.z + bv.z, v4.w + bv.w};
    const int gy = ty0 + row0 + jj;
    const size_t off = ((size_t)(b * H + gy) * W + gx) * a.Cout + (m0 + mf) * 16 + kq * 4;
    if (add) {
      float ad[4];
      load4<T>(add + off, ad);
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] += ad[r];
    }
    store4<T>(out + off, v);
#pragma unroll
    for (int r = 0; r < 4; ++r) { ssum[r] += v[r]; ssq[r] += v[r] * v[r]; }
  }
  if (a.ostats) {
    // a lane's four channels fall into one group (channels per group: a multiple of 4)
    const float s1 = wave16_sum((ssum[0] + ssum[1]) + (ssum[2] + ssum[3]));
    const float s2 = wave16_sum((ssq[0] + ssq[1]) + (ssq[2] + ssq[3]));
    if (px == 0) {
      s_stat[(wv * 2 + 0) * 4 + kq] = (double)s1;
      s_stat[(wv * 2 + 1) * 4 + kq] = (double)s2;
    }
    __syncthreads();
    const int gs = a.Cout / a.ogroups;                              // channels per group, 4 <= gs, gs | 32 or 32 | gs
    const int ngrp_blk = gs >= 32 ? 1 : 32 / gs;
    if (tid < 2 * ngrp_blk) {
      const int gi = tid >> 1, k = tid & 1;
      double acc1 = 0.0;
      for (int w = 0; w < NWAVE; ++w)
        for (int q4 = 0; q4 < 4; ++q4) {
          const int c = (w >> 2) * 16 + q4 * 4;                     // first channel (within the 32-channel tile) of that value
          if ((gs >= 32 ? 0 : c / gs) == gi) acc1 += s_stat[(w * 2 + k) * 4 + q4];
        }
      const int g = (m0 * 16) / gs + gi;
      const int stripe = blockIdx.x % LD_STAT_STRIPES;
      atomicAdd(&a.ostats[(((size_t)b * LD_STAT_STRIPES + stripe) * a.ogroups + g) * 2 + k], acc1);
    }
  }
}

template <typename T, int NCW, int KS>
int launch_ws(const WsDev& a, hipStream_t st) {
  constexpr int NCH = NCW * KS;
  const size_t region = (size_t)NCH * 12 * 1024 > (size_t)KS * 16384 ? (size_t)NCH * 12 * 1024 : (size_t)KS * 16384;
  const size_t lds = region + 2 * NCH * 32 * sizeof(float) + 64 * sizeof(double);
  if (lds > 65536) LD_HIP(ld_allow_lds((conv3x3_ws_kernel<T, NCW, KS>), lds));
  LD_LAUNCH((conv3x3_ws_kernel<T, NCW, KS>), dim3(a.nwg), dim3(512), lds, st, a);
  LD_LAUNCH_CHECK("conv3x3_ws");
  return LD_OK;
}

template <typename T>
int dispatch_ws(const WsDev& a, int nch, hipStream_t st) {
  switch (nch) {
    case 2: return launch_ws<T, 1, 2>(a, st);
    case 4: return launch_ws<T, 1, 4>(a, st);
    default: return launch_ws<T, 2, 4>(a, st);
  }
}

}  // namespace

// Returns 1 if this launch is handled here, 0 if another kernel must take it, <0 on error.
int ld_conv3x3_ws_try(const ld_conv3x3_args* p, hipStream_t st) {
  static const int disabled = getenv("LD_CONV_NO_WS") ? atoi(getenv("LD_CONV_NO_WS")) : 0;   // tuning override (A/B)
  static const long max_px = getenv("LD_CONV_WS_MAX_PX") ? atol(getenv("LD_CONV_WS_MAX_PX")) : 64 * 64;
  if (disabled || p->dtype == LD_F32 || p->nsrc != 1) return 0;
  const int cin = p->src[0].C;
  if (cin != 64 && cin != 128 && cin != 256) return 0;
  if (p->H % 8 != 0 || p->W % 16 != 0 || (long)p->H * p->W > max_px) return 0;
  if (p->out_stats) {
    if (p->out_groups <= 0 || p->Cout % p->out_groups != 0) return 0;
    const int gs = p->Cout / p->out_groups;
    if (gs % 4 != 0 || !(gs >= 32 ? gs % 32 == 0 : 32 % gs == 0)) return 0;
  }
  const ld_src& S = p->src[0];
  const long ld = S.pix_stride > 0 ? S.pix_stride : S.C;
  if ((long)p->B * p->H * p->W * ld >= (1L << 31)) return 0;
  WsDev a;
  a.s = to_dev(S);
  a.w = p->weight; a.bias = p->bias; a.addend = p->addend; a.out = p->out; a.ostats = p->out_stats;
  a.ogroups = p->out_groups > 0 ? p->out_groups : 1;
  a.B = p->B; a.H = p->H; a.W = p->W; a.Cout = p->Cout; a.t_ptr = p->t_ptr;
  a.tiles_x = p->W / 16;
  a.ntile = a.tiles_x * (p->H / 8);
  a.ncout = p->Cout / 32;
  a.nwg = a.ntile * a.ncout * p->B;
  const int rc = LD_DISPATCH16(p->dtype, dispatch_ws<T>(a, cin / 32, st));
  if (rc == LD_OK) ld_count(LD_COUNTER_CONV3X3_WS);
  return rc == LD_OK ? 1 : rc;
}
