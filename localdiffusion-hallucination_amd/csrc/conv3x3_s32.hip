// 3x3 convolution for the ResBlock conv path of the north star: Cout = 32, Cin = 32 or 64 (one or two K-chunks), large
// maps (256^2 / 128^2), 16-bit storage.  Same arithmetic, fragment layouts and fused prologue / epilogue as conv3x3.hip
// (Block.proj, ddpm.py:173, with the producer's GroupNorm + FiLM + SiLU applied on load, :179-185, and the result's
// GroupNorm statistics in the epilogue); what differs is everything AROUND the 72 MFMAs of a wave.
//
// Why (round 5; cycle stamps of one mid-launch workgroup of the generic kernel, 32 -> 32 @256^2, 4 patches, GroupNorm
// prologue: 25.0k cycles): 5.3k until the tile's requests have left, 4.8k for the coefficient chain (statistics words ->
// fp64 stripe sums on 8 threads -> LDS -> barrier -> per-channel coefficients -> LDS -> barrier), 7.3k for the
// normalise + SiLU pass over the halo tile (three co-resident workgroups per CU do it at the same time: the VALU is
// the busy unit), 2.2k for the MFMAs, 1.0k stores, 3.5k for the statistics (16-lane sums -> LDS -> barrier -> fp64 group
// sums -> atomics); and the launch is 1,024 workgroups on 768 resident slots: 1.33 rounds.  Here:
//   * FOUR workgroups per CU (<= 128 registers, 39 KB of LDS: no coefficient or statistics scratch): the 1,024 tiles
//     of a 4-patch launch are resident at once -- one round;
//   * the GroupNorm coefficients are built in REGISTERS by every wave for the 8 channels its lanes stage: the 16 lanes of
//     a row load one statistics stripe each (two groups = 32 contiguous bytes), the stripes meet by DPP inside the row
//     (fp64), and gamma / beta / FiLM for the lane's channels are two 16-byte loads each -- no LDS, no barrier, nothing
//     that waits for another wave;
//   * the statistics of the result leave as fp64 atomics straight from each wave (16-lane DPP sums, lanes px == 0 add
//     their (m-tile, kq) entry -- one group each at 4 channels per group): no LDS, no barrier, and a wave that has
//     issued its atomics is finished.
//   * the output leaves by write-through stores (store16_out, common.hip.h; finding 98): 32->32 @256^2 x 4 patches alone
//     9.3 / 9.6 / 13.6 us (plain / statistics / prologue) = 0.50 / 0.48 / 0.34 of 8 TB/s, x 8 patches 13.4 / 15.2 / 23.5 (0.63 /
//     0.55 / 0.36) -- which also retired the persistent LDS-DMA kernel (conv3x3_c32.hip) from the default routing (finding 99).
// Routing: ld_conv3x3_s32_try (below); everything it does not take falls through to the generic kernel.
#include "common.hip.h"

namespace {

struct S32Dev {                  // the argument block behind the preloaded head (one scalar batch)
  const void* data1;             // second source of a two-source launch (NCH == 2); else unused
  const double* stats;           // PRO: source 0's GroupNorm operands
  const float* gamma;
  const float* beta;
  const float* film;             // the timestep's FiLM row for this block ([scale C | shift C], + b * film_bstride) or null
  const float* bias;
  void* out;
  double* ostats;
  int ld1, film_bstride, act, ogroups;
};

constexpr int TR = 16, TC = 16, HR = 18, HC = 18, NPIX = HR * HC, NPIXP = 336, PLANE = NPIXP * 16, ITER = 6, WU = 5, MT = 2, NW = 4;
constexpr int S32_LDS = 4 * PLANE + 9 * MT * 1024;

typedef const S32Dev __attribute__((address_space(4)))* S32KernargPtr;

template <typename T, int NCH, bool PRO>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4)))
void conv3x3_s32_kernel(const void* data0, const void* wts, int hw, int ld0, int c0, S32Dev rest_) {
  static_assert(sizeof(T) == 2, "16-bit storage only");
  static_assert(!PRO || NCH == 1, "the prologue variant stages one chunk");
  constexpr int E = 8;
  __shared__ __attribute__((aligned(256))) char smem[S32_LDS];
  char* s_x = smem;
  char* s_w = smem + 4 * PLANE;

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, px = lane & 15, kq = lane >> 4;
  const int H = hw & 0xffff, W = hw >> 16;
  const int tiles_x = W >> 4;
  const int bx = blockIdx.x, b = blockIdx.z;
  int ty_, tx_;
  tile_of(bx, tiles_x, H >> 4, ty_, tx_);
  const int ty0 = ty_ * TR, tx0 = tx_ * TC;

  // ---- the tile's requests: weights (no per-lane geometry: they leave first), then the halo fragments
  unsigned hvalid = 0, hoff[ITER];
#pragma unroll
  for (int it = 0; it < ITER; ++it) {
    const int q = (it * 4 + wv) * 16 + px;
    const int hy = (q * 3641) >> 16, hx_ = q - hy * HC;             // q / 18 for q < 400
    const int gy = ty0 - 1 + hy, gx = tx0 - 1 + hx_;
    const bool in = q < NPIX && gy >= 0 && gy < H && gx >= 0 && gx < W;
    if (in) hvalid |= 1u << it;
    hoff[it] = in ? (unsigned)(__umul24(gy, W) + gx) : 0u;           // pixel index inside the image (clamped: never dereferenced when !in)
  }
  const long img_px = (long)b * H * W;
  u32x4 hx[ITER], wx[WU];
  auto issue_w = [&](int ch) {
    const char* wc = reinterpret_cast<const char*>(wts) + (long)ch * (9 * MT * 1024);
#pragma unroll
    for (int k = 0; k < WU; ++k) {
      const int u = k * 256 + tid;
      wx[k] = u32x4{0u, 0u, 0u, 0u};
      if (u < 9 * MT * 64) wx[k] = *reinterpret_cast<const u32x4*>(wc + u * 16);
    }
  };
  auto issue_h = [&](const void* data, int ld, int coff) {
    const char* sp = reinterpret_cast<const char*>(data) + (img_px * ld + coff + kq * E) * (long)sizeof(T);
    const unsigned ldb = (unsigned)ld * (unsigned)sizeof(T);
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      hx[it] = u32x4{0u, 0u, 0u, 0u};
      if ((hvalid >> it) & 1u) hx[it] = load16_act(sp + (size_t)hoff[it] * ldb);
    }
  };
  issue_w(0);
  issue_h(data0, ld0, 0);

  // ---- the rest of the argument block: ONE scalar batch, behind the requests above (scalar loads return out of order: a
  // wait for any of them is a wait for all; the laundered pointer keeps hipcc from hoisting them in front of the requests)
  constexpr unsigned REST_OFF = ld_kernarg_offset<const void*, const void*, int, int, int>(alignof(S32Dev));
  typedef const char __attribute__((address_space(4)))* KChar;
  S32KernargPtr pr = (S32KernargPtr)((KChar)__builtin_amdgcn_kernarg_segment_ptr() + REST_OFF);
  asm volatile("" : "+s"(pr));
  S32Dev a;
  a.data1 = pr->data1; a.stats = pr->stats; a.gamma = pr->gamma; a.beta = pr->beta; a.film = pr->film; a.bias = pr->bias;
  a.out = pr->out; a.ostats = pr->ostats; a.ld1 = pr->ld1; a.film_bstride = pr->film_bstride; a.act = pr->act; a.ogroups = pr->ogroups;
  (void)rest_;

  // ---- PRO: coefficient operands (requested behind the tile: the data is the long transfer) and the coefficients in registers
  float ca[PRO ? E : 1], cs[PRO ? E : 1];
  if constexpr (PRO) {
    const float* fp = a.film ? a.film + (long)b * a.film_bstride : a.gamma;
    const int c = kq * E;
    const f32x4 g0 = *reinterpret_cast<const f32x4*>(a.gamma + c), g1 = *reinterpret_cast<const f32x4*>(a.gamma + c + 4);
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(a.beta + c), b1 = *reinterpret_cast<const f32x4*>(a.beta + c + 4);
    const f32x4 fs0 = *reinterpret_cast<const f32x4*>(fp + c), fs1 = *reinterpret_cast<const f32x4*>(fp + c + 4);
    const f32x4 fh0 = *reinterpret_cast<const f32x4*>(fp + (a.film ? 32 : 0) + c), fh1 = *reinterpret_cast<const f32x4*>(fp + (a.film ? 32 : 0) + c + 4);
    // statistics: [B, stripes, 8 groups, 2] fp64; lane (px, kq) takes groups 2kq, 2kq + 1 of stripes px, px + 16, ...
    constexpr int SPL = LD_STAT_STRIPES / 16;
    typedef __attribute__((ext_vector_type(2))) double f64x2;
    f64x2 q0[SPL], q1[SPL];
#pragma unroll
    for (int k = 0; k < SPL; ++k) {
      const double* sp = a.stats + (((size_t)b * LD_STAT_STRIPES + k * 16 + px) * 8 + 2 * kq) * 2;
      q0[k] = *reinterpret_cast<const f64x2*>(sp);
      q1[k] = *reinterpret_cast<const f64x2*>(sp + 2);
    }
    __builtin_amdgcn_sched_barrier(0);
    f64x2 t0 = q0[0], t1 = q1[0];
#pragma unroll
    for (int k = 1; k < SPL; ++k) { t0 += q0[k]; t1 += q1[k]; }
    double s1[2] = {row_group_sum_d(t0[0], 16), row_group_sum_d(t1[0], 16)};
    double s2[2] = {row_group_sum_d(t0[1], 16), row_group_sum_d(t1[1], 16)};
    float mean[2], rstd[2];
    const double inv_n = (double)__builtin_amdgcn_rcpf((float)((long)H * W) * 4.0f);      // as build_gn_coef (16-bit storage)
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const double m = s1[g] * inv_n;
      double var = s2[g] * inv_n - m * m;
      var = var > 0.0 ? var : 0.0;
      mean[g] = (float)m;
      rstd[g] = __builtin_amdgcn_rsqf((float)(var + 1e-5));
    }
    const float gam[E] = {g0[0], g0[1], g0[2], g0[3], g1[0], g1[1], g1[2], g1[3]};
    const float bet[E] = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
    const float fsc[E] = {fs0[0], fs0[1], fs0[2], fs0[3], fs1[0], fs1[1], fs1[2], fs1[3]};
    const float fsh[E] = {fh0[0], fh0[1], fh0[2], fh0[3], fh1[0], fh1[1], fh1[2], fh1[3]};
#pragma unroll
    for (int e = 0; e < E; ++e) {
      float aa = rstd[e >> 2] * gam[e];
      float ss = bet[e] - mean[e >> 2] * aa;
      if (a.film) {
        const float sc = fsc[e] + 1.0f;
        aa *= sc;
        ss = ss * sc + fsh[e];
      }
      ca[e] = aa;
      cs[e] = ss;
    }
  }
  f32x4 bias[MT];
  f32x4 acc[MT][NW];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int j = 0; j < NW; ++j) acc[m][j] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) {
    if (ch > 0) __syncthreads();                         // the previous chunk's fragments have been read
    // ---- stage: weights as they are, halo fragments normalised + activated (zero padding stays exactly zero)
#pragma unroll
    for (int k = 0; k < WU; ++k) {
      const int u = k * 256 + tid;
      if (u < 9 * MT * 64) *reinterpret_cast<u32x4*>(s_w + u * 16) = wx[k];
    }
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      const int q = (it * 4 + wv) * 16 + px;
      if (q < NPIXP) {                                   // (wave-uniform: the last iteration belongs to wave 0)
        u32x4 raw = hx[it];
        if constexpr (PRO) {
          if ((hvalid >> it) & 1u) {
            float v[E];
            unpack16<T>(make_uint4(raw[0], raw[1], raw[2], raw[3]), v);
            affine_act_n<false, E>(v, ca, cs, a.act);
            const uint4 p = pack16<T>(v);
            raw = u32x4{p.x, p.y, p.z, p.w};
          }
        }
        *reinterpret_cast<u32x4*>(s_x + kq * PLANE + q * 16) = raw;
      }
    }
    __syncthreads();
    if (ch + 1 < NCH) {                                  // the second chunk's halo: behind the first one's staging, under its MFMAs
      if (c0 > 32) issue_h(data0, ld0, 32);
      else issue_h(a.data1, a.ld1, 0);
    } else {                                             // the epilogue's operand: under the last chunk's MFMAs
#pragma unroll
      for (int m = 0; m < MT; ++m) bias[m] = *reinterpret_cast<const f32x4*>(a.bias + m * 16 + kq * 4);
    }
    // ---- 72 MFMAs: tap column dx outer, the six activation fragments of the column live, two weight fragments at a time
    // (per accumulator the taps arrive in the generic kernel's order -- dx, then dy: bit-identical sums)
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      uint4 Bq[NW + 2];
#pragma unroll
      for (int rr = 0; rr < NW + 2; ++rr)
        Bq[rr] = *reinterpret_cast<const uint4*>(s_x + kq * PLANE + (((wv * NW + rr) * HC + dx + px) * 16));
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
        uint4 A[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) A[m] = *reinterpret_cast<const uint4*>(s_w + ((dy * 3 + dx) * MT + m) * 1024 + lane * 16);
#pragma unroll
        for (int j = 0; j < NW; ++j)
#pragma unroll
          for (int m = 0; m < MT; ++m) mma16<T>(acc[m][j], A[m], Bq[j + dy]);
      }
    }
    // (the next chunk's weights are L2 hits: requested behind the MFMAs, their registers are not held across them --
    //  with both operands of chunk 1 in flight under chunk 0's MFMAs the kernel needs 140 registers: 3 workgroups per CU)
    if (ch + 1 < NCH) issue_w(ch + 1);
  }

  // ---- epilogue: bias, statistics, NHWC store (tiles are never ragged: H, W multiples of 16).  Lane holds channels
  // 16m + 4kq .. + 3 of pixel px of rows 4wv .. 4wv + 3; two m-tiles leave as ONE 16-byte store per lane (pair_frag16).
  asm volatile("" ::"v"(bias[0]), "v"(bias[1]));       // retired before the first store (in-order counter: finding 59)
  char* outp = reinterpret_cast<char*>(a.out) + ((img_px + (long)(ty0 + wv * NW) * W + tx0) * 32) * (long)sizeof(T);
  const unsigned lane_off = (unsigned)px * 64u + pair_frag16_off(kq);
  const unsigned row_off = (unsigned)W * 64u;
  float s1[MT] = {0.f, 0.f}, s2[MT] = {0.f, 0.f};
#pragma unroll
  for (int j = 0; j < NW; ++j) {
    float v[MT][4];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) v[m][r] = acc[m][j][r] + bias[m][r];
    store16_out(outp + lane_off + j * row_off, pair_frag16<T>(v[0], v[1]));
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      // (the generic kernel's order: per channel over the rows, then the four channels of the fragment)
      s1[m] += (v[m][0] + v[m][1]) + (v[m][2] + v[m][3]);
      s2[m] += (v[m][0] * v[m][0] + v[m][1] * v[m][1]) + (v[m][2] * v[m][2] + v[m][3] * v[m][3]);
    }
  }
  if (a.ostats) {
    // A lane's four channels fall into ONE group (4 | channels per group): 16-lane sums by DPP, one LDS value per
    // (wave, m-tile, kq), ONE barrier, then lanes 0-7 / 16-23 of wave 0 add the four waves' values (sums / sums of squares)
    // and the entries of a group meet by DPP: 16 fp64 atomics per workgroup on one 128-byte line.  (Measured: atomics
    // straight from every wave -- 64 per workgroup, no barrier -- cost 3.8 us per launch where these cost 1.9: same-line
    // fp64 atomics serialise at the memory side, ~60 ns each, and all 1,024 workgroups of a launch end together.)
    double* s_stat = reinterpret_cast<double*>(s_w);    // [4 waves][2][8]: the weights are dead
    __syncthreads();                                    // every wave has read its last weight fragment
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const float t1 = wave16_sum(s1[m]), t2 = wave16_sum(s2[m]);
      if (px == 0) {
        s_stat[(wv * 2 + 0) * 8 + m * 4 + kq] = (double)t1;
        s_stat[(wv * 2 + 1) * 8 + m * 4 + kq] = (double)t2;
      }
    }
    __syncthreads();
    if (tid < 32) {
      const int k = tid >> 4, e = tid & 15;             // entry e = 4m + kq (e < 8)
      const int gs = 32 / a.ogroups, q4 = gs >> 2;      // entries per group: 1, 2, 4 or 8 (checked on the host)
      double v = 0.0;
      if (e < 8) {
#pragma unroll
        for (int w4 = 0; w4 < 4; ++w4) v += s_stat[(w4 * 2 + k) * 8 + e];
      }
      v = row_group_sum_d(v, q4);
      if (e < 8 && (e & (q4 - 1)) == 0)
        atomicAdd(&a.ostats[(((size_t)b * LD_STAT_STRIPES + (bx % LD_STAT_STRIPES)) * a.ogroups + e / q4) * 2 + k], v);
    }
  }
}

template <typename T, int NCH, bool PRO>
int launch_s32(const ld_conv3x3_args* p, hipStream_t st) {
  S32Dev d{};
  const ld_src& s0 = p->src[0];
  d.data1 = p->nsrc > 1 ? p->src[1].data : nullptr;
  d.ld1 = p->nsrc > 1 ? (p->src[1].pix_stride > 0 ? p->src[1].pix_stride : p->src[1].C) : 0;
  d.stats = s0.gn_stats; d.gamma = s0.gn_gamma; d.beta = s0.gn_beta; d.film = s0.film;
  d.film_bstride = s0.film_bstride; d.act = s0.act;
  d.bias = p->bias; d.out = p->out; d.ostats = p->out_stats; d.ogroups = p->out_groups > 0 ? p->out_groups : 1;
  const int ld0 = s0.pix_stride > 0 ? s0.pix_stride : s0.C;
  const dim3 grid((p->W / 16) * (p->H / 16), 1, p->B);
  LD_LAUNCH((conv3x3_s32_kernel<T, NCH, PRO>), grid, dim3(256), 0, st, s0.data, p->weight, p->H | (p->W << 16), ld0, s0.C, d);
  LD_LAUNCH_CHECK("conv3x3_s32");
  return LD_OK;
}

}  // namespace

// Returns 1 if this launch is handled here, 0 if another kernel must take it, < 0 on error.
int ld_conv3x3_s32_try(const ld_conv3x3_args* p, hipStream_t st) {
  const LdTuning& tn = ld_tuning();
  if (!tn.conv_s32 || !ld_dtype_16(p->dtype) || p->Cout != 32 || p->addend || p->weight_terms == 2) return 0;
  if (p->H % 16 != 0 || p->W % 16 != 0 || p->H >= 65536 || p->W >= 32768) return 0;
  const long tiles = (long)(p->W / 16) * (p->H / 16) * p->B;
  if (tiles < tn.conv_s32_min_tiles) return 0;
  int ctot = 0;
  bool pro = false;
  for (int s = 0; s < p->nsrc; ++s) {
    const ld_src& S = p->src[s];
    if (S.upsample) return 0;
    const long ld = S.pix_stride > 0 ? S.pix_stride : S.C;
    if ((long)p->H * p->W * ld * 2 >= (1L << 32)) return 0;          // 32-bit byte offsets inside an image
    ctot += S.C;
    pro = pro || S.gn_stats != nullptr;
  }
  if (ctot != 32 && ctot != 64) return 0;
  if (p->nsrc == 2 && (p->src[0].C != 32 || p->src[1].C != 32)) return 0;
  if (pro) {
    // the register-coefficient prologue: one 32-channel source, 8 groups, the timestep's FiLM row at a fixed address
    const ld_src& S = p->src[0];
    if (p->nsrc != 1 || S.C != 32 || S.gn_groups != 8 || p->t_ptr != nullptr) return 0;
    if (!S.gn_gamma || !S.gn_beta) return 0;
  }
  if (p->out_stats) {
    const int og = p->out_groups;
    if (og <= 0 || 32 % og != 0 || 32 / og < 4) return 0;
  }
  // conv_s32 is a bit mask: 1 = single-chunk launches without a prologue, 2 = with the GroupNorm prologue, 4 = two chunks
  if (!(tn.conv_s32 & (pro ? 2 : (ctot == 32 ? 1 : 4)))) return 0;
  int rc;
  if (pro) rc = LD_DISPATCH16(p->dtype, launch_s32<T, 1, true>(p, st));
  else if (ctot == 32) rc = LD_DISPATCH16(p->dtype, launch_s32<T, 1, false>(p, st));
  else rc = LD_DISPATCH16(p->dtype, launch_s32<T, 2, false>(p, st));
  if (rc == LD_OK) ld_count(LD_COUNTER_CONV3X3_S32);
  return rc == LD_OK ? 1 : rc;
}
