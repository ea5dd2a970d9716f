// Wave-specialised 3x3 convolution for the C >= 64 stages (32^2 ... 128^2 maps), gfx950.
//
// Same arithmetic, fragment layouts, fused prologue (concat, nearest x2, GroupNorm-apply + FiLM + SiLU/ReLU of
// the producer) and epilogue (bias, GroupNorm statistics, NHWC store) as conv3x3.hip; different machine mapping.
// Measured on the register-staged kernel (rocprofv3 kernel trace, LD_CONV_DEBUG ablations, 256->256 @ 32^2, B=8):
// 18.3 us, of which MFMA issue is 3.8; an EMPTY chunk loop (no loads, no MFMA, no stores) still takes 9.6 us
// (ds_write_b128 of 48 KB per chunk at ~79 B/clk between two barriers), the GroupNorm prologue adds 8 us of VALU
// that nothing overlaps, and each chunk's loads are waited for in the open.  Here:
//   * workgroup = 8 waves: waves 0-3 CONSUME (LDS fragment reads + MFMA + epilogue, exactly the inner loop of
//     conv3x3.hip), waves 4-7 PRODUCE; every SIMD holds one of each, so the producer's VALU and address work
//     issues in the shadow of the consumer's MFMAs;
//   * a 3-stage LDS ring of K-chunks, one stage = [9*MT KiB weights][halo blocks of 16 pixels x 64 B], filled by
//     LDS-DMA (global_load_lds_dwordx4, no staging registers, no ds_write): at item i the producers issue the
//     DMA of item i+2, wait (counted vmcnt: every wave issues exactly NDMA instructions per item, out-of-image
//     lanes read a clamped in-image address) for item i+1, and normalise / activate / zero-pad it in place;
//   * ONE workgroup barrier per item; items = (tile, K-chunk) pairs of a persistent workgroup (grid = one
//     workgroup per CU, each walking tiles of one image), so the ring also runs across tile boundaries and the
//     consumers' epilogue overlaps the next tile's DMA;
//   * GroupNorm statistics: per-wave partials go to a parity-double-buffered LDS slot and are flushed (fp64
//     sum, one striped atomic per group) by consumer wave 0 behind the next barrier.
#include "common.hip.h"
#include <stdlib.h>

namespace {

struct WsDev {
  SrcDev s[2];
  int nsrc;
  const void* w;
  const float* bias;
  void* out;
  double* ostats;
  int ogroups;
  int B, H, W, Cout;
  const int* t_ptr;
  int tiles_x, nct, ntiles;   // pixel-tile columns, cout tiles, tiles per image (pixel tiles x cout tiles)
  int dbg;                    // LD_CONV_DEBUG ablation bits: 1 no DMA, 4 no MFMA, 8 no stores, 16 no prologue transform
};

// LD_CONV_DEBUG bit 32: workgroup (0,0) records (cycle counter, 100 MHz real-time counter) pairs at phase
// boundaries of consumer wave 0 and producer wave 4 into LDS and dumps them here at the end (ld_debug_ws_trace).
constexpr int TRACE_EV = 40;
__device__ unsigned long long g_ws_trace[8][TRACE_EV];
__device__ unsigned g_ws_hwid[8];     // HW_ID register of the 8 waves of workgroup (0,0): SIMD placement
#define LD_TRACE(role)                                                                  \
  do {                                                                                  \
    if (tracing && s_tn < TRACE_EV) s_trace[wid * TRACE_EV + s_tn++] = __builtin_readcyclecounter(); \
  } while (0)
#define LD_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define LD_BARRIER()                      \
  do {                                    \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
    __builtin_amdgcn_s_barrier();         \
    asm volatile("" ::: "memory");        \
  } while (0)

template <typename T, int MT, int NW>
__global__ __launch_bounds__(512) void conv3x3_ws_kernel(WsDev a) {
  constexpr int E = DT<T>::E, CK = DT<T>::CK;
  constexpr int TR = 4 * NW, TC = 16, HR = TR + 2, HC = TC + 2, NPIX = HR * HC;
  constexpr int HBLK = ((NPIX + 15) / 16 + 3) / 4 * 4, HPW = HBLK / 4;   // halo blocks (1 KiB each), per producer wave
  constexpr int WBLK = 9 * MT, WPW = WBLK / 4;                           // weight blocks, per producer wave
  static_assert(WBLK % 4 == 0, "weight blocks must split evenly over the 4 producer waves");
  constexpr int STAGE = (WBLK + HBLK) * 1024, NST = 3, NDMA = WPW + HPW;
  constexpr bool P = DT<T>::precise;

  extern __shared__ __attribute__((aligned(1024))) char smem[];
  char* ring = smem;
  const int ctot = a.s[0].C + (a.nsrc > 1 ? a.s[1].C : 0);
  float* s_coef = reinterpret_cast<float*>(smem + NST * STAGE);   // [src0: a[C0] s[C0]][src1: a[C1] s[C1]]
  float* s_part = s_coef + 2 * ctot;                              // [parity][wave][sum|sumsq][16*MT]
  float* s_g = s_part + 2 * 4 * 2 * 16 * MT;                      // [src][mean 32 | rstd 32]
  unsigned long long* s_trace = reinterpret_cast<unsigned long long*>(s_g + 128);

  const int tid = threadIdx.x, lane = tid & 63, px = lane & 15, kq = lane >> 4;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);       // provably wave-uniform role
  const bool prod = wid >= 4;
  const int wv = wid & 3;
  const int b = blockIdx.y, H = a.H, W = a.W, G = gridDim.x;
  const int nch0 = a.s[0].C / CK;
  const int nch = nch0 + (a.nsrc > 1 ? a.s[1].C / CK : 0);
  const int mt_total = a.Cout / 16;
  const int ntl = (a.ntiles - (int)blockIdx.x + G - 1) / G;       // tiles blockIdx.x, +G, ...
  const int total = ntl * nch;
  const unsigned ring_a = lds_addr(ring);
  const bool tracing = (a.dbg & 32) && blockIdx.x == 0 && blockIdx.y == 0 && lane == 0;
  int s_tn = 0;
  LD_TRACE(prod ? 1 : 0);
  if ((a.dbg & 32) && blockIdx.x == 0 && blockIdx.y == 0 && lane == 0) g_ws_hwid[wid] = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));

  // tile-independent halo coordinates of the slots this producer thread owns
  int qy[HPW], qx[HPW];
#pragma unroll
  for (int r = 0; r < HPW; ++r) {
    const int q = (r * 4 + wv) * 16 + px;
    qy[r] = q / HC;
    qx[r] = q - qy[r] * HC;
  }
  // position in the item sequence (tile ti of this workgroup, K-chunk ch) with the tile origin cached:
  // the three integer divisions run once per tile, not per item
  struct Cursor { int ti, ch, ty0, tx0, ct; };
  auto locate = [&](Cursor& c) {
    const int t = blockIdx.x + c.ti * G;
    c.ct = t % a.nct;
    const int pt = t / a.nct;
    c.ty0 = (pt / a.tiles_x) * TR;
    c.tx0 = (pt % a.tiles_x) * TC;
  };
  auto advance = [&](Cursor& c) {
    if (++c.ch == nch) { c.ch = 0; ++c.ti; locate(c); }
  };
  // K-chunk order is rotated per workgroup so that the CUs of an XCD do not all request the same weight
  // lines from L2 in the same microsecond
  const int rot = a.dbg & 64 ? 0 : (int)((blockIdx.x + blockIdx.y * G) % (unsigned)nch);

  // ---- producer: LDS-DMA of item (ti, ch) into ring stage `st`
  auto dma = [&](const Cursor& c, int st) {
    if (a.dbg & 1) return;
    const int ty0 = c.ty0, tx0 = c.tx0, ct = c.ct;
    const int ch = c.ch + rot >= nch ? c.ch + rot - nch : c.ch + rot;
    const unsigned sa = ring_a + st * STAGE;
    const char* wsrc = reinterpret_cast<const char*>(a.w) + (size_t)ch * 9 * mt_total * 1024 + lane * 16;
#pragma unroll
    for (int r = 0; r < WPW; ++r) {
      const int wb = r * 4 + wv, tap = wb / MT, m = wb - tap * MT;
      glds16(wsrc + (size_t)(tap * mt_total + ct * MT + m) * 1024, __builtin_amdgcn_readfirstlane(sa + wb * 1024));
    }
    const int si = ch >= nch0 ? 1 : 0;
    const SrcDev S = si ? a.s[1] : a.s[0];
    const T* sdata = reinterpret_cast<const T*>(S.data) + (ch - si * nch0) * CK + kq * E;
    const int Hs = S.ups ? H / 2 : H, Ws = S.ups ? W / 2 : W;
#pragma unroll
    for (int r = 0; r < HPW; ++r) {
      int gy = ty0 - 1 + qy[r], gx = tx0 - 1 + qx[r];
      gy = gy < 0 ? 0 : (gy > H - 1 ? H - 1 : gy);               // out-of-image lanes read an in-image pixel;
      gx = gx < 0 ? 0 : (gx > W - 1 ? W - 1 : gx);               // fixup() zeroes their slots after landing
      const int sy = S.ups ? gy >> 1 : gy, sx = S.ups ? gx >> 1 : gx;
      glds16(sdata + (((size_t)b * Hs + sy) * Ws + sx) * S.ld,
             __builtin_amdgcn_readfirstlane(sa + (WBLK + r * 4 + wv) * 1024));
    }
  };
  // ---- producer: in-place prologue of the landed halo blocks this wave loaded (zero padding stays exactly zero)
  auto fixup = [&](const Cursor& c, int st) {
    const int ty0 = c.ty0, tx0 = c.tx0;
    const int ch = c.ch + rot >= nch ? c.ch + rot - nch : c.ch + rot;
    const int si = ch >= nch0 ? 1 : 0;
    const SrcDev S = si ? a.s[1] : a.s[0];
    const bool tr = S.stats != nullptr && !(a.dbg & 16);
    float ca[E], cs[E];
    if (tr) {
      const float* cap = s_coef + (si ? 2 * a.s[0].C : 0) + (ch - si * nch0) * CK + kq * E;
#pragma unroll
      for (int e = 0; e < E; ++e) { ca[e] = cap[e]; cs[e] = cap[S.C + e]; }
    }
    char* hb = ring + st * STAGE + WBLK * 1024 + lane * 16;
#pragma unroll
    for (int r = 0; r < HPW; ++r) {
      const int q = (r * 4 + wv) * 16 + px;
      if (q < NPIX) {
        const int gy = ty0 - 1 + qy[r], gx = tx0 - 1 + qx[r];
        const bool valid = gy >= 0 && gy < H && gx >= 0 && gx < W;
        uint4* ptr = reinterpret_cast<uint4*>(hb + (r * 4 + wv) * 1024);
        if (!valid) {
          *ptr = make_uint4(0u, 0u, 0u, 0u);
        } else if (tr) {
          float v[E];
          unpack16<T>(*ptr, v);
          affine_act_n<P, E>(v, ca, cs, S.act);
          *ptr = pack16<T>(v);
        }
      }
    }
  };

  // ---- setup: producers start the ring, consumers build the GroupNorm coefficients
  Cursor iss{0, 0, 0, 0, 0}, fix{0, 0, 0, 0, 0}, cur{0, 0, 0, 0, 0};   // next to request / to post-process / consumed
  locate(iss);
  fix = cur = iss;
  if (prod) {
    if (total > 0) { dma(iss, 0); advance(iss); }
    if (total > 1) { dma(iss, 1); advance(iss); }
  } else {
    for (int s = 0; s < a.nsrc; ++s) {
      const SrcDev S = s ? a.s[1] : a.s[0];
      if (S.stats && tid < S.groups) {
        const int Gn = S.groups;
        const double* p = S.stats + (size_t)b * LD_STAT_STRIPES * Gn * 2 + 2 * tid;
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int k = 0; k < LD_STAT_STRIPES; ++k) { s1 += p[(size_t)k * Gn * 2]; s2 += p[(size_t)k * Gn * 2 + 1]; }
        const long npix = S.ups ? (long)(H / 2) * (W / 2) : (long)H * W;
        const double inv_n = 1.0 / ((double)npix * (S.C / Gn));
        const double mean = s1 * inv_n;
        double var = s2 * inv_n - mean * mean;
        var = var > 0.0 ? var : 0.0;
        s_g[s * 64 + tid] = (float)mean;
        s_g[s * 64 + 32 + tid] = (float)(1.0 / sqrt(var + 1e-5));
      }
    }
  }
  LD_BARRIER();
  if (!prod) {
    const int trow = a.t_ptr ? *a.t_ptr : 0;
    int off = 0;
    for (int s = 0; s < a.nsrc; ++s) {
      const SrcDev S = s ? a.s[1] : a.s[0];
      if (S.stats) {
        const int C = S.C, gs = C / S.groups;
        const float* film = S.film ? S.film + (long)trow * S.film_tstride + (long)b * S.film_bstride : nullptr;
        for (int c = tid; c < C; c += 256) {
          const int g = c / gs;
          float ga = s_g[s * 64 + 32 + g] * S.gamma[c];
          float sh = S.beta[c] - s_g[s * 64 + g] * ga;
          if (film) {
            const float sc = film[c] + 1.0f;
            ga *= sc;
            sh = sh * sc + film[C + c];
          }
          s_coef[off + c] = ga;
          s_coef[off + C + c] = sh;
        }
      }
      off += 2 * S.C;
    }
  }
  LD_BARRIER();
  if (prod && total > 0) {
    if (total > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    fixup(fix, 0);
    advance(fix);
  }
  LD_BARRIER();
  LD_TRACE(prod ? 1 : 0);

  // ---- consumer state
  f32x4 acc[MT][NW];
  float4 bias[MT];
  T* out = reinterpret_cast<T*>(a.out);
  bool pend = false;
  int pend_par = 0, pend_m0 = 0, par = 0;
  auto flush = [&](int pr, int m0) {
    const int gs = a.Cout / a.ogroups, ngrp_blk = (16 * MT) / gs;
    if (tid < 2 * ngrp_blk) {
      const int gi = tid >> 1, k = tid & 1;
      double acc1 = 0.0;
      for (int w4 = 0; w4 < 4; ++w4)
        for (int c = 0; c < gs; ++c) acc1 += (double)s_part[((pr * 4 + w4) * 2 + k) * 16 * MT + gi * gs + c];
      const int g = (m0 * 16) / gs + gi;
      const int stripe = (blockIdx.x + m0) % LD_STAT_STRIPES;
      atomicAdd(&a.ostats[(((size_t)b * LD_STAT_STRIPES + stripe) * a.ogroups + g) * 2 + k], acc1);
    }
  };
  int st = 0;
  for (int i = 0; i < total; ++i) {
    if (prod) {
      const int st2 = st == 0 ? 2 : st - 1, st1 = st == 2 ? 0 : st + 1;       // stages of items i+2, i+1
      if (i + 2 < total) {
        dma(iss, st2);
        advance(iss);
      }
      LD_TRACE(1);
      if (i + 1 < total) {
        if (i + 2 < total) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        LD_TRACE(1);
        fixup(fix, st1);
        advance(fix);
      }
      LD_LGKM0();
      LD_TRACE(1);
    } else {
      if (pend) {
        if (wid == 0) flush(pend_par, pend_m0);
        pend = false;
      }
      const int ty0 = cur.ty0, tx0 = cur.tx0, m0 = cur.ct * MT, ch = cur.ch;
      if (ch == 0) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          bias[m] = *reinterpret_cast<const float4*>(a.bias + (m0 + m) * 16 + kq * 4);
#pragma unroll
          for (int j = 0; j < NW; ++j) acc[m][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
      }
      LD_TRACE(0);
      if (!(a.dbg & 4)) {
        const char* wb = ring + st * STAGE;
        const char* xb = wb + WBLK * 1024 + kq * 256;
        uint4 A[2][3][MT], Bq[2][NW + 2];
        // opaque per-item copy of the pixel lane: without it hipcc hoists the 3*(NW+2) block-layout fragment
        // addresses (two registers each) out of the item loop and spills
        int pxo = px;
        asm volatile("" : "+v"(pxo));
        auto load_frags = [&](int dx, int set) {
#pragma unroll
          for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int m = 0; m < MT; ++m)
              A[set][dy][m] = *reinterpret_cast<const uint4*>(wb + ((dy * 3 + dx) * MT + m) * 1024 + lane * 16);
#pragma unroll
          for (int rr = 0; rr < NW + 2; ++rr) {
            const int q = (wv * NW + rr) * HC + dx + pxo;
            Bq[set][rr] = *reinterpret_cast<const uint4*>(xb + ((q >> 4) << 10) + ((q & 15) << 4));
          }
        };
        load_frags(0, 0);
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          if (dx + 1 < 3) load_frags(dx + 1, (dx + 1) & 1);
          __builtin_amdgcn_sched_barrier(0);   // keep the next column's reads ABOVE this column's MFMAs (hipcc sinks them to their uses otherwise)
#pragma unroll
          for (int rr = 0; rr < NW + 2; ++rr) {
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
              const int j = rr - dy;
              if (j >= 0 && j < NW) {
#pragma unroll
                for (int m = 0; m < MT; ++m) mma16<T>(acc[m][j], A[dx & 1][dy][m], Bq[dx & 1][rr]);
              }
            }
          }
        }
      }
      LD_TRACE(0);
      if (ch == nch - 1) {   // tile finished: bias, statistics partials, NHWC store
        const int gx = tx0 + px;
        float ssum[MT][4], ssq[MT][4];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          const int co = (m0 + m) * 16 + kq * 4;
          const float4 bv = bias[m];
#pragma unroll
          for (int r = 0; r < 4; ++r) ssum[m][r] = ssq[m][r] = 0.f;
#pragma unroll
          for (int j = 0; j < NW; ++j) {
            const int gy = ty0 + wv * NW + j;
            if (gy < H && gx < W) {
              float v[4] = {acc[m][j][0] + bv.x, acc[m][j][1] + bv.y, acc[m][j][2] + bv.z, acc[m][j][3] + bv.w};
              if (!(a.dbg & 8)) store4<T>(out + (((size_t)b * H + gy) * W + gx) * a.Cout + co, v);
#pragma unroll
              for (int r = 0; r < 4; ++r) { ssum[m][r] += v[r]; ssq[m][r] += v[r] * v[r]; }
            }
          }
        }
        if (a.ostats) {
          float* sp = s_part + (par * 4 + wv) * 2 * 16 * MT;
#pragma unroll
          for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float s1 = wave16_sum(ssum[m][r]), s2 = wave16_sum(ssq[m][r]);
              if (px == 0) {
                sp[m * 16 + kq * 4 + r] = s1;
                sp[16 * MT + m * 16 + kq * 4 + r] = s2;
              }
            }
          pend = true; pend_par = par; pend_m0 = m0;
          par ^= 1;
        }
      }
    }
    LD_TRACE(prod ? 1 : 0);
    LD_BARRIER();
    LD_TRACE(prod ? 1 : 0);
    if (!prod) advance(cur);
    st = st == 2 ? 0 : st + 1;
  }
  if (!prod && pend && wid == 0) flush(pend_par, pend_m0);
  if (tracing)
    for (int k = 0; k < TRACE_EV; ++k) g_ws_trace[wid][k] = k < s_tn ? s_trace[wid * TRACE_EV + k] : 0ull;
}

template <typename T, int MT, int NW>
int launch_ws(const WsDev& a0, hipStream_t st) {
  WsDev a = a0;
  constexpr int TR = 4 * NW, HR = TR + 2, HC = 18;
  constexpr int HBLK = ((HR * HC + 15) / 16 + 3) / 4 * 4;
  const int ctot = a.s[0].C + (a.nsrc > 1 ? a.s[1].C : 0);
  const size_t lds = (size_t)3 * (9 * MT + HBLK) * 1024 + (2 * ctot + 2 * 4 * 2 * 16 * MT + 128) * sizeof(float) + 8 * TRACE_EV * 8;
  if (lds > 160 * 1024) return 0;
  static size_t allowed = 0;
  if (lds > allowed) {
    LD_HIP(ld_allow_lds(conv3x3_ws_kernel<T, MT, NW>, lds));
    allowed = lds;
  }
  a.tiles_x = (a.W + 15) / 16;
  a.nct = a.Cout / (16 * MT);
  a.ntiles = a.tiles_x * ((a.H + TR - 1) / TR) * a.nct;
  int G = (256 + a.B - 1) / a.B;                           // one workgroup per CU over the whole launch
  if (G > a.ntiles) G = a.ntiles;
  hipLaunchKernelGGL((conv3x3_ws_kernel<T, MT, NW>), dim3(G, a.B), dim3(512), lds, st, a);
  LD_LAUNCH_CHECK("conv3x3_ws");
  return 1;
}

}  // namespace

// Returns 1 if this launch is handled here, 0 if another kernel must take it, <0 on error.
int ld_conv3x3_ws_try(const ld_conv3x3_args* p, hipStream_t st) {
  static const int disabled = getenv("LD_CONV_NO_WS") ? atoi(getenv("LD_CONV_NO_WS")) : 0;
  if (disabled || p->Cout % 64 != 0) return 0;
  for (int s = 0; s < p->nsrc; ++s)
    if (p->src[s].gn_stats && p->src[s].gn_groups > 32) return 0;
  if (p->out_stats) {
    const int gs = p->out_groups > 0 ? p->Cout / p->out_groups : 0;
    if (gs < 2 || gs > 64 || 64 % gs != 0) return 0;
  }
  WsDev a;
  a.nsrc = p->nsrc;
  for (int s = 0; s < p->nsrc; ++s) a.s[s] = to_dev(p->src[s]);
  if (p->nsrc == 1) a.s[1] = a.s[0];
  a.w = p->weight; a.bias = p->bias; a.out = p->out; a.ostats = p->out_stats;
  a.ogroups = p->out_groups > 0 ? p->out_groups : 1;
  a.B = p->B; a.H = p->H; a.W = p->W; a.Cout = p->Cout; a.t_ptr = p->t_ptr;
  a.tiles_x = a.nct = a.ntiles = 0;
  static const int dbg = getenv("LD_CONV_DEBUG") ? atoi(getenv("LD_CONV_DEBUG")) : 0;
  a.dbg = dbg;
  return p->dtype == LD_F32 ? launch_ws<float, 4, 2>(a, st) : launch_ws<bf16, 4, 2>(a, st);
}

// Debug hook (not part of the public ABI): cycle-counter trace of the last LD_CONV_DEBUG&32 launch,
// [8 waves][40 events] uint64 followed by the 8 HW_ID values.
extern "C" int ld_debug_ws_trace(unsigned long long* host) {
  LD_HIP(hipDeviceSynchronize());
  LD_HIP(hipMemcpyFromSymbol(host, HIP_SYMBOL(g_ws_trace), sizeof(unsigned long long) * 8 * TRACE_EV));
  unsigned hw[8];
  LD_HIP(hipMemcpyFromSymbol(hw, HIP_SYMBOL(g_ws_hwid), sizeof(hw)));
  for (int i = 0; i < 8; ++i) host[8 * TRACE_EV + i] = hw[i];
  return LD_OK;
}
