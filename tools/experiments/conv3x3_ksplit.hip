// SHELVED EXPERIMENT (round 3, DESIGN finding 52; built only by csrc/build.sh --debug-variants, enabled by LD_CONV_KSPLIT=1).
// Small-map 3x3 convolution, "weights in registers, whole K in LDS, K split over waves" (16-bit storage).
//
// Same op as conv3x3.hip (nn.Conv2d(k=3,p=1) of Block.proj ddpm.py:173, Upsample :117, the last-stage convs :372,:391),
// same fragment layouts, different schedule, for the launches the generic kernel runs as a latency chain: the 32^2 and
// 64^2 maps with 64-256 input channels (a 128-pixel x 32-channel workgroup of the generic kernel walks
// 8 K-chunks with two barriers and one dependent global round trip each -- 2,300 cycles per chunk around 576 cycles
// of MFMA issue, one wave per SIMD, no unit more than a quarter busy).
//
// Here a workgroup of KS waves (KS = number of K-chunks: 8 for 256 input channels, 4 for 128, 2 for 64) owns ONE
// 8 x 16-pixel x 32-channel output tile and
//   * issues EVERY operand load of the tile up front: the halo tile for ALL K-chunks by LDS-DMA
//     (global_load_lds_dwordx4, blocks of 16 pixels x 64 B = 1 KiB per instruction, layout [chunk][block][kq][px][16 B]
//     as in conv3x3_c32.hip: conflict-free fragment reads for every tap), and the weights straight into REGISTERS --
//     they are packed in fragment order (ld_pack_conv_weight), so a wave's A operand of (chunk, tap, m) is one
//     coalesced 1-KiB load.  One wait, one barrier, and the whole K extent is resident;
//   * splits K over the waves: wave ks owns chunk ks (its 18 weight fragments never touch LDS, and no two waves load
//     the same weights: the first version split the pixel rows as well and was bound by the 2x 147 KB of weight loads
//     through the CU's 64 B/clk vector memory path) and all 8 pixel rows of the tile; it runs its 144 MFMAs with no
//     barrier at all;
//   * joins the KS partial sums through LDS (the halo region is dead by then) in a FIXED order -- results do not depend
//     on timing, replays are bitwise equal: with 8 slices, waves 4-7 hand their sums to waves 0-3 first (64 KB of LDS
//     instead of 128); the epilogue (bias, optional addend, GroupNorm statistics, NHWC store) is spread over all waves.
// A handful of barriers per workgroup instead of 2 * nch, and a memory phase whose length is bytes / bandwidth instead
// of nch dependent round trips.
//
// Scope (ld_conv3x3_ksplit_try returns 0 for anything else and the generic kernel takes the launch): bf16 / fp16 storage,
// one source, Cin = 64 / 128 / 256 (2 / 4 / 8 chunks), Cout % 32 == 0, H % 8 == 0, W % 16 == 0, maps of at most
// LD_CONV_KSPLIT_MAX_PX pixels; nearest-x2 upsample of the source, addend and output statistics are supported; a GroupNorm
// prologue on the source is applied in place in LDS after the tile has landed.
#include "../../localdiffusion-hallucination_amd/csrc/common.hip.h"
#include <stdlib.h>

namespace {

struct WsDev {
  SrcDev s;
  const void* w;
  const float* bias;
  const void* addend;
  void* out;
  double* ostats;
  int ogroups, gs_shift;               // gs_shift: log2(channels per group)
  int B, H, W, Cout;
  const int* t_ptr;
  int tiles_x, ntile, ncout, nwg;      // tiles per image, cout tiles, total workgroups
};

// 64 zero bytes in global memory: out-of-image halo lanes of the LDS-DMA read from here, so zero padding needs no
// pass over LDS afterwards (the first version's fix-up pass: 3,300 cycles on every border tile, and at 32^2 every
// tile is one)
__device__ uint4 g_ws_zero[4];

// TRACE (library built with --debug-variants, LD_CONV_KSPLIT_TRACE=1): cycle stamps of one mid-launch workgroup's wave 0
__device__ unsigned long long g_ws_trace[16];
#define WS_STAMP(k) do { if (TRACE && tracing) tr_t[k] = __builtin_readcyclecounter(); } while (0)

template <typename T, int KS, bool TRACE = false>
__global__ __launch_bounds__(KS * 64) void conv3x3_ksplit_kernel(WsDev a) {
  constexpr int E = DT<T>::E, CK = DT<T>::CK;
  constexpr int MT = 2, NWAVE = KS, TR = 8, TC = 16, RW = TR, HR = TR + 2, HC = TC + 2;
  constexpr int NPIX = HR * HC, NBLK = (NPIX + 15) / 16;          // 180 halo pixels, 12 blocks of 16
  constexpr int NCH = KS;                                          // K-chunks of the launch = waves
  constexpr int XCH = NBLK * 1024;                                 // bytes of halo per chunk
  constexpr int NDMA = NBLK;                                       // DMA instructions per wave: NCH*NBLK blocks over NCH waves
  constexpr int NSLOT = KS > 4 ? 4 : KS;                           // 16-KiB partial-sum slots in LDS
  constexpr int NFR = 16 / NWAVE;                                  // output fragments a wave finalises (2, 4 or 8)
  constexpr bool P = DT<T>::precise;
  static_assert(sizeof(T) == 2, "16-bit storage only");

  extern __shared__ __attribute__((aligned(1024))) char smem[];
  char* s_x = smem;                                                // [NCH][NBLK][kq][16 px][16 B]; later the partial sums
  float* s_coef = reinterpret_cast<float*>(smem + (NCH * XCH > NSLOT * 16384 ? NCH * XCH : NSLOT * 16384));
  double* s_stat = reinterpret_cast<double*>(s_coef + 2 * NCH * CK);

  const int tid = threadIdx.x, lane = tid & 63, px = lane & 15, kq = lane >> 4;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  unsigned long long tr_t[16] = {0};
  const bool tracing = TRACE && tid == 0 && blockIdx.x == gridDim.x / 2 + 3;
  WS_STAMP(0);

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

  // ---- (1) the halo tile, every chunk, by LDS-DMA: block g = r*NWAVE + wave -> (chunk g / NBLK, block g % NBLK).
  //          Issued BEFORE the weights: the barrier below needs only these to have landed.
  const SrcDev S = a.s;
  const int Hs = S.ups ? H / 2 : H, Ws = S.ups ? W / 2 : W;
  const unsigned x_a = lds_addr(s_x);
  {
    const T* sdata = reinterpret_cast<const T*>(S.data) + kq * E;
    const T* zero = reinterpret_cast<const T*>(g_ws_zero) + kq * E;
#pragma unroll
    for (int r = 0; r < NDMA; ++r) {
      const int g = r * NWAVE + wv;
      const int ch = g / NBLK, blk = g - ch * NBLK;
      const int q = blk * 16 + px;
      const int hy = (q * 3641) >> 16, hx = q - hy * HC;            // q / 18
      const int gy = ty0 - 1 + hy, gx = tx0 - 1 + hx;
      const bool valid = gy >= 0 && gy < H && gx >= 0 && gx < W;    // (slots q >= 180 are never read)
      const int sy = S.ups ? gy >> 1 : gy, sx = S.ups ? gx >> 1 : gx;
      const T* src = valid ? sdata + ((size_t)(b * Hs + sy) * Ws + sx) * S.ld + ch * CK : zero;
      glds16(src, __builtin_amdgcn_readfirstlane(x_a + g * 1024));
    }
  }
  // ---- (2) weights of this wave's chunk -> registers (fragment order in HBM: one coalesced KiB per load)
  uint4 Areg[9][MT];
  {
    const uint4* wg = reinterpret_cast<const uint4*>(a.w);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int m = 0; m < MT; ++m) Areg[tap][m] = wg[(size_t)((wv * 9 + tap) * mt_total + m0 + m) * 64 + lane];
  }
  WS_STAMP(1);
  // ---- (3) GroupNorm + FiLM + activation coefficients of the source (overlaps the loads in flight)
  const bool has_coef = S.stats != nullptr;
  if (has_coef) {
    const int trow = a.t_ptr ? *a.t_ptr : 0;
    const long npix = S.ups ? (long)(H / 2) * (W / 2) : (long)H * W;
    build_gn_coef(S, b, trow, npix, s_coef, s_stat, tid, NWAVE * 64);
  }
  WS_STAMP(2);
  if (has_coef) {
    // normalise + activate this wave's blocks in place (out-of-image pixels stay exactly zero); needs every load retired
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int r = 0; r < NDMA; ++r) {
      const int g = r * NWAVE + wv;
      const int ch = g / NBLK, blk = g - ch * NBLK;
      const int q = blk * 16 + px;
      const int hy = (q * 3641) >> 16, hx = q - hy * HC;
      const int gy = ty0 - 1 + hy, gx = tx0 - 1 + hx;
      if (q < NPIX && gy >= 0 && gy < H && gx >= 0 && gx < W) {
        uint4* ptr = reinterpret_cast<uint4*>(s_x + g * 1024 + lane * 16);
        float v[E], ca[E], cs[E];
        const float* cap = s_coef + ch * CK + kq * E;
#pragma unroll
        for (int e = 0; e < E; ++e) { ca[e] = cap[e]; cs[e] = cap[S.C + e]; }
        unpack16<T>(*ptr, v);
        affine_act_n<P, E>(v, ca, cs, S.act);
        *ptr = pack16<T>(v);
      }
    }
  } else {
    // the DMAs are the 12 OLDEST entries of this wave's in-order vector-memory queue: they have landed once at most
    // the 18 weight loads are outstanding (hipcc's own waits for the weight registers do not know about the DMAs and
    // can only over-wait)
    asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
  }
  WS_STAMP(3);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  WS_STAMP(4);

  // ---- (4) MFMAs of this wave's K-chunk over the whole tile: no barrier, fragment reads of tap column dx+1 in
  //          flight during the MFMAs of column dx
  f32x4 acc[MT][RW];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int j = 0; j < RW; ++j) acc[m][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  {
    const char* xb = s_x + wv * XCH + kq * 256;
    uint4 Bq[2][RW + 2];
    auto load_frags = [&](int dx, int set) {
#pragma unroll
      for (int rr = 0; rr < RW + 2; ++rr) {
        const int q = rr * HC + dx + px;
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
            for (int m = 0; m < MT; ++m) mma16<T>(acc[m][j], Areg[dy * 3 + dx][m], Bq[dx & 1][rr]);
          }
        }
      }
    }
  }
  WS_STAMP(5);

  // ---- (5) join the K-chunks.  Slot s (16 KiB) = [fragment f = m*8 + row][lane] float4.
  __syncthreads();                                                  // every wave is done reading the halo tile
  WS_STAMP(6);
  float4* s_red = reinterpret_cast<float4*>(smem);
  auto put = [&](int slot) {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int j = 0; j < RW; ++j)
        s_red[(slot * 16 + m * 8 + j) * 64 + lane] = make_float4(acc[m][j][0], acc[m][j][1], acc[m][j][2], acc[m][j][3]);
  };
  if constexpr (KS == 8) {
    // 8 partial sums in 4 slots: waves 4-7 hand theirs to waves 0-3 (wave w reads and then rewrites only slot w)
    if (wv >= 4) put(wv - 4);
    __syncthreads();
    if (wv < 4) {
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int j = 0; j < RW; ++j) {
          const float4 o = s_red[(wv * 16 + m * 8 + j) * 64 + lane];
          acc[m][j][0] += o.x; acc[m][j][1] += o.y; acc[m][j][2] += o.z; acc[m][j][3] += o.w;
        }
      put(wv);
    }
  } else {
    put(wv);
  }
  __syncthreads();
  WS_STAMP(7);

  // ---- (6) epilogue: wave w finalises fragments NFR*w .. NFR*w + NFR-1 (all of one m-tile): fixed order over the slots
  const int f0 = NFR * wv, mf = f0 >> 3;
  const float4 bv = *reinterpret_cast<const float4*>(a.bias + (m0 + mf) * 16 + kq * 4);
  const int gx = tx0 + px;
  T* out = reinterpret_cast<T*>(a.out);
  const T* add = reinterpret_cast<const T*>(a.addend);
  float ssum[4] = {0.f, 0.f, 0.f, 0.f}, ssq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int jj = 0; jj < NFR; ++jj) {
    const int f = f0 + jj;
    float4 v4 = s_red[f * 64 + lane];
#pragma unroll
    for (int k = 1; k < NSLOT; ++k) {
      const float4 o = s_red[(k * 16 + f) * 64 + lane];
      v4.x += o.x; v4.y += o.y; v4.z += o.z; v4.w += o.w;
    }
    float v[4] = {v4.x + bv.x, v4.y + bv.y, v4.z + bv.z, v4.w + bv.w};
    const int gy = ty0 + (f & 7);
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
  WS_STAMP(8);
  if (a.ostats) {
    // a lane's four channels fall into one group (channels per group: a power of two >= 4)
    const float s1 = wave16_sum((ssum[0] + ssum[1]) + (ssum[2] + ssum[3]));
    const float s2 = wave16_sum((ssq[0] + ssq[1]) + (ssq[2] + ssq[3]));
    if (px == 0) {
      s_stat[(wv * 2 + 0) * 4 + kq] = (double)s1;
      s_stat[(wv * 2 + 1) * 4 + kq] = (double)s2;
    }
    __syncthreads();
    const int ngrp_blk = a.gs_shift >= 5 ? 1 : 32 >> a.gs_shift;    // groups inside the 32-channel tile
    if (tid < 2 * ngrp_blk) {
      const int gi = tid >> 1, k = tid & 1;
      double acc1 = 0.0;
#pragma unroll
      for (int w = 0; w < NWAVE; ++w)
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          const int c = ((NFR * w) >> 3) * 16 + q4 * 4;            // first channel (within the tile) behind that value
          if ((c >> a.gs_shift) == gi) acc1 += s_stat[(w * 2 + k) * 4 + q4];
        }
      const int g = ((m0 * 16) >> a.gs_shift) + gi;
      const int stripe = blockIdx.x % LD_STAT_STRIPES;
      atomicAdd(&a.ostats[(((size_t)b * LD_STAT_STRIPES + stripe) * a.ogroups + g) * 2 + k], acc1);
    }
  }
  if (TRACE && tracing) {
    tr_t[9] = __builtin_readcyclecounter();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // this wave's stores have left
    tr_t[10] = __builtin_readcyclecounter();
#pragma unroll
    for (int k = 0; k < 16; ++k) g_ws_trace[k] = tr_t[k];
  }
}

template <typename T, int KS, bool TRACE = false>
int launch_ws(const WsDev& a, hipStream_t st) {
  constexpr int NSLOT = KS > 4 ? 4 : KS;
  const size_t region = (size_t)KS * 12 * 1024 > (size_t)NSLOT * 16384 ? (size_t)KS * 12 * 1024 : (size_t)NSLOT * 16384;
  const size_t lds = region + 2 * KS * 32 * sizeof(float) + 64 * sizeof(double);
  if (lds > 65536) LD_HIP(ld_allow_lds((conv3x3_ksplit_kernel<T, KS, TRACE>), lds));
  LD_LAUNCH((conv3x3_ksplit_kernel<T, KS, TRACE>), dim3(a.nwg), dim3(KS * 64), lds, st, a);
  LD_LAUNCH_CHECK("conv3x3_ws");
  return LD_OK;
}

template <typename T>
int dispatch_ws(const WsDev& a, int nch, hipStream_t st) {
#ifdef LD_DEBUG_VARIANTS
  static const int trace = getenv("LD_CONV_KSPLIT_TRACE") ? atoi(getenv("LD_CONV_KSPLIT_TRACE")) : 0;
  if (trace && nch == 8 && std::is_same<T, bf16>::value) return launch_ws<bf16, 8, true>(a, st);
#endif
  switch (nch) {
    case 2: return launch_ws<T, 2>(a, st);
    case 4: return launch_ws<T, 4>(a, st);
    default: return launch_ws<T, 8>(a, st);
  }
}

}  // namespace

// Returns 1 if this launch is handled here, 0 if another kernel must take it, <0 on error.
int ld_conv3x3_ksplit_try(const ld_conv3x3_args* p, hipStream_t st) {
  static const int disabled = getenv("LD_CONV_KSPLIT_OFF") ? atoi(getenv("LD_CONV_KSPLIT_OFF")) : 0;   // tuning override (A/B)
  static const long max_px = getenv("LD_CONV_KSPLIT_MAX_PX") ? atol(getenv("LD_CONV_KSPLIT_MAX_PX")) : 64 * 64;
  static const int min_cin = getenv("LD_CONV_KSPLIT_MIN_CIN") ? atoi(getenv("LD_CONV_KSPLIT_MIN_CIN")) : 64;
  if (disabled || p->dtype == LD_F32 || p->nsrc != 1) return 0;
  const int cin = p->src[0].C;
  if ((cin != 64 && cin != 128 && cin != 256) || cin < min_cin) return 0;
  if (p->H % 8 != 0 || p->W % 16 != 0 || (long)p->H * p->W > max_px) return 0;
  int gs_shift = 5;
  if (p->out_stats) {
    if (p->out_groups <= 0 || p->Cout % p->out_groups != 0) return 0;
    const int gs = p->Cout / p->out_groups;                 // channels per group: a power of two >= 4
    if (gs < 4 || (gs & (gs - 1)) != 0) return 0;
    gs_shift = __builtin_ctz(gs);
  }
  const ld_src& S = p->src[0];
  const long ld = S.pix_stride > 0 ? S.pix_stride : S.C;
  if ((long)p->B * p->H * p->W * ld >= (1L << 31)) return 0;
  WsDev a;
  a.s = to_dev(S);
  a.w = p->weight; a.bias = p->bias; a.addend = p->addend; a.out = p->out; a.ostats = p->out_stats;
  a.ogroups = p->out_groups > 0 ? p->out_groups : 1;
  a.gs_shift = gs_shift;
  a.B = p->B; a.H = p->H; a.W = p->W; a.Cout = p->Cout; a.t_ptr = p->t_ptr;
  a.tiles_x = p->W / 16;
  a.ntile = a.tiles_x * (p->H / 8);
  a.ncout = p->Cout / 32;
  a.nwg = a.ntile * a.ncout * p->B;
  const int rc = LD_DISPATCH16(p->dtype, dispatch_ws<T>(a, cin / 32, st));
  return rc == LD_OK ? 1 : rc;
}

// Debug hook (not part of the public ABI): cycle stamps of the last traced launch (16 uint64).
extern "C" int ld_debug_ksplit_trace(unsigned long long* host) {
  LD_HIP(hipDeviceSynchronize());
  LD_HIP(hipMemcpyFromSymbol(host, HIP_SYMBOL(g_ws_trace), sizeof(unsigned long long) * 16));
  return LD_OK;
}
