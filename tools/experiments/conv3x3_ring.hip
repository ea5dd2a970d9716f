// EXPERIMENT (round 3, DESIGN finding 58; built only by csrc/build.sh --debug-variants, enabled by LD_CONV_RING=1).
// Small-map 3x3 convolution with BOTH operands streamed into LDS by LDS-DMA, one barrier per K-chunk (16-bit storage).
//
// Same op as conv3x3.hip (nn.Conv2d(k=3,p=1) of Block.proj ddpm.py:173, Upsample :117, the last-stage convs :372,:391),
// same tile (8 x 16 pixels x 16*MT channels, 4 waves of 2 pixel rows), same fragment layouts, same order of MFMAs
// (results are bit-identical to the generic kernel), different staging.  The generic kernel stages a K-chunk through
// REGISTERS: global loads of chunk k+1 before the MFMAs of chunk k, then wait, barrier, 7-8 ds_write_b128 per
// thread, barrier (DESIGN finding 31: 2,300 cycles per chunk around 576 cycles of MFMA issue; finding 34: the
// staging writes and the two barriers are part of the 2/3 of a launch that remains when every load, MFMA and store
// is removed).  Here a chunk's operands -- 12 halo blocks of 16 pixels x 64 B and 9*MT weight blocks, 1 KiB each --
// go from L2 to an R-slot LDS ring with global_load_lds_dwordx4 (inline asm, counted s_waitcnt vmcnt as in
// conv3x3_c32.hip): no staging registers, no LDS writes by the waves, ONE barrier per chunk:
//     wait own DMAs of chunk k -> barrier -> request chunk k+R-1 into the slot chunk k-1 vacated -> MFMAs of chunk k.
// Out-of-image halo lanes read 64 zero bytes from a global constant, so borders need no pass over LDS.
//
// Scope (ld_conv3x3_ring_try returns 0 for anything else and the generic kernel takes the launch): bf16 / fp16 storage,
// no GroupNorm prologue on any source (the RAW launches of finding 42), one or two sources (channel concat), nearest-x2
// upsample, addend and output statistics supported; H % 8 == 0, W % 16 == 0, maps of at most LD_CONV_RING_MAX_PX pixels.
#include "../../localdiffusion-hallucination_amd/csrc/common.hip.h"
#include <stdlib.h>

namespace {

struct RingDev {
  SrcDev s[2];
  int nsrc;
  const void* w;
  const float* bias;
  const void* addend;
  void* out;
  double* ostats;
  int ogroups;
  int B, H, W, Cout;
  int tiles_x, ntile, ncout, nwg;      // tiles per image, cout tiles, total workgroups
};

__device__ uint4 g_ring_zero[4];       // 64 zero bytes: the source of every out-of-image halo lane
__device__ unsigned long long g_ring_trace[16];
#define RG_STAMP(k) do { if (TRACE && tracing) tr_t[k] = __builtin_readcyclecounter(); } while (0)

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// IL: the BPW requests of the next chunk are issued one at a time BETWEEN the MFMA groups of the current chunk instead of
// in one burst in front of them (issuing a chunk's 30 KB takes ~650 cycles -- the CU's vector-memory path accepts ~47 B
// per clock -- during which a wave that issues them back to back does nothing else).
template <typename T, int MT, int R, bool TRACE = false, bool IL = false>
__global__ __launch_bounds__(256) void conv3x3_ring_kernel(RingDev a) {
  constexpr int E = DT<T>::E, CK = DT<T>::CK;
  constexpr int NW = 2, TR = 8, TC = 16, HC = TC + 2;
  constexpr int NBLK = 12, HB = NBLK / 4;                          // 180 halo pixels in 12 blocks of 16; 3 per wave
  constexpr int WBLK = 9 * MT, NB = NBLK + WBLK, BPW = (NB + 3) / 4;   // blocks per chunk, DMA instructions per wave per chunk
  constexpr int WB = BPW - HB;                                     // weight blocks per wave (the last ones may be duplicates)
  constexpr int SLOT = NB * 1024;
  static_assert(sizeof(T) == 2, "16-bit storage only");
  static_assert(R == 2 || R == 3, "ring depth");

  extern __shared__ __attribute__((aligned(1024))) char smem[];
  double* s_stat = reinterpret_cast<double*>(smem + R * SLOT);

  const int tid = threadIdx.x, lane = tid & 63, px = lane & 15, kq = lane >> 4;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  unsigned long long tr_t[16] = {0};
  const bool tracing = TRACE && tid == 0 && blockIdx.x == gridDim.x / 2 + 3;
  RG_STAMP(0);

  // Workgroup -> (image, tile, cout tile).  Workgroups are dealt round-robin over the 8 XCDs; when the grid divides by
  // 8 the logical index is permuted so that the workgroups of ONE XCD cover a contiguous range of (image, tile) with
  // all their cout tiles (they share the halo tile in that XCD's L2).
  int L = blockIdx.x;
  if ((a.nwg & 7) == 0) L = (L & 7) * (a.nwg >> 3) + (L >> 3);
  const int ct = L % a.ncout;
  const int til = (L / a.ncout) % a.ntile;
  const int b = L / (a.ncout * a.ntile);
  const int ty0 = (til / a.tiles_x) * TR, tx0 = (til % a.tiles_x) * TC;
  const int H = a.H, W = a.W;
  const int m0 = ct * MT, mt_total = a.Cout / 16;
  const int nch0 = a.s[0].C / CK;
  const int nch = nch0 + (a.nsrc > 1 ? a.s[1].C / CK : 0);

  // ---- per-lane source addresses of this wave's halo blocks (chunk 0 of each source) and weight blocks
  const char* hp0[HB];
  const char* hp1[HB];
  unsigned hstep[HB];                                              // bytes per chunk: 64 inside the image, 0 for the zero source
#pragma unroll
  for (int r = 0; r < HB; ++r) {
    const int q = (r * 4 + wv) * 16 + px;
    const int hy = (q * 3641) >> 16, hx = q - hy * HC;             // q / 18
    const int gy = ty0 - 1 + hy, gx = tx0 - 1 + hx;
    const bool valid = gy >= 0 && gy < H && gx >= 0 && gx < W;     // (slots q >= 180 are never read)
    const char* zero = reinterpret_cast<const char*>(g_ring_zero) + kq * 16;
    auto src_ptr = [&](const SrcDev& S) -> const char* {
      const int Hs = S.ups ? H / 2 : H, Ws = S.ups ? W / 2 : W;
      const int sy = S.ups ? gy >> 1 : gy, sx = S.ups ? gx >> 1 : gx;
      return reinterpret_cast<const char*>(S.data) + ((((long)b * Hs + sy) * Ws + sx) * S.ld + kq * E) * (long)sizeof(T);
    };
    hp0[r] = valid ? src_ptr(a.s[0]) : zero;
    hp1[r] = valid ? src_ptr(a.s[1]) : zero;
    hstep[r] = valid ? CK * (unsigned)sizeof(T) : 0u;
  }
  const char* wp[WB];
  int wj[WB];
#pragma unroll
  for (int r = 0; r < WB; ++r) {
    int j = r * 4 + wv;
    if (j > WBLK - 1) j = WBLK - 1;                                // (duplicate request: same bytes to the same place)
    const int tap = j / MT, m = j - tap * MT;
    wj[r] = j;
    wp[r] = reinterpret_cast<const char*>(a.w) + ((long)(tap * mt_total + m0 + m) * 64 + lane) * 16;
  }
  const long wstride = 9L * mt_total * 1024;                       // bytes per chunk of packed weights
  const unsigned ring_a = lds_addr(smem);
  auto dma_one = [&](int ch, int slot, int r) {                    // request r (0 .. BPW-1) of chunk ch
    const int si = ch >= nch0 ? 1 : 0;
    const unsigned c = (unsigned)(ch - si * nch0);
    const unsigned sa = ring_a + slot * SLOT;
    if (r < HB) glds16((si ? hp1[r] : hp0[r]) + c * hstep[r], __builtin_amdgcn_readfirstlane(sa + (r * 4 + wv) * 1024));
    else glds16(wp[r - HB] + ch * wstride, __builtin_amdgcn_readfirstlane(sa + (NBLK + wj[r - HB]) * 1024));
  };
  auto dma = [&](int ch, int slot) {                               // always BPW instructions per wave
#pragma unroll
    for (int r = 0; r < BPW; ++r) dma_one(ch, slot, r);
  };

  float4 bias[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) bias[m] = *reinterpret_cast<const float4*>(a.bias + (m0 + m) * 16 + kq * 4);
  RG_STAMP(1);
  dma(0, 0);
  if (R == 3 && nch > 1) dma(1, 1);
  RG_STAMP(2);

  f32x4 acc[MT][NW];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int j = 0; j < NW; ++j) acc[m][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment reads of tap column dx+1 are in flight during the MFMAs of column dx (same order as conv3x3_body.hip.h)
  auto compute = [&](int slot, bool req, int rch, int rslot) {
    const char* xb = smem + slot * SLOT + kq * 256;
    const char* wb = smem + slot * SLOT + NBLK * 1024 + lane * 16;
    uint4 A[2][3][MT], Bq[2][NW + 2];
    auto load_frags = [&](int dx, int set) {
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int m = 0; m < MT; ++m) A[set][dy][m] = *reinterpret_cast<const uint4*>(wb + ((dy * 3 + dx) * MT + m) * 1024);
#pragma unroll
      for (int rr = 0; rr < NW + 2; ++rr) {
        const int q = (wv * NW + rr) * HC + dx + px;
        Bq[set][rr] = *reinterpret_cast<const uint4*>(xb + ((q >> 4) << 10) + ((q & 15) << 4));
      }
    };
    load_frags(0, 0);
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      if (dx + 1 < 3) load_frags(dx + 1, (dx + 1) & 1);
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
        if constexpr (IL) {                       // 12 MFMA groups per chunk: requests 0 .. BPW-1 behind the first BPW
          constexpr int PER = (BPW + 11) / 12;
#pragma unroll
          for (int u = 0; u < PER; ++u) {
            const int r = (dx * (NW + 2) + rr) * PER + u;
            if (r < BPW && req) dma_one(rch, rslot, r);
          }
        }
      }
    }
  };

  int slot = 0;
  for (int ch = 0; ch < nch; ++ch) {
    // own DMAs of chunk ch have landed once only the requests of chunk ch+1 (R = 3) are outstanding
    if (R == 3 && ch + 1 < nch) wait_vm<BPW>();
    else wait_vm<0>();
    if (ch == 2) RG_STAMP(4);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();          // chunk ch is in LDS for every wave; every wave is done with chunk ch-1
    asm volatile("" ::: "memory");
    if (ch == 0) RG_STAMP(3);
    if (ch == 2) RG_STAMP(5);
    const bool req = ch + R - 1 < nch;
    int ns = slot + R - 1;
    if (ns >= R) ns -= R;
    if (!IL && req) dma(ch + R - 1, ns);
    if (ch == 2) RG_STAMP(6);
    compute(slot, req, ch + R - 1, ns);
    if (ch == 2) RG_STAMP(7);
    slot = slot == R - 1 ? 0 : slot + 1;
  }
  RG_STAMP(8);

  // ---- epilogue: bias, optional addend, statistics, NHWC store (as conv3x3_body.hip.h).  lane holds channels
  //      16m+4kq..+3 of pixel px.
  const long obase = (((long)b * H + ty0) * W + tx0) * a.Cout + m0 * 16;
  char* outb = reinterpret_cast<char*>(a.out) + obase * (long)sizeof(T);
  const char* addb = reinterpret_cast<const char*>(a.addend) + obase * (long)sizeof(T);
  const unsigned lane_off = (__umul24(__umul24(wv * NW, W) + px, a.Cout) + kq * 4) * (unsigned)sizeof(T);
  const unsigned row_off = __umul24(W, a.Cout) * (unsigned)sizeof(T);
  float ssum[MT][4], ssq[MT][4];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) ssum[m][r] = ssq[m][r] = 0.f;
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const float4 bv = bias[m];
#pragma unroll
    for (int j = 0; j < NW; ++j) {
      const unsigned off = lane_off + j * row_off + m * 16 * (unsigned)sizeof(T);
      float v[4] = {acc[m][j][0] + bv.x, acc[m][j][1] + bv.y, acc[m][j][2] + bv.z, acc[m][j][3] + bv.w};
      if (a.addend) {
        float ad[4];
        load4<T>(reinterpret_cast<const T*>(addb + off), ad);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += ad[r];
      }
      store4<T>(reinterpret_cast<T*>(outb + off), v);
#pragma unroll
      for (int r = 0; r < 4; ++r) { ssum[m][r] += v[r]; ssq[m][r] += v[r] * v[r]; }
    }
  }
  RG_STAMP(9);
  if (a.ostats) {
    const int gs = a.Cout / a.ogroups;          // channels per group; gs <= 16*MT by construction
    const int ngrp_blk = (16 * MT) / gs;
    const int stripe = til % LD_STAT_STRIPES;
    if ((gs & 3) == 0) {
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const float s1 = wave16_sum((ssum[m][0] + ssum[m][1]) + (ssum[m][2] + ssum[m][3]));
        const float s2 = wave16_sum((ssq[m][0] + ssq[m][1]) + (ssq[m][2] + ssq[m][3]));
        if (px == 0) {
          s_stat[(wv * 2 + 0) * 4 * MT + m * 4 + kq] = (double)s1;
          s_stat[(wv * 2 + 1) * 4 * MT + m * 4 + kq] = (double)s2;
        }
      }
      __syncthreads();
      if (tid < 2 * ngrp_blk) {
        const int gi = tid >> 1, k = tid & 1, q4 = gs >> 2;
        double acc1 = 0.0;
        for (int w4 = 0; w4 < 4; ++w4)
          for (int c = 0; c < q4; ++c) acc1 += s_stat[(w4 * 2 + k) * 4 * MT + gi * q4 + c];
        const int g = (m0 * 16) / gs + gi;
        atomicAdd(&a.ostats[(((size_t)b * LD_STAT_STRIPES + stripe) * a.ogroups + g) * 2 + k], acc1);
      }
    } else {
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float s1 = wave16_sum(ssum[m][r]), s2 = wave16_sum(ssq[m][r]);
          if (px == 0) {
            s_stat[(wv * 2 + 0) * 16 * MT + m * 16 + kq * 4 + r] = (double)s1;
            s_stat[(wv * 2 + 1) * 16 * MT + m * 16 + kq * 4 + r] = (double)s2;
          }
        }
      __syncthreads();
      if (tid < 2 * ngrp_blk) {
        const int gi = tid >> 1, k = tid & 1;
        double acc1 = 0.0;
        for (int w4 = 0; w4 < 4; ++w4)
          for (int c = 0; c < gs; ++c) acc1 += s_stat[(w4 * 2 + k) * 16 * MT + gi * gs + c];
        const int g = (m0 * 16) / gs + gi;
        atomicAdd(&a.ostats[(((size_t)b * LD_STAT_STRIPES + stripe) * a.ogroups + g) * 2 + k], acc1);
      }
    }
  }
  if (TRACE && tracing) {
    tr_t[10] = __builtin_readcyclecounter();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // this wave's stores have left
    tr_t[11] = __builtin_readcyclecounter();
#pragma unroll
    for (int k = 0; k < 16; ++k) g_ring_trace[k] = tr_t[k];
  }
}

template <typename T, int MT, int R, bool TRACE = false, bool IL = false>
int launch_ring(const RingDev& a, hipStream_t st) {
  const size_t lds = (size_t)R * (12 + 9 * MT) * 1024 + 4 * 2 * 16 * MT * sizeof(double);
  if (lds > 65536) LD_HIP(ld_allow_lds((conv3x3_ring_kernel<T, MT, R, TRACE, IL>), lds));
  LD_LAUNCH((conv3x3_ring_kernel<T, MT, R, TRACE, IL>), dim3(a.nwg), dim3(256), lds, st, a);
  LD_LAUNCH_CHECK("conv3x3_ring");
  return LD_OK;
}

template <typename T>
int dispatch_ring(const RingDev& a0, bool mt4, int ring, hipStream_t st) {
  RingDev a = a0;
  a.ncout = a.Cout / (mt4 ? 64 : 32);
  a.nwg = a.ntile * a.ncout * a.B;
  static const int trace = getenv("LD_CONV_RING_TRACE") ? atoi(getenv("LD_CONV_RING_TRACE")) : 0;
  static const int il = getenv("LD_CONV_RING_IL") ? atoi(getenv("LD_CONV_RING_IL")) : 0;
  if (trace && !mt4 && std::is_same<T, bf16>::value) {
    if (il) return launch_ring<bf16, 2, 2, true, true>(a, st);
    return ring == 3 ? launch_ring<bf16, 2, 3, true>(a, st) : launch_ring<bf16, 2, 2, true>(a, st);
  }
  if (il && ring != 3) return mt4 ? launch_ring<T, 4, 2, false, true>(a, st) : launch_ring<T, 2, 2, false, true>(a, st);
  if (mt4) return ring == 3 ? launch_ring<T, 4, 3>(a, st) : launch_ring<T, 4, 2>(a, st);
  return ring == 3 ? launch_ring<T, 2, 3>(a, st) : launch_ring<T, 2, 2>(a, st);
}

}  // namespace

static int g_ring_on = -1;            // -1: take LD_CONV_RING from the environment at the first launch
// Debug hook (not part of the public ABI): switch the kernel on / off inside one process (A/B against the generic kernel).
extern "C" int ld_debug_ring_enable(int on) { g_ring_on = on ? 1 : 0; return LD_OK; }

// Returns 1 if this launch is handled here, 0 if another kernel must take it, <0 on error.
int ld_conv3x3_ring_try(const ld_conv3x3_args* p, hipStream_t st) {
  if (g_ring_on < 0) g_ring_on = getenv("LD_CONV_RING") && atoi(getenv("LD_CONV_RING")) ? 1 : 0;
  if (!g_ring_on) return 0;
  static const long max_px = getenv("LD_CONV_RING_MAX_PX") ? atol(getenv("LD_CONV_RING_MAX_PX")) : 64 * 64;
  static const int min_ch = getenv("LD_CONV_RING_MIN_CHUNKS") ? atoi(getenv("LD_CONV_RING_MIN_CHUNKS")) : 2;
  static const int ring = getenv("LD_CONV_RING_R") ? atoi(getenv("LD_CONV_RING_R")) : 2;
  static const int force_mt = getenv("LD_CONV_RING_MT") ? atoi(getenv("LD_CONV_RING_MT")) : 0;
  static const long mt4_min = getenv("LD_CONV_RING_MT4_MIN_WGS") ? atol(getenv("LD_CONV_RING_MT4_MIN_WGS")) : 256;
  if (p->dtype == LD_F32 || p->weight_terms == 2) return 0;
  if (p->H % 8 != 0 || p->W % 16 != 0 || (long)p->H * p->W > max_px) return 0;
  int ctot = 0;
  for (int s = 0; s < p->nsrc; ++s) {
    if (p->src[s].gn_stats) return 0;                        // RAW launches only
    ctot += p->src[s].C;
    const long ld = p->src[s].pix_stride > 0 ? p->src[s].pix_stride : p->src[s].C;
    if ((long)p->B * p->H * p->W * ld >= (1L << 31)) return 0;
  }
  if (ctot / 32 < min_ch) return 0;
  if (p->out_stats && (p->out_groups <= 0 || p->Cout % p->out_groups != 0 || p->Cout / p->out_groups > 32)) return 0;
  RingDev a;
  a.nsrc = p->nsrc;
  for (int s = 0; s < p->nsrc; ++s) a.s[s] = to_dev(p->src[s]);
  if (p->nsrc == 1) a.s[1] = a.s[0];
  a.w = p->weight; a.bias = p->bias; a.addend = p->addend; a.out = p->out; a.ostats = p->out_stats;
  a.ogroups = p->out_groups > 0 ? p->out_groups : 1;
  a.B = p->B; a.H = p->H; a.W = p->W; a.Cout = p->Cout;
  a.tiles_x = p->W / 16;
  a.ntile = a.tiles_x * (p->H / 8);
  bool mt4 = p->Cout % 64 == 0 && (long)a.ntile * (p->Cout / 64) * p->B >= mt4_min;
  if (force_mt == 2) mt4 = false;
  if (force_mt == 4 && p->Cout % 64 == 0) mt4 = true;
  const int rc = LD_DISPATCH16(p->dtype, dispatch_ring<T>(a, mt4, ring, st));
  return rc == LD_OK ? 1 : rc;
}

// Debug hook (not part of the public ABI): cycle stamps of the last traced launch (16 uint64).
extern "C" int ld_debug_ring_trace(unsigned long long* host) {
  LD_HIP(hipDeviceSynchronize());
  LD_HIP(hipMemcpyFromSymbol(host, HIP_SYMBOL(g_ring_trace), sizeof(unsigned long long) * 16));
  return LD_OK;
}
