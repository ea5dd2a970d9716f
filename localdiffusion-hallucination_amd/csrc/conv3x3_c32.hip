// Persistent deep-ring 3x3 convolution for the C=32 output stages (32->32 and 64->32 at 256^2 / 128^2: the
// "ResBlock conv path" of the north star, AI 144-192 FLOP/B => HBM-bound if nothing else stalls).
//
// Same arithmetic, fragment layouts and fused prologue/epilogue as conv3x3.hip, different schedule.  What
// bounds these launches is bytes in flight: with one 21-KiB halo tile prefetched per workgroup (the previous
// version: 2 workgroups/CU) a CU keeps ~42 KiB outstanding, which at the loaded HBM latency of ~2-4 us is
// ~2.5 TB/s chip-wide -- exactly what was measured, for every schedule tried.  Here:
//   * grid = one 512-thread workgroup per CU (G x B, G*B ~ 256); a workgroup walks tiles g, g+G, ... of ONE
//     image, so weights (<= 2 K-chunks, 36 KiB), bias and GroupNorm coefficients are set up once and the
//     output statistics stay in registers until one flush at the end;
//   * items = (tile, K-chunk) pairs; the halo tile of an item lives in one slot of an R-deep LDS ring
//     (R = 6, 21 KiB per slot) filled by LDS-DMA.  Slot = 21 blocks of 16 halo pixels,
//     each block [kq 0..3][pixel 0..15][16 B] = 1 KiB = ONE global_load_lds_dwordx4: lane l = kq*16 + p reads
//     fragment kq of pixel p (16 pixels x 64 B contiguous in HBM), and a fragment read of 16 consecutive
//     pixels hits 16 distinct 16-B slots (conflict-free for every tap);
//   * at item i every wave (a) requests its blocks of item i+R-1 into the slot item i-1 just vacated, (b) runs
//     the MFMAs of item i, (c) stores the tile if it is finished, (d) waits -- with an exact counted vmcnt: the
//     DMA is issued as inline asm (common.hip.h), every wave issues a fixed number of DMA and store instructions
//     per item because out-of-image lanes read a clamped in-image address and tiles are never ragged
//     (H, W multiples of 16) -- for its OWN blocks of item i+1 and normalises / activates / zero-pads them in
//     place, then ONE workgroup barrier.  R-2 items (63-84 KiB per CU) stay in flight across that barrier and
//     the VALU prologue of item i+1 overlaps the MFMAs of item i of the other wave on the SIMD.
#include "common.hip.h"
#include <stdlib.h>

namespace {

struct C32Dev {
  SrcDev s[2];
  int nsrc;
  const void* w;
  const float* bias;
  void* out;
  double* ostats;
  int ogroups;
  int B, H, W;
  const int* t_ptr;
  int tiles_x, ntiles;
  int dbg;     // LD_CONV_DEBUG ablation bits (0 in production): 1 no DMA, 4 no MFMA, 8 no stores, 16 no transform
};

// LD_CONV_DEBUG bit 32: wave 0 of workgroup (0,0) keeps cycle-counter stamps (start, setup done, ring filled, end
// of each of the first 10 items, end) in registers and dumps them here when it finishes (ld_debug_c32_trace).
__device__ unsigned long long g_c32_trace[16];

// s_waitcnt vmcnt(n) for a wave-uniform runtime n (the instruction takes an immediate): binary decision tree,
// 6 scalar compares for any n in 0..63
template <int LO, int HI>
__device__ __forceinline__ void wait_vmcnt_range(int n) {
  if constexpr (LO == HI) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LO) : "memory");
  } else {
    constexpr int MID = (LO + HI) / 2;
    if (n <= MID) wait_vmcnt_range<LO, MID>(n);
    else wait_vmcnt_range<MID + 1, HI>(n);
  }
}
__device__ __forceinline__ void wait_vmcnt(int n) {
  if (n < 0 || n > 63) n = 0;
  wait_vmcnt_range<0, 63>(n);
}

// NWAVE = 8: the 512-thread workgroup of the description (16 x 16-pixel tiles, 21-KiB ring slots).  NWAVE = 4 ("lite",
// experiment LD_CONV_C32_LITE): 256 threads, 8 x 16-pixel tiles, 12-KiB slots -- a footprint (R = 4: 48 KB of LDS) that
// lets the persistent workgroups of BOTH sub-batch streams, and other kernels, share a CU.
template <typename T, int NCH, int R, int DBG, int NWAVE = 8>
__global__ __launch_bounds__(64 * NWAVE) void conv3x3_c32_kernel(C32Dev a) {
  constexpr int E = DT<T>::E, CK = DT<T>::CK;
  constexpr int NTHR = 64 * NWAVE;
  constexpr int MT = 2, NW = 2, TR = NW * NWAVE, TC = 16, HR = TR + 2, HC = TC + 2;
  constexpr int NPIX = HR * HC, NBLK = (NPIX + 15) / 16, BPW = (NBLK + NWAVE - 1) / NWAVE;   // 324 px, 21 blocks, 3
  constexpr int XBUF = NBLK * 1024;                                                         // bytes per ring slot
  constexpr int WCH = 9 * MT * 1024;                                                        // bytes per weight chunk
  constexpr int MS = sizeof(T) == 2 ? 2 : 1;                                                // m-tiles per store (pair_frag16: 16-byte stores)
  static_assert(MT % MS == 0, "m-tiles are stored in pairs");
  constexpr int NST = MT / MS * NW;                                                         // store instructions per wave per tile
  constexpr bool P = DT<T>::precise;

  // Single-chunk launches (32 -> 32) keep the wave's 18 weight fragments in REGISTERS for the whole launch: every wave
  // re-reading them from LDS for every tile was 144 of the 240 KB of LDS operand traffic per item (1,100 of ~1,900
  // LDS-bandwidth cycles per item against 1,150 cycles of MFMA), and it frees 18 KB of LDS.
  constexpr bool WREG = NCH == 1;
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  char* s_w = smem;                                   // [NCH][9][MT][1 KiB]  (WREG: not allocated)
  char* s_x = smem + (WREG ? 0 : NCH * WCH);          // [R slots][NBLK][kq][16 px][16 B]
  float* s_coef = reinterpret_cast<float*>(s_x + R * XBUF);
  double* s_stat = reinterpret_cast<double*>(s_x);    // [8 waves][2][32] / coefficient scratch (before / after the ring)

  const int tid = threadIdx.x, lane = tid & 63, px = lane & 15, kq = lane >> 4;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.y, H = a.H, W = a.W, G = gridDim.x;
  const int nch0 = a.s[0].C / CK;
  const int ntl = (a.ntiles - (int)blockIdx.x + G - 1) / G;       // tiles blockIdx.x, +G, ...
  const int total = ntl * NCH;
  const int nb_w = (NBLK - wv + NWAVE - 1) / NWAVE;                // DMA instructions of this wave per item (3 or 2)
  const unsigned ring_a = lds_addr(s_x);
  const bool tracing = (DBG & 32) && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0;
  unsigned long long tr_t[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) tr_t[k] = 0;
  if (tracing) tr_t[0] = __builtin_readcyclecounter();

  // ---- one-time setup: weights -> LDS, bias -> registers, GroupNorm coefficients -> LDS
  uint4 Areg[WREG ? 9 : 1][MT];
  {
    const uint4* wg = reinterpret_cast<const uint4*>(a.w);
    if constexpr (WREG) {
#pragma unroll
      for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int m = 0; m < MT; ++m) Areg[tap][m] = wg[(tap * MT + m) * 64 + lane];
    } else {
      for (int u = tid; u < NCH * 9 * MT * 64; u += NTHR) *reinterpret_cast<uint4*>(s_w + (size_t)u * 16) = wg[u];
    }
  }
  float4 bias[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) bias[m] = *reinterpret_cast<const float4*>(a.bias + m * 16 + kq * 4);
  if (a.s[0].stats != nullptr || (a.nsrc > 1 && a.s[1].stats != nullptr)) {
    const int trow = a.t_ptr ? *a.t_ptr : 0;
    int off = 0;
    for (int s = 0; s < a.nsrc; ++s) {
      const SrcDev S = s ? a.s[1] : a.s[0];
      if (S.stats) {
        const long npix = S.ups ? (long)(H / 2) * (W / 2) : (long)H * W;
        build_gn_coef<DT<T>::precise>(S, b, trow, npix, s_coef + off, s_stat, tid, NTHR);
      }
      off += 2 * S.C;
    }
  }
  __syncthreads();                                    // weights / coefficients visible; s_stat scratch is dead
  // Retire every compiler-visible global load HERE: hipcc's s_waitcnt bookkeeping does not see the asm DMAs, so
  // a wait it placed at the first use of `bias` inside the item loop (vmcnt(1..2), counting only its own stores)
  // would drain the whole ring on every tile.
#pragma unroll
  for (int m = 0; m < MT; ++m) asm volatile("" ::"v"(bias[m].x), "v"(bias[m].y), "v"(bias[m].z), "v"(bias[m].w));
  if constexpr (WREG) {
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int m = 0; m < MT; ++m)
        asm volatile("" ::"v"(Areg[tap][m].x), "v"(Areg[tap][m].y), "v"(Areg[tap][m].z), "v"(Areg[tap][m].w));
  }

  if (tracing) tr_t[1] = __builtin_readcyclecounter();
  // tile-independent halo coordinates of the ring blocks this thread loads and post-processes
  int qy[BPW], qx[BPW];
#pragma unroll
  for (int r = 0; r < BPW; ++r) {
    const int q = (r * NWAVE + wv) * 16 + px;
    qy[r] = q / HC;
    qx[r] = q - qy[r] * HC;
  }
  // Position in the item sequence.  Tiles step by G through the image; the step is applied as a (row, column)
  // increment so that no integer division runs inside the item loop (measured: the divisions, a 49-way waitcnt
  // switch and 64-bit address products made an EMPTY item cost 2,000 cycles, as much as its useful work).
  struct Cursor { int ch, ty, tx, ty0, tx0; };
  const int step_y = G / a.tiles_x, step_x = G % a.tiles_x;
  auto advance = [&](Cursor& c) {
    if (++c.ch == NCH) {
      c.ch = 0;
      c.ty += step_y;
      c.tx += step_x;
      if (c.tx >= a.tiles_x) { c.tx -= a.tiles_x; ++c.ty; }
      c.ty0 = c.ty * TR;
      c.tx0 = c.tx * TC;
    }
  };
  // ---- LDS-DMA of this wave's blocks of one item into ring slot `slot` (always nb_w instructions)
  auto dma = [&](const Cursor& c, int slot) {
    if (DBG & 1) return;
    const int si = c.ch >= nch0 ? 1 : 0;
    const SrcDev S = si ? a.s[1] : a.s[0];
    const T* sdata = reinterpret_cast<const T*>(S.data) + (c.ch - si * nch0) * CK + kq * E;
    const int Hs = S.ups ? H / 2 : H, Ws = S.ups ? W / 2 : W;
    const int row0 = b * Hs;                                       // element offsets fit 32 bits (checked on the host)
    const unsigned sa = ring_a + slot * XBUF;
#pragma unroll
    for (int r = 0; r < BPW; ++r) {
      const int blk = r * NWAVE + wv;
      if (blk < NBLK) {                                            // wave-uniform
        int gy = c.ty0 - 1 + qy[r], gx = c.tx0 - 1 + qx[r];
        gy = gy < 0 ? 0 : (gy > H - 1 ? H - 1 : gy);               // out-of-image (and padding) lanes read an in-image
        gx = gx < 0 ? 0 : (gx > W - 1 ? W - 1 : gx);               // pixel; fixup() zeroes their slots after landing
        const int sy = S.ups ? gy >> 1 : gy, sx = S.ups ? gx >> 1 : gx;
        glds16(sdata + ((row0 + sy) * Ws + sx) * S.ld, __builtin_amdgcn_readfirstlane(sa + blk * 1024));
      }
    }
  };
  // ---- in-place prologue of the landed blocks this wave loaded (zero padding stays exactly zero)
  auto fixup = [&](const Cursor& c, int slot) {
    const int si = c.ch >= nch0 ? 1 : 0;
    const SrcDev S = si ? a.s[1] : a.s[0];
    const bool tr = S.stats != nullptr && !(DBG & 16);
    const bool border = c.ty0 == 0 || c.tx0 == 0 || c.ty0 + TR >= H || c.tx0 + TC >= W;
    if (!tr && !border) return;
    float ca[E], cs[E];
    if (tr) {
      const float* cap = s_coef + (si ? 2 * a.s[0].C : 0) + (c.ch - si * nch0) * CK + kq * E;
#pragma unroll
      for (int e = 0; e < E; ++e) { ca[e] = cap[e]; cs[e] = cap[S.C + e]; }
    }
    char* xb = s_x + slot * XBUF + lane * 16;
#pragma unroll
    for (int r = 0; r < BPW; ++r) {
      const int blk = r * NWAVE + wv, q = blk * 16 + px;
      if (blk < NBLK && q < NPIX) {
        const int gy = c.ty0 - 1 + qy[r], gx = c.tx0 - 1 + qx[r];
        const bool valid = gy >= 0 && gy < H && gx >= 0 && gx < W;
        uint4* ptr = reinterpret_cast<uint4*>(xb + blk * 1024);
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

  float ssum[MT][4], ssq[MT][4];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) ssum[m][r] = ssq[m][r] = 0.f;
  f32x4 acc[MT][NW];
  T* out = reinterpret_cast<T*>(a.out);

  // ---- fill the ring: items 0 .. R-2 requested, item 0 landed and post-processed
  Cursor iss{0, (int)blockIdx.x / a.tiles_x, (int)blockIdx.x % a.tiles_x, 0, 0}, fix, cur;
  iss.ty0 = iss.ty * TR;
  iss.tx0 = iss.tx * TC;
  fix = cur = iss;
  int requested = 0;
  for (; requested < R - 1 && requested < total; ++requested) { dma(iss, requested); advance(iss); }
  if (total > 0) {
    wait_vmcnt((DBG & 1) ? 0 : (requested - 1) * nb_w);
    fixup(fix, 0);
    advance(fix);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  if (tracing) tr_t[2] = __builtin_readcyclecounter();
  // Phase order inside an item is skewed between the two waves that share a SIMD (wave w and w+4): waves 0-3 run
  // MFMA -> store -> prologue of the next item, waves 4-7 run prologue -> MFMA -> store, so one wave's VALU / LDS /
  // store phase overlaps the other's MFMAs instead of both queueing on the same pipe (measured in lockstep:
  // 1,300 cycles of MFMA + 2,760 of prologue + 650 of epilogue per item, strictly one after the other).
  const bool prologue_first = wv >= NWAVE / 2;
  int slot = 0;
  for (int i = 0; i < total; ++i) {
    if (tracing && i == 3) tr_t[3] = __builtin_readcyclecounter();
    // (a) request item i+R-1 into the slot item i-1 vacated (all waves are past barrier i-1)
    if (i + R - 1 < total) {
      dma(iss, slot == 0 ? R - 1 : slot - 1);
      advance(iss);
    }
    if (tracing && i == 3) tr_t[4] = __builtin_readcyclecounter();
    // (b)+(c) MFMAs of item i (fragment reads of tap column dx+1 are in flight during the MFMAs of column dx);
    //         tile finished: bias, statistics, store (exactly NST store instructions: tiles are never ragged)
    auto mfma_and_store = [&]() {
      if (cur.ch == 0) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int j = 0; j < NW; ++j) acc[m][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      if (!(DBG & 4)) {
        const char* xb = s_x + slot * XBUF + kq * 256;
        const char* wb = s_w + cur.ch * WCH + lane * 16;
        uint4 A[WREG ? 1 : 2][WREG ? 1 : 3][MT], Bq[2][NW + 2];
        auto load_frags = [&](int dx, int set) {
          if constexpr (!WREG) {
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
              for (int m = 0; m < MT; ++m) A[set][dy][m] = *reinterpret_cast<const uint4*>(wb + ((dy * 3 + dx) * MT + m) * 1024);
          }
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
                for (int m = 0; m < MT; ++m) {
                  if constexpr (WREG) mma16<T>(acc[m][j], Areg[dy * 3 + dx][m], Bq[dx & 1][rr]);
                  else mma16<T>(acc[m][j], A[dx & 1][dy][m], Bq[dx & 1][rr]);
                }
              }
            }
          }
        }
      }
      if (cur.ch == NCH - 1 && !(DBG & 8)) {
        const int gx = cur.tx0 + px;
#pragma unroll
        for (int m = 0; m < MT; m += MS) {
#pragma unroll
          for (int j = 0; j < NW; ++j) {
            const int gy = cur.ty0 + wv * NW + j;
            float v[MS][4];
#pragma unroll
            for (int mm = 0; mm < MS; ++mm) {
              const float4 bv = bias[m + mm];
              v[mm][0] = acc[m + mm][j][0] + bv.x; v[mm][1] = acc[m + mm][j][1] + bv.y;
              v[mm][2] = acc[m + mm][j][2] + bv.z; v[mm][3] = acc[m + mm][j][3] + bv.w;
            }
            T* pix = out + (size_t)(((b * H + gy) * W + gx) * 32 + m * 16);
            if constexpr (MS == 2)      // ONE 16-byte store per lane for the pair of m-tiles: a wave writes 1 KiB contiguous
              store16_out(reinterpret_cast<char*>(pix) + pair_frag16_off(kq), pair_frag16<T>(v[0], v[1]));
            else
              store4<T>(pix + kq * 4, v[0]);
#pragma unroll
            for (int mm = 0; mm < MS; ++mm)
#pragma unroll
              for (int r = 0; r < 4; ++r) { ssum[m + mm][r] += v[mm][r]; ssq[m + mm][r] += v[mm][r] * v[mm][r]; }
          }
        }
      }
    };
    // (d) own blocks of item i+1: wait, post-process.  Younger than that DMA in this wave's in-order VM queue:
    //     the DMAs of items i+2 .. min(i+R-1, total-1) and the stores of every tile this wave finished since the
    //     DMA was issued (at iteration i+2-R, or in the ring fill) -- iterations first_it .. i when the stores of
    //     this iteration are already out (MFMA first), first_it .. i-1 otherwise.
    auto prologue_next = [&](bool stored_this_iteration) {
      if (i + 1 >= total) return;
      const int last = i + R - 1 < total - 1 ? i + R - 1 : total - 1;
      const int first_it = i + 2 - R > 0 ? i + 2 - R : 0;
      const int hi_it = stored_this_iteration ? i : i - 1;          // last iteration whose stores are counted
      int tile_ends;
      if (NCH == 1) tile_ends = hi_it - first_it + 1;
      else tile_ends = (hi_it + 1) / 2 - first_it / 2;              // iterations k in [first_it, hi_it] with k odd
      if (tile_ends < 0) tile_ends = 0;
      const int nst = (DBG & 8) ? 0 : NST * tile_ends;
      if (!(DBG & 128)) wait_vmcnt((DBG & 1) ? nst : (last - (i + 1)) * nb_w + nst);
      if (!(DBG & 64)) fixup(fix, slot == R - 1 ? 0 : slot + 1);
      advance(fix);
    };
    if (prologue_first) {
      prologue_next(false);
      if (tracing && i == 3) tr_t[5] = __builtin_readcyclecounter();
      mfma_and_store();
    } else {
      mfma_and_store();
      if (tracing && i == 3) tr_t[5] = __builtin_readcyclecounter();
      prologue_next(true);
    }
    if (tracing && i == 3) tr_t[8] = __builtin_readcyclecounter();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (tracing && i == 3) tr_t[9] = __builtin_readcyclecounter();
    advance(cur);
    slot = slot == R - 1 ? 0 : slot + 1;
  }
  if (tracing) tr_t[13] = __builtin_readcyclecounter();

  if (a.ostats) {                                                 // one flush per workgroup
    // (every DMA of this wave has landed -- each item waited for its own -- so only stores are outstanding: they need
    //  no wait here; a vmcnt(0) would expose their round trip in front of the statistics)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float s1 = wave16_sum(ssum[m][r]), s2 = wave16_sum(ssq[m][r]);
        if (px == 0) {
          s_stat[(wv * 2 + 0) * 32 + m * 16 + kq * 4 + r] = (double)s1;
          s_stat[(wv * 2 + 1) * 32 + m * 16 + kq * 4 + r] = (double)s2;
        }
      }
    __syncthreads();
    const int gs = 32 / a.ogroups;
    const int stripe = blockIdx.x % LD_STAT_STRIPES;
    if (gs <= 16) {
      // lanes 0-31 of wave 0: the sums of channel `lane`, lanes 32-63: the sums of squares; eight independent LDS reads
      // (one per wave), then the gs channels of a group meet by DPP inside their 16-lane row (conv3x3_body.hip.h)
      if (tid < 64) {
        const int k = tid >> 5, ch = tid & 31;
        double v = 0.0;
#pragma unroll
        for (int w8 = 0; w8 < NWAVE; ++w8) v += s_stat[(w8 * 2 + k) * 32 + ch];
        v = row_group_sum_d(v, gs);
        if ((ch & (gs - 1)) == 0)
          atomicAdd(&a.ostats[(((size_t)b * LD_STAT_STRIPES + stripe) * a.ogroups + ch / gs) * 2 + k], v);
      }
    } else if (tid < 2 * a.ogroups) {
      const int gi = tid >> 1, k = tid & 1;
      double acc1 = 0.0;
      for (int w8 = 0; w8 < NWAVE; ++w8)
        for (int c = 0; c < gs; ++c) acc1 += s_stat[(w8 * 2 + k) * 32 + gi * gs + c];
      atomicAdd(&a.ostats[(((size_t)b * LD_STAT_STRIPES + stripe) * a.ogroups + gi) * 2 + k], acc1);
    }
  }
  if (tracing) {
    tr_t[14] = __builtin_readcyclecounter();
#pragma unroll
    for (int k = 0; k < 16; ++k) g_c32_trace[k] = tr_t[k];
  }
}

template <typename T, int NCH, int R, int DBG>
int launch_c32_dbg(C32Dev& a, size_t lds, dim3 grid, hipStream_t st) {
  if (lds > 65536) LD_HIP(ld_allow_lds((conv3x3_c32_kernel<T, NCH, R, DBG>), lds));   // cached per device
  LD_LAUNCH((conv3x3_c32_kernel<T, NCH, R, DBG>), grid, dim3(512), lds, st, a);
  LD_LAUNCH_CHECK("conv3x3_c32");
  return LD_OK;
}

template <typename T, int NCH, int R>
int launch_c32(const C32Dev& a0, hipStream_t st) {
  C32Dev a = a0;
  a.tiles_x = a.W / 16;
  a.ntiles = a.tiles_x * (a.H / 16);
  const int ctot = a.s[0].C + (a.nsrc > 1 ? a.s[1].C : 0);
  const size_t lds = (NCH == 1 ? 0 : (size_t)NCH * 9 * 2 * 1024) + (size_t)R * 21 * 1024 + 2 * ctot * sizeof(float);
#ifdef LD_DEBUG_VARIANTS          // experiment-only (finding 53): a grid for part of the chip
  static const int cus = getenv("LD_CONV_C32_CUS") ? atoi(getenv("LD_CONV_C32_CUS")) : 256;
#else
  constexpr int cus = 256;
#endif
  int G = (cus + a.B - 1) / a.B;                       // one workgroup per CU over the whole launch
  if (G > a.ntiles) G = a.ntiles;
  const dim3 grid(G, a.B);
  // the LD_CONV_DEBUG ablation / trace variants are separate instantiations: the production kernel carries none
  // of their branches (only a few bit patterns are built; anything else runs the production kernel)
#ifdef LD_DEBUG_VARIANTS
  if (std::is_same<T, bf16>::value && NCH == 1) {
    switch (a.dbg) {
      case 32: return launch_c32_dbg<T, NCH, R, 32>(a, lds, grid, st);
      case 13: return launch_c32_dbg<T, NCH, R, 13>(a, lds, grid, st);
      case 45: return launch_c32_dbg<T, NCH, R, 45>(a, lds, grid, st);
      case 205: return launch_c32_dbg<T, NCH, R, 205>(a, lds, grid, st);
      case 1: return launch_c32_dbg<T, NCH, R, 1>(a, lds, grid, st);
      case 4: return launch_c32_dbg<T, NCH, R, 4>(a, lds, grid, st);
      case 8: return launch_c32_dbg<T, NCH, R, 8>(a, lds, grid, st);
      case 12: return launch_c32_dbg<T, NCH, R, 12>(a, lds, grid, st);
      default: break;
    }
  }
#endif
  return launch_c32_dbg<T, NCH, R, 0>(a, lds, grid, st);
}

#ifdef LD_DEBUG_VARIANTS
// "lite" geometry (NWAVE = 4; experiment, finding 66: slower in the sampler): 8 x 16-pixel tiles, 256-thread
// workgroups, LD_CONV_C32_LITE_WGS of them per launch
template <typename T, int R>
int launch_c32_lite(const C32Dev& a0, hipStream_t st) {
  C32Dev a = a0;
  a.tiles_x = a.W / 16;
  a.ntiles = a.tiles_x * (a.H / 8);
  const size_t lds = (size_t)R * 12 * 1024 + 2 * a.s[0].C * sizeof(float);
  static const int wgs = getenv("LD_CONV_C32_LITE_WGS") ? atoi(getenv("LD_CONV_C32_LITE_WGS")) : 256;
  int G = (wgs + a.B - 1) / a.B;
  if (G > a.ntiles) G = a.ntiles;
  if (lds > 65536) LD_HIP(ld_allow_lds((conv3x3_c32_kernel<T, 1, R, 0, 4>), lds));
  LD_LAUNCH((conv3x3_c32_kernel<T, 1, R, 0, 4>), dim3(G, a.B), dim3(256), lds, st, a);
  LD_LAUNCH_CHECK("conv3x3_c32 (lite)");
  return LD_OK;
}
#endif

}  // namespace

// Returns 1 if this launch is handled here, 0 if the generic kernel must take it, <0 on error.
int ld_conv3x3_c32_try(const ld_conv3x3_args* p, hipStream_t st) {
  const LdTuning& tn = ld_tuning();
  if (!tn.conv_c32 || p->addend || p->Cout != 32 || p->H < 32 || p->W < 32 || p->H % 16 != 0 || p->W % 16 != 0) return 0;
  const int ck = p->dtype == LD_F32 ? 16 : 32;
  int ctot = 0;
  for (int s = 0; s < p->nsrc; ++s) ctot += p->src[s].C;
  // Single K-chunk only: measured in situ (cfg3) the ring gains 1-4 us per 32->32 @256^2 launch over the
  // register-staged kernel but LOSES 1-6 us on the two-chunk 64->32 ones (the kernel template still carries NCH).
  if (ctot != ck || p->nsrc != 1) return 0;
  if (p->out_stats && (p->out_groups <= 0 || 32 % p->out_groups != 0)) return 0;
  const long min_tiles = tn.conv_c32_min_tiles;
  const long tiles = (long)(p->W / 16) * (p->H / 16) * p->B;
#ifdef LD_DEBUG_VARIANTS
  // experiment (finding 66): launches of LD_CONV_C32_LITE .. min_tiles - 1 tiles -- the 4-patch launches of the
  // two-sub-batch regime -- on the 256-thread geometry (0 = off)
  static const long lite_min = getenv("LD_CONV_C32_LITE") ? atol(getenv("LD_CONV_C32_LITE")) : 0;
  static const int lite_ring = getenv("LD_CONV_C32_LITE_R") ? atoi(getenv("LD_CONV_C32_LITE_R")) : 4;
  const bool lite = lite_min > 0 && tiles >= lite_min && tiles < min_tiles && p->dtype != LD_F32;
#else
  constexpr bool lite = false;
#endif
  if (tiles < min_tiles && !lite) return 0;            // too few tiles to amortise a persistent workgroup
  for (int s = 0; s < p->nsrc; ++s) {                  // the kernel uses 32-bit element offsets
    const long ld = p->src[s].pix_stride > 0 ? p->src[s].pix_stride : p->src[s].C;
    if ((long)p->B * p->H * p->W * ld >= (1L << 31)) return 0;
  }
  C32Dev a;
  a.nsrc = p->nsrc;
  for (int s = 0; s < p->nsrc; ++s) a.s[s] = to_dev(p->src[s]);
  if (p->nsrc == 1) a.s[1] = a.s[0];
  a.w = p->weight; a.bias = p->bias; a.out = p->out; a.ostats = p->out_stats;
  a.ogroups = p->out_groups > 0 ? p->out_groups : 1;
  a.B = p->B; a.H = p->H; a.W = p->W; a.t_ptr = p->t_ptr; a.tiles_x = a.ntiles = 0;
  a.dbg = 0;
  int rc;
#ifdef LD_DEBUG_VARIANTS          // ablation switches, ring depths 2-4 and the lite geometry: --debug-variants builds only
  static const int dbg = getenv("LD_CONV_DEBUG") ? atoi(getenv("LD_CONV_DEBUG")) : 0;
  a.dbg = dbg;
  static const int ring = getenv("LD_CONV_C32_R") ? atoi(getenv("LD_CONV_C32_R")) : 6;
  if (lite) {
    if (p->dtype == LD_F16) rc = lite_ring == 6 ? launch_c32_lite<f16, 6>(a, st) : launch_c32_lite<f16, 4>(a, st);
    else rc = lite_ring == 6 ? launch_c32_lite<bf16, 6>(a, st) : (lite_ring == 3 ? launch_c32_lite<bf16, 3>(a, st) : launch_c32_lite<bf16, 4>(a, st));
  } else if (p->dtype == LD_BF16 && ring == 4) rc = launch_c32<bf16, 1, 4>(a, st);
  else if (p->dtype == LD_F16 && ring == 4) rc = launch_c32<f16, 1, 4>(a, st);
  else if (p->dtype == LD_BF16 && ring == 2) rc = launch_c32<bf16, 1, 2>(a, st);
  else if (p->dtype == LD_BF16 && ring == 3) rc = launch_c32<bf16, 1, 3>(a, st);
  else
#endif
  if (p->dtype == LD_F32) rc = launch_c32<float, 1, 6>(a, st);
  else if (p->dtype == LD_F16) rc = launch_c32<f16, 1, 6>(a, st);
  else rc = launch_c32<bf16, 1, 6>(a, st);
  if (rc == LD_OK) ld_count(LD_COUNTER_CONV3X3_C32);
  return rc == LD_OK ? 1 : rc;
}

// Debug hook (not part of the public ABI): cycle stamps of the last LD_CONV_DEBUG&32 launch (16 uint64).
extern "C" int ld_debug_c32_trace(unsigned long long* host) {
  LD_HIP(hipDeviceSynchronize());
  LD_HIP(hipMemcpyFromSymbol(host, HIP_SYMBOL(g_c32_trace), sizeof(unsigned long long) * 16));
  return LD_OK;
}
