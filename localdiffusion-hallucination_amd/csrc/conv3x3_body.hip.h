// Device side of the 3x3 convolution (conv3x3.hip has the description, the launch heuristics and the C entry point):
// the argument block, the tile function and its __global__ wrapper.  Included by conv3x3.hip (and by the archived stage-program experiment, tools/experiments/stage_programs.hip).
#pragma once
#include "common.hip.h"

namespace {

struct Conv3Dev {    // pointers together, scalars together: the argument block arrives in few, wide s_loads (finding 67)
  SrcDev s[2];
  const void* w;
  const float* bias;
  void* out;
  double* ostats;
  const int* t_ptr;
  const void* addend;
  const void* w2;      // SIDE: the ResnetBlock's res_conv (1x1 over the same concatenated input, ddpm.py:198) as a second output of this launch
  const float* bias2;  //   packed like a 1x1 weight ([chunk][cout-tile][kq][row][e]: one fragment per m-tile and chunk), its bias,
  void* out2;          //   and where res_conv(x) + bias2 goes (NHWC [B,H,W,Cout], no statistics)
  int nsrc;
  int ogroups;
  int B, H, W, Cout;
  int tiles_x;
  int wsplit;  // 1: two-term weights (pack.hip): 2*nch virtual chunks, source chunk v >> 1, weight chunk v
  int dbg;     // ablation switches (LD_CONV_DEBUG env, 0 in production): 1 no halo loads, 2 no weight loads, 4 no MFMA, 8 no stores
};

// LD_CONV_DEBUG=64: cycle stamps of one workgroup from the middle of the launch (tools/trace_conv.py)
__device__ unsigned long long g_conv_trace[24];
// LD_CONV_DEBUG=128 (debug-variants builds): every launch owns a slot (a.dbg >> 8) and leaves its position on the 100 MHz
// real-time clock there -- [0] when workgroup (0,0,0) started (last replay wins), [1] the latest end of any workgroup (the
// clock is monotonic, so the last replay wins as well).  One clock for every stream: the overlap of launches of the two
// sub-batch streams INSIDE replayed graphs can be read off (tools/exp_conv_path_overlap.py).
__device__ unsigned long long g_conv_span[1024][2];
#define TR_STAMP(k) do { if ((DBG & 64) && tracing) tr_t[k] = __builtin_readcyclecounter(); } while (0)

// SK ("split-K halves", bf16 small-map launches): a 512-thread workgroup whose two halves each own every other
// K-chunk with their own staging buffers.  The barrier schedule is shared and the halves run it in opposite phase:
// while one half waits for its loads, transforms and writes them to LDS, the other reads fragments and issues MFMAs,
// so every SIMD holds two waves whose staging and matrix phases overlap (a 256-thread workgroup alone on a CU is one
// dependent chain per SIMD with the matrix pipe 25 % busy, DESIGN finding 24).  The halves' partial sums meet in LDS.
// RAW: no source carries a GroupNorm prologue (32 of the 45 launches of a cfg3 step: block1 convolutions, resampling
// convolutions and the 32^2 block2 convolutions whose input a separate gn_apply pass materialised).  The staging code of
// the general kernel tests `stats != nullptr` and the activation kind per fragment at run time; with one wave per SIMD
// those scalar tests, branches and register copies sit on the chunk loop's critical path (DESIGN finding 42).
// The kernel body as a device function of the (virtual) workgroup index: conv3x3_kernel calls it with its own index,
// the archived stage-program experiment (tools/experiments/stage_programs.hip) with the tiles it takes from its work counter.
// SIDE (round 6; RAW launches of 16-bit storage, register tiles below 16 fragments): the block's res_conv -- a 1x1 convolution of the
// SAME concatenated input (ddpm.py:198, 212) -- rides in this launch.  Its B operand is the centre tap's activation fragment, which
// is in registers anyway; per chunk it costs one more weight fragment per m-tile (1 KB of the 18-36 KB a chunk stages) and MT * NW
// of the chunk's 9 * MT * NW + MT * NW MFMAs.  The block's tail then needs no GEMM of its own: `out = res_conv(x) + SiLU(GN(h2))` was a
// conv1x1 launch walking the block's 6-12 input chunks a second time (12-14 us at 32^2 where gn_apply takes 4.5), and x / skip die
// after block1 instead of after the tail (the pool's peak live set: six 256^2 tensors -> five).
typedef const Conv3Dev __attribute__((address_space(4)))* Conv3KernargPtr;   // the block in kernel-argument (constant) memory: scalar loads
template <typename T, int MT, int NW, bool DEEP, int DBG, bool SK = false, bool RAW = false, bool SIDE = false>
__device__ __forceinline__ void conv3x3_tile(const Conv3Dev& head, Conv3KernargPtr rest, const int bx, const int by, const int bz,
                                             const int gdx, const int gdz, char* smem) {
  // `head`: the fields the first requests need (conv3x3_kernel's preloaded arguments; everything else unset).  `rest`:
  // the whole block in kernel-argument memory, read in ONE batch behind those requests (below).  Direct callers that
  // hold a complete block pass it as `head` and rest = nullptr.
  Conv3Dev a = head;
  constexpr int E = DT<T>::E, CK = DT<T>::CK;
  constexpr int TR = 4 * NW, TC = 16, HR = TR + 2, HC = TC + 2;
  constexpr int NPIX = HR * HC, NPIXP = (NPIX + 15) / 16 * 16, PLANE = NPIXP * 16;
  constexpr int ITER = (NPIXP + 63) / 64;
  constexpr int UNITS = 9 * MT * 64, WU = (UNITS + 255) / 256;
  constexpr bool P = DT<T>::precise;

  static_assert(!SIDE || (RAW && !SK && !DEEP && MT * NW < 16 && sizeof(T) == 2), "SIDE: RAW 16-bit launches of the small register tiles");
  constexpr int STAGE = 4 * PLANE + (9 + (SIDE ? 1 : 0)) * MT * 1024;         // one half's staging buffers (+ the res_conv fragments)
  const int half = SK ? (int)(threadIdx.x >> 8) : 0;       // wave-uniform
  char* s_x = smem + half * STAGE;
  char* s_w = s_x + 4 * PLANE;
  char* s_w2 = s_w + 9 * MT * 1024;                        // SIDE
  float* s_coef = reinterpret_cast<float*>(smem + (SK ? 2 : 1) * STAGE);
  // (everything in front of the first requests is computed from the PRELOADED arguments only -- conv3x3_kernel: scalar
  //  loads return out of order, so the first use of ANY field of the argument block waits for all of it; the fields of
  //  the second source, the channel total and the chunk count are therefore taken behind issue_loads)
  // fp64 scratch: [4 waves][2][16*MT] per-wave channel sums (also the stripe-reduction scratch of
  // build_gn_coef, 32 doubles).  Per-lane/per-wave partials are fp32 over <= 16*NW values; every
  // sum across waves and workgroups is fp64, so E[x^2]-mean^2 does not see fp32 partial-sum rounding.
  double* s_stat;

  const int tid = threadIdx.x & 255, lane = tid & 63, wv = tid >> 6, px = lane & 15, kq = lane >> 4;   // within the half
  unsigned long long tr_t[24] = {0};
  const bool tracing = (DBG & 64) && threadIdx.x == 0 && bz == gdz / 2 && by == 0 &&
                       bx == (gdx * 5) / 8;
  TR_STAMP(0);
  if ((DBG & 128) && threadIdx.x == 0 && bx == 0 && by == 0 && bz == 0)
    g_conv_span[(a.dbg >> 8) & 1023][0] = __builtin_amdgcn_s_memrealtime();
  const int b = bz, m0 = by * MT;
  int ty_, tx_;
  tile_of(bx, a.tiles_x, gdx / a.tiles_x, ty_, tx_);
  const int ty0 = ty_ * TR, tx0 = tx_ * TC;
  const int H = a.H, W = a.W;
  const int nch0 = a.s[0].C / CK;
  const int ws = a.wsplit;
  int nch = 0;                                          // virtual chunks (two per source chunk with two-term weights)
  if constexpr (SK) nch = (nch0 + (a.nsrc > 1 ? a.s[1].C / CK : 0)) << ws;
  const int mt_total = a.Cout / 16;
  const uint4* wg = reinterpret_cast<const uint4*>(a.w);

  // ---- register-staged pipeline (cdna_hip_programming.md T14): the global loads of chunk k+1
  // (halo fragments + weight fragments) are issued before the MFMAs of chunk k and written to LDS
  // after them, so a workgroup pays ONE exposed global round trip instead of one per chunk.
  uint4 hxA[ITER], wxA[WU], hxB[DEEP ? ITER : 1], wxB[DEEP ? WU : 1];   // B set: prefetch distance 2 (DEEP)
  uint4 wx2 = make_uint4(0u, 0u, 0u, 0u);                               // SIDE: this thread's 16 bytes of the chunk's res_conv fragments (tid < 64 MT)
  // Loop-invariant addressing (with one wave per SIMD every VALU instruction in the chunk loop is on the
  // critical path, PMC: MFMA busy ~20 % of wave cycles): per-thread element offsets of the halo pixels for
  // both source geometries and of the weight units are computed once; a chunk only adds a scalar stride.
  // Addresses are a workgroup-uniform 64-bit base (scalar registers) plus a non-negative 32-bit per-lane byte
  // offset built from 24-bit multiplies: the first version spent ~1,500 VALU instructions per wave around 72 MFMAs
  // (3.5k cycles of 64-bit / quarter-rate integer address arithmetic before the first load), which made the
  // single-chunk launches instruction-issue-bound (in-kernel trace, tools/trace_conv.py).
  unsigned hvalid = 0;                                  // bit it: halo item `it` is inside the image
  unsigned hoffb0[ITER], hoffb1[ITER];                  // byte offsets from sbase0 / sbase1
  const char* sbase0;
  const char* sbase1;
  auto src_base = [&](const SrcDev& S, unsigned (&hoffb)[ITER]) -> const char* {
    const int Hs = S.ups ? H / 2 : H, Ws = S.ups ? W / 2 : W;
    const int row0 = S.ups ? (ty0 - 1) >> 1 : ty0 - 1, col0 = S.ups ? (tx0 - 1) >> 1 : tx0 - 1;
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      const int q = (it * 4 + wv) * 16 + px;
      const int hy = (q * 3641) >> 16, hx_ = q - hy * HC;        // q / 18 for q < 400
      const int gy = ty0 - 1 + hy, gx = tx0 - 1 + hx_;
      const int r = (S.ups ? gy >> 1 : gy) - row0, c = (S.ups ? gx >> 1 : gx) - col0;   // 0 .. HR, 0 .. HC
      hoffb[it] = (__umul24(__umul24(r, Ws) + c, S.ld) + kq * E) * (unsigned)sizeof(T);
    }
    return reinterpret_cast<const char*>(S.data) + (((long)b * Hs + row0) * Ws + col0) * S.ld * (long)sizeof(T);
  };
  auto second_source = [&]() {     // SK: in front of the first requests (half 1's first chunk may belong to it); else behind them (note at s_coef)
    if (a.nsrc > 1) sbase1 = src_base(a.s[1], hoffb1);
    else {
#pragma unroll
      for (int it = 0; it < ITER; ++it) hoffb1[it] = hoffb0[it];
    }
  };
#pragma unroll
  for (int it = 0; it < ITER; ++it) {
    const int q = (it * 4 + wv) * 16 + px;
    const int hy = (q * 3641) >> 16, hx_ = q - hy * HC;
    const int gy = ty0 - 1 + hy, gx = tx0 - 1 + hx_;
    if (q < NPIX && gy >= 0 && gy < H && gx >= 0 && gx < W && !(DBG & 1)) hvalid |= 1u << it;
  }
  sbase0 = src_base(a.s[0], hoffb0);
  sbase1 = sbase0;
  if constexpr (SK) second_source();
  unsigned woffb[WU];                                   // byte offsets into a chunk's packed weights; ~0u = none
#pragma unroll
  for (int k = 0; k < WU; ++k) {
    const int u = k * 256 + tid;
    const int tap = u / (MT * 64), r = u - tap * (MT * 64);
    woffb[k] = (u < UNITS && !(DBG & 2)) ? (__umul24(tap, mt_total) + m0) * 1024u + r * 16u : ~0u;
  }
  const long wstride = 9L * mt_total * 1024;            // bytes per chunk
  auto issue_loads = [&](int ch, uint4 (&hx)[ITER], uint4 (&wx)[WU], auto first_tag) {
    constexpr bool FIRST = decltype(first_tag)::value;    // chunk 0 of the first source: nothing of the second source is touched
    const int cs = ch >> ws;                              // source chunk of this (virtual) chunk
    const int si = FIRST ? 0 : (cs >= nch0 ? 1 : 0);
    const char* sp = (si ? sbase1 : sbase0) + (long)(cs - si * nch0) * CK * (long)sizeof(T);
    // two-term weights: the lo pass (odd virtual chunk) multiplies the SAME halo tile, which is still in LDS
    const bool same_halo = !SK && !DEEP && ws && (ch & 1);
    const char* wc = reinterpret_cast<const char*>(wg) + (long)ch * wstride;
#pragma unroll
    for (int k = 0; k < WU; ++k) {
      wx[k] = make_uint4(0u, 0u, 0u, 0u);
      if (woffb[k] != ~0u) wx[k] = *reinterpret_cast<const uint4*>(wc + woffb[k]);
    }
    if constexpr (SIDE && !FIRST) {                       // (chunk 0's are requested behind the argument block: a.w2 is not a preloaded argument)
      if (tid < MT * 64) wx2 = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(a.w2) + ((long)ch * mt_total + m0) * 1024 + tid * 16);
    }
    if (!same_halo) {
#pragma unroll
      for (int it = 0; it < ITER; ++it) {
        hx[it] = make_uint4(0u, 0u, 0u, 0u);
        if ((hvalid >> it) & 1u) {
          const u32x4 v = load16_act(sp + (FIRST ? hoffb0[it] : (si ? hoffb1[it] : hoffb0[it])));
          hx[it] = make_uint4(v[0], v[1], v[2], v[3]);
        }
      }
    }
  };
  auto write_lds = [&](int ch, const uint4 (&hx)[ITER], const uint4 (&wx)[WU]) {
    const int csrc = ch >> ws;                            // source chunk of this (virtual) chunk
    const int si = csrc >= nch0 ? 1 : 0;
    const SrcDev S = si ? a.s[1] : a.s[0];
    const int c0 = (csrc - si * nch0) * CK;
    const int coef_off = si ? 2 * a.s[0].C : 0;
    const bool has_coef = !RAW && S.stats != nullptr;
    // this thread always stages the same E channels of the chunk: keep their (a, s) in registers
    float ca[E], cs[E];
    if (has_coef) {
      const float* cap = s_coef + coef_off + c0 + kq * E;
#pragma unroll
      for (int e = 0; e < E; ++e) { ca[e] = cap[e]; cs[e] = cap[S.C + e]; }
    }
    const bool same_halo = !SK && !DEEP && ws && (ch & 1);
#pragma unroll
    for (int k = 0; k < WU; ++k) {
      const int u = k * 256 + tid;
      if (u < UNITS) *reinterpret_cast<uint4*>(s_w + (size_t)u * 16) = wx[k];
    }
    if constexpr (SIDE) {
      if (tid < MT * 64) *reinterpret_cast<uint4*>(s_w2 + tid * 16) = wx2;
    }
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      const int q = (it * 4 + wv) * 16 + px;
      if (q < NPIXP && !same_halo) {
        uint4 raw = hx[it];
        if (has_coef && ((hvalid >> it) & 1u)) {        // zero padding stays exactly zero
          float v[E];
          unpack16<T>(raw, v);
          affine_act_n<P, E>(v, ca, cs, S.act);
          raw = pack16<T>(v);
        }
        *reinterpret_cast<uint4*>(s_x + kq * PLANE + q * 16) = raw;
      }
    }
  };

  TR_STAMP(1);
  if constexpr (SK) { if (half < nch) issue_loads(half, hxA, wxA, std::false_type{}); }     // half h owns chunks h, h+2, ...
  else issue_loads(0, hxA, wxA, std::true_type{});
  TR_STAMP(2);
  if (rest) {
    // the pointer is laundered HERE: loads through it cannot be hoisted in front of the requests above (scalar loads
    // return out of order -- a wait for any of them is a wait for all of them)
    Conv3KernargPtr pr = rest;
    asm volatile("" : "+s"(pr));
    a.s[1].data = pr->s[1].data; a.s[1].stats = pr->s[1].stats; a.s[1].gamma = pr->s[1].gamma; a.s[1].beta = pr->s[1].beta;
    a.s[1].film = pr->s[1].film; a.s[1].C = pr->s[1].C; a.s[1].ld = pr->s[1].ld; a.s[1].ups = pr->s[1].ups;
    a.s[1].groups = pr->s[1].groups; a.s[1].act = pr->s[1].act; a.s[1].film_tstride = pr->s[1].film_tstride;
    a.s[1].film_bstride = pr->s[1].film_bstride;
    a.s[0].stats = pr->s[0].stats; a.s[0].gamma = pr->s[0].gamma; a.s[0].beta = pr->s[0].beta; a.s[0].film = pr->s[0].film;
    a.s[0].groups = pr->s[0].groups; a.s[0].act = pr->s[0].act; a.s[0].film_tstride = pr->s[0].film_tstride;
    a.s[0].film_bstride = pr->s[0].film_bstride;
    a.bias = pr->bias; a.out = pr->out; a.ostats = pr->ostats; a.t_ptr = pr->t_ptr; a.addend = pr->addend;
    a.ogroups = pr->ogroups; a.B = pr->B; a.dbg = pr->dbg;
    if constexpr (SIDE) { a.w2 = pr->w2; a.bias2 = pr->bias2; a.out2 = pr->out2; }
  }
  if constexpr (!SK) second_source();
  {
    const int c1 = a.nsrc > 1 ? a.s[1].C : 0;
    nch = (nch0 + c1 / CK) << ws;
    s_stat = reinterpret_cast<double*>(s_coef + 2 * (a.s[0].C + c1));
  }
  // the rest of the argument block in ONE scalar batch
  asm volatile("" ::"s"(a.bias), "s"(a.out), "s"(a.ostats), "s"(a.ogroups), "s"(a.addend), "s"(a.s[1].C), "s"(a.s[1].ld), "s"(a.s[1].ups),
               "s"(a.s[1].data), "s"(a.s[0].stats), "s"(a.s[0].gamma), "s"(a.s[0].beta), "s"(a.s[0].film), "s"(a.s[0].groups), "s"(a.s[0].act));
  float4 bias[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) bias[m] = *reinterpret_cast<const float4*>(a.bias + (m0 + m) * 16 + kq * 4);
  float4 bias2[SIDE ? MT : 1];
  if constexpr (SIDE) {
    asm volatile("" ::"s"(a.w2), "s"(a.bias2), "s"(a.out2));
    if (tid < MT * 64) wx2 = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(a.w2) + (long)m0 * 1024 + tid * 16);   // chunk 0
#pragma unroll
    for (int m = 0; m < MT; ++m) bias2[m] = *reinterpret_cast<const float4*>(a.bias2 + (m0 + m) * 16 + kq * 4);
  }

  // ---- prologue coefficients (overlaps the loads above)
  {
    const bool any = !RAW && (a.s[0].stats != nullptr || (a.nsrc > 1 && a.s[1].stats != nullptr));
    if (any) {
      const int trow = a.t_ptr ? *a.t_ptr : 0;
      int off = 0;
      for (int s = 0; s < a.nsrc; ++s) {
        const SrcDev S = s ? a.s[1] : a.s[0];
        if (S.stats) {
          const long npix = S.ups ? (long)(H / 2) * (W / 2) : (long)H * W;
          build_gn_coef<DT<T>::precise>(S, b, trow, npix, s_coef + off, s_stat, (int)threadIdx.x, SK ? 512 : 256);
        }
        off += 2 * S.C;
      }
    }
  }

  TR_STAMP(3);
  f32x4 acc[MT][NW];
  f32x4 acc2[SIDE ? MT : 1][SIDE ? NW : 1];                // SIDE: res_conv's accumulators
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int j = 0; j < NW; ++j) {
      acc[m][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      if constexpr (SIDE) acc2[m][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

  // LDS fragment reads are software-pipelined against the MFMAs: all fragments of tap column dx+1 (3*MT
  // weight + NW+2 activation fragments) are requested before the MFMAs of column dx issue, so the ~100-cycle
  // ds_read latency is paid once per chunk instead of once per activation fragment (PMC on 256->256@32^2:
  // 38 % of wave cycles were s_waitcnt stalls with the read-then-use order).
  auto compute = [&]() {
    if (DBG & 4) return;
    if constexpr (MT * NW >= 16) {
      // 64 channels x 16 rows: m-tile outer inside a tap column, so only the 3 weight fragments of ONE m-tile (and the
      // next one's, in flight) are live instead of 3*MT per column -- 24 registers instead of 96, which is what lets two
      // of these workgroups share a CU.  Per (m, row) the accumulation order (dx, then dy) is the one of the loop below.
      uint4 A[2][3], Bq[2][NW + 2];
      auto load_b = [&](int dx, int set) {
#pragma unroll
        for (int rr = 0; rr < NW + 2; ++rr)
          Bq[set][rr] = *reinterpret_cast<const uint4*>(s_x + kq * PLANE + (((wv * NW + rr) * HC + dx + px) * 16));
      };
      auto load_a = [&](int dx, int m, int set) {
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
          A[set][dy] = *reinterpret_cast<const uint4*>(s_w + ((dy * 3 + dx) * MT + m) * 1024 + lane * 16);
      };
      load_b(0, 0);
      load_a(0, 0, 0);
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        if (dx + 1 < 3) load_b(dx + 1, (dx + 1) & 1);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          const int idx = dx * MT + m;
          if (idx + 1 < 3 * MT) load_a((idx + 1) / MT, (idx + 1) % MT, (idx + 1) & 1);
#pragma unroll
          for (int rr = 0; rr < NW + 2; ++rr) {
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
              const int j = rr - dy;
              if (j >= 0 && j < NW) mma16<T>(acc[m][j], A[idx & 1][dy], Bq[dx & 1][rr]);
            }
          }
        }
      }
      return;
    }
    uint4 A[2][3][MT], Bq[2][NW + 2];
    auto load_frags = [&](int dx, int set) {
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int m = 0; m < MT; ++m)
          A[set][dy][m] = *reinterpret_cast<const uint4*>(s_w + ((dy * 3 + dx) * MT + m) * 1024 + lane * 16);
#pragma unroll
      for (int rr = 0; rr < NW + 2; ++rr)
        Bq[set][rr] = *reinterpret_cast<const uint4*>(s_x + kq * PLANE + (((wv * NW + rr) * HC + dx + px) * 16));
    };
    load_frags(0, 0);
    uint4 A2[SIDE ? MT : 1];
    if constexpr (SIDE) {
#pragma unroll
      for (int m = 0; m < MT; ++m) A2[m] = *reinterpret_cast<const uint4*>(s_w2 + m * 1024 + lane * 16);
    }
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
            if constexpr (SIDE) {                           // the centre tap's activation fragment IS the 1x1 convolution's operand
              if (dx == 1 && dy == 1) {
#pragma unroll
                for (int m = 0; m < MT; ++m) mma16<T>(acc2[m][j], A2[m], Bq[dx & 1][rr]);
              }
            }
          }
        }
      }
    }
  };
  if constexpr (SK) {
    for (int k = 0; k <= nch; ++k) {
      __syncthreads();               // first time: coefficients visible; then: the other half's phase is complete
      if ((k & 1) == half) {         // staging phase: chunk k (mine) into my buffers, request chunk k+2
        if (k < nch) {
          write_lds(k, hxA, wxA);
          if (k + 2 < nch) issue_loads(k + 2, hxA, wxA, std::false_type{});
        }
      } else if (k >= 1) {           // matrix phase: chunk k-1 (mine, staged in the previous interval)
        compute();
      }
    }
    // join the partial sums: half 1 -> LDS (its own staging buffers are dead) -> half 0
    __syncthreads();
    float4* s_red = reinterpret_cast<float4*>(smem + STAGE);
    if (half == 1) {
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int j = 0; j < NW; ++j)
          s_red[(m * NW + j) * 256 + tid] = make_float4(acc[m][j][0], acc[m][j][1], acc[m][j][2], acc[m][j][3]);
    }
    __syncthreads();
    if (half == 0) {
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int j = 0; j < NW; ++j) {
          const float4 o = s_red[(m * NW + j) * 256 + tid];
          acc[m][j][0] += o.x; acc[m][j][1] += o.y; acc[m][j][2] += o.z; acc[m][j][3] += o.w;
        }
    }
  } else if constexpr (!DEEP) {
    for (int ch = 0; ch < nch; ++ch) {
      __syncthreads();               // previous chunk fully consumed (first time: coefficients visible)
      if (ch == 0) TR_STAMP(4);
      if (ch == 2) TR_STAMP(11);     // 11..15: one steady-state chunk (tools/trace_conv.py)
      write_lds(ch, hxA, wxA);
      if (ch == 0) TR_STAMP(5);
      if (ch == 2) TR_STAMP(12);
      __syncthreads();
      if (ch == 0) TR_STAMP(6);
      if (ch == 2) TR_STAMP(13);
      if (ch + 1 < nch) issue_loads(ch + 1, hxA, wxA, std::false_type{});
      if (ch == 2) TR_STAMP(14);
      compute();
      if (ch == 0) TR_STAMP(7);
      if (ch == 2) TR_STAMP(15);
    }
    TR_STAMP(8);
  } else {
    // prefetch distance 2 with two register sets (the launches that use this variant run one wave per SIMD,
    // so the 512-entry register file is theirs): chunk k+2 is requested before chunk k is computed
    if (nch > 1) issue_loads(1, hxB, wxB, std::false_type{});
    for (int ch = 0; ch < nch; ch += 2) {
      __syncthreads();
      write_lds(ch, hxA, wxA);
      __syncthreads();
      if (ch + 2 < nch) issue_loads(ch + 2, hxA, wxA, std::false_type{});
      compute();
      if (ch + 1 < nch) {
        __syncthreads();
        write_lds(ch + 1, hxB, wxB);
        __syncthreads();
        if (ch + 3 < nch) issue_loads(ch + 3, hxB, wxB, std::false_type{});
        compute();
      }
    }
  }

  // ---- epilogue: bias, statistics, NHWC store.  lane holds channels 16m+4kq..+3 of pixel px.
  const int gx = tx0 + px;
  // uniform base of the tile + one 32-bit lane offset; rows and m-tiles add constants
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
  // Every load of the epilogue is retired BEFORE its first store.  The vector-memory counter is in order and counts
  // stores too: hipcc placed its wait for bias[m] (requested before the chunk loop, long since landed) in front of
  // m-tile m's first use, behind the stores of the m-tiles before it, and across the `if (valid)` branches its
  // bookkeeping falls back to small counts -- vmcnt(1) behind four stores, vmcnt(0) behind six in the 64-channel
  // tile: a store round trip exposed per pair of m-tiles; with an addend, one load -> wait -> store chain per fragment
  // (tools/scan_store_waits.py on the hipcc -S output).
#pragma unroll
  for (int m = 0; m < MT; ++m) asm volatile("" ::"v"(bias[m].x), "v"(bias[m].y), "v"(bias[m].z), "v"(bias[m].w));
  if constexpr (SIDE) {                                     // ... the side output's bias too: its first use is behind the main output's stores
#pragma unroll
    for (int m = 0; m < MT; ++m) asm volatile("" ::"v"(bias2[m].x), "v"(bias2[m].y), "v"(bias2[m].z), "v"(bias2[m].w));
  }
  const bool rows_in = ty0 + TR <= H && tx0 + TC <= W;      // workgroup-uniform: no ragged row / column in this tile
  // FULL: every pixel of the tile is inside the image and there is no addend (all but one launch of a cfg3 step): the
  // stores and the statistics run without per-fragment execution masks and selects.
  auto emit = [&](auto full_tag) {
    constexpr bool FULL = decltype(full_tag)::value;
    // the addend (conv_fusion's pre-statistics constants): raw loads from an always-valid address (rows / columns past the
    // image are never stored), all requested before the first wait, converted where they are used (finding 63)
    typename Raw4<T>::type ad[FULL ? 1 : MT][FULL ? 1 : NW];
    if constexpr (!FULL) {
      if (a.addend) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int j = 0; j < NW; ++j) {
            const int gy = ty0 + wv * NW + j;
            const bool in = rows_in || (gy < H && gx < W);
            const unsigned off = lane_off + j * row_off + m * 16 * (unsigned)sizeof(T);
            ad[m][j] = load4_raw<T>(reinterpret_cast<const T*>(addb + (in ? off : 0u)));
          }
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int j = 0; j < NW; ++j) pin_raw4(ad[m][j]);
      }
    }
    // 16-bit storage: the fragments of two adjacent m-tiles leave as ONE 16-byte store per lane (pair_frag16)
    constexpr int MS = sizeof(T) == 2 ? 2 : 1;
    static_assert(MT % 2 == 0, "m-tiles are stored in pairs");
#pragma unroll
    for (int m = 0; m < MT; m += MS) {
#pragma unroll
      for (int j = 0; j < NW; ++j) {
        const int gy = ty0 + wv * NW + j;
        const bool valid = FULL || ((rows_in || (gy < H && gx < W)) && half == 0);   // (SK: half 0 holds the joined sums)
        float v[MS][4];
#pragma unroll
        for (int mm = 0; mm < MS; ++mm) {
          const float4 bv = bias[m + mm];
          v[mm][0] = acc[m + mm][j][0] + bv.x; v[mm][1] = acc[m + mm][j][1] + bv.y;
          v[mm][2] = acc[m + mm][j][2] + bv.z; v[mm][3] = acc[m + mm][j][3] + bv.w;
          if constexpr (!FULL) {
            if (a.addend) {
              float av[4];
              unpack4<T>(ad[m + mm][j], av);
#pragma unroll
              for (int r = 0; r < 4; ++r) v[mm][r] += av[r];
            }
          }
        }
        if constexpr (MS == 2) {
          const uint4 w16 = pair_frag16<T>(v[0], v[1]);      // all lanes: the exchange is unconditional
          const unsigned off = lane_off - kq * 4 * (unsigned)sizeof(T) + j * row_off + m * 16 * (unsigned)sizeof(T) + pair_frag16_off(kq);
          if (valid && !(DBG & 8)) store16_out(outb + off, w16);
        } else {
          const unsigned off = lane_off + j * row_off + m * 16 * (unsigned)sizeof(T);
          if (valid && !(DBG & 8)) store4<T>(reinterpret_cast<T*>(outb + off), v[0]);
        }
        if (valid) {
#pragma unroll
          for (int mm = 0; mm < MS; ++mm)
#pragma unroll
            for (int r = 0; r < 4; ++r) { ssum[m + mm][r] += v[mm][r]; ssq[m + mm][r] += v[mm][r] * v[mm][r]; }
        }
      }
    }
  };
  if (rows_in && !a.addend && half == 0) emit(std::true_type{});
  else emit(std::false_type{});
  if constexpr (SIDE) {       // res_conv(x) + bias2 -> out2: the same fragment pairs, the same write-through stores, no statistics
    char* out2b = reinterpret_cast<char*>(a.out2) + obase * (long)sizeof(T);
#pragma unroll
    for (int m = 0; m < MT; m += 2) {
#pragma unroll
      for (int j = 0; j < NW; ++j) {
        const int gy = ty0 + wv * NW + j;
        const bool valid = rows_in || (gy < H && gx < W);
        float v[2][4];
#pragma unroll
        for (int mm = 0; mm < 2; ++mm) {
          const float4 bv = bias2[m + mm];
          v[mm][0] = acc2[m + mm][j][0] + bv.x; v[mm][1] = acc2[m + mm][j][1] + bv.y;
          v[mm][2] = acc2[m + mm][j][2] + bv.z; v[mm][3] = acc2[m + mm][j][3] + bv.w;
        }
        const uint4 w16 = pair_frag16<T>(v[0], v[1]);
        const unsigned off = lane_off - kq * 4 * (unsigned)sizeof(T) + j * row_off + m * 16 * (unsigned)sizeof(T) + pair_frag16_off(kq);
        if (valid) store16_out(out2b + off, w16);
      }
    }
  }
  TR_STAMP(9);
  if (a.ostats) {
    __syncthreads();                            // s_stat may still be read as build_gn_coef scratch
    TR_STAMP(16);
    const int gs = a.Cout / a.ogroups;          // channels per group; gs <= 16*MT by construction
    const int ngrp_blk = (16 * MT) / gs;
    const int stripe = bx % LD_STAT_STRIPES;
    if (!P && (gs & 3) == 0) {
      // a lane's four channels (4kq .. 4kq+3 of m-tile m) always fall into ONE group when gs is a multiple of 4:
      // add them before the cross-lane reduction -- 4*MT row reductions and LDS values per wave instead of 16*MT
      // (the statistics were 20 % of a workgroup's cycles, tools/trace_conv.py).  bf16 storage only: the fp32
      // path keeps its short fp32 partial sums.
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const float s1 = wave16_sum((ssum[m][0] + ssum[m][1]) + (ssum[m][2] + ssum[m][3]));
        const float s2 = wave16_sum((ssq[m][0] + ssq[m][1]) + (ssq[m][2] + ssq[m][3]));
        if (px == 0 && half == 0) {
          s_stat[(wv * 2 + 0) * 4 * MT + m * 4 + kq] = (double)s1;
          s_stat[(wv * 2 + 1) * 4 * MT + m * 4 + kq] = (double)s2;
        }
      }
      TR_STAMP(17);
      __syncthreads();
      TR_STAMP(18);
      const int q4 = gs >> 2;                     // tile entries (m, kq) per group
      if ((q4 & (q4 - 1)) == 0) {
        // Lanes 0-15 of wave 0 take the sums, lanes 16-31 the sums of squares; lane e owns tile entry e = 4m + kq: four
        // independent LDS reads (one per wave), then the group's q4 entries meet by DPP inside the 16-lane row.  One
        // thread per group walking its 4*q4 values (up to 32 dependent LDS read -> fp64 add pairs) took 1,400-2,000
        // cycles of a workgroup's ~5,000-cycle epilogue (tools/trace_conv.py).
        if (tid < 32 && half == 0) {
          const int k = tid >> 4, e = tid & 15;
          double v = 0.0;
          if (e < 4 * MT) {
#pragma unroll
            for (int w4 = 0; w4 < 4; ++w4) v += s_stat[(w4 * 2 + k) * 4 * MT + e];
          }
          v = row_group_sum_d(v, q4);
          if (e < 4 * MT && (e & (q4 - 1)) == 0) {
            const int g = (m0 * 16) / gs + e / q4;
            atomicAdd(&a.ostats[(((size_t)b * LD_STAT_STRIPES + stripe) * a.ogroups + g) * 2 + k], v);
          }
        }
      } else if (tid < 2 * ngrp_blk && half == 0) {
        const int gi = tid >> 1, k = tid & 1;
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
          if (px == 0 && half == 0) {
            s_stat[(wv * 2 + 0) * 16 * MT + m * 16 + kq * 4 + r] = (double)s1;
            s_stat[(wv * 2 + 1) * 16 * MT + m * 16 + kq * 4 + r] = (double)s2;
          }
        }
      __syncthreads();
      if (tid < 2 * ngrp_blk && half == 0) {
        const int gi = tid >> 1, k = tid & 1;
        double acc1 = 0.0;
        for (int w4 = 0; w4 < 4; ++w4)
          for (int c = 0; c < gs; ++c) acc1 += s_stat[(w4 * 2 + k) * 16 * MT + gi * gs + c];
        const int g = (m0 * 16) / gs + gi;
        atomicAdd(&a.ostats[(((size_t)b * LD_STAT_STRIPES + stripe) * a.ogroups + g) * 2 + k], acc1);
      }
    }
  }
  if ((DBG & 128) && threadIdx.x == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's stores have left
    atomicMax(&g_conv_span[(a.dbg >> 8) & 1023][1], (unsigned long long)__builtin_amdgcn_s_memrealtime());
  }
  if ((DBG & 64) && tracing) {
    tr_t[19] = __builtin_readcyclecounter();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's stores have left
    tr_t[10] = __builtin_readcyclecounter();
#pragma unroll
    for (int k = 0; k < 24; ++k) g_conv_trace[k] = tr_t[k];
  }
}

// What the first requests of a workgroup need -- the first source, the weights, the geometry: 13 dwords -- are leading
// SCALAR kernel arguments: the file is built with -amdgpu-kernarg-preload-count, so the command processor puts them
// (up to 14 dwords) into SGPRs before the wave starts and the halo / weight requests leave without a scalar round trip
// in front of them (a by-value struct is never preloaded: finding 67).  The rest of the block (`rest`: second source,
// GroupNorm operands, bias, outputs) is fetched in ONE batch that is pinned BEHIND those requests (conv3x3_tile).
template <typename T, int MT, int NW, bool DEEP, int DBG, bool SK = false, bool RAW = false, bool SIDE = false>
__global__ __launch_bounds__(SK ? 512 : 256) __attribute__((amdgpu_waves_per_eu(SK ? 2 : (MT == 2 ? 3 : 1), SK ? 2 : (MT == 2 ? 3 : 2))))
void conv3x3_kernel(const void* data0, const void* w, int H, int W, int tiles_x, int C0, int ld0, int ups0, int nsrc, int wsplit, int Cout,
                    Conv3Dev rest) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  Conv3Dev a{};
  a.s[0].data = data0; a.w = w; a.H = H; a.W = W; a.tiles_x = tiles_x; a.s[0].C = C0; a.s[0].ld = ld0; a.s[0].ups = ups0;
  a.nsrc = nsrc; a.wsplit = wsplit; a.Cout = Cout;
  // `rest` in kernel-argument memory: behind the two pointers and nine ints, at the struct's 8-byte alignment
  constexpr unsigned REST_OFF = ld_kernarg_offset<const void*, const void*, int, int, int, int, int, int, int, int, int>(alignof(Conv3Dev));
  static_assert(REST_OFF == 56, "conv3x3_kernel: leading arguments changed");
  typedef const char __attribute__((address_space(4)))* KChar;
  Conv3KernargPtr pr = (Conv3KernargPtr)((KChar)__builtin_amdgcn_kernarg_segment_ptr() + REST_OFF);
  if constexpr (SK) {       // (the split-K variant sets its second source up in front of its first requests: whole block now)
    a = rest;
    a.tiles_x = tiles_x;
    conv3x3_tile<T, MT, NW, DEEP, DBG, SK, RAW>(a, nullptr, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.x, gridDim.z, smem);
  } else {
    conv3x3_tile<T, MT, NW, DEEP, DBG, SK, RAW, SIDE>(a, pr, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.x, gridDim.z, smem);
  }
}

}  // namespace
