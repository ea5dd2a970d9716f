// 3x3 convolution (pad 1) as an implicit GEMM on MFMA, NHWC, gfx950.
//
// Replaces nn.Conv2d(k=3,p=1) in Block.proj (ddpm.py:173), Upsample (:117), the last-stage
// convs (:372,:391) and BasicBlock (unet_model.py:20,24,30), with
//   - prologue fused into the input staging: channel concat of two sources (torch.cat,
//     ddpm.py:435-448), nearest x2 upsample (:116), GroupNorm-apply + FiLM + SiLU/ReLU of the
//     producer (:179-185, unet_model.py:21-22);
//   - epilogue: bias, GroupNorm statistics of the result (sum, sum^2 per (batch, group), fp64
//     atomics -- one per group per workgroup), NHWC store.
//
// Tiling.  Workgroup = 256 threads = 4 waves; output tile = (4*NW rows) x 16 cols of pixels x
// (16*MT) output channels.  Wave w owns rows [w*NW, (w+1)*NW) (each row = one 16-pixel MFMA
// column tile) and all MT channel tiles: acc[MT][NW] fragments of 16x16.
// K loop over 64-byte channel chunks (32 bf16 / 16 fp32 channels):
//   LDS input image  [kq 0..3][halo pixel q][16 B]   -- plane stride = multiple of 256 B, so the
//       16 lanes of a ds_read_b128 group (consecutive q, same kq) hit 16 distinct 16-B slots for
//       every tap offset: conflict-free without a swizzle;
//   LDS weights      [tap][m][lane][16 B]            -- already in fragment order in HBM
//       (ld_pack_conv_weight), read linearly.
// Inner order: dx outer (3*MT weight fragments live), then halo rows rr: one activation fragment
// feeds the (up to) 3 taps dy that touch it => 9*MT + 3*(NW+2) LDS reads per 9*MT*NW MFMAs.
#include "common.hip.h"
#include <stdlib.h>

int ld_conv3x3_c32_try(const ld_conv3x3_args* p, hipStream_t st);   // conv3x3_c32.hip
int ld_conv3x3_ws_try(const ld_conv3x3_args* p, hipStream_t st);    // conv3x3_ws.hip

namespace {

struct Conv3Dev {
  SrcDev s[2];
  int nsrc;
  const void* w;
  const float* bias;
  void* out;
  double* ostats;
  int ogroups;
  int B, H, W, Cout;
  const int* t_ptr;
  const void* addend;
  int tiles_x;
  int dbg;     // ablation switches (LD_CONV_DEBUG env, 0 in production): 1 no halo loads, 2 no weight loads, 4 no MFMA, 8 no stores
};

// LD_CONV_DEBUG=64: cycle stamps of one workgroup from the middle of the launch (tools/trace_conv.py)
__device__ unsigned long long g_conv_trace[16];
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
template <typename T, int MT, int NW, bool DEEP, int DBG, bool SK = false, bool RAW = false>
__global__ __launch_bounds__(SK ? 512 : 256) __attribute__((amdgpu_waves_per_eu(SK ? 2 : (MT == 2 ? 3 : 1), SK ? 2 : (MT == 2 ? 3 : 2)))) void conv3x3_kernel(Conv3Dev a) {
  constexpr int E = DT<T>::E, CK = DT<T>::CK;
  constexpr int TR = 4 * NW, TC = 16, HR = TR + 2, HC = TC + 2;
  constexpr int NPIX = HR * HC, NPIXP = (NPIX + 15) / 16 * 16, PLANE = NPIXP * 16;
  constexpr int ITER = (NPIXP + 63) / 64;
  constexpr int UNITS = 9 * MT * 64, WU = (UNITS + 255) / 256;
  constexpr bool P = DT<T>::precise;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int STAGE = 4 * PLANE + 9 * MT * 1024;         // one half's staging buffers
  const int half = SK ? (int)(threadIdx.x >> 8) : 0;       // wave-uniform
  char* s_x = smem + half * STAGE;
  char* s_w = s_x + 4 * PLANE;
  float* s_coef = reinterpret_cast<float*>(smem + (SK ? 2 : 1) * STAGE);
  const int ctot = a.s[0].C + (a.nsrc > 1 ? a.s[1].C : 0);
  // fp64 scratch: [4 waves][2][16*MT] per-wave channel sums (also the stripe-reduction scratch of
  // build_gn_coef, 32 doubles).  Per-lane/per-wave partials are fp32 over <= 16*NW values; every
  // sum across waves and workgroups is fp64, so E[x^2]-mean^2 does not see fp32 partial-sum rounding.
  double* s_stat = reinterpret_cast<double*>(s_coef + 2 * ctot);

  const int tid = threadIdx.x & 255, lane = tid & 63, wv = tid >> 6, px = lane & 15, kq = lane >> 4;   // within the half
  unsigned long long tr_t[16] = {0};
  const bool tracing = (DBG & 64) && threadIdx.x == 0 && blockIdx.z == gridDim.z / 2 && blockIdx.y == 0 &&
                       blockIdx.x == (gridDim.x * 5) / 8;
  TR_STAMP(0);
  const int b = blockIdx.z, m0 = blockIdx.y * MT;
  const int ty0 = (blockIdx.x / a.tiles_x) * TR, tx0 = (blockIdx.x % a.tiles_x) * TC;
  const int H = a.H, W = a.W;
  const int nch0 = a.s[0].C / CK;
  const int nch = nch0 + (a.nsrc > 1 ? a.s[1].C / CK : 0);
  const int mt_total = a.Cout / 16;
  const uint4* wg = reinterpret_cast<const uint4*>(a.w);

  // ---- register-staged pipeline (cdna_hip_programming.md T14): the global loads of chunk k+1
  // (halo fragments + weight fragments) are issued before the MFMAs of chunk k and written to LDS
  // after them, so a workgroup pays ONE exposed global round trip instead of one per chunk.
  uint4 hxA[ITER], wxA[WU], hxB[DEEP ? ITER : 1], wxB[DEEP ? WU : 1];   // B set: prefetch distance 2 (DEEP)
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
  {
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
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      const int q = (it * 4 + wv) * 16 + px;
      const int hy = (q * 3641) >> 16, hx_ = q - hy * HC;
      const int gy = ty0 - 1 + hy, gx = tx0 - 1 + hx_;
      if (q < NPIX && gy >= 0 && gy < H && gx >= 0 && gx < W && !(DBG & 1)) hvalid |= 1u << it;
    }
    sbase0 = src_base(a.s[0], hoffb0);
    sbase1 = sbase0;
    if (a.nsrc > 1) sbase1 = src_base(a.s[1], hoffb1);
    else {
#pragma unroll
      for (int it = 0; it < ITER; ++it) hoffb1[it] = hoffb0[it];
    }
  }
  unsigned woffb[WU];                                   // byte offsets into a chunk's packed weights; ~0u = none
#pragma unroll
  for (int k = 0; k < WU; ++k) {
    const int u = k * 256 + tid;
    const int tap = u / (MT * 64), r = u - tap * (MT * 64);
    woffb[k] = (u < UNITS && !(DBG & 2)) ? (__umul24(tap, mt_total) + m0) * 1024u + r * 16u : ~0u;
  }
  const long wstride = 9L * mt_total * 1024;            // bytes per chunk
  auto issue_loads = [&](int ch, uint4 (&hx)[ITER], uint4 (&wx)[WU]) {
    const int si = ch >= nch0 ? 1 : 0;
    const char* sp = (si ? sbase1 : sbase0) + (long)(ch - si * nch0) * CK * (long)sizeof(T);
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      hx[it] = make_uint4(0u, 0u, 0u, 0u);
      if ((hvalid >> it) & 1u) hx[it] = *reinterpret_cast<const uint4*>(sp + (si ? hoffb1[it] : hoffb0[it]));
    }
    const char* wc = reinterpret_cast<const char*>(wg) + (long)ch * wstride;
#pragma unroll
    for (int k = 0; k < WU; ++k) {
      wx[k] = make_uint4(0u, 0u, 0u, 0u);
      if (woffb[k] != ~0u) wx[k] = *reinterpret_cast<const uint4*>(wc + woffb[k]);
    }
  };
  auto write_lds = [&](int ch, const uint4 (&hx)[ITER], const uint4 (&wx)[WU]) {
    const int si = ch >= nch0 ? 1 : 0;
    const SrcDev S = si ? a.s[1] : a.s[0];
    const int c0 = (ch - si * nch0) * CK;
    const int coef_off = si ? 2 * a.s[0].C : 0;
    const bool has_coef = !RAW && S.stats != nullptr;
    // this thread always stages the same E channels of the chunk: keep their (a, s) in registers
    float ca[E], cs[E];
    if (has_coef) {
      const float* cap = s_coef + coef_off + c0 + kq * E;
#pragma unroll
      for (int e = 0; e < E; ++e) { ca[e] = cap[e]; cs[e] = cap[S.C + e]; }
    }
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      const int q = (it * 4 + wv) * 16 + px;
      if (q < NPIXP) {
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
#pragma unroll
    for (int k = 0; k < WU; ++k) {
      const int u = k * 256 + tid;
      if (u < UNITS) *reinterpret_cast<uint4*>(s_w + (size_t)u * 16) = wx[k];
    }
  };

  TR_STAMP(1);
  if (!SK || half < nch) issue_loads(half, hxA, wxA);     // half h owns chunks h, h+2, ...
  TR_STAMP(2);
  float4 bias[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) bias[m] = *reinterpret_cast<const float4*>(a.bias + (m0 + m) * 16 + kq * 4);

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
          build_gn_coef(S, b, trow, npix, s_coef + off, s_stat, (int)threadIdx.x, SK ? 512 : 256);
        }
        off += 2 * S.C;
      }
    }
  }

  TR_STAMP(3);
  f32x4 acc[MT][NW];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int j = 0; j < NW; ++j) acc[m][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // LDS fragment reads are software-pipelined against the MFMAs: all fragments of tap column dx+1 (3*MT
  // weight + NW+2 activation fragments) are requested before the MFMAs of column dx issue, so the ~100-cycle
  // ds_read latency is paid once per chunk instead of once per activation fragment (PMC on 256->256@32^2:
  // 38 % of wave cycles were s_waitcnt stalls with the read-then-use order).
  auto compute = [&]() {
    if (DBG & 4) return;
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
      }
    }
  };
  if constexpr (SK) {
    for (int k = 0; k <= nch; ++k) {
      __syncthreads();               // first time: coefficients visible; then: the other half's phase is complete
      if ((k & 1) == half) {         // staging phase: chunk k (mine) into my buffers, request chunk k+2
        if (k < nch) {
          write_lds(k, hxA, wxA);
          if (k + 2 < nch) issue_loads(k + 2, hxA, wxA);
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
      if (ch + 1 < nch) issue_loads(ch + 1, hxA, wxA);
      if (ch == 2) TR_STAMP(14);
      compute();
      if (ch == 0) TR_STAMP(7);
      if (ch == 2) TR_STAMP(15);
    }
    TR_STAMP(8);
  } else {
    // prefetch distance 2 with two register sets (the launches that use this variant run one wave per SIMD,
    // so the 512-entry register file is theirs): chunk k+2 is requested before chunk k is computed
    if (nch > 1) issue_loads(1, hxB, wxB);
    for (int ch = 0; ch < nch; ch += 2) {
      __syncthreads();
      write_lds(ch, hxA, wxA);
      __syncthreads();
      if (ch + 2 < nch) issue_loads(ch + 2, hxA, wxA);
      compute();
      if (ch + 1 < nch) {
        __syncthreads();
        write_lds(ch + 1, hxB, wxB);
        __syncthreads();
        if (ch + 3 < nch) issue_loads(ch + 3, hxB, wxB);
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
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const float4 bv = bias[m];
#pragma unroll
    for (int j = 0; j < NW; ++j) {
      const int gy = ty0 + wv * NW + j;
      const bool valid = gy < H && gx < W && half == 0;   // (SK: half 0 holds the joined sums)
      const unsigned off = lane_off + j * row_off + m * 16 * (unsigned)sizeof(T);
      float v[4] = {acc[m][j][0] + bv.x, acc[m][j][1] + bv.y, acc[m][j][2] + bv.z, acc[m][j][3] + bv.w};
      if (valid && a.addend) {
        float ad[4];
        load4<T>(reinterpret_cast<const T*>(addb + off), ad);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += ad[r];
      }
      if (valid) {
        if (!(DBG & 8)) store4<T>(reinterpret_cast<T*>(outb + off), v);
#pragma unroll
        for (int r = 0; r < 4; ++r) { ssum[m][r] += v[r]; ssq[m][r] += v[r] * v[r]; }
      }
    }
  }
  TR_STAMP(9);
  if (a.ostats) {
    __syncthreads();                            // s_stat may still be read as build_gn_coef scratch
    const int gs = a.Cout / a.ogroups;          // channels per group; gs <= 16*MT by construction
    const int ngrp_blk = (16 * MT) / gs;
    const int stripe = blockIdx.x % LD_STAT_STRIPES;
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
      __syncthreads();
      if (tid < 2 * ngrp_blk && half == 0) {
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
  if ((DBG & 64) && tracing) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's stores have left
    tr_t[10] = __builtin_readcyclecounter();
#pragma unroll
    for (int k = 0; k < 16; ++k) g_conv_trace[k] = tr_t[k];
  }
}

template <typename T, int MT, int NW, bool DEEP, int DBG, bool SK = false, bool RAW = false>
int launch_dbg(const Conv3Dev& a, hipStream_t st) {
  constexpr int TR = 4 * NW, HR = TR + 2, HC = 18;
  constexpr int NPIXP = (HR * HC + 15) / 16 * 16;
  const int ctot = a.s[0].C + (a.nsrc > 1 ? a.s[1].C : 0);
  const size_t lds = (SK ? 2 : 1) * (4 * NPIXP * 16 + 9 * MT * 1024) + 2 * ctot * sizeof(float) + 4 * 2 * 16 * MT * sizeof(double);
  if (lds > 65536) LD_HIP(ld_allow_lds(conv3x3_kernel<T, MT, NW, DEEP, DBG, SK, RAW>, lds));   // cached per device
  Conv3Dev d = a;
  d.tiles_x = (a.W + 15) / 16;
  const int tiles_y = (a.H + TR - 1) / TR;
  dim3 grid(d.tiles_x * tiles_y, a.Cout / (16 * MT), a.B);
  LD_LAUNCH((conv3x3_kernel<T, MT, NW, DEEP, DBG, SK, RAW>), grid, dim3(SK ? 512 : 256), lds, st, d);
  LD_LAUNCH_CHECK("conv3x3");
  return LD_OK;
}

// The LD_CONV_DEBUG ablation variants are separate instantiations (bf16, non-DEEP only) that exist only in a
// library built with -DLD_DEBUG_VARIANTS (build.sh --debug-variants): the production library carries neither their
// branches nor their code objects.
template <typename T, int MT, int NW, bool DEEP>
int launch(const Conv3Dev& a, hipStream_t st) {
#ifdef LD_DEBUG_VARIANTS
  if constexpr (std::is_same<T, bf16>::value && !DEEP) {
    switch (a.dbg) {
      case 1: return launch_dbg<T, MT, NW, DEEP, 1>(a, st);
      case 2: return launch_dbg<T, MT, NW, DEEP, 2>(a, st);
      case 3: return launch_dbg<T, MT, NW, DEEP, 3>(a, st);
      case 4: return launch_dbg<T, MT, NW, DEEP, 4>(a, st);
      case 7: return launch_dbg<T, MT, NW, DEEP, 7>(a, st);
      case 8: return launch_dbg<T, MT, NW, DEEP, 8>(a, st);
      case 12: return launch_dbg<T, MT, NW, DEEP, 12>(a, st);
      case 15: return launch_dbg<T, MT, NW, DEEP, 15>(a, st);
      case 64: return launch_dbg<T, MT, NW, DEEP, 64>(a, st);
      default: break;
    }
  }
#endif
  if constexpr (!DEEP) {
    static const bool no_raw = getenv("LD_CONV_NO_RAW") != nullptr;      // tuning override: always the general kernel
    const bool raw = !a.s[0].stats && !(a.nsrc > 1 && a.s[1].stats);
    if (raw && !no_raw) return launch_dbg<T, MT, NW, DEEP, 0, false, true>(a, st);
  }
  return launch_dbg<T, MT, NW, DEEP, 0>(a, st);
}

template <typename T>
int dispatch(const Conv3Dev& a, hipStream_t st) {
  static const int force_mt = getenv("LD_CONV_MT") ? atoi(getenv("LD_CONV_MT")) : 0;   // tuning overrides
  static const int force_nw = getenv("LD_CONV_NW") ? atoi(getenv("LD_CONV_NW")) : 0;
  bool mt4 = (a.Cout % 64) == 0;
  // a launch that would not even give every CU one 64-channel-tile workgroup uses 32-channel tiles instead
  // (128->128 @32^2, B=8: 128 workgroups -> 256; rocprofv3: 10.3 -> 8.3 us)
  static const long mt4_min = getenv("LD_CONV_MT4_MIN_WGS") ? atol(getenv("LD_CONV_MT4_MIN_WGS")) : 256;
  if (mt4 && (long)((a.W + 15) / 16) * ((a.H + 7) / 8) * (a.Cout / 64) * a.B < mt4_min) mt4 = false;
  // enough workgroups to fill 256 CUs a couple of times over with the big tile?
  const long blocks16 = (long)((a.W + 15) / 16) * ((a.H + 15) / 16) * (a.Cout / (mt4 ? 64 : 32)) * a.B;
  // 64-channel tiles keep 2 pixel rows per wave: with the prefetch registers the 4-row variant
  // drops to one wave per SIMD and measured slower (64->64@128^2: 23.3 vs 19.7 us)
  static const long big_min = getenv("LD_CONV_BIG_MIN") ? atol(getenv("LD_CONV_BIG_MIN")) : 512;   // tuning override
  bool big = blocks16 >= big_min && a.H >= 16 && !mt4;
  if (force_mt == 2) mt4 = false;
  if (force_mt == 4 && (a.Cout % 64) == 0) mt4 = true;
  if (force_nw == 2) big = false;
  if (force_nw == 4) big = true;
#ifdef LD_DEBUG_VARIANTS          // experiment-only (finding 15): exists in a library built with --debug-variants
  static const int force_deep = getenv("LD_CONV_DEEP") ? atoi(getenv("LD_CONV_DEEP")) : -1;
#else
  constexpr int force_deep = -1;
#endif
  const int ck = sizeof(T) == 4 ? 16 : 32;
  const int nch = (a.s[0].C + (a.nsrc > 1 ? a.s[1].C : 0)) / ck;
  const long wg = (long)((a.W + 15) / 16) * ((a.H + 7) / 8) * (a.Cout / (mt4 ? 64 : 32)) * a.B;
  // measured: distance-2 prefetch is within noise of distance 1 on every small-map shape (the waits there are
  // barrier skew, not load latency), so it stays an opt-in experiment (LD_CONV_DEEP=1)
  bool deep = false;
  if (force_deep >= 0) deep = force_deep != 0;
  // split-K halves (opt-in, LD_CONV_SK=1: launches with >= 8 chunks and at most LD_CONV_SK_MAX_WGS workgroups;
  // LD_CONV_SK=2: every eligible launch).  Measured: 256->256 @32^2 18.0 -> 17.1 us, with the GroupNorm prologue
  // 25.5 -> 22.5, 512->256 46.0 -> 37.6; four-chunk launches and grids beyond one workgroup per CU lose.  Over a
  // step: +0.6 % for one batch of 8 on one stream, -1 % with two concurrent sub-batches (the second stream already
  // fills the gaps this variant closes), hence off by default.
  static const int force_sk = getenv("LD_CONV_SK") ? atoi(getenv("LD_CONV_SK")) : 0;
  static const long sk_max_wgs = getenv("LD_CONV_SK_MAX_WGS") ? atol(getenv("LD_CONV_SK_MAX_WGS")) : 256;
  bool sk = force_sk >= 1 && sizeof(T) == 2 && !big && !deep && ((nch >= 8 && wg <= sk_max_wgs) || force_sk == 2);
  if constexpr (sizeof(T) == 2) {
    if (sk) return mt4 ? launch_dbg<T, 4, 2, false, 0, true>(a, st) : launch_dbg<T, 2, 2, false, 0, true>(a, st);
  }
#ifdef LD_DEBUG_VARIANTS
  if (deep && !big) return mt4 ? launch<T, 4, 2, true>(a, st) : launch<T, 2, 2, true>(a, st);
#endif
  (void)deep;
  if (mt4) return big ? launch<T, 4, 4, false>(a, st) : launch<T, 4, 2, false>(a, st);
  return big ? launch<T, 2, 4, false>(a, st) : launch<T, 2, 2, false>(a, st);
}

}  // namespace

// Debug hook (not part of the public ABI): cycle stamps of the last LD_CONV_DEBUG=64 launch (16 uint64).
extern "C" int ld_debug_conv_trace(unsigned long long* host) {
  LD_HIP(hipDeviceSynchronize());
  LD_HIP(hipMemcpyFromSymbol(host, HIP_SYMBOL(g_conv_trace), sizeof(unsigned long long) * 16));
  return LD_OK;
}

extern "C" int ld_conv3x3(const ld_conv3x3_args* p, void* stream) {
  LD_REQUIRE(p != nullptr, "ld_conv3x3: null args");
  LD_REQUIRE(p->nsrc == 1 || p->nsrc == 2, "ld_conv3x3: nsrc must be 1 or 2 (got %d)", p->nsrc);
  LD_REQUIRE(ld_dtype_ok(p->dtype), "ld_conv3x3: bad dtype %d", p->dtype);
  LD_REQUIRE(p->Cout > 0 && p->Cout % 32 == 0, "ld_conv3x3: Cout %d must be a multiple of 32", p->Cout);
  LD_REQUIRE(p->B > 0 && p->H > 0 && p->W > 0, "ld_conv3x3: bad shape");
  LD_REQUIRE(p->weight && p->bias && p->out, "ld_conv3x3: null weight/bias/out");
  Conv3Dev a;
  a.nsrc = p->nsrc;
  for (int s = 0; s < p->nsrc; ++s) {
    const ld_src& S = p->src[s];
    LD_REQUIRE(S.data != nullptr, "ld_conv3x3: src[%d] null", s);
    LD_REQUIRE(S.C > 0 && S.C % 32 == 0, "ld_conv3x3: src[%d].C=%d must be a multiple of 32", s, S.C);
    if (S.upsample) LD_REQUIRE(p->H % 2 == 0 && p->W % 2 == 0, "ld_conv3x3: upsample needs even H,W");
    if (S.gn_stats) {
      LD_REQUIRE(S.gn_gamma && S.gn_beta && S.gn_groups > 0 && S.C % S.gn_groups == 0,
                 "ld_conv3x3: src[%d] GroupNorm prologue incomplete", s);
    }
    a.s[s] = to_dev(S);
  }
  if (p->nsrc == 1) a.s[1] = a.s[0];
  if (p->out_stats) {
    LD_REQUIRE(p->out_groups > 0 && p->Cout % p->out_groups == 0 && (p->Cout / p->out_groups) <= 32,
               "ld_conv3x3: out_groups %d incompatible with Cout %d", p->out_groups, p->Cout);
  }
  a.w = p->weight; a.bias = p->bias; a.out = p->out; a.ostats = p->out_stats;
  a.ogroups = p->out_groups > 0 ? p->out_groups : 1;
  a.B = p->B; a.H = p->H; a.W = p->W; a.Cout = p->Cout; a.t_ptr = p->t_ptr; a.tiles_x = 0;
  a.addend = p->addend;
  static const int dbg = getenv("LD_CONV_DEBUG") ? atoi(getenv("LD_CONV_DEBUG")) : 0;
  a.dbg = dbg;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  {
    const int rc = ld_conv3x3_c32_try(p, st);        // persistent LDS-DMA kernel for the Cout=32 stages
    if (rc != 0) return rc < 0 ? rc : LD_OK;
  }
  {
    const int rc = ld_conv3x3_ws_try(p, st);         // small maps, 64-256 input channels: weights in registers, K split over waves
    if (rc != 0) return rc < 0 ? rc : LD_OK;
  }
  ld_count(LD_COUNTER_CONV3X3_GENERIC);
  return LD_DISPATCH(p->dtype, dispatch<T>(a, st));
}
