// Persistent 3x3 convolution for the C=32 output stages (32->32 and 64->32 at 256^2 / 128^2: the
// "ResBlock conv path" of the north star, AI 144-192 FLOP/B => HBM-bound if nothing else stalls).
//
// Same arithmetic, fragment layouts and fused prologue/epilogue as conv3x3.hip, different schedule:
//   * grid = (G, B) with G*B ~ 2 workgroups per CU; a workgroup walks tiles g, g+G, ... of ONE image,
//     so the launch has no tail of half-filled waves and per-workgroup setup happens once:
//     weights (<= 2 K-chunks, 36 KiB) go to LDS once, bias and GroupNorm coefficients are built once,
//     GroupNorm statistics of the output stay in registers across tiles and are flushed once;
//   * halo tiles are double-buffered in LDS and filled by LDS-DMA (global_load_lds_dwordx4: no staging
//     registers, so the kernel keeps >= 2 workgroups per CU).  LDS image = 21 blocks of 16 halo pixels,
//     each block [kq 0..3][pixel 0..15][16 B] = 1 KiB = ONE DMA instruction: lane l = kq*16 + p reads
//     fragment kq of pixel p, i.e. the wave reads 16 pixels x 64 B = one contiguous KiB of HBM (the
//     LDS destination is wave-base + lane*16, the per-lane SOURCE does the halo/upsample/concat
//     gather), and a fragment read of 16 consecutive pixels still hits 16 distinct 16-B slots
//     (conflict-free for every tap).  Slots outside the image are zero-filled by LDS writes;
//   * the DMA of item i+1 is issued right after the barrier that retires item i-1's readers and flies
//     during item i's MFMAs (raw s_barrier + explicit s_waitcnt: __syncthreads() would drain it,
//     cdna_hip_programming.md "Pipelining across barriers");
//   * with a GroupNorm prologue the landed tile is transformed in place (LDS -> regs -> LDS) by the
//     wave that owns the plane: its 8 channels are wave-uniform, so the coefficients are scalars.
#include "common.cuh"
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

template <typename T>
__global__ __launch_bounds__(256, 2) void conv3x3_c32_kernel(C32Dev a) {
  constexpr int E = DT<T>::E, CK = DT<T>::CK;
  constexpr int MT = 2, NW = 4, TR = 16, TC = 16, HR = TR + 2, HC = TC + 2;
  constexpr int NPIX = HR * HC, NPIXP = (NPIX + 15) / 16 * 16, PLANE = NPIXP * 16;   // 336 slots
  constexpr int NBLK = NPIXP / 16, BPW = (NBLK + 3) / 4;                              // 16-pixel blocks, per wave
  constexpr int XBUF = NBLK * 1024;                                                  // bytes per halo buffer
  constexpr int WCH = 9 * MT * 1024;                                                 // bytes per weight chunk
  constexpr bool P = DT<T>::precise;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  // LDS budget: two workgroups per CU need <= 81,920 B each: weights nch*18 KiB + 2 halo buffers 42 KiB +
  // coefficients; the fp64 statistics scratch aliases the halo buffers (used only before the first DMA
  // and after the last MFMA).
  const int nch_w = a.s[0].C / CK + (a.nsrc > 1 ? a.s[1].C / CK : 0);
  char* s_w = smem;                                   // [nch][9][MT][1 KiB]
  char* s_x = smem + nch_w * WCH;                     // [2 buffers][NBLK][kq][16 px][16 B]
  float* s_coef = reinterpret_cast<float*>(s_x + 2 * XBUF);
  double* s_stat = reinterpret_cast<double*>(s_x);    // [4 waves][2][32] / coef scratch

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, px = lane & 15, kq = lane >> 4;
  const int b = blockIdx.y, H = a.H, W = a.W;
  const int nch0 = a.s[0].C / CK;
  const int nch = nch0 + (a.nsrc > 1 ? a.s[1].C / CK : 0);
  const int G = gridDim.x;
  const int ntl = (a.ntiles - (int)blockIdx.x + G - 1) / G;       // tiles g, g+G, ...
  const int total = ntl * nch;

  // ---- one-time setup: weights -> LDS, bias -> registers, GroupNorm coefficients -> LDS
  {
    const uint4* wg = reinterpret_cast<const uint4*>(a.w);
    for (int u = tid; u < nch * 9 * MT * 64; u += 256) *reinterpret_cast<uint4*>(s_w + (size_t)u * 16) = wg[u];
  }
  float4 bias[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) bias[m] = *reinterpret_cast<const float4*>(a.bias + m * 16 + kq * 4);
  const bool any_coef = a.s[0].stats != nullptr || (a.nsrc > 1 && a.s[1].stats != nullptr);
  if (any_coef) {
    const int trow = a.t_ptr ? *a.t_ptr : 0;
    int off = 0;
    for (int s = 0; s < a.nsrc; ++s) {
      const SrcDev S = s ? a.s[1] : a.s[0];
      if (S.stats) {
        const long npix = S.ups ? (long)(H / 2) * (W / 2) : (long)H * W;
        build_gn_coef(S, b, trow, npix, s_coef + off, s_stat, tid, 256);
      }
      off += 2 * S.C;
    }
    __syncthreads();                                  // s_stat scratch is about to be overwritten by the first DMA
  }

  // ---- DMA of one (tile, chunk) item into buffer `buf`: wave wv owns plane kq = wv
  auto dma = [&](int item, int buf) {
    if (a.dbg & 1) return;
    const int ti = item / nch, ch = item - ti * nch;
    const int tile = blockIdx.x + ti * G;
    const int ty0 = (tile / a.tiles_x) * TR, tx0 = (tile % a.tiles_x) * TC;
    const int si = ch >= nch0 ? 1 : 0;
    const SrcDev S = si ? a.s[1] : a.s[0];
    const int c0 = (ch - si * nch0) * CK;
    const T* sdata = reinterpret_cast<const T*>(S.data);
    const int Hs = S.ups ? H / 2 : H, Ws = S.ups ? W / 2 : W;
    char* xb = s_x + buf * XBUF;
#pragma unroll
    for (int r = 0; r < BPW; ++r) {
      const int blk = r * 4 + wv;
      if (blk < NBLK) {
        const int q = blk * 16 + px;
        bool inb = false;
        size_t idx = 0;
        if (q < NPIX) {
          const int hy = q / HC, hx = q - hy * HC;
          const int gy = ty0 - 1 + hy, gx = tx0 - 1 + hx;
          if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
            const int sy = S.ups ? gy >> 1 : gy, sx = S.ups ? gx >> 1 : gx;
            idx = (((size_t)b * Hs + sy) * Ws + sx) * S.ld + c0 + kq * E;
            inb = true;
          }
        }
        if (inb) {
          glds16(sdata + idx, __builtin_amdgcn_readfirstlane(lds_addr(xb + blk * 1024)));
        } else {
          *reinterpret_cast<uint4*>(xb + blk * 1024 + lane * 16) = make_uint4(0u, 0u, 0u, 0u);
        }
      }
    }
  };
  // ---- in-place normalise + FiLM + activation of the landed plane (prologue sources only)
  auto transform = [&](int item, int buf) {
    const int ti = item / nch, ch = item - ti * nch;
    const int si = ch >= nch0 ? 1 : 0;
    const SrcDev S = si ? a.s[1] : a.s[0];
    if (S.stats == nullptr || (a.dbg & 16)) return false;
    const int tile = blockIdx.x + ti * G;
    const int ty0 = (tile / a.tiles_x) * TR, tx0 = (tile % a.tiles_x) * TC;
    const int c0 = (ch - si * nch0) * CK;
    const float* cap = s_coef + (si ? 2 * a.s[0].C : 0) + c0 + kq * E;
    float ca[E], cs[E];
#pragma unroll
    for (int e = 0; e < E; ++e) { ca[e] = cap[e]; cs[e] = cap[S.C + e]; }
    char* xb = s_x + buf * XBUF;
#pragma unroll
    for (int r = 0; r < BPW; ++r) {
      const int blk = r * 4 + wv;
      const int q = blk * 16 + px;
      if (blk < NBLK && q < NPIX) {
        const int hy = q / HC, hx = q - hy * HC;
        const int gy = ty0 - 1 + hy, gx = tx0 - 1 + hx;
        if (gy >= 0 && gy < H && gx >= 0 && gx < W) {     // zero padding stays exactly zero
          char* ptr = xb + blk * 1024 + lane * 16;
          uint4 raw = *reinterpret_cast<const uint4*>(ptr);
          float v[E];
          unpack16<T>(raw, v);
          affine_act_n<P, E>(v, ca, cs, S.act);
          *reinterpret_cast<uint4*>(ptr) = pack16<T>(v);
        }
      }
    }
    return true;
  };

  float ssum[MT][4], ssq[MT][4];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) ssum[m][r] = ssq[m][r] = 0.f;
  f32x4 acc[MT][NW];
  T* out = reinterpret_cast<T*>(a.out);

  if (total > 0) dma(0, 0);
  bool stores_behind = false;      // the previous iteration issued exactly 8 stores AFTER this item's DMA
  for (int i = 0; i < total; ++i) {
    const int buf = i & 1;
    const int ti = i / nch, ch = i - ti * nch;
    // my DMA / zero-fill of item i has landed.  vmcnt retires in order and the previous tile's 8 epilogue
    // stores are younger than that DMA, so vmcnt(8) waits for the DMA without draining the stores.
    if (stores_behind) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    stores_behind = false;
    __builtin_amdgcn_s_barrier();                                 // ... and everybody's; item i-1 fully read
    asm volatile("" ::: "memory");
    if (i + 1 < total) dma(i + 1, buf ^ 1);
    if (transform(i, buf)) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
    if (ch == 0) {
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int j = 0; j < NW; ++j) acc[m][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const char* xb = s_x + buf * XBUF;
    const char* wb = s_w + ch * WCH;
    if (!(a.dbg & 4)) {   // fragment reads of tap column dx+1 are in flight during the MFMAs of column dx (see conv3x3.hip)
      uint4 A[2][3][MT], Bq[2][NW + 2];
      auto load_frags = [&](int dx, int set) {
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
          for (int m = 0; m < MT; ++m)
            A[set][dy][m] = *reinterpret_cast<const uint4*>(wb + ((dy * 3 + dx) * MT + m) * 1024 + lane * 16);
#pragma unroll
        for (int rr = 0; rr < NW + 2; ++rr) {
          const int q = (wv * NW + rr) * HC + dx + px;
          Bq[set][rr] = *reinterpret_cast<const uint4*>(xb + ((q >> 4) << 10) + kq * 256 + ((q & 15) << 4));
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
        }
      }
    }
    if (ch == nch - 1) {                                          // tile finished: bias, stats, store
      const int tile = blockIdx.x + ti * G;
      const int ty0 = (tile / a.tiles_x) * TR, tx0 = (tile % a.tiles_x) * TC;
      const int gx = tx0 + px;
      // all 8 store instructions execute in this wave iff its 4 rows are inside the image (uniform test)
      stores_behind = (ty0 + wv * NW + NW <= H) && (i + 1 < total) && !(a.dbg & 8);
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const int co = m * 16 + kq * 4;
        const float4 bv = bias[m];
#pragma unroll
        for (int j = 0; j < NW; ++j) {
          const int gy = ty0 + wv * NW + j;
          if (gy < H && gx < W) {
            float v[4] = {acc[m][j][0] + bv.x, acc[m][j][1] + bv.y, acc[m][j][2] + bv.z, acc[m][j][3] + bv.w};
            if (!(a.dbg & 8)) store4<T>(out + (((size_t)b * H + gy) * W + gx) * 32 + co, v);
#pragma unroll
            for (int r = 0; r < 4; ++r) { ssum[m][r] += v[r]; ssq[m][r] += v[r] * v[r]; }
          }
        }
      }
    }
  }

  if (a.ostats) {                                                 // one flush per workgroup
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
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
    if (tid < 2 * a.ogroups) {
      const int gi = tid >> 1, k = tid & 1;
      double acc1 = 0.0;
      for (int w4 = 0; w4 < 4; ++w4)
        for (int c = 0; c < gs; ++c) acc1 += s_stat[(w4 * 2 + k) * 32 + gi * gs + c];
      const int stripe = blockIdx.x % LD_STAT_STRIPES;
      atomicAdd(&a.ostats[(((size_t)b * LD_STAT_STRIPES + stripe) * a.ogroups + gi) * 2 + k], acc1);
    }
  }
}

template <typename T>
int launch_c32(const C32Dev& a0, hipStream_t st) {
  C32Dev a = a0;
  a.tiles_x = (a.W + 15) / 16;
  a.ntiles = a.tiles_x * ((a.H + 15) / 16);
  const int ctot = a.s[0].C + (a.nsrc > 1 ? a.s[1].C : 0);
  const int ck = sizeof(T) == 4 ? 16 : 32;
  const size_t lds = (size_t)(ctot / ck) * 9 * 2 * 1024 + 2 * 4 * 336 * 16 + 2 * ctot * sizeof(float);
  static size_t allowed = 0;
  if (lds > allowed) {
    LD_HIP(ld_allow_lds(conv3x3_c32_kernel<T>, lds));
    allowed = lds;
  }
  int G = (512 + a.B - 1) / a.B;                       // ~2 workgroups per CU over the whole launch
  if (G > a.ntiles) G = a.ntiles;
  hipLaunchKernelGGL((conv3x3_c32_kernel<T>), dim3(G, a.B), dim3(256), lds, st, a);
  LD_LAUNCH_CHECK("conv3x3_c32");
  return LD_OK;
}

}  // namespace

// Returns 1 if this launch is handled here, 0 if the generic kernel must take it, <0 on error.
int ld_conv3x3_c32_try(const ld_conv3x3_args* p, hipStream_t st) {
  static const int disabled = getenv("LD_CONV_NO_C32") ? 1 : 0;
  if (disabled || p->Cout != 32 || p->H < 32 || p->W < 32) return 0;
  const int ck = p->dtype == LD_F32 ? 16 : 32;
  int ctot = 0;
  for (int s = 0; s < p->nsrc; ++s) ctot += p->src[s].C;
  // measured in situ (cfg3): single-chunk convs gain 3-8 us over the generic kernel, two-chunk ones lose ~5 us
  static const int max_chunks = getenv("LD_CONV_C32_CHUNKS") ? atoi(getenv("LD_CONV_C32_CHUNKS")) : 1;
  if (ctot / ck > max_chunks || ctot / ck > 2) return 0;
  if (p->out_stats && (p->out_groups <= 0 || 32 % p->out_groups != 0)) return 0;
  const long tiles = (long)((p->W + 15) / 16) * ((p->H + 15) / 16) * p->B;
  if (tiles < 1024) return 0;                          // too few tiles to amortise a persistent workgroup
  C32Dev a;
  a.nsrc = p->nsrc;
  for (int s = 0; s < p->nsrc; ++s) a.s[s] = to_dev(p->src[s]);
  if (p->nsrc == 1) a.s[1] = a.s[0];
  a.w = p->weight; a.bias = p->bias; a.out = p->out; a.ostats = p->out_stats;
  a.ogroups = p->out_groups > 0 ? p->out_groups : 1;
  a.B = p->B; a.H = p->H; a.W = p->W; a.t_ptr = p->t_ptr; a.tiles_x = a.ntiles = 0;
  static const int dbg = getenv("LD_CONV_DEBUG") ? atoi(getenv("LD_CONV_DEBUG")) : 0;
  a.dbg = dbg;
  const int rc = p->dtype == LD_F32 ? launch_c32<float>(a, st) : launch_c32<bf16>(a, st);
  return rc == LD_OK ? 1 : rc;
}
