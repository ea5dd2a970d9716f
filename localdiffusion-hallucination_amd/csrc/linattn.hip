// Linear-attention core (LinearAttention.forward, ddpm.py:234-251) on the NHWC qkv tensor
// [B, n, 3*hidden] (q | k | v, hidden = heads*32) written by ld_conv1x1(LD_EPI_QKV_LINEAR):
//
//   k = softmax_n(k)                         (:243)   -> max pass (kmax) + exp/sum in the ctx pass
//   ctx[b,h,d,e] = sum_n k[d,n] v[e,n]       (:247)   -> ld_linattn_ctx, fp32 MFMA 32x32x2
//   out[e,n]     = sum_d ctx[d,e] q[d,n]     (:249)   \  folded: to_out(out) = M_b q  with
//   to_out conv1x1 (hidden -> C)             (:229)   /  M_b[c, h*32+d] = sum_e Wout[c,h*32+e] ctx[d,e]/Z[d]
//
// The n-contraction of ctx has its K index on the *strided* (pixel) axis of NHWC, which is what
// v_mfma_f32_32x32x2_f32 wants: each lane supplies ONE element A[d][n] / B[n][e], so a wave reads
// two pixel rows of 32 contiguous channels per instruction -- coalesced, no transpose, exact fp32
// (the instruction is a k-ordered fmaf chain).  Partial results per pixel chunk are written out
// and reduced in the fold kernel (deterministic; no float atomics).
#include "common.hip.h"

typedef __attribute__((ext_vector_type(16))) float f32x16;

namespace {

// ------------------------------------------------------------------ per-channel max of k over n
template <typename T>
__global__ __launch_bounds__(256) void kmax_kernel(const T* __restrict__ qkv, unsigned* __restrict__ kmax_enc, int n,
                                                   int hidden, int nparts) {
  constexpr int E = DT<T>::E;
  extern __shared__ float s_max[];                      // [rows][hidden]
  const int b = blockIdx.y, pt = blockIdx.x, tid = threadIdx.x;
  const int fpr = hidden / E;                           // fragments per pixel row of k
  const int rows = 256 / fpr;
  const int fr = tid % fpr, row = tid / fpr;
  const int npp = (n + nparts - 1) / nparts;
  const int lo = pt * npp, hi = min(n, lo + npp);
  float mx[E];
#pragma unroll
  for (int e = 0; e < E; ++e) mx[e] = -INFINITY;
  if (row < rows) {
    for (int p = lo + row; p < hi; p += rows) {
      const uint4 r = *reinterpret_cast<const uint4*>(qkv + ((size_t)b * n + p) * 3 * hidden + hidden + fr * E);
      float v[E];
      unpack16<T>(r, v);
#pragma unroll
      for (int e = 0; e < E; ++e) mx[e] = fmaxf(mx[e], v[e]);
    }
#pragma unroll
    for (int e = 0; e < E; ++e) s_max[row * hidden + fr * E + e] = mx[e];
  }
  __syncthreads();
  for (int c = tid; c < hidden; c += 256) {
    float m = -INFINITY;
    for (int r = 0; r < rows; ++r) m = fmaxf(m, s_max[r * hidden + c]);
    if (m > -INFINITY) atomicMax(kmax_enc + ((size_t)b * LD_STAT_STRIPES + pt % LD_STAT_STRIPES) * hidden + c, enc_max(m));
  }
}

// ------------------------------------------------------------------ ctx partials
constexpr int CTX_STRIDE = 32 * 32 + 64;   // floats per (b, h, chunk): ctx[d][e], Z[d], then the max m[d] the chunk's
                                           // exponentials were taken against (see ctx_reduce_kernel)

template <typename T>
__global__ __launch_bounds__(256) void ctx_kernel(const T* __restrict__ qkv, const unsigned* __restrict__ kmax_enc,
                                                  float* __restrict__ ctx_part, int n, int heads, int nchunks) {
  __shared__ float s_red[4][CTX_STRIDE];
  const int hidden = heads * 32;
  const int ck = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, r = lane & 31, half = lane >> 5;
  unsigned kme = 0u;
#pragma unroll
  for (int sp = 0; sp < LD_STAT_STRIPES; ++sp)
    kme = max(kme, kmax_enc[((size_t)b * LD_STAT_STRIPES + sp) * hidden + h * 32 + r]);
  const float km = dec_max(kme);
  const int npc = (n + nchunks - 1) / nchunks;           // pixels per chunk
  const int lo = ck * npc, hi = min(n, lo + npc);
  const int npw = (hi - lo + 3) / 4;                     // pixels per wave
  const int wlo = lo + wv * npw, whi = min(hi, wlo + npw);
  const T* kbase = qkv + (size_t)b * n * 3 * hidden + hidden + h * 32 + r;
  const T* vbase = kbase + hidden;
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  float z = 0.f;
  constexpr int U = 8;
  for (int p0 = wlo; p0 < whi; p0 += 2 * U) {
    float av[U], bv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int p = p0 + 2 * u + half;
      const bool ok = p < whi;
      const size_t off = (size_t)(ok ? p : wlo) * 3 * hidden;
      const float kk = to_f<T>(kbase[off]), vv = to_f<T>(vbase[off]);
      av[u] = ok ? (DT<T>::precise ? expf(kk - km) : __expf(kk - km)) : 0.f;
      bv[u] = ok ? vv : 0.f;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u], acc, 0, 0, 0);
      z += av[u];
    }
  }
  // D layout: col e = lane&31, row d = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int d = (i & 3) + 8 * (i >> 2) + 4 * half;
    s_red[wv][d * 32 + r] = acc[i];
  }
  z += __shfl_xor(z, 32);
  if (half == 0) s_red[wv][1024 + r] = z;
  __syncthreads();
  float* dst = ctx_part + (((size_t)b * heads + h) * nchunks + ck) * CTX_STRIDE;
  for (int i = tid; i < 1056; i += 256) dst[i] = s_red[0][i] + s_red[1][i] + s_red[2][i] + s_red[3][i];
  if (wv == 0 && half == 0) dst[1056 + r] = km;       // every chunk used the global max
}

// ------------------------------------------------------------------ ctx partials, bf16 MFMA
// bf16 storage: ctx = P^T V with the contraction over pixels on v_mfma_f32_16x16x32_bf16.  Both
// operands have their K index (the pixel) on the strided axis of the row-major [pixel][32 ch] LDS
// tiles, so both fragments come from ds_read_b64_tr_b16 (transposed LDS read; rows padded to 96 B
// so the 8 rows a half-wave touches hit disjoint bank octets).  Wave w owns the 16x16 tile
// (d-tile w>>1, e-tile w&1) of the 32x32 context.  P = exp(k - max) is formed while staging
// (16-B loads, 8 channels per lane), its column sums Z come from the same registers.
typedef __attribute__((ext_vector_type(4))) short s16x4;
constexpr int TN = 256;          // pixels per LDS tile
constexpr int CROW = 96;         // bytes per LDS row (64 data + 32 pad)

__device__ __forceinline__ uint2 tr_read8(const char* p) {
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) s16x4*)(const_cast<char*>(p)));
  return __builtin_bit_cast(uint2, v);
}

template <typename T>
__global__ __launch_bounds__(256) void ctx_mfma_kernel(const T* __restrict__ qkv, const unsigned* __restrict__ kmax_enc,
                                                       float* __restrict__ ctx_part, int n, int heads, int nchunks) {
  __shared__ __attribute__((aligned(16))) char s_p[TN * CROW];
  __shared__ __attribute__((aligned(16))) char s_v[TN * CROW];
  const int hidden = heads * 32;
  const int ck = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, li = lane & 15, kg = lane >> 4;
  const int c8 = tid & 3, prow = tid >> 2;               // staging role: 8 channels of pixel prow (+64*it)
  float km[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    unsigned kme = 0u;
#pragma unroll
    for (int sp = 0; sp < LD_STAT_STRIPES; ++sp)
      kme = max(kme, kmax_enc[((size_t)b * LD_STAT_STRIPES + sp) * hidden + h * 32 + c8 * 8 + e]);
    km[e] = dec_max(kme);
  }
  float z[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) z[e] = 0.f;
  const int npc = (n + nchunks - 1) / nchunks;
  const int lo = ck * npc, hi = min(n, lo + npc);
  const T* kbase = qkv + (size_t)b * n * 3 * hidden + hidden + h * 32 + c8 * 8;
  const int dt = wv >> 1, et = wv & 1;
  const int tq = (lane >> 2) & 3, tp = lane & 3;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int p0 = lo; p0 < hi; p0 += TN) {
    __syncthreads();
#pragma unroll
    for (int it = 0; it < TN / 64; ++it) {
      const int r = prow + 64 * it, p = p0 + r;
      uint4 pk = make_uint4(0u, 0u, 0u, 0u), vv = pk;
      if (p < hi) {
        const T* rp = kbase + (size_t)p * 3 * hidden;
        const uint4 kk = *reinterpret_cast<const uint4*>(rp);
        vv = *reinterpret_cast<const uint4*>(rp + hidden);
        float f[8];
        unpack16<T>(kk, f);
        // P is stored scaled by 2^pshift (fp16: keeps small weights out of the subnormals; Z carries the same factor)
#pragma unroll
        for (int e = 0; e < 8; ++e) { f[e] = __expf(f[e] - km[e] + DT<T>::pshift * 0.6931471805599453f); z[e] += f[e]; }
        pk = pack16<T>(f);
      }
      *reinterpret_cast<uint4*>(s_p + r * CROW + c8 * 16) = pk;
      *reinterpret_cast<uint4*>(s_v + r * CROW + c8 * 16) = vv;
    }
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < TN / 32; ++ks) {
      const int row = ks * 32 + kg * 4 + tq;
      const char* pa = s_p + row * CROW + dt * 32 + tp * 8;
      const char* pb = s_v + row * CROW + et * 32 + tp * 8;
      const uint2 a1 = tr_read8(pa), a2 = tr_read8(pa + 16 * CROW);
      const uint2 b1 = tr_read8(pb), b2 = tr_read8(pb + 16 * CROW);
      mma16<T>(acc, make_uint4(a1.x, a1.y, a2.x, a2.y), make_uint4(b1.x, b1.y, b2.x, b2.y));
    }
  }
  float* dst = ctx_part + (((size_t)b * heads + h) * nchunks + ck) * CTX_STRIDE;
  // D: row d = dt*16 + 4*kg + r, col e = et*16 + li
#pragma unroll
  for (int r = 0; r < 4; ++r) dst[(dt * 16 + kg * 4 + r) * 32 + et * 16 + li] = acc[r];
  // Z[d]: deterministic tree over the 64 threads that share c8
  __syncthreads();
  float* s_z = reinterpret_cast<float*>(s_p);            // [256][8]
#pragma unroll
  for (int e = 0; e < 8; ++e) s_z[tid * 8 + e] = z[e];
  __syncthreads();
  if (tid < 32) {
    const int cc = tid >> 3, e = tid & 7;                 // channel = cc*8 + e
    float t = 0.f;
    for (int k = 0; k < 64; ++k) t += s_z[(k * 4 + cc) * 8 + e];
    dst[1024 + tid] = t;
  }
  if (tid < 4) {                                          // c8 == tid: this thread's km[] are channels 8*tid..
#pragma unroll
    for (int e = 0; e < 8; ++e) dst[1056 + tid * 8 + e] = km[e];
  }
}

// ------------------------------------------------------------------ reduce the chunk partials
// grid = (heads, B, 4): ctxn[b,h,d,e] = sum_chunks ctx / sum_chunks Z[d]   (the k-softmax normaliser)
constexpr int MAXCH = 128;               // chunks the reduce kernel can combine
__global__ __launch_bounds__(256) void ctx_reduce_kernel(const float* __restrict__ ctx_part, int nchunks,
                                                         float* __restrict__ ctxn, int heads) {
  // partial c was accumulated against its own max m_c[d]; rescale to M[d] = max_c m_c[d]:
  //   ctx[d][e] = sum_c w_c[d] ctx_c[d][e],  Z[d] = sum_c w_c[d] Z_c[d],  w_c[d] = exp(m_c[d]-M[d])
  __shared__ float s_w[8][MAXCH + 1], s_zc[8][MAXCH + 1], s_z[8];
  const int h = blockIdx.x, b = blockIdx.y, quarter = blockIdx.z, tid = threadIdx.x;
  const float* src = ctx_part + ((size_t)b * heads + h) * nchunks * CTX_STRIDE;
  const int i = quarter * 256 + tid;                     // element d*32+e; this block covers d = 8*quarter..+7
  {
    const int dl = tid >> 5, d = quarter * 8 + dl;
    for (int c = tid & 31; c < MAXCH; c += 32) {
      const bool ok = c < nchunks;
      s_w[dl][c] = ok ? src[(size_t)c * CTX_STRIDE + 1056 + d] : -INFINITY;
      s_zc[dl][c] = ok ? src[(size_t)c * CTX_STRIDE + 1024 + d] : 0.f;
    }
  }
  __syncthreads();
  if (tid < 8) {
    float M = -INFINITY;
    for (int c = 0; c < nchunks; ++c) M = fmaxf(M, s_w[tid][c]);
    float z = 0.f;
    for (int c = 0; c < nchunks; ++c) {
      const float wgt = expf(s_w[tid][c] - M);
      s_w[tid][c] = wgt;
      z += wgt * s_zc[tid][c];
    }
    s_z[tid] = z;
  }
  __syncthreads();
  const int dl = tid >> 5;
  float s = 0.f;
#pragma unroll 8
  for (int c = 0; c < nchunks; ++c) s += s_w[dl][c] * src[(size_t)c * CTX_STRIDE + i];
  ctxn[((size_t)b * heads + h) * 1024 + i] = s / s_z[dl];
}

// ------------------------------------------------------------------ fold ctx into per-batch 1x1 weights
// grid = (C/16 channel tiles, B): M_b[c, h*32+d] = sum_e Wout[c, h*32+e] * ctxn[b,h,d,e], written in
// the packed fragment order ld_conv1x1 reads (k=1 layout of pack.hip).
template <typename T>
__global__ __launch_bounds__(256) void fold_kernel(const float* __restrict__ ctxn, const float* __restrict__ w_out,
                                                   T* __restrict__ w_packed, int C, int heads, int perm) {
  constexpr int E = DT<T>::E, CK = DT<T>::CK;
  extern __shared__ float s_ctx[];                       // [heads][32][33] (padded: conflict-free rows)
  const int hidden = heads * 32, b = blockIdx.y, mt = blockIdx.x, tid = threadIdx.x;
  for (int i = tid; i < heads * 1024; i += 256) s_ctx[(i >> 5) * 33 + (i & 31)] = ctxn[(size_t)b * heads * 1024 + i];
  __syncthreads();
  const int mt_total = C / 16;
  T* dst = w_packed + (size_t)b * C * hidden;
  for (int i = tid; i < 16 * hidden; i += 256) {
    const int ii = i / hidden, ci = i - ii * hidden;     // ci = h*32 + d
    const int co = mt * 16 + ii;
    const int h = ci >> 5;
    const float* wrow = w_out + (size_t)co * hidden + h * 32;
    const float* crow = s_ctx + ci * 33;
    float m = 0.f;
#pragma unroll 8
    for (int e = 0; e < 32; ++e) m = fmaf(wrow[e], crow[e], m);
    int ch = ci / CK, kq = (ci % CK) / E, e = ci % E;
    if (perm) {   // chained-MFMA operand order (linattn_fused.hip): ci = 32s + 16(e>>2) + 4kq + (e&3), bf16 only
      ch = ci >> 5; kq = (ci >> 2) & 3; e = ((ci >> 4) & 1) * 4 + (ci & 3);
    }
    store_elem_out<T>(&dst[((((size_t)ch * mt_total + mt) * 4 + kq) * 16 + ii) * E + e], from_f<T>(m));
  }
}

// ------------------------------------------------------------------ reduce + fold in one launch
// grid = (heads, B, 4): combines the chunk partials of 8 k-channels d of one (batch, head) (ctx_reduce_kernel's
// arithmetic), keeps the normalised 8x32 context slice in LDS and immediately folds it into the 8 columns
// h*32+d of M_b = W_out . blockdiag(ctx^T) (fold_kernel's arithmetic and output order): a column of M_b only
// needs row d of the context, so the quarters are independent.  Two dependent launches of ~10-25 us of latency
// each become one; the chunk loop keeps 16 loads per thread in flight.
// DR = k-channels d per workgroup (grid.z = 32 / DR).  DR = 8: every thread owns one element of the 8x32 slice and
// walks ALL chunk partials (nchunks / 16 dependent load batches: 8 at the 128 partials of a 256^2 block).  DR = 2:
// the 256 threads are 4 chunk groups x (2 x 32 elements); a thread sums every fourth partial (2 batches at 128
// partials), the groups meet in LDS -- four times the workgroups, a quarter of the dependent round trips.
// MC = the most chunk partials the instantiation combines (128: every launch of cfg3 / cfg4; 512: one large image per
// launch, where 128 chunks would leave kvctx with fewer workgroups than CUs -- cfg5's 512^2 maps at B = 1).
template <typename T, int DR, int MC = MAXCH>
__global__ __launch_bounds__(256) void ctxfold_kernel(const float* __restrict__ ctx_part, int nchunks,
                                                      const float* __restrict__ w_out, T* __restrict__ w_packed,
                                                      int C, int heads, int perm) {
  constexpr int E = DT<T>::E, CK = DT<T>::CK;
  constexpr int CG = 8 / DR, EL = DR * 32;                 // chunk groups, elements of the slice
  __shared__ float s_w[DR][MC + 1], s_z[DR], s_ctx[DR * 33], s_part[CG][EL];
  const int h = blockIdx.x, b = blockIdx.y, part = blockIdx.z, tid = threadIdx.x;
  const float* src = ctx_part + ((size_t)b * heads + h) * nchunks * CTX_STRIDE;
  const int hidden = heads * 32, mt_total = C / 16;
  // Every request that does not depend on an earlier result leaves before the first wait: the chunk maxima / Z of
  // phase 1, the first 16 context partials of phase 2 and the first W_out row of phase 3.  Written phase by phase the
  // launch was four dependent round trips (tools/scan_head_chain.py); now it is one, plus one per further batch of 16.
  const int r1 = tid % EL, dl1 = r1 >> 5, sub = r1 & 31, d1 = part * DR + dl1;       // phase 1 (threads >= EL: redundant copies)
  float mc[MC / 32], zc[MC / 32];
#pragma unroll
  for (int k = 0; k < MC / 32; ++k) {
    const int c = k * 32 + sub;
    const bool ok = c < nchunks;
    mc[k] = src[(size_t)(ok ? c : 0) * CTX_STRIDE + 1056 + d1];
    zc[k] = src[(size_t)(ok ? c : 0) * CTX_STRIDE + 1024 + d1];
    if (!ok) { mc[k] = -INFINITY; zc[k] = 0.f; }
  }
  const int cg = tid / EL, rem = tid - cg * EL, dl = rem >> 5;
  const int i = part * EL + rem;                           // element d*32+e of the head's context
  float v[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int c = cg + k * CG;
    v[k] = src[(size_t)(c < nchunks ? c : 0) * CTX_STRIDE + i];
  }
  const int nout = C * DR;
  const int idx0 = tid < nout ? tid : nout - 1;
  const int co0 = idx0 / DR;
  f32x4 wr[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) wr[q] = *reinterpret_cast<const f32x4*>(w_out + (size_t)co0 * hidden + h * 32 + q * 4);
  __builtin_amdgcn_sched_barrier(0);
  {   // per k-channel d (32 lanes each): global max of the chunk maxima, rescale weights, Z
    float M = -INFINITY;
#pragma unroll
    for (int k = 0; k < MC / 32; ++k) M = fmaxf(M, mc[k]);
#pragma unroll
    for (int o = 1; o < 32; o <<= 1) M = fmaxf(M, __shfl_xor(M, o));
    float z = 0.f;
#pragma unroll
    for (int k = 0; k < MC / 32; ++k) {
      const int c = k * 32 + sub;
      const float wgt = c < nchunks ? expf(mc[k] - M) : 0.f;
      if (tid < EL) s_w[dl1][c] = wgt;
      z += wgt * zc[k];
    }
#pragma unroll
    for (int o = 1; o < 32; o <<= 1) z += __shfl_xor(z, o);
    if (tid < EL && sub == 0) s_z[dl1] = z;
  }
  __syncthreads();
  {
    float s = 0.f;
    for (int c0 = cg; c0 < nchunks; c0 += 16 * CG) {       // chunks cg, cg + CG, ...: 16 loads in flight
      if (c0 != cg) {
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = src[(size_t)(c0 + k * CG < nchunks ? c0 + k * CG : 0) * CTX_STRIDE + i];
      }
#pragma unroll
      for (int k = 0; k < 16; ++k) s = fmaf(c0 + k * CG < nchunks ? s_w[dl][c0 + k * CG] : 0.f, v[k], s);
    }
    if constexpr (CG == 1) {
      s_ctx[dl * 33 + (rem & 31)] = s / s_z[dl];
    } else {
      s_part[cg][rem] = s;
      __syncthreads();
      if (tid < EL) {
        float t = s_part[0][tid];
#pragma unroll
        for (int g = 1; g < CG; ++g) t += s_part[g][tid];
        s_ctx[(tid >> 5) * 33 + (tid & 31)] = t / s_z[tid >> 5];
      }
    }
  }
  __syncthreads();
  T* dst = w_packed + (size_t)b * C * hidden;
  for (int idx = tid; idx < nout; idx += 256) {
    const int co = idx / DR, d8 = idx - co * DR, ci = h * 32 + part * DR + d8;
    const float* wrow = w_out + (size_t)co * hidden + h * 32;
    const float* crow = s_ctx + d8 * 33;
    float m = 0.f;
    if (idx == tid) {
#pragma unroll
      for (int e = 0; e < 32; ++e) m = fmaf(wr[e >> 2][e & 3], crow[e], m);
    } else {
#pragma unroll 8
      for (int e = 0; e < 32; ++e) m = fmaf(wrow[e], crow[e], m);
    }
    const int mt = co >> 4, ii = co & 15;
    int ch = ci / CK, kq = (ci % CK) / E, e = ci % E;
    if (perm) { ch = ci >> 5; kq = (ci >> 2) & 3; e = ((ci >> 4) & 1) * 4 + (ci & 3); }
    store_elem_out<T>(&dst[((((size_t)ch * mt_total + mt) * 4 + kq) * 16 + ii) * E + e], from_f<T>(m));
  }
}

}  // namespace

extern "C" int ld_linattn_ctxfold(const float* ctx_part, int nchunks, const float* w_out, void* w_packed, int B, int C,
                                  int heads, int dim_head, int perm, int dtype, void* stream) {
  LD_REQUIRE(ctx_part && w_out && w_packed && B > 0 && heads > 0 && nchunks > 0 && nchunks <= 4 * MAXCH,
             "ld_linattn_ctxfold: bad args (nchunks 1..512)");
  LD_REQUIRE(dim_head == 32 && C % 16 == 0, "ld_linattn_ctxfold: dim_head 32, C %% 16 == 0");
  LD_REQUIRE(ld_dtype_ok(dtype), "ld_linattn_ctxfold: bad dtype %d", dtype);
  LD_REQUIRE(!(perm && !ld_dtype_16(dtype)), "ld_linattn_ctxfold: perm=1 is the 16-bit chained-operand order");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  LD_DISPATCH(dtype, [&] {
    // many partials: four chunk groups per workgroup (tuning table: fold_split_min, default 32 partials; 0 = never)
    const int split_min = (int)ld_tuning().fold_split_min;
    if (nchunks > MAXCH)      // one k-channel per workgroup, eight chunk groups: 2-4 dependent batches of 16 loads at 256-512 partials
      LD_LAUNCH((ctxfold_kernel<T, 1, 4 * MAXCH>), dim3(heads, B, 32), dim3(256), 0, st, ctx_part, nchunks, w_out, (T*)w_packed, C, heads, perm);
    else if (split_min > 0 && nchunks >= split_min)
      LD_LAUNCH((ctxfold_kernel<T, 2>), dim3(heads, B, 16), dim3(256), 0, st, ctx_part, nchunks, w_out, (T*)w_packed, C, heads, perm);
    else
      LD_LAUNCH((ctxfold_kernel<T, 8>), dim3(heads, B, 4), dim3(256), 0, st, ctx_part, nchunks, w_out, (T*)w_packed, C, heads, perm);
    return 0;
  }());
  LD_LAUNCH_CHECK("linattn_ctxfold");
  return LD_OK;
}

extern "C" size_t ld_linattn_ctx_part_floats(int B, int heads, int dim_head, int nchunks) {
  (void)dim_head;
  return (size_t)B * heads * nchunks * CTX_STRIDE;
}

extern "C" int ld_linattn_kmax(const void* qkv, uint32_t* kmax_enc, int B, int n, int heads, int dim_head,
                               int dtype, void* stream) {
  LD_REQUIRE(qkv && kmax_enc && B > 0 && n > 0, "ld_linattn_kmax: bad args");
  LD_REQUIRE(dim_head == 32, "ld_linattn_*: dim_head must be 32 (got %d)", dim_head);
  const int hidden = heads * dim_head;
  const int nparts = n >= 16384 ? 64 : (n >= 256 ? n / 256 : 1);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  dim3 grid(nparts, B);
  if (dtype == LD_F32) {
    const int rows = 256 / (hidden / 4);
    LD_REQUIRE(rows >= 1, "ld_linattn_kmax: hidden %d too large", hidden);
    LD_LAUNCH(kmax_kernel<float>, grid, dim3(256), rows * hidden * sizeof(float), st,
                       (const float*)qkv, kmax_enc, n, hidden, nparts);
  } else if (ld_dtype_16(dtype)) {
    const int rows = 256 / (hidden / 8);
    LD_DISPATCH16(dtype, [&] {
      LD_LAUNCH(kmax_kernel<T>, grid, dim3(256), rows * hidden * sizeof(float), st, (const T*)qkv, kmax_enc, n, hidden, nparts);
      return 0;
    }());
  } else {
    return ld_fail(LD_EINVAL, "ld_linattn_kmax: bad dtype %d", dtype);
  }
  LD_LAUNCH_CHECK("linattn_kmax");
  return LD_OK;
}

extern "C" int ld_linattn_ctx(const void* qkv, const uint32_t* kmax_enc, float* ctx_part, int B,
                              int n, int heads, int dim_head, int nchunks, int dtype, void* stream) {
  LD_REQUIRE(qkv && kmax_enc && ctx_part && B > 0 && n > 0 && nchunks > 0, "ld_linattn_ctx: bad args");
  LD_REQUIRE(dim_head == 32, "ld_linattn_*: dim_head must be 32 (got %d)", dim_head);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  dim3 grid(nchunks, heads, B);
  if (dtype == LD_F32)
    LD_LAUNCH(ctx_kernel<float>, grid, dim3(256), 0, st, (const float*)qkv, kmax_enc, ctx_part, n, heads, nchunks);
  else if (ld_dtype_16(dtype))
    LD_DISPATCH16(dtype, [&] {
      LD_LAUNCH(ctx_mfma_kernel<T>, grid, dim3(256), 0, st, (const T*)qkv, kmax_enc, ctx_part, n, heads, nchunks);
      return 0;
    }());
  else
    return ld_fail(LD_EINVAL, "ld_linattn_ctx: bad dtype %d", dtype);
  LD_LAUNCH_CHECK("linattn_ctx");
  return LD_OK;
}

extern "C" int ld_linattn_ctx_reduce(const float* ctx_part, int nchunks, float* ctxn, int B, int heads,
                                     int dim_head, void* stream) {
  LD_REQUIRE(ctx_part && ctxn && B > 0 && nchunks > 0 && nchunks <= MAXCH && heads > 0, "ld_linattn_ctx_reduce: bad args (nchunks 1..128)");
  LD_REQUIRE(dim_head == 32, "ld_linattn_*: dim_head must be 32 (got %d)", dim_head);
  LD_LAUNCH(ctx_reduce_kernel, dim3(heads, B, 4), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     ctx_part, nchunks, ctxn, heads);
  LD_LAUNCH_CHECK("linattn_ctx_reduce");
  return LD_OK;
}

extern "C" int ld_linattn_fold(const float* ctxn, const float* w_out, void* w_packed, int B, int C, int heads,
                               int dim_head, int perm, int dtype, void* stream) {
  LD_REQUIRE(ld_dtype_ok(dtype), "ld_linattn_fold: bad dtype %d", dtype);
  LD_REQUIRE(!(perm && !ld_dtype_16(dtype)), "ld_linattn_fold: perm=1 is the 16-bit chained-operand order");
  LD_REQUIRE(ctxn && w_out && w_packed && B > 0, "ld_linattn_fold: bad args");
  LD_REQUIRE(dim_head == 32 && C % 16 == 0, "ld_linattn_fold: dim_head 32, C %% 16 == 0");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const size_t lds = (size_t)heads * 32 * 33 * sizeof(float);
  LD_DISPATCH(dtype, [&] {
    LD_LAUNCH(fold_kernel<T>, dim3(C / 16, B), dim3(256), lds, st, ctxn, w_out, (T*)w_packed, C, heads, perm);
    return 0;
  }());
  LD_LAUNCH_CHECK("linattn_fold");
  return LD_OK;
}
