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
#include "common.cuh"

typedef __attribute__((ext_vector_type(16))) float f32x16;

namespace {

// ------------------------------------------------------------------ per-channel max of k over n
template <typename T>
__global__ __launch_bounds__(256) void kmax_kernel(const T* __restrict__ qkv, float* __restrict__ part, int n,
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
    part[((size_t)b * nparts + pt) * hidden + c] = m;
  }
}

// ------------------------------------------------------------------ ctx partials
constexpr int CTX_STRIDE = 32 * 32 + 32;   // floats per (b, h, chunk): ctx[d][e] then Z[d]

template <typename T>
__global__ __launch_bounds__(256) void ctx_kernel(const T* __restrict__ qkv, const float* __restrict__ kmax_part,
                                                  int nparts, float* __restrict__ ctx_part, int n, int heads,
                                                  int nchunks) {
  __shared__ float s_red[4][CTX_STRIDE];
  const int hidden = heads * 32;
  const int ck = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, r = lane & 31, half = lane >> 5;
  float km = -INFINITY;
  for (int p = 0; p < nparts; ++p) km = fmaxf(km, kmax_part[((size_t)b * nparts + p) * hidden + h * 32 + r]);
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
      av[u] = ok ? expf(kk - km) : 0.f;
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
  for (int i = tid; i < CTX_STRIDE; i += 256) dst[i] = s_red[0][i] + s_red[1][i] + s_red[2][i] + s_red[3][i];
}

// ------------------------------------------------------------------ fold ctx into per-batch 1x1 weights
// grid = (C/16 channel tiles, B): every workgroup reduces the chunk partials of ctx (L2-resident,
// heads*1056 floats per chunk) into LDS, then produces one 16-row tile of M_b.
template <typename T>
__global__ __launch_bounds__(256) void fold_kernel(const float* __restrict__ ctx_part, int nchunks,
                                                   const float* __restrict__ w_out, T* __restrict__ w_packed,
                                                   int C, int heads) {
  constexpr int E = DT<T>::E, CK = DT<T>::CK;
  extern __shared__ float s_ctx[];                       // [heads][32][32] normalised, then [heads][32] Z
  const int hidden = heads * 32, b = blockIdx.y, mt = blockIdx.x, tid = threadIdx.x;
  float* s_z = s_ctx + heads * 1024;
  for (int i = tid; i < heads * 32; i += 256) {
    const int h = i / 32, d = i - h * 32;
    float z = 0.f;
    for (int c = 0; c < nchunks; ++c) z += ctx_part[(((size_t)b * heads + h) * nchunks + c) * CTX_STRIDE + 1024 + d];
    s_z[i] = z;
  }
  __syncthreads();
  for (int i = tid; i < heads * 1024; i += 256) {
    const int h = i / 1024, de = i - h * 1024;
    float s = 0.f;
    for (int c = 0; c < nchunks; ++c) s += ctx_part[(((size_t)b * heads + h) * nchunks + c) * CTX_STRIDE + de];
    s_ctx[i] = s / s_z[h * 32 + (de >> 5)];
  }
  __syncthreads();
  const int mt_total = C / 16;
  T* dst = w_packed + (size_t)b * C * hidden;
  for (int i = tid; i < 16 * hidden; i += 256) {
    const int ii = i / hidden, ci = i - ii * hidden;     // ci = h*32 + d
    const int co = mt * 16 + ii;
    const int h = ci >> 5, d = ci & 31;
    const float* wrow = w_out + (size_t)co * hidden + h * 32;
    const float* crow = s_ctx + h * 1024 + d * 32;
    float m = 0.f;
#pragma unroll 8
    for (int e = 0; e < 32; ++e) m = fmaf(wrow[e], crow[e], m);
    const int ch = ci / CK, kq = (ci % CK) / E, e = ci % E;
    dst[((((size_t)ch * mt_total + mt) * 4 + kq) * 16 + ii) * E + e] = from_f<T>(m);
  }
}

}  // namespace

extern "C" size_t ld_linattn_ctx_part_floats(int B, int heads, int dim_head, int nchunks) {
  (void)dim_head;
  return (size_t)B * heads * nchunks * CTX_STRIDE;
}

extern "C" int ld_linattn_kmax(const void* qkv, float* kmax_part, int B, int n, int heads, int dim_head,
                               int nparts, int dtype, void* stream) {
  LD_REQUIRE(qkv && kmax_part && B > 0 && n > 0 && nparts > 0, "ld_linattn_kmax: bad args");
  LD_REQUIRE(dim_head == 32, "ld_linattn_*: dim_head must be 32 (got %d)", dim_head);
  const int hidden = heads * dim_head;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  dim3 grid(nparts, B);
  if (dtype == LD_F32) {
    const int rows = 256 / (hidden / 4);
    LD_REQUIRE(rows >= 1, "ld_linattn_kmax: hidden %d too large", hidden);
    hipLaunchKernelGGL(kmax_kernel<float>, grid, dim3(256), rows * hidden * sizeof(float), st,
                       (const float*)qkv, kmax_part, n, hidden, nparts);
  } else if (dtype == LD_BF16) {
    const int rows = 256 / (hidden / 8);
    hipLaunchKernelGGL(kmax_kernel<bf16>, grid, dim3(256), rows * hidden * sizeof(float), st,
                       (const bf16*)qkv, kmax_part, n, hidden, nparts);
  } else {
    return ld_fail(LD_EINVAL, "ld_linattn_kmax: bad dtype %d", dtype);
  }
  LD_LAUNCH_CHECK("linattn_kmax");
  return LD_OK;
}

extern "C" int ld_linattn_ctx(const void* qkv, const float* kmax_part, int nparts, float* ctx_part, int B,
                              int n, int heads, int dim_head, int nchunks, int dtype, void* stream) {
  LD_REQUIRE(qkv && kmax_part && ctx_part && B > 0 && n > 0 && nchunks > 0, "ld_linattn_ctx: bad args");
  LD_REQUIRE(dim_head == 32, "ld_linattn_*: dim_head must be 32 (got %d)", dim_head);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  dim3 grid(nchunks, heads, B);
  if (dtype == LD_F32)
    hipLaunchKernelGGL(ctx_kernel<float>, grid, dim3(256), 0, st, (const float*)qkv, kmax_part, nparts, ctx_part, n, heads, nchunks);
  else if (dtype == LD_BF16)
    hipLaunchKernelGGL(ctx_kernel<bf16>, grid, dim3(256), 0, st, (const bf16*)qkv, kmax_part, nparts, ctx_part, n, heads, nchunks);
  else
    return ld_fail(LD_EINVAL, "ld_linattn_ctx: bad dtype %d", dtype);
  LD_LAUNCH_CHECK("linattn_ctx");
  return LD_OK;
}

extern "C" int ld_linattn_fold(const float* ctx_part, int nchunks, const float* w_out, void* w_packed, int B,
                               int C, int heads, int dim_head, int dtype, void* stream) {
  LD_REQUIRE(ctx_part && w_out && w_packed && B > 0 && nchunks > 0, "ld_linattn_fold: bad args");
  LD_REQUIRE(dim_head == 32 && C % 16 == 0, "ld_linattn_fold: dim_head 32, C %% 16 == 0");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const size_t lds = (size_t)heads * (1024 + 32) * sizeof(float);
  if (dtype == LD_F32)
    hipLaunchKernelGGL(fold_kernel<float>, dim3(C / 16, B), dim3(256), lds, st, ctx_part, nchunks, w_out, (float*)w_packed, C, heads);
  else if (dtype == LD_BF16)
    hipLaunchKernelGGL(fold_kernel<bf16>, dim3(C / 16, B), dim3(256), lds, st, ctx_part, nchunks, w_out, (bf16*)w_packed, C, heads);
  else
    return ld_fail(LD_EINVAL, "ld_linattn_fold: bad dtype %d", dtype);
  LD_LAUNCH_CHECK("linattn_fold");
  return LD_OK;
}
