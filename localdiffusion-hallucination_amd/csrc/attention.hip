// Full softmax attention (Attention.forward, ddpm.py:271-282 + Attend.forward, attend.py:84-113)
// on the NHWC qkv tensor [B, n, 3*hidden] with q already scaled by dim_head^-0.5
// (ld_conv1x1 LD_EPI_QKV_FULL);  out [B, n, hidden].
//
// Flash-style: the n x n similarity matrix (attend.py:102) is never materialised.  One workgroup
// = 64 queries of one (batch, head); K/V tiles of 64 keys go through LDS as fp32; wave w scores
// keys 16w..16w+15 of every tile for all 64 queries (lane = query) with an online softmax
// (running max m, normaliser l), and the four partial (m, l, o) are merged at the end.
// Round-1 version: fp32 VALU arithmetic (exact-ish parity path).  The MFMA (bf16 QK^T / PV)
// version is the planned replacement -- this op is ~2 % of the forward's FLOPs.
#include "common.cuh"

namespace {
constexpr int D = 32;     // dim_head
constexpr int TK = 64;    // keys per tile
constexpr int KW = 16;    // keys per wave per tile

template <typename T>
__global__ __launch_bounds__(256) void attention_kernel(const T* __restrict__ qkv, T* __restrict__ out, int n,
                                                        int heads) {
  __shared__ __attribute__((aligned(16))) float s_k[TK][D];
  __shared__ __attribute__((aligned(16))) float s_v[TK][D];
  __shared__ float s_m[4][64], s_l[4][64];
  __shared__ float s_o[4][D][64];                       // [wave][d][query] -> conflict-free
  const int hidden = heads * D;
  const int b = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * 64;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const size_t rowstride = (size_t)3 * hidden;
  const T* base = qkv + (size_t)b * n * rowstride;
  const int qi = min(q0 + lane, n - 1);
  float q[D], o[D];
  {
    const T* qp = base + (size_t)qi * rowstride + h * D;
#pragma unroll
    for (int d = 0; d < D; d += 4) load4<T>(qp + d, q + d);
#pragma unroll
    for (int d = 0; d < D; ++d) o[d] = 0.f;
  }
  float m = -1e30f, l = 0.f;
  for (int j0 = 0; j0 < n; j0 += TK) {
    __syncthreads();
    // stage K and V tiles: 64 keys x 32 dims x 2 = 4096 floats, 256 threads x 4 float4... (4 floats each x4)
    for (int i = tid; i < TK * D / 4; i += 256) {
      const int key = i / (D / 4), d4 = (i - key * (D / 4)) * 4;
      const int j = j0 + key;
      float kv[4] = {0.f, 0.f, 0.f, 0.f}, vv[4] = {0.f, 0.f, 0.f, 0.f};
      if (j < n) {
        load4<T>(base + (size_t)j * rowstride + hidden + h * D + d4, kv);
        load4<T>(base + (size_t)j * rowstride + 2 * hidden + h * D + d4, vv);
      }
      *reinterpret_cast<float4*>(&s_k[key][d4]) = make_float4(kv[0], kv[1], kv[2], kv[3]);
      *reinterpret_cast<float4*>(&s_v[key][d4]) = make_float4(vv[0], vv[1], vv[2], vv[3]);
    }
    __syncthreads();
    float s[KW];
    float tmax = -1e30f;
#pragma unroll
    for (int kk = 0; kk < KW; ++kk) {
      const int key = wv * KW + kk;
      float acc = 0.f;
#pragma unroll
      for (int d = 0; d < D; d += 4) {
        const float4 kvv = *reinterpret_cast<const float4*>(&s_k[key][d]);   // broadcast read
        acc = fmaf(q[d], kvv.x, acc); acc = fmaf(q[d + 1], kvv.y, acc);
        acc = fmaf(q[d + 2], kvv.z, acc); acc = fmaf(q[d + 3], kvv.w, acc);
      }
      const bool ok = (j0 + key) < n;
      s[kk] = ok ? acc : -1e30f;
      tmax = fmaxf(tmax, s[kk]);
    }
    const float mn = fmaxf(m, tmax);
    const float alpha = expf(m - mn);
    l *= alpha;
#pragma unroll
    for (int d = 0; d < D; ++d) o[d] *= alpha;
    m = mn;
#pragma unroll
    for (int kk = 0; kk < KW; ++kk) {
      const int key = wv * KW + kk;
      const bool ok = (j0 + key) < n;
      const float p = ok ? expf(s[kk] - m) : 0.f;
      l += p;
#pragma unroll
      for (int d = 0; d < D; d += 4) {
        const float4 vvv = *reinterpret_cast<const float4*>(&s_v[key][d]);
        o[d] = fmaf(p, vvv.x, o[d]); o[d + 1] = fmaf(p, vvv.y, o[d + 1]);
        o[d + 2] = fmaf(p, vvv.z, o[d + 2]); o[d + 3] = fmaf(p, vvv.w, o[d + 3]);
      }
    }
  }
  s_m[wv][lane] = m;
  s_l[wv][lane] = l;
#pragma unroll
  for (int d = 0; d < D; ++d) s_o[wv][d][lane] = o[d];
  __syncthreads();
  if (wv == 0) {
    float M = fmaxf(fmaxf(s_m[0][lane], s_m[1][lane]), fmaxf(s_m[2][lane], s_m[3][lane]));
    float f[4], L = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) { f[w] = expf(s_m[w][lane] - M); L += s_l[w][lane] * f[w]; }
    const float inv = 1.0f / L;
    float r[D];
#pragma unroll
    for (int d = 0; d < D; ++d)
      r[d] = (s_o[0][d][lane] * f[0] + s_o[1][d][lane] * f[1] + s_o[2][d][lane] * f[2] + s_o[3][d][lane] * f[3]) * inv;
    if (q0 + lane < n) {
      T* op = out + ((size_t)b * n + q0 + lane) * hidden + h * D;
#pragma unroll
      for (int d = 0; d < D; d += 4) store4<T>(op + d, r + d);
    }
  }
}
}  // namespace

extern "C" int ld_attention(const void* qkv, void* out, int B, int n, int heads, int dim_head, int dtype,
                            void* stream) {
  LD_REQUIRE(qkv && out && B > 0 && n > 0 && heads > 0, "ld_attention: bad args");
  LD_REQUIRE(dim_head == D, "ld_attention: dim_head must be 32 (got %d)", dim_head);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  dim3 grid((n + 63) / 64, heads, B);
  if (dtype == LD_F32)
    hipLaunchKernelGGL(attention_kernel<float>, grid, dim3(256), 0, st, (const float*)qkv, (float*)out, n, heads);
  else if (dtype == LD_BF16)
    hipLaunchKernelGGL(attention_kernel<bf16>, grid, dim3(256), 0, st, (const bf16*)qkv, (bf16*)out, n, heads);
  else
    return ld_fail(LD_EINVAL, "ld_attention: bad dtype %d", dtype);
  LD_LAUNCH_CHECK("attention");
  return LD_OK;
}
