// Pointwise kernels of the reverse process (NCHW fp32 boundary tensors) and the final 1x1 conv.
// All are HBM-streaming; the per-step coefficients come from a device schedule table indexed
// through t_ptr so the whole step is HIP-graph replayable.
#include "common.hip.h"

namespace {
constexpr int BS = 256;
inline unsigned nblocks(long n, int per = 1) {
  long b = (n + (long)BS * per - 1) / ((long)BS * per);
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (unsigned)b;
}
#define GRID_STRIDE(i, n) for (long i = (long)blockIdx.x * BS + threadIdx.x; i < (n); i += (long)gridDim.x * BS)

__device__ __forceinline__ float clampf(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }

__device__ __forceinline__ float to_x0(float x, float mo, const float* row, int obj) {
  if (obj == LD_OBJ_X0) return mo;
  if (obj == LD_OBJ_NOISE) return row[LD_SCHED_SQRT_RECIP] * x - row[LD_SCHED_SQRT_RECIPM1] * mo;
  return row[LD_SCHED_SQRT_AB] * x - row[LD_SCHED_SQRT_1MAB] * mo;
}

// ---------------------------------------------------------------- RNG (rng.py, same integers)
__device__ __forceinline__ unsigned long long mix64(unsigned long long z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__global__ void randn_kernel(float* out, long n, long first, unsigned long long seed, long stream_base, long stream_tmul,
                             const int* t_ptr) {
  const long stream = stream_base + (t_ptr ? stream_tmul * (long)(*t_ptr) : 0);
  const unsigned long long base = mix64(seed ^ ((unsigned long long)stream * 0xD1B54A32D192ED03ull));
  GRID_STRIDE(i, n) {
    const unsigned long long h = mix64(base + (unsigned long long)(first + i + 1) * 0x9E3779B97F4A7C15ull);
    const float u1 = ((float)(unsigned)(h >> 40) + 1.0f) * 5.9604644775390625e-8f;          // 2^-24
    const float u2 = (float)(unsigned)((h >> 16) & 0xFFFFFFu) * 5.9604644775390625e-8f;
    out[i] = sqrtf(-2.0f * logf(u1)) * cosf(6.283185307179586f * u2);
  }
}
__global__ void step_add_kernel(int* t, int delta) { if (threadIdx.x == 0 && blockIdx.x == 0) *t += delta; }
// start of a denoiser evaluation in ONE launch: zero the statistics arena(s) and advance the device step counter
// (as two hipMemsetAsync + ld_step_add these were three dependent ~5-8 us nodes at the head of every replayed step)
// + (film_rows != nullptr) the step's FiLM row: every ResnetBlock's (scale, shift) vector for the NEW timestep is copied
// from the [T, row_floats] table into a fixed buffer, so that the ~20 launches of a step that apply FiLM read it from a
// known address instead of first loading the step counter and then the row it selects -- one dependent global round trip
// less at the head of every workgroup of those launches.  Block 0 does it: it is the one that knows the new timestep.
__global__ void step_begin_kernel(StepBeginDev sb) {
  __shared__ int s_t;
  step_begin_work(sb, blockIdx.x, gridDim.x, &s_t);
}

// ---------------------------------------------------------------- DDPM / DDIM steps
__global__ void ddpm_step_kernel(const float* x, const float* mo, const float* z, float* xp, float* x0o,
                                 const float* sched, const int* t_ptr, float lo, float hi, int obj, long n) {
  const int t = t_ptr ? *t_ptr : 0;
  const float* row = sched + (size_t)t * LD_SCHED_COLS;
  const float c1 = row[LD_SCHED_COEF1], c2 = row[LD_SCHED_COEF2], sg = row[LD_SCHED_SIGMA];
  // the draw is added at t > 0 (ddpm.py:857).  Row mode (t_ptr null: `sched` IS the step's row, the kernel does not know t):
  // the caller says it with the noise pointer -- null at t = 0.  (Round 5: row mode used to read t = 0 and dropped the draw at
  // every step; GaussianDiffusion.p_sample, the one caller, had no test that drew noise.)
  // Both modes are null-safe (round 6): a table-mode call at t > 0 without a noise buffer adds no draw instead of faulting.
  const bool draw = (t_ptr ? t > 0 : true) && z != nullptr;
  GRID_STRIDE(i, n) {
    const float xi = x[i];
    const float x0 = clampf(to_x0(xi, mo[i], row, obj), lo, hi);
    const float mean = c1 * x0 + c2 * xi;
    xp[i] = draw ? mean + sg * z[i] : mean;
    if (x0o) x0o[i] = x0;
  }
}
__global__ void posterior_step_kernel(const float* x, const float* x0, const float* z, float* xp,
                                      const float* sched, const int* t_ptr, long n) {
  const int t = t_ptr ? *t_ptr : 0;
  const float* row = sched + (size_t)t * LD_SCHED_COLS;
  const float c1 = row[LD_SCHED_COEF1], c2 = row[LD_SCHED_COEF2], sg = row[LD_SCHED_SIGMA];
  const bool draw = (t_ptr ? t > 0 : true) && z != nullptr;          // (as in ddpm_step_kernel: null-safe in both modes)
  GRID_STRIDE(i, n) {
    const float mean = c1 * x0[i] + c2 * x[i];
    xp[i] = draw ? mean + sg * z[i] : mean;
  }
}
struct DdimK { float sr, srm1, sab, s1mab, san, c, sigma, lo, hi; int obj, last; };
__global__ void ddim_step_kernel(const float* x, const float* mo, const float* z, float* xn, DdimK k, long n) {
  GRID_STRIDE(i, n) {
    const float xi = x[i], m = mo[i];
    float x0;
    if (k.obj == LD_OBJ_X0) x0 = m;
    else if (k.obj == LD_OBJ_NOISE) x0 = k.sr * xi - k.srm1 * m;
    else x0 = k.sab * xi - k.s1mab * m;
    x0 = clampf(x0, k.lo, k.hi);
    if (k.last) { xn[i] = x0; continue; }
    const float eps = (k.sr * xi - x0) / k.srm1;
    xn[i] = x0 * k.san + k.c * eps + k.sigma * (z ? z[i] : 0.f);
  }
}

// the same update with the pair's scalars read from a device table row {sr, srm1, sab, s1mab, san, c, sigma, last}
// selected by a device pair counter: one captured launch serves every DDIM step of a replayed graph
__global__ void ddim_step_at_kernel(const float* x, const float* mo, const float* z, float* xn, const float* table,
                                    const int* idx, float lo, float hi, int obj, long n) {
  const float* r = table + 8 * (size_t)(*idx);
  const float sr = r[0], srm1 = r[1], sab = r[2], s1mab = r[3], san = r[4], c = r[5], sigma = r[6];
  const bool last = r[7] != 0.0f;
  GRID_STRIDE(i, n) {
    const float xi = x[i], m = mo[i];
    float x0;
    if (obj == LD_OBJ_X0) x0 = m;
    else if (obj == LD_OBJ_NOISE) x0 = sr * xi - srm1 * m;
    else x0 = sab * xi - s1mab * m;
    x0 = clampf(x0, lo, hi);
    if (last) { xn[i] = x0; continue; }
    const float eps = (sr * xi - x0) / srm1;
    xn[i] = x0 * san + c * eps + sigma * (z ? z[i] : 0.f);
  }
}

// ---------------------------------------------------------------- branch / fusion
__global__ void branch_cond_kernel(const float* cond, const float* mask, float* co, float* ci, float lo_clip,
                                   int C, int HW, long n) {
  GRID_STRIDE(i, n) {
    const long bc = i / HW, p = i - bc * HW, b = bc / C;
    const float bin = mask[b * HW + p] >= 1.0f ? 1.0f : 0.0f;
    const float c = cond[i];
    co[i] = c * bin;
    ci[i] = c * clampf(1.0f - bin, lo_clip, 1.0f);
  }
}
__global__ void mask_out_kernel(float* mo, const float* mask, float min_val, int C, int HW, long n) {
  GRID_STRIDE(i, n) {
    const long bc = i / HW, p = i - bc * HW, b = bc / C;
    const float bin = mask[b * HW + p] >= 1.0f ? 1.0f : 0.0f;
    mo[i] = (bin == 0.0f) ? min_val : mo[i] * bin;
  }
}
__global__ void fuse_ddpm_kernel(const float* xo, const float* xi, const float* x0o, const float* x0i,
                                 const float* mask, float* x, float* x0, float lo, float hi, int C, int HW, long n) {
  GRID_STRIDE(i, n) {
    const long bc = i / HW, p = i - bc * HW, b = bc / C;
    const float m = mask[b * HW + p] >= 1.0f ? 1.0f : 0.0f;
    // per-branch clamp (ddpm.py:775-776) then recomposition + clamp (:785-786, :803-804)
    x0[i] = clampf(clampf(x0i[i], lo, hi) * (1.0f - m) + clampf(x0o[i], lo, hi), lo, hi);
    const float a = xo[i] * m, c = xi[i] * (1.0f - m);
    x[i] = (a == 0.0f) ? c : a;
  }
}
// K-mask generalisation: branch 0 = OOD-style, branches 1..K-1 = IND-style; masks [B,K,HW]
__global__ void branch_cond_k_kernel(const float* cond, const float* masks, float* out, float lo_clip, int B, int C,
                                     int K, int HW, long n) {
  GRID_STRIDE(i, n) {                       // i over [B, C, HW]
    const long bc = i / HW, p = i - bc * HW, b = bc / C;
    const float c = cond[i];
    for (int k = 0; k < K; ++k) {
      const float bin = masks[((size_t)b * K + k) * HW + p] >= 1.0f ? 1.0f : 0.0f;
      out[(size_t)k * n + i] = c * (k == 0 ? bin : clampf(bin, lo_clip, 1.0f));
    }
  }
}
__global__ void fuse_ddpm_k_kernel(const float* x_first, const float* x_rest, const float* x0_first, const float* x0_rest,
                                   const float* masks, float* x, float* x0, float lo, float hi, int C, int K, int HW, long n) {
  GRID_STRIDE(i, n) {
    const long bc = i / HW, p = i - bc * HW, b = bc / C;
    const float* mrow = masks + (size_t)b * K * HW + p;
    // x0 = clamp(sum_{k>=1} clamp(x0_k) m_k + clamp(x0_0));  x = first non-zero of x_k m_k  (ddpm.py:784-804 per branch)
    float m = mrow[HW] >= 1.0f ? 1.0f : 0.0f;
    float acc = clampf(x0_rest[i], lo, hi) * m;
    const float a0 = x_first[i] * (mrow[0] >= 1.0f ? 1.0f : 0.0f);
    float xv = (a0 == 0.0f) ? x_rest[i] * m : a0;
    for (int k = 2; k < K; ++k) {
      m = mrow[(size_t)k * HW] >= 1.0f ? 1.0f : 0.0f;
      acc += clampf(x0_rest[(size_t)(k - 1) * n + i], lo, hi) * m;
      if (xv == 0.0f) xv = x_rest[(size_t)(k - 1) * n + i] * m;
    }
    x0[i] = clampf(acc + clampf(x0_first[i], lo, hi), lo, hi);
    x[i] = xv;
  }
}
struct FuseDdimK { float sr, srm1, san, c, sigma, lo, hi; };
__global__ void fuse_ddim_kernel(const float* xo, const float* xi, const float* x0o, const float* x0i,
                                 const float* mask, const float* z, float* xn, FuseDdimK k, int C, int HW, long n) {
  GRID_STRIDE(i, n) {
    const long bc = i / HW, p = i - bc * HW, b = bc / C;
    const float m = mask[b * HW + p] >= 1.0f ? 1.0f : 0.0f;
    const float a0 = clampf(x0o[i], k.lo, k.hi), b0 = clampf(x0i[i], k.lo, k.hi);   // clip_x_start (:726-727)
    const float eo = (k.sr * xo[i] - a0) / k.srm1, ei = (k.sr * xi[i] - b0) / k.srm1;
    const float x0 = clampf((a0 == 0.0f) ? b0 : a0, k.lo, k.hi);
    const float po = eo * m, pi = ei * (1.0f - m);
    const float eps = (po == 0.0f) ? pi : po;
    xn[i] = x0 * k.san + k.c * eps + k.sigma * (z ? z[i] : 0.f);
  }
}
// K-branch DDIM fusion (ddpm.py:1022-1041 read per branch; K = 2 with m_1 = 1 - (m_0 >= 1) is fuse_ddim_kernel bit for bit)
__global__ void fuse_ddim_k_kernel(const float* x_first, const float* x_rest, const float* x0_first, const float* x0_rest,
                                   const float* masks, const float* z, float* xn, FuseDdimK k, int C, int K, int HW, long n) {
  GRID_STRIDE(i, n) {
    const long bc = i / HW, p = i - bc * HW, b = bc / C;
    const float* mrow = masks + (size_t)b * K * HW + p;
    const float a0 = clampf(x0_first[i], k.lo, k.hi);
    const float e0 = (k.sr * x_first[i] - a0) / k.srm1;
    float eps = e0 * (mrow[0] >= 1.0f ? 1.0f : 0.0f);
    // x0 of the IND branch that owns the pixel: the first k >= 1 with m_k >= 1, the last branch if none does
    float alt = 0.f;
    bool have = false;
    for (int kk = 1; kk < K; ++kk) {
      const float m = mrow[(size_t)kk * HW] >= 1.0f ? 1.0f : 0.0f;
      const float ak = clampf(x0_rest[(size_t)(kk - 1) * n + i], k.lo, k.hi);
      const float ek = (k.sr * x_rest[(size_t)(kk - 1) * n + i] - ak) / k.srm1;
      if (!have && (m == 1.0f || kk == K - 1)) { alt = ak; have = true; }
      if (eps == 0.0f) eps = ek * m;
    }
    const float x0 = clampf((a0 == 0.0f) ? alt : a0, k.lo, k.hi);
    xn[i] = x0 * k.san + k.c * eps + k.sigma * (z ? z[i] : 0.f);
  }
}
__global__ void q_sample_kernel(const float* x0, const float* z, float* out, float sab, float s1mab, long n) {
  GRID_STRIDE(i, n) out[i] = sab * x0[i] + s1mab * z[i];
}
// q_sample with a timestep per sample (training-side forward, ddpm.py:1147-1154 with t [B]): grid.y = sample
__global__ void q_sample_t_kernel(const float* x0, const float* z, float* out, const int* t, const float* sab, const float* s1mab,
                                  long per) {
  const int b = blockIdx.y;
  const float a = sab[t[b]], c = s1mab[t[b]];
  const long base = (long)b * per;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < per; i += (long)gridDim.x * blockDim.x)
    out[base + i] = a * x0[base + i] + c * z[base + i];
}
// Per-sample training loss (ddpm.py:1186-1201): loss_b = loss_weight[t_b] * mean_{c,h,w} (model_out - target)^2 with the
// target of the objective -- noise (pred_noise) | x_start (pred_x0) | sqrt_ab[t_b] noise - sqrt_1mab[t_b] x_start
// (pred_v, :643-647).  One workgroup per sample; every thread sums its strided elements in fp64 and the workgroup
// combines them in a fixed order: the result does not depend on scheduling.
__global__ __launch_bounds__(256) void p_losses_kernel(const float* mo, const float* x0, const float* z, const int* t,
                                                       const float* sab, const float* s1mab, const float* lw, float* loss,
                                                       long per, int objective) {
  __shared__ double s_part[256];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int tb = t[b];
  const float a = sab[tb], c = s1mab[tb];
  const long base = (long)b * per;
  double acc = 0.0;
  for (long i = tid; i < per; i += 256) {
    const float xs = x0[base + i], nz = z[base + i];
    const float target = objective == LD_OBJ_NOISE ? nz : (objective == LD_OBJ_X0 ? xs : a * nz - c * xs);
    const float d = mo[base + i] - target;
    acc += (double)(d * d);
  }
  s_part[tid] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) s_part[tid] += s_part[tid + s];
    __syncthreads();
  }
  if (tid == 0) loss[b] = (float)(s_part[0] / (double)per) * lw[tb];
}
__global__ void recompose_kernel(const float* patches, const float* masks, float* out, int K, int C, int HW, long n) {
  GRID_STRIDE(i, n) {                       // i over [B, C, HW]
    const long bc = i / HW, p = i - bc * HW, b = bc / C, c = bc - b * C;
    float acc = 0.f;
    for (int k = 0; k < K; ++k) {
      const float m = masks[(size_t)k * HW + p] >= 1.0f ? 1.0f : 0.0f;
      acc += patches[(((size_t)b * K + k) * C + c) * HW + p] * m;
    }
    out[i] = acc;
  }
}

// ---------------------------------------------------------------- final 1x1 conv -> NCHW fp32
// One thread per pixel.  The pixel's Cin channels are fetched with 16-byte loads (a 64-byte bf16 row in 4
// requests instead of 8) before any arithmetic; the weights sit in LDS (same address in every lane: broadcast).
// The accumulation order (groups of 4 channels, fmaf chain inside a group) is fixed, so the stand-alone kernel and
// the fused final step below give bitwise the same model output.
template <typename T>
__device__ __forceinline__ void final_conv_pixel(const T* xp, const float* s_w, int Cin, int Cout, float* acc) {
  constexpr int E = 16 / sizeof(T);
  acc[0] = acc[1] = acc[2] = acc[3] = 0.f;
  for (int c0 = 0; c0 < Cin; c0 += 4 * E) {             // 4 x 16 B in flight per pass
    uint4 raw[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
      raw[q] = (c0 + q * E < Cin) ? *reinterpret_cast<const uint4*>(xp + c0 + q * E) : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (c0 + q * E >= Cin) break;
      float v[E];
      unpack16<T>(raw[q], v);
#pragma unroll
      for (int g = 0; g < E; g += 4) {
        const int c = c0 + q * E + g;
        for (int o = 0; o < Cout; ++o) {
          const float* wr = s_w + o * Cin + c;
          acc[o] = fmaf(v[g], wr[0], fmaf(v[g + 1], wr[1], fmaf(v[g + 2], wr[2], fmaf(v[g + 3], wr[3], acc[o]))));
        }
      }
    }
  }
}

template <typename T>
__global__ void final_conv_kernel(const T* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                  float* __restrict__ out, int HW, int Cin, int Cout, long npix) {
  extern __shared__ float s_fw[];
  for (int i = threadIdx.x; i < Cin * Cout; i += BS) s_fw[i] = w[i];
  __syncthreads();
  GRID_STRIDE(i, npix) {                    // i over [B, HW]
    const long b = i / HW, p = i - b * HW;
    float acc[4];
    final_conv_pixel<T>(x + (size_t)i * Cin, s_fw, Cin, Cout, acc);
    for (int o = 0; o < Cout; ++o) out[((size_t)b * Cout + o) * HW + p] = acc[o] + bias[o];
  }
}

// final 1x1 conv + the whole ancestral update of the same pixel (ld_ddpm_step's arithmetic, ddpm.py:817-858) with
// the noise of ld_randn generated in place (same counter -> same integers): one launch and one pass over the
// feature map instead of three launches.
template <typename T>
__global__ void final_step_kernel(const T* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                  float* __restrict__ model_out, float* __restrict__ x_t, float* __restrict__ x0o,
                                  const float* __restrict__ sched, const int* __restrict__ t_ptr, float lo, float hi,
                                  int obj, unsigned long long seed, long stream_base, long stream_tmul, long first,
                                  const float* __restrict__ keep_mask, int HW, int Cin, int Cout, long npix) {
  extern __shared__ float s_fw[];
  for (int i = threadIdx.x; i < Cin * Cout; i += BS) s_fw[i] = w[i];
  const int t = t_ptr ? *t_ptr : 0;
  const long stream = stream_base + stream_tmul * (long)t;
  const float* row = sched + (size_t)t * LD_SCHED_COLS;
  const float c1 = row[LD_SCHED_COEF1], c2 = row[LD_SCHED_COEF2], sg = row[LD_SCHED_SIGMA];
  const unsigned long long base = mix64(seed ^ ((unsigned long long)stream * 0xD1B54A32D192ED03ull));
  __syncthreads();
  GRID_STRIDE(i, npix) {
    const long b = i / HW, p = i - b * HW;
    float acc[4];
    final_conv_pixel<T>(x + (size_t)i * Cin, s_fw, Cin, Cout, acc);
    // ld_mask_out folded in: outside the mask the prediction is replaced by the range minimum (ddpm.py:693-696)
    const bool keep = keep_mask ? keep_mask[i] >= 1.0f : true;
    for (int o = 0; o < Cout; ++o) {
      const size_t idx = ((size_t)b * Cout + o) * HW + p;
      const float mo = keep ? acc[o] + bias[o] : lo;
      store_f32_out(&model_out[idx], mo);
      const float xi = x_t[idx];
      const float x0 = clampf(to_x0(xi, mo, row, obj), lo, hi);
      const float mean = c1 * x0 + c2 * xi;
      float r = mean;
      if (t > 0) {
        const unsigned long long h = mix64(base + (unsigned long long)(first + idx + 1) * 0x9E3779B97F4A7C15ull);
        const float u1 = ((float)(unsigned)(h >> 40) + 1.0f) * 5.9604644775390625e-8f;          // 2^-24
        const float u2 = (float)(unsigned)((h >> 16) & 0xFFFFFFu) * 5.9604644775390625e-8f;
        r = mean + sg * (sqrtf(-2.0f * logf(u1)) * cosf(6.283185307179586f * u2));
      }
      store_f32_out(&x_t[idx], r);
      if (x0o) store_f32_out(&x0o[idx], x0);
    }
  }
}
}  // namespace

#define ST(s) reinterpret_cast<hipStream_t>(s)

extern "C" int ld_randn_at(float* out, int64_t n, int64_t first, uint64_t seed, int64_t stream_base,
                           int64_t stream_tmul, const int32_t* t_ptr, void* stream) {
  LD_REQUIRE(out && n > 0 && first >= 0, "ld_randn: bad args");
  LD_LAUNCH(randn_kernel, dim3(nblocks(n)), dim3(BS), 0, ST(stream), out, (long)n, (long)first,
                     (unsigned long long)seed, (long)stream_base, (long)stream_tmul, t_ptr);
  LD_LAUNCH_CHECK("randn");
  return LD_OK;
}
extern "C" int ld_randn(float* out, int64_t n, uint64_t seed, int64_t stream_base, int64_t stream_tmul,
                        const int32_t* t_ptr, void* stream) {
  return ld_randn_at(out, n, 0, seed, stream_base, stream_tmul, t_ptr, stream);
}
extern "C" int ld_step_add(int32_t* t_ptr, int delta, void* stream) {
  LD_REQUIRE(t_ptr, "ld_step_add: null");
  LD_LAUNCH(step_add_kernel, dim3(1), dim3(64), 0, ST(stream), t_ptr, delta);
  LD_LAUNCH_CHECK("step_add");
  return LD_OK;
}
int ld_step_begin_check(const void* zero_a, size_t bytes_a, const void* zero_b, size_t bytes_b, const void* t_ptr, const void* idx_ptr,
                        const void* t_table, const void* film_rows, int row_floats, const void* film_cur) {
  LD_REQUIRE((idx_ptr == nullptr) == (t_table == nullptr), "ld_step_begin: idx_ptr and t_table go together");
  LD_REQUIRE((film_rows == nullptr) == (film_cur == nullptr), "ld_step_begin_film: film_rows and film_cur go together");
  LD_REQUIRE(!film_rows || (t_ptr && row_floats > 0 && row_floats % 4 == 0 && ((size_t)film_rows % 16) == 0 && ((size_t)film_cur % 16) == 0),
             "ld_step_begin_film: needs the step counter, row_floats %% 4 == 0 and 16-byte aligned rows");
  LD_REQUIRE((zero_a || bytes_a == 0) && (zero_b || bytes_b == 0), "ld_step_begin: null arena");
  LD_REQUIRE(bytes_a % 16 == 0 && bytes_b % 16 == 0 && ((size_t)zero_a % 16) == 0 && ((size_t)zero_b % 16) == 0,
             "ld_step_begin: arenas must be 16-byte aligned and sized");
  return LD_OK;
}
extern "C" int ld_step_begin(void* zero_a, size_t bytes_a, void* zero_b, size_t bytes_b, int32_t* t_ptr, int delta,
                             int32_t* idx_ptr, const int32_t* t_table, void* stream) {
  return ld_step_begin_film(zero_a, bytes_a, zero_b, bytes_b, t_ptr, delta, idx_ptr, t_table, nullptr, 0, nullptr, stream);
}
extern "C" int ld_step_begin_film(void* zero_a, size_t bytes_a, void* zero_b, size_t bytes_b, int32_t* t_ptr, int delta,
                                  int32_t* idx_ptr, const int32_t* t_table, const float* film_rows, int row_floats,
                                  float* film_cur, void* stream) {
  if (int rc = ld_step_begin_check(zero_a, bytes_a, zero_b, bytes_b, t_ptr, idx_ptr, t_table, film_rows, row_floats, film_cur)) return rc;
  const long na = (long)(bytes_a / 16), nb = (long)(bytes_b / 16);
  const StepBeginDev sb{(uint4*)zero_a, na, (uint4*)zero_b, nb, t_ptr, delta, idx_ptr, t_table, film_rows, row_floats, film_cur};
  LD_LAUNCH(step_begin_kernel, dim3(nblocks(na + nb + 1)), dim3(BS), 0, ST(stream), sb);
  LD_LAUNCH_CHECK("step_begin");
  return LD_OK;
}
extern "C" int ld_ddpm_step(const float* x_t, const float* model_out, const float* noise, float* x_prev,
                            float* x0_out, const float* sched, const int32_t* t_ptr, float lo, float hi,
                            int objective, int64_t n, void* stream) {
  LD_REQUIRE(x_t && model_out && x_prev && sched && n > 0, "ld_ddpm_step: null pointer");
  LD_REQUIRE(objective >= 0 && objective <= 2, "ld_ddpm_step: objective %d", objective);
  LD_LAUNCH(ddpm_step_kernel, dim3(nblocks(n)), dim3(BS), 0, ST(stream), x_t, model_out, noise, x_prev,
                     x0_out, sched, t_ptr, lo, hi, objective, (long)n);
  LD_LAUNCH_CHECK("ddpm_step");
  return LD_OK;
}
extern "C" int ld_posterior_step(const float* x_t, const float* x0, const float* noise, float* x_prev,
                                 const float* sched, const int32_t* t_ptr, int64_t n, void* stream) {
  LD_REQUIRE(x_t && x0 && x_prev && sched && n > 0, "ld_posterior_step: null pointer");
  LD_LAUNCH(posterior_step_kernel, dim3(nblocks(n)), dim3(BS), 0, ST(stream), x_t, x0, noise, x_prev,
                     sched, t_ptr, (long)n);
  LD_LAUNCH_CHECK("posterior_step");
  return LD_OK;
}
extern "C" int ld_ddim_step(const float* x_t, const float* model_out, const float* noise, float* x_next,
                            float sqrt_recip, float sqrt_recipm1, float sqrt_ab, float sqrt_1mab,
                            float sqrt_abar_next, float c, float sigma, float lo, float hi, int objective,
                            int last, int64_t n, void* stream) {
  LD_REQUIRE(x_t && model_out && x_next && n > 0, "ld_ddim_step: null pointer");
  DdimK k{sqrt_recip, sqrt_recipm1, sqrt_ab, sqrt_1mab, sqrt_abar_next, c, sigma, lo, hi, objective, last};
  LD_LAUNCH(ddim_step_kernel, dim3(nblocks(n)), dim3(BS), 0, ST(stream), x_t, model_out, noise, x_next, k, (long)n);
  LD_LAUNCH_CHECK("ddim_step");
  return LD_OK;
}
extern "C" int ld_ddim_step_at(const float* x_t, const float* model_out, const float* noise, float* x_next,
                               const float* pair_table, const int32_t* idx_ptr, float lo, float hi, int objective,
                               int64_t n, void* stream) {
  LD_REQUIRE(x_t && model_out && x_next && pair_table && idx_ptr && n > 0, "ld_ddim_step_at: null pointer");
  LD_REQUIRE(objective >= 0 && objective <= 2, "ld_ddim_step_at: objective %d", objective);
  LD_LAUNCH(ddim_step_at_kernel, dim3(nblocks(n)), dim3(BS), 0, ST(stream), x_t, model_out, noise, x_next, pair_table, idx_ptr,
            lo, hi, objective, (long)n);
  LD_LAUNCH_CHECK("ddim_step_at");
  return LD_OK;
}
extern "C" int ld_branch_conditions(const float* cond, const float* mask, float* cond_out, float* cond_in,
                                    float lo_clip, int B, int C, int HW, void* stream) {
  LD_REQUIRE(cond && mask && cond_out && cond_in, "ld_branch_conditions: null pointer");
  const long n = (long)B * C * HW;
  LD_LAUNCH(branch_cond_kernel, dim3(nblocks(n)), dim3(BS), 0, ST(stream), cond, mask, cond_out, cond_in, lo_clip, C, HW, n);
  LD_LAUNCH_CHECK("branch_conditions");
  return LD_OK;
}
extern "C" int ld_mask_out(float* model_out, const float* mask, float min_val, int B, int C, int HW, void* stream) {
  LD_REQUIRE(model_out && mask, "ld_mask_out: null pointer");
  const long n = (long)B * C * HW;
  LD_LAUNCH(mask_out_kernel, dim3(nblocks(n)), dim3(BS), 0, ST(stream), model_out, mask, min_val, C, HW, n);
  LD_LAUNCH_CHECK("mask_out");
  return LD_OK;
}
extern "C" int ld_fuse_ddpm(const float* x_out, const float* x_in, const float* x0_out, const float* x0_in,
                            const float* mask, float* x, float* x0, float lo, float hi, int B, int C, int HW,
                            void* stream) {
  LD_REQUIRE(x_out && x_in && x0_out && x0_in && mask && x && x0, "ld_fuse_ddpm: null pointer");
  const long n = (long)B * C * HW;
  LD_LAUNCH(fuse_ddpm_kernel, dim3(nblocks(n)), dim3(BS), 0, ST(stream), x_out, x_in, x0_out, x0_in, mask, x, x0, lo, hi, C, HW, n);
  LD_LAUNCH_CHECK("fuse_ddpm");
  return LD_OK;
}
extern "C" int ld_branch_conditions_k(const float* cond, const float* masks, float* cond_k, float lo_clip, int B, int C,
                                      int K, int HW, void* stream) {
  LD_REQUIRE(cond && masks && cond_k && K >= 2, "ld_branch_conditions_k: bad args (K >= 2)");
  const long n = (long)B * C * HW;
  LD_LAUNCH(branch_cond_k_kernel, dim3(nblocks(n)), dim3(BS), 0, ST(stream), cond, masks, cond_k, lo_clip, B, C, K, HW, n);
  LD_LAUNCH_CHECK("branch_conditions_k");
  return LD_OK;
}
extern "C" int ld_fuse_ddpm_k(const float* x_first, const float* x_rest, const float* x0_first, const float* x0_rest,
                              const float* masks, float* x, float* x0, float lo, float hi, int B, int C, int K, int HW,
                              void* stream) {
  LD_REQUIRE(x_first && x_rest && x0_first && x0_rest && masks && x && x0 && K >= 2, "ld_fuse_ddpm_k: bad args (K >= 2)");
  const long n = (long)B * C * HW;
  LD_LAUNCH(fuse_ddpm_k_kernel, dim3(nblocks(n)), dim3(BS), 0, ST(stream), x_first, x_rest, x0_first, x0_rest, masks, x, x0,
            lo, hi, C, K, HW, n);
  LD_LAUNCH_CHECK("fuse_ddpm_k");
  return LD_OK;
}
extern "C" int ld_fuse_ddim(const float* x_out, const float* x_in, const float* x0_out, const float* x0_in,
                            const float* mask, const float* noise, float* x_next, float sqrt_recip,
                            float sqrt_recipm1, float sqrt_abar_next, float c, float sigma, float lo, float hi,
                            int B, int C, int HW, void* stream) {
  LD_REQUIRE(x_out && x_in && x0_out && x0_in && mask && x_next, "ld_fuse_ddim: null pointer");
  const long n = (long)B * C * HW;
  FuseDdimK k{sqrt_recip, sqrt_recipm1, sqrt_abar_next, c, sigma, lo, hi};
  LD_LAUNCH(fuse_ddim_kernel, dim3(nblocks(n)), dim3(BS), 0, ST(stream), x_out, x_in, x0_out, x0_in, mask, noise, x_next, k, C, HW, n);
  LD_LAUNCH_CHECK("fuse_ddim");
  return LD_OK;
}
extern "C" int ld_fuse_ddim_k(const float* x_first, const float* x_rest, const float* x0_first, const float* x0_rest,
                              const float* masks, const float* noise, float* x_next, float sqrt_recip, float sqrt_recipm1,
                              float sqrt_abar_next, float c, float sigma, float lo, float hi, int B, int C, int K, int HW,
                              void* stream) {
  LD_REQUIRE(x_first && x_rest && x0_first && x0_rest && masks && x_next && K >= 2, "ld_fuse_ddim_k: bad args (K >= 2)");
  const long n = (long)B * C * HW;
  FuseDdimK k{sqrt_recip, sqrt_recipm1, sqrt_abar_next, c, sigma, lo, hi};
  LD_LAUNCH(fuse_ddim_k_kernel, dim3(nblocks(n)), dim3(BS), 0, ST(stream), x_first, x_rest, x0_first, x0_rest, masks, noise,
            x_next, k, C, K, HW, n);
  LD_LAUNCH_CHECK("fuse_ddim_k");
  return LD_OK;
}
extern "C" int ld_q_sample(const float* x0, const float* noise, float* out, float sqrt_ab, float sqrt_1mab,
                           int64_t n, void* stream) {
  LD_REQUIRE(x0 && noise && out && n > 0, "ld_q_sample: null pointer");
  LD_LAUNCH(q_sample_kernel, dim3(nblocks(n)), dim3(BS), 0, ST(stream), x0, noise, out, sqrt_ab, sqrt_1mab, (long)n);
  LD_LAUNCH_CHECK("q_sample");
  return LD_OK;
}
extern "C" int ld_q_sample_t(const float* x0, const float* noise, float* out, const int* t, const float* sqrt_ab,
                             const float* sqrt_1mab, int B, int64_t elems_per_sample, void* stream) {
  LD_REQUIRE(x0 && noise && out && t && sqrt_ab && sqrt_1mab && B > 0 && elems_per_sample > 0, "ld_q_sample_t: bad args");
  long bx = (elems_per_sample + BS - 1) / BS;
  if (bx > 1024) bx = 1024;
  LD_LAUNCH(q_sample_t_kernel, dim3((unsigned)bx, B), dim3(BS), 0, ST(stream), x0, noise, out, t, sqrt_ab, sqrt_1mab, (long)elems_per_sample);
  LD_LAUNCH_CHECK("q_sample_t");
  return LD_OK;
}
extern "C" int ld_p_losses(const float* model_out, const float* x_start, const float* noise, const int* t, const float* sqrt_ab,
                           const float* sqrt_1mab, const float* loss_weight, float* loss_out, int B, int64_t elems_per_sample,
                           int objective, void* stream) {
  LD_REQUIRE(model_out && x_start && noise && t && sqrt_ab && sqrt_1mab && loss_weight && loss_out, "ld_p_losses: null pointer");
  LD_REQUIRE(B > 0 && elems_per_sample > 0, "ld_p_losses: empty batch");
  LD_REQUIRE(objective == LD_OBJ_X0 || objective == LD_OBJ_NOISE || objective == LD_OBJ_V, "ld_p_losses: objective %d", objective);
  LD_LAUNCH(p_losses_kernel, dim3(B), dim3(256), 0, ST(stream), model_out, x_start, noise, t, sqrt_ab, sqrt_1mab, loss_weight, loss_out,
            (long)elems_per_sample, objective);
  LD_LAUNCH_CHECK("p_losses");
  return LD_OK;
}
extern "C" int ld_recompose(const float* patches, const float* masks, float* out, int B, int K, int C, int HW,
                            void* stream) {
  LD_REQUIRE(patches && masks && out && K > 0, "ld_recompose: bad args");
  const long n = (long)B * C * HW;
  LD_LAUNCH(recompose_kernel, dim3(nblocks(n)), dim3(BS), 0, ST(stream), patches, masks, out, K, C, HW, n);
  LD_LAUNCH_CHECK("recompose");
  return LD_OK;
}
extern "C" int ld_final_conv(const void* x, const float* w, const float* b, float* out_nchw, int B, int H, int W,
                             int Cin, int Cout, int dtype, void* stream) {
  LD_REQUIRE(x && w && b && out_nchw, "ld_final_conv: null pointer");
  LD_REQUIRE(Cout >= 1 && Cout <= 4 && Cin % 8 == 0, "ld_final_conv: Cout %d (1..4), Cin %d (multiple of 8)", Cout, Cin);
  const long npix = (long)B * H * W;
  LD_REQUIRE(ld_dtype_ok(dtype), "ld_final_conv: bad dtype %d", dtype);
  LD_DISPATCH(dtype, [&] {
    LD_LAUNCH(final_conv_kernel<T>, dim3(nblocks(npix)), dim3(BS), Cin * Cout * sizeof(float), ST(stream), (const T*)x, w, b, out_nchw, H * W, Cin, Cout, npix);
    return 0;
  }());
  LD_LAUNCH_CHECK("final_conv");
  return LD_OK;
}

extern "C" int ld_final_step_at(const void* x, const float* w, const float* b, float* model_out, float* x_t,
                                float* x0_out, const float* sched, const int32_t* t_ptr, float lo, float hi,
                                int objective, uint64_t seed, int64_t noise_base, int64_t noise_tmul,
                                int64_t noise_first, const float* keep_mask, int B, int H, int W, int Cin, int Cout,
                                int dtype, void* stream) {
  LD_REQUIRE(x && w && b && model_out && x_t && sched, "ld_final_step: null pointer");
  LD_REQUIRE(noise_first >= 0 && (noise_tmul == 0 || t_ptr), "ld_final_step: noise_first %ld, noise_tmul without t_ptr",
             (long)noise_first);
  LD_REQUIRE(Cout >= 1 && Cout <= 4 && Cin % 8 == 0, "ld_final_step: Cout %d (1..4), Cin %d (multiple of 8)", Cout, Cin);
  LD_REQUIRE(objective >= 0 && objective <= 2, "ld_final_step: objective %d", objective);
  LD_REQUIRE(ld_dtype_ok(dtype), "ld_final_step: bad dtype %d", dtype);
  const long npix = (long)B * H * W;
  const size_t lds = (size_t)Cin * Cout * sizeof(float);
  LD_DISPATCH(dtype, [&] {
    LD_LAUNCH(final_step_kernel<T>, dim3(nblocks(npix)), dim3(BS), lds, ST(stream), (const T*)x, w, b, model_out,
              x_t, x0_out, sched, t_ptr, lo, hi, objective, (unsigned long long)seed, (long)noise_base, (long)noise_tmul,
              (long)noise_first, keep_mask, H * W, Cin, Cout, npix);
    return 0;
  }());
  LD_LAUNCH_CHECK("final_step");
  return LD_OK;
}
extern "C" int ld_final_step(const void* x, const float* w, const float* b, float* model_out, float* x_t, float* x0_out,
                             const float* sched, const int32_t* t_ptr, float lo, float hi, int objective, uint64_t seed,
                             int64_t noise_stream, int B, int H, int W, int Cin, int Cout, int dtype, void* stream) {
  return ld_final_step_at(x, w, b, model_out, x_t, x0_out, sched, t_ptr, lo, hi, objective, seed, noise_stream, 0, 0,
                          nullptr, B, H, W, Cin, Cout, dtype, stream);
}
