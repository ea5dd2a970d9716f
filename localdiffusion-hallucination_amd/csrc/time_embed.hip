// Timestep embedding: SinusoidalPosEmb (ddpm.py:136-149) -> Linear -> GELU(erf) -> Linear
// (:339-344), and the per-ResnetBlock FiLM projection SiLU -> Linear(time_dim, 2*C) (:191-206).
// Both depend only on t, so the host evaluates them ONCE for all T timesteps into tables
// ([T, time_dim] and [T, 2C] per block) and the per-step kernels index the table through t_ptr.
// One workgroup per timestep; tiny GEMVs, fp32 throughout.
#include "common.hip.h"

namespace {
// FOURIER = false: SinusoidalPosEmb, emb = [sin(t f_k) | cos(t f_k)], `dim` wide (ddpm.py:136-149).
// FOURIER = true: RandomOrLearnedSinusoidalPosEmb (ddpm.py:151-165), emb = [t | sin(2 pi w_k t) | cos(2 pi w_k t)], `dim` =
// learned_sinusoidal_dim + 1 wide, `freqs` = the module's `weights`; the angle is ((t * w) * 2) * pi in fp32, the reference's order.
template <bool FOURIER>
__global__ void time_mlp_kernel(const int* __restrict__ times, const float* __restrict__ freqs, int dim,
                                const float* __restrict__ w1, const float* __restrict__ b1,
                                const float* __restrict__ w2, const float* __restrict__ b2, int td,
                                float* __restrict__ temb) {
  extern __shared__ float sm[];            // emb[dim] | h[td]
  float* emb = sm;
  float* hbuf = sm + dim;
  const int i = blockIdx.x, tid = threadIdx.x, half = dim / 2;      // (FOURIER: dim is odd, half = learned_sinusoidal_dim / 2)
  const float t = (float)times[i];
  if (tid < half) {
    if (FOURIER) {
      const float ang = ((t * freqs[tid]) * 2.0f) * 3.14159265358979323846f;
      emb[1 + tid] = sinf(ang);
      emb[1 + half + tid] = cosf(ang);
      if (tid == 0) emb[0] = t;
    } else {
      const float ang = t * freqs[tid];
      emb[tid] = sinf(ang);
      emb[half + tid] = cosf(ang);
    }
  }
  __syncthreads();
  for (int o = tid; o < td; o += blockDim.x) {
    float acc = b1[o];
    for (int k = 0; k < dim; ++k) acc = fmaf(w1[o * dim + k], emb[k], acc);
    hbuf[o] = 0.5f * acc * (1.0f + erff(acc * 0.70710678118654752440f));    // exact GELU
  }
  __syncthreads();
  for (int o = tid; o < td; o += blockDim.x) {
    float acc = b2[o];
    for (int k = 0; k < td; ++k) acc = fmaf(w2[o * td + k], hbuf[k], acc);
    temb[(size_t)i * td + o] = acc;
  }
}

__global__ void film_kernel(const float* __restrict__ temb, int td, const float* __restrict__ w,
                            const float* __restrict__ b, int two_c, float* __restrict__ film) {
  extern __shared__ float act[];           // SiLU(temb[i])
  const int i = blockIdx.x, tid = threadIdx.x;
  for (int k = tid; k < td; k += blockDim.x) {
    const float v = temb[(size_t)i * td + k];
    act[k] = v / (1.0f + expf(-v));
  }
  __syncthreads();
  for (int o = tid; o < two_c; o += blockDim.x) {
    float acc = b[o];
    for (int k = 0; k < td; ++k) acc = fmaf(w[(size_t)o * td + k], act[k], acc);
    film[(size_t)i * two_c + o] = acc;
  }
}
}  // namespace

extern "C" int ld_time_mlp(const int32_t* times, int n, const float* freqs, int dim, const float* w1,
                           const float* b1, const float* w2, const float* b2, int time_dim, float* temb,
                           void* stream) {
  LD_REQUIRE(times && freqs && w1 && b1 && w2 && b2 && temb && n > 0, "ld_time_mlp: bad args");
  LD_REQUIRE(dim % 2 == 0 && dim / 2 <= 128, "ld_time_mlp: dim %d", dim);
  LD_LAUNCH(time_mlp_kernel<false>, dim3(n), dim3(128), (dim + time_dim) * sizeof(float),
                     reinterpret_cast<hipStream_t>(stream), times, freqs, dim, w1, b1, w2, b2, time_dim, temb);
  LD_LAUNCH_CHECK("time_mlp");
  return LD_OK;
}

extern "C" int ld_time_mlp_fourier(const int32_t* times, int n, const float* weights, int learned_dim, const float* w1,
                                   const float* b1, const float* w2, const float* b2, int time_dim, float* temb,
                                   void* stream) {
  LD_REQUIRE(times && weights && w1 && b1 && w2 && b2 && temb && n > 0, "ld_time_mlp_fourier: bad args");
  LD_REQUIRE(learned_dim % 2 == 0 && learned_dim >= 2 && learned_dim / 2 <= 128, "ld_time_mlp_fourier: learned_sinusoidal_dim %d", learned_dim);
  const int dim = learned_dim + 1;
  LD_LAUNCH(time_mlp_kernel<true>, dim3(n), dim3(128), (dim + time_dim) * sizeof(float),
                     reinterpret_cast<hipStream_t>(stream), times, weights, dim, w1, b1, w2, b2, time_dim, temb);
  LD_LAUNCH_CHECK("time_mlp_fourier");
  return LD_OK;
}

extern "C" int ld_film(const float* temb, int n, int time_dim, const float* w, const float* b, int two_c,
                       float* film, void* stream) {
  LD_REQUIRE(temb && w && b && film && n > 0, "ld_film: bad args");
  LD_LAUNCH(film_kernel, dim3(n), dim3(128), time_dim * sizeof(float),
                     reinterpret_cast<hipStream_t>(stream), temb, time_dim, w, b, two_c, film);
  LD_LAUNCH_CHECK("film");
  return LD_OK;
}
