// GroupNorm-apply (+FiLM) + activation + second operand + final activation (+ 2x2 max-pool).
//
//   ResnetBlock tail (ddpm.py:210-212):      out = SiLU(GN(conv2_raw)) + res          (b raw)
//   BasicBlock tail (unet_model.py:38-51):   out = ReLU(GN(conv2_raw) + GN(id_raw))    (b normalised)
//   followed by nn.MaxPool2d(2) in the conditioning encoder (unet_model.py:120,123,129) -> pool=1
//
// Pure HBM streaming: 16 B per lane, coefficient tables (2*C floats per operand) built per block
// from the fp64 statistics.  grid = (pixel blocks, B).
#include "common.cuh"

namespace {
struct GnDev {
  SrcDev a, b;
  int has_b, final_act, pool;
  void* out;
  int B, H, W;
  const int* t_ptr;
};

template <typename T>
__device__ __forceinline__ void eval_pixel(const GnDev& g, const float* coefA, const float* coefB, int bidx,
                                           size_t pix, int c, float* v) {
  constexpr int E = DT<T>::E;
  constexpr bool P = DT<T>::precise;
  const int C = g.a.C;
  const T* xa = reinterpret_cast<const T*>(g.a.data);
  uint4 ra = *reinterpret_cast<const uint4*>(xa + pix * C + c);
  unpack16<T>(ra, v);
  affine_act_n<P, E>(v, coefA + c, coefA + C + c, g.a.act);
  if (g.has_b) {
    const T* xb = reinterpret_cast<const T*>(g.b.data);
    uint4 rb = *reinterpret_cast<const uint4*>(xb + pix * C + c);
    float u[E];
    unpack16<T>(rb, u);
    if (g.b.stats) affine_act_n<P, E>(u, coefB + c, coefB + C + c, g.b.act);
#pragma unroll
    for (int e = 0; e < E; ++e) v[e] += u[e];
  }
  act_n<P, E>(v, g.final_act);
}

template <typename T>
__global__ __launch_bounds__(256) void gn_apply_kernel(GnDev g) {
  constexpr int E = DT<T>::E;
  extern __shared__ __attribute__((aligned(16))) float s_coef[];   // [2C] for a, [2C] for b
  const int C = g.a.C, b = blockIdx.y, tid = threadIdx.x;
  const int trow = g.t_ptr ? *g.t_ptr : 0;
  const long npix_in = (long)g.H * g.W;
  double* red = reinterpret_cast<double*>(s_coef + 4 * C);
  build_gn_coef(g.a, b, trow, npix_in, s_coef, red, tid, 256);
  if (g.has_b && g.b.stats) build_gn_coef(g.b, b, trow, npix_in, s_coef + 2 * C, red, tid, 256);
  const int fpp = C / E;                               // fragments per pixel
  const int Ho = g.pool ? g.H / 2 : g.H, Wo = g.pool ? g.W / 2 : g.W;
  const long nfrag = (long)Ho * Wo * fpp;
  T* out = reinterpret_cast<T*>(g.out);
  for (long f = (long)blockIdx.x * 256 + tid; f < nfrag; f += (long)gridDim.x * 256) {
    const long opix = f / fpp;
    const int c = (int)(f - opix * fpp) * E;
    float v[E];
    if (!g.pool) {
      eval_pixel<T>(g, s_coef, s_coef + 2 * C, b, (size_t)b * npix_in + opix, c, v);
    } else {
      const int oy = (int)(opix / Wo), ox = (int)(opix - (long)oy * Wo);
      float u[E];
      eval_pixel<T>(g, s_coef, s_coef + 2 * C, b, (size_t)b * npix_in + (size_t)(2 * oy) * g.W + 2 * ox, c, v);
#pragma unroll
      for (int k = 1; k < 4; ++k) {
        eval_pixel<T>(g, s_coef, s_coef + 2 * C, b,
                      (size_t)b * npix_in + (size_t)(2 * oy + (k >> 1)) * g.W + 2 * ox + (k & 1), c, u);
#pragma unroll
        for (int e = 0; e < E; ++e) v[e] = fmaxf(v[e], u[e]);
      }
    }
    *reinterpret_cast<uint4*>(out + ((size_t)b * Ho * Wo + opix) * C + c) = pack16<T>(v);
  }
}

template <typename T>
int run(const GnDev& g, hipStream_t st) {
  const int E = DT<T>::E;
  const int Ho = g.pool ? g.H / 2 : g.H, Wo = g.pool ? g.W / 2 : g.W;
  const long nfrag = (long)Ho * Wo * (g.a.C / E);
  long blocks = (nfrag + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  dim3 grid((unsigned)blocks, g.B);
  const size_t lds = 4 * g.a.C * sizeof(float) + 64 * sizeof(double);
  hipLaunchKernelGGL((gn_apply_kernel<T>), grid, dim3(256), lds, st, g);
  LD_LAUNCH_CHECK("gn_apply");
  return LD_OK;
}
}  // namespace

extern "C" int ld_gn_apply(const ld_gn_apply_args* p, void* stream) {
  LD_REQUIRE(p && p->a.data && p->out, "ld_gn_apply: null pointer");
  LD_REQUIRE(p->a.gn_stats && p->a.gn_gamma && p->a.gn_beta && p->a.gn_groups > 0, "ld_gn_apply: operand a needs GroupNorm data");
  LD_REQUIRE(p->a.C % 32 == 0 && p->a.C % p->a.gn_groups == 0, "ld_gn_apply: C=%d groups=%d", p->a.C, p->a.gn_groups);
  LD_REQUIRE(p->a.pix_stride == 0 || p->a.pix_stride == p->a.C, "ld_gn_apply: operands must be dense");
  LD_REQUIRE(!p->pool || (p->H % 2 == 0 && p->W % 2 == 0), "ld_gn_apply: pool needs even H,W");
  GnDev g;
  g.a = to_dev(p->a);
  g.has_b = p->b.data != nullptr;
  if (g.has_b) {
    LD_REQUIRE(p->b.C == p->a.C, "ld_gn_apply: operand channel mismatch");
    g.b = to_dev(p->b);
  } else {
    g.b = g.a;
  }
  g.final_act = p->final_act; g.pool = p->pool; g.out = p->out;
  g.B = p->B; g.H = p->H; g.W = p->W; g.t_ptr = p->t_ptr;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (p->dtype == LD_F32) return run<float>(g, st);
  if (p->dtype == LD_BF16) return run<bf16>(g, st);
  return ld_fail(LD_EINVAL, "ld_gn_apply: bad dtype %d", p->dtype);
}
