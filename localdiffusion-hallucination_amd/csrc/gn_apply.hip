// GroupNorm-apply (+FiLM) + activation + second operand + final activation (+ 2x2 max-pool).
//
//   ResnetBlock tail (ddpm.py:210-212):      out = SiLU(GN(conv2_raw)) + res          (b raw)
//   BasicBlock tail (unet_model.py:38-51):   out = ReLU(GN(conv2_raw) + GN(id_raw))    (b normalised)
//   followed by nn.MaxPool2d(2) in the conditioning encoder (unet_model.py:120,123,129) -> pool=1
//
// Pure HBM streaming: 16 B per lane, coefficient tables (2*C floats per operand) built per block
// from the fp64 statistics.  grid = (pixel blocks, B).
#include "common.hip.h"

namespace {
struct GnDev {
  SrcDev a, b;
  int has_b, final_act, pool;
  void* out;
  int B, H, W;
  const int* t_ptr;
};

// HAS_B / POOL are template parameters and, when the fragments-per-pixel count divides the block size, every
// thread owns ONE channel fragment for the whole launch: its coefficients live in registers and the pixel index
// advances by a constant -- no 64-bit division, no LDS coefficient reads and no feature branches in the
// streaming loop (measured on the generic loop: 8-10 us for 8 MB launches whose traffic is worth 3 us).
template <typename T, bool HAS_B>
__device__ __forceinline__ void eval_pixel(const GnDev& g, const float* ca, const float* sa, const float* cb, const float* sb,
                                           size_t elem, float* v) {
  constexpr int E = DT<T>::E;
  constexpr bool P = DT<T>::precise;
  const T* xa = reinterpret_cast<const T*>(g.a.data);
  uint4 ra = *reinterpret_cast<const uint4*>(xa + elem);
  uint4 rb = make_uint4(0u, 0u, 0u, 0u);
  if (HAS_B) rb = *reinterpret_cast<const uint4*>(reinterpret_cast<const T*>(g.b.data) + elem);
  unpack16<T>(ra, v);
  affine_act_n<P, E>(v, ca, sa, g.a.act);
  if (HAS_B) {
    float u[E];
    unpack16<T>(rb, u);
    if (g.b.stats) affine_act_n<P, E>(u, cb, sb, g.b.act);
#pragma unroll
    for (int e = 0; e < E; ++e) v[e] += u[e];
  }
  act_n<P, E>(v, g.final_act);
}

template <typename T, bool HAS_B, bool POOL>
__global__ __launch_bounds__(256) void gn_apply_kernel(GnDev g) {
  constexpr int E = DT<T>::E;
  extern __shared__ __attribute__((aligned(16))) float s_coef[];   // [2C] for a, [2C] for b
  const int C = g.a.C, b = blockIdx.y, tid = threadIdx.x;
  const int trow = g.t_ptr ? *g.t_ptr : 0;
  const long npix_in = (long)g.H * g.W;
  double* red = reinterpret_cast<double*>(s_coef + 4 * C);
  build_gn_coef(g.a, b, trow, npix_in, s_coef, red, tid, 256);
  if (HAS_B && g.b.stats) build_gn_coef(g.b, b, trow, npix_in, s_coef + 2 * C, red, tid, 256);
  const int fpp = C / E;                               // fragments per pixel
  const int Ho = POOL ? g.H / 2 : g.H, Wo = POOL ? g.W / 2 : g.W;
  const int npix_out = Ho * Wo;
  T* out = reinterpret_cast<T*>(g.out) + (size_t)b * npix_out * C;
  const size_t in0 = (size_t)b * npix_in * C;
  auto one = [&](int opix, int c, const float* ca, const float* sa, const float* cb, const float* sb) {
    float v[E];
    if (!POOL) {
      eval_pixel<T, HAS_B>(g, ca, sa, cb, sb, in0 + (size_t)opix * C + c, v);
    } else {
      const int oy = opix / Wo, ox = opix - oy * Wo;
      float u[E];
      eval_pixel<T, HAS_B>(g, ca, sa, cb, sb, in0 + ((size_t)(2 * oy) * g.W + 2 * ox) * C + c, v);
#pragma unroll
      for (int k = 1; k < 4; ++k) {
        eval_pixel<T, HAS_B>(g, ca, sa, cb, sb, in0 + ((size_t)(2 * oy + (k >> 1)) * g.W + 2 * ox + (k & 1)) * C + c, u);
#pragma unroll
        for (int e = 0; e < E; ++e) v[e] = fmaxf(v[e], u[e]);
      }
    }
    *reinterpret_cast<uint4*>(out + (size_t)opix * C + c) = pack16<T>(v);
  };
  if (256 % fpp == 0) {
    const int c = (tid % fpp) * E, ppb = 256 / fpp;    // pixels per block-iteration
    float ca[E], sa[E], cb[E], sb[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
      ca[e] = s_coef[c + e]; sa[e] = s_coef[C + c + e];
      cb[e] = HAS_B ? s_coef[2 * C + c + e] : 0.f; sb[e] = HAS_B ? s_coef[3 * C + c + e] : 0.f;
    }
    const int step = gridDim.x * ppb;
    int opix = blockIdx.x * ppb + tid / fpp;
    // two pixels per iteration: both fragments' loads are in flight before the first is consumed
    for (; opix + step < npix_out; opix += 2 * step) {
      one(opix, c, ca, sa, cb, sb);
      one(opix + step, c, ca, sa, cb, sb);
    }
    if (opix < npix_out) one(opix, c, ca, sa, cb, sb);
  } else {
    const int nfrag = npix_out * fpp;
    for (int f = blockIdx.x * 256 + tid; f < nfrag; f += gridDim.x * 256) {
      const int opix = f / fpp, c = (f - opix * fpp) * E;
      one(opix, c, s_coef + c, s_coef + C + c, s_coef + 2 * C + c, s_coef + 3 * C + c);
    }
  }
}

template <typename T>
int run(const GnDev& g, hipStream_t st) {
  const int E = DT<T>::E;
  const int Ho = g.pool ? g.H / 2 : g.H, Wo = g.pool ? g.W / 2 : g.W;
  const long nfrag = (long)Ho * Wo * (g.a.C / E);
  LD_REQUIRE(nfrag < (1L << 31) / 16, "ld_gn_apply: image too large for 32-bit fragment indices");
  long blocks = (nfrag + 511) / 512;                     // two fragments per thread and iteration
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  dim3 grid((unsigned)blocks, g.B);
  const size_t lds = 4 * g.a.C * sizeof(float) + 64 * sizeof(double);
  if (g.has_b) {
    if (g.pool) LD_LAUNCH((gn_apply_kernel<T, true, true>), grid, dim3(256), lds, st, g);
    else LD_LAUNCH((gn_apply_kernel<T, true, false>), grid, dim3(256), lds, st, g);
  } else {
    if (g.pool) LD_LAUNCH((gn_apply_kernel<T, false, true>), grid, dim3(256), lds, st, g);
    else LD_LAUNCH((gn_apply_kernel<T, false, false>), grid, dim3(256), lds, st, g);
  }
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
  LD_REQUIRE(ld_dtype_ok(p->dtype), "ld_gn_apply: bad dtype %d", p->dtype);
  return LD_DISPATCH(p->dtype, run<T>(g, st));
}
