// Device side of gn_apply (gn_apply.hip has the description and the C entry point).  Included by gn_apply.hip (and by the archived tools/experiments/stage_programs.hip).
#pragma once
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
__device__ __forceinline__ void load_pixel(const GnDev& g, size_t elem, uint4& ra, uint4& rb) {
  ra = *reinterpret_cast<const uint4*>(reinterpret_cast<const T*>(g.a.data) + elem);
  rb = make_uint4(0u, 0u, 0u, 0u);
  if (HAS_B) rb = *reinterpret_cast<const uint4*>(reinterpret_cast<const T*>(g.b.data) + elem);
}
template <typename T, bool HAS_B>
__device__ __forceinline__ void finish_pixel(const GnDev& g, const float* ca, const float* sa, const float* cb, const float* sb,
                                             const uint4& ra, const uint4& rb, float* v) {
  constexpr int E = DT<T>::E;
  constexpr bool P = DT<T>::precise;
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
template <typename T, bool HAS_B>
__device__ __forceinline__ void eval_pixel(const GnDev& g, const float* ca, const float* sa, const float* cb, const float* sb,
                                           size_t elem, float* v) {
  uint4 ra, rb;
  load_pixel<T, HAS_B>(g, elem, ra, rb);
  finish_pixel<T, HAS_B>(g, ca, sa, cb, sb, ra, rb, v);
}

// The kernel body as a device function of the (virtual) workgroup index (gn_apply_kernel: its own index; tools/experiments/stage_programs.hip: the
// blocks a persistent workgroup takes from its work counter).  bx / gdx: block and number of blocks of image `b`.
typedef const GnDev __attribute__((address_space(4)))* GnKernargPtr;   // the block in kernel-argument (constant) memory
// `rest` (gn_apply_lead_kernel): `head` holds only what the head's requests need (preloaded kernel arguments); the
// activation kinds and the output pointer are read from the block BEHIND those requests (finding 83).
template <typename T, bool HAS_B, bool POOL>
__device__ __forceinline__ void gn_apply_tile(const GnDev& head, const int bx, const int gdx, const int b, float* s_coef,
                                              GnKernargPtr rest = nullptr) {
  constexpr int E = DT<T>::E;
  GnDev g = head;
  const int C = g.a.C, tid = threadIdx.x;
  const int trow = g.t_ptr ? *g.t_ptr : 0;
  const long npix_in = (long)g.H * g.W;
  double* red = reinterpret_cast<double*>(s_coef + 4 * C);
  const int fpp = C / E;                               // fragments per pixel
  const int Ho = POOL ? g.H / 2 : g.H, Wo = POOL ? g.W / 2 : g.W;
  const int npix_out = Ho * Wo;
  const size_t in0 = (size_t)b * npix_in * C;
  // The thread's first pixel pair is requested BEFORE the coefficients are built: the data does not depend on them,
  // and behind them it was a second dependent round trip in a launch that is two round trips long (nearly every
  // launch is one pair per thread).  Branch-free (clamped addresses, native vectors): a conditional load costs a wait.
  u32x4 pa0, pa1, pb0, pb1;
  {
    const int fq = fpp <= 256 ? fpp : 256, cq = (tid % fq) * E, ppq = 256 / fq;
    const int o0 = bx * ppq + tid / fq, o1 = o0 + gdx * ppq;
    const size_t e0 = in0 + (size_t)(o0 < npix_out ? o0 : npix_out - 1) * C + cq;
    const size_t e1 = in0 + (size_t)(o1 < npix_out ? o1 : npix_out - 1) * C + cq;
    pa0 = *reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(g.a.data) + e0);
    pa1 = *reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(g.a.data) + e1);
    pb0 = pa0; pb1 = pa1;
    if (HAS_B) {
      pb0 = *reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(g.b.data) + e0);
      pb1 = *reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(g.b.data) + e1);
    }
  }
  build_gn_coef<DT<T>::precise>(g.a, b, trow, npix_in, s_coef, red, tid, 256);
  if (HAS_B && g.b.stats) build_gn_coef<DT<T>::precise>(g.b, b, trow, npix_in, s_coef + 2 * C, red, tid, 256);
  if (rest) {
    GnKernargPtr pr = rest;
    asm volatile("" : "+s"(pr));
    g.a.act = pr->a.act; g.b.act = pr->b.act; g.final_act = pr->final_act; g.out = pr->out;
  }
  T* out = reinterpret_cast<T*>(g.out) + (size_t)b * npix_out * C;
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
    store16_out(out + (size_t)opix * C + c, pack16<T>(v));
  };
  if (256 % fpp == 0) {
    const int c = (tid % fpp) * E, ppb = 256 / fpp;    // pixels per block-iteration
    float ca[E], sa[E], cb[E], sb[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
      ca[e] = s_coef[c + e]; sa[e] = s_coef[C + c + e];
      cb[e] = HAS_B ? s_coef[2 * C + c + e] : 0.f; sb[e] = HAS_B ? s_coef[3 * C + c + e] : 0.f;
    }
    const int step = gdx * ppb;
    int opix = bx * ppb + tid / fpp;
    // two pixels per iteration, every load of the pair requested before the first store: written as two calls of
    // one(), hipcc kept the second pixel's loads BEHIND the first pixel's store (it cannot prove that `out` does not
    // alias the inputs), and the wait for them then also drains that store -- the vector-memory counter is in order and
    // counts stores (tools/scan_store_waits.py)
    bool first = true;
    for (; opix + step < npix_out; opix += 2 * step) {
      if constexpr (!POOL) {
        uint4 ra0, rb0, ra1, rb1;
        if (first) {
          ra0 = make_uint4(pa0[0], pa0[1], pa0[2], pa0[3]); rb0 = make_uint4(pb0[0], pb0[1], pb0[2], pb0[3]);
          ra1 = make_uint4(pa1[0], pa1[1], pa1[2], pa1[3]); rb1 = make_uint4(pb1[0], pb1[1], pb1[2], pb1[3]);
        } else {
          load_pixel<T, HAS_B>(g, in0 + (size_t)opix * C + c, ra0, rb0);
          load_pixel<T, HAS_B>(g, in0 + (size_t)(opix + step) * C + c, ra1, rb1);
        }
        first = false;
        float v0[E], v1[E];
        finish_pixel<T, HAS_B>(g, ca, sa, cb, sb, ra0, rb0, v0);
        finish_pixel<T, HAS_B>(g, ca, sa, cb, sb, ra1, rb1, v1);
        store16_out(out + (size_t)opix * C + c, pack16<T>(v0));
        store16_out(out + (size_t)(opix + step) * C + c, pack16<T>(v1));
      } else {
        one(opix, c, ca, sa, cb, sb);
        one(opix + step, c, ca, sa, cb, sb);
      }
    }
    if (opix < npix_out) {
      if (!POOL && first) {                            // a single pixel, already here
        float v0[E];
        finish_pixel<T, HAS_B>(g, ca, sa, cb, sb, make_uint4(pa0[0], pa0[1], pa0[2], pa0[3]), make_uint4(pb0[0], pb0[1], pb0[2], pb0[3]), v0);
        store16_out(out + (size_t)opix * C + c, pack16<T>(v0));
      } else {
        one(opix, c, ca, sa, cb, sb);
      }
    }
  } else {
    const int nfrag = npix_out * fpp;
    for (int f = bx * 256 + tid; f < nfrag; f += gdx * 256) {
      const int opix = f / fpp, c = (f - opix * fpp) * E;
      one(opix, c, s_coef + c, s_coef + C + c, s_coef + 2 * C + c, s_coef + 3 * C + c);
    }
  }
}

// The launches of the sampling loop (no pooling, one normalised operand, FiLM shared by the batch, no step counter):
// 14 leading scalar dwords -- both operands, the GroupNorm tables, C | groups << 10 | gridDim.x << 16 (the grid size is a
// HIDDEN kernel argument: read through gridDim it is a scalar load in front of everything), pixels per image -- are
// preloaded into SGPRs with the wave, so the pixel requests and the coefficient requests leave without a scalar round trip.
template <typename T, bool HAS_B>
__global__ __launch_bounds__(256) void gn_apply_lead_kernel(const void* a_data, const void* b_data, const double* a_stats, const float* a_gamma,
                                                            const float* a_beta, const float* a_film, int c_groups, int hw, GnDev rest) {
  extern __shared__ __attribute__((aligned(16))) float s_coef[];
  GnDev g{};
  g.a.data = a_data; g.a.stats = a_stats; g.a.gamma = a_gamma; g.a.beta = a_beta; g.a.film = a_film;
  g.a.C = c_groups & 0x3ff; g.a.groups = (c_groups >> 10) & 0x3f; g.a.ld = g.a.C;
  g.b.data = b_data; g.b.C = g.a.C; g.b.ld = g.a.C;
  g.has_b = HAS_B; g.H = hw; g.W = 1;
  constexpr unsigned REST_OFF = ld_kernarg_offset<const void*, const void*, const double*, const float*, const float*, const float*, int, int>(alignof(GnDev));
  static_assert(REST_OFF == 56, "gn_apply_lead_kernel: leading arguments changed");
  typedef const char __attribute__((address_space(4)))* KChar;
  gn_apply_tile<T, HAS_B, false>(g, blockIdx.x, (int)((unsigned)c_groups >> 16), blockIdx.y, s_coef,
                                 (GnKernargPtr)((KChar)__builtin_amdgcn_kernarg_segment_ptr() + REST_OFF));
}

template <typename T, bool HAS_B, bool POOL>
__global__ __launch_bounds__(256) void gn_apply_kernel(GnDev g) {
  extern __shared__ __attribute__((aligned(16))) float s_coef[];   // [2C] for a, [2C] for b
  gn_apply_tile<T, HAS_B, POOL>(g, blockIdx.x, gridDim.x, blockIdx.y, s_coef);
}

}  // namespace
