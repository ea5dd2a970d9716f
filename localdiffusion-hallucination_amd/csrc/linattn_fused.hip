// Fused linear attention for 16-bit storage (bf16 / fp16): q, k and v are NEVER written to HBM.
//
// LinearAttention.forward (ddpm.py:234-251) reads x and produces to_out(ctx^T softmax_d(q)) with
// ctx = softmax_n(k) v^T.  The unfused path materialises the [B,n,384] qkv tensor (403 MB at
// 256^2, B=8, bf16) and reads it back twice; here both consumers RE-COMPUTE their slice of the 1x1
// projection from x (32..128 channels per pixel), which is far cheaper than the round trip:
//
//   ld_linattn_kvctx : per (batch, head, pixel chunk)  k,v = W_kv[h] rms(x)  (MFMA), two sweeps over
//        the chunk: (0) per-channel max of k, (1) P = exp(k - max), ctx += P^T V (MFMA, transposed
//        LDS reads), Z += colsum(P).  Emits partial ctx, Z and the chunk's max per channel; the
//        reduce kernel rescales partials to the global max (softmax over n, ddpm.py:243,247).
//   ld_linattn_out   : per pixel tile  q = softmax_d(W_q rms(x)) * scale  (MFMA, ddpm.py:242,245),
//        then out = M_b q with the q ACCUMULATOR REGISTERS used directly as the next MFMA's B
//        operand (cdna_hip_programming.md section 3: k-slot (lane>>4, e) of step s is q channel
//        32s + 16(e>>2) + 4(lane>>4) + (e&3); ld_linattn_fold(perm=1) writes M_b in that order),
//        + bias, RMSNorm, + x (ddpm.py:229-232,249,251,425).
#include "common.hip.h"
#include <stdlib.h>

typedef __attribute__((ext_vector_type(4))) short s16x4;

namespace {
constexpr int CTX_STRIDE = 32 * 32 + 64;   // ctx[d][e], Z[d], m[d]   (must match linattn.hip)
constexpr int KTN = 256;                   // pixels per tile in kvctx
constexpr int PROW = 96;                   // LDS row bytes of the P / V tiles (64 data + 32 pad)

__device__ __forceinline__ uint2 tr8(const char* p) {
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(const_cast<char*>(p)));
  return __builtin_bit_cast(uint2, v);
}

struct KvCtxArgs {
  const void* x;
  const uint4* wkv;     // [heads][NCH][4 tiles: k0 k1 v0 v1][64] fragments (g*sqrt(C) folded in)
  float* ctx_part;
  int n, C, heads, nchunks;
  const float* kshift;  // optional [heads*32]: softmax_n(k) shift per k-channel (>= max_n k): single-sweep mode
  int wsplit;           // 1: two-term weights (wave-per-head kernel, C = 32 / 64)
};
// The kernels take the 13 dwords as SCALAR arguments: all of them are preloaded into SGPRs with the wave (finding 83) --
// no scalar round trip in front of the first requests.  A by-value struct is never preloaded.
#define KVCTX_PARAMS const void* x_, const uint4* wkv_, float* ctx_part_, int n_, int C_, int heads_, int nchunks_, const float* kshift_, int wsplit_
#define KVCTX_ARGS(a) (a).x, (a).wkv, (a).ctx_part, (a).n, (a).C, (a).heads, (a).nchunks, (a).kshift, (a).wsplit

template <typename T, int NCH>
__global__ __launch_bounds__(256) void kvctx_kernel(KVCTX_PARAMS) {
  const KvCtxArgs a{x_, wkv_, ctx_part_, n_, C_, heads_, nchunks_, kshift_, wsplit_};
  constexpr int PLANE = KTN * 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* s_x = smem;                                     // [NCH][4][KTN][16 B]
  char* s_p = s_x + NCH * 4 * PLANE;                    // [KTN][96 B]
  char* s_v = s_p + KTN * PROW;
  float* s_rinv = reinterpret_cast<float*>(s_v + KTN * PROW);   // [KTN]
  float* s_m = s_rinv + KTN;                            // [4][32]
  const int ck = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, li = lane & 15, kq = lane >> 4;
  const int n = a.n, C = a.C;
  const int npc = (n + a.nchunks - 1) / a.nchunks;
  const int lo = ck * npc, hi = min(n, lo + npc);
  uint4 A[4][NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int m = 0; m < 4; ++m) A[m][c] = a.wkv[(((size_t)h * NCH + c) * 4 + m) * 64 + lane];
  float cmax[2][4], mloc[2][4], zs[2][4];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) { cmax[m][r] = -INFINITY; mloc[m][r] = 0.f; zs[m][r] = 0.f; }
  f32x4 cacc = {0.f, 0.f, 0.f, 0.f};
  const int dt = wv >> 1, et = wv & 1, tq = (lane >> 2) & 3, tp = lane & 3;
  const T* xb = reinterpret_cast<const T*>(a.x) + (size_t)b * n * C;

  // register-staged prefetch of the next x tile (T14): loads of tile k+1 fly during tile k's MFMAs.
  // The tile sequence is pass 0: lo..hi, then pass 1: lo..hi again.
  uint4 xr[NCH][4];
  auto issue = [&](int p0) {
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int p = p0 + (it * 4 + wv) * 16 + li;
        xr[c][it] = make_uint4(0u, 0u, 0u, 0u);
        if (p < hi) xr[c][it] = *reinterpret_cast<const uint4*>(xb + (size_t)p * C + c * 32 + kq * 8);
      }
  };
  issue(lo);
  for (int pass = 0; pass < 2; ++pass) {
    for (int p0 = lo; p0 < hi; p0 += KTN) {
      __syncthreads();
      float rs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int qq = (it * 4 + wv) * 16 + li;
          const uint4 raw = xr[c][it];
          float v[8];
          unpack16<T>(raw, v);
#pragma unroll
          for (int e = 0; e < 8; ++e) rs[it] = fmaf(v[e], v[e], rs[it]);
          *reinterpret_cast<uint4*>(s_x + (c * 4 + kq) * PLANE + qq * 16) = raw;
        }
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        float r = rs[it];
        r = kq4_sum(r);
        if (kq == 0) s_rinv[(it * 4 + wv) * 16 + li] = rms_rinv<false>(r);
      }
      __syncthreads();
      {
        const int nxt = p0 + KTN;
        if (nxt < hi) issue(nxt);
        else if (pass == 0) issue(lo);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int qq = (wv * 4 + j) * 16 + li;
        const bool valid = (p0 + qq) < hi;
        f32x4 acc[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
          const uint4 Bf = *reinterpret_cast<const uint4*>(s_x + (c * 4 + kq) * PLANE + qq * 16);
#pragma unroll
          for (int m = 0; m < 4; ++m) mma16<T>(acc[m], A[m][c], Bf);
        }
        const float rinv = s_rinv[qq];
        if (pass == 0) {
          if (valid) {
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
              for (int r = 0; r < 4; ++r) cmax[m][r] = fmaxf(cmax[m][r], acc[m][r] * rinv);
          }
        } else {
#pragma unroll
          for (int m = 0; m < 2; ++m) {
            float pv[4], vv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              pv[r] = valid ? __expf(acc[m][r] * rinv - mloc[m][r] + DT<T>::pshift * 0.6931471805599453f) : 0.f;
              vv[r] = valid ? acc[2 + m][r] * rinv : 0.f;
              zs[m][r] += pv[r];
            }
            *reinterpret_cast<uint2*>(s_p + qq * PROW + (16 * m + 4 * kq) * 2) =
                make_uint2(pack2<T>(pv[0], pv[1]), pack2<T>(pv[2], pv[3]));
            *reinterpret_cast<uint2*>(s_v + qq * PROW + (16 * m + 4 * kq) * 2) =
                make_uint2(pack2<T>(vv[0], vv[1]), pack2<T>(vv[2], vv[3]));
          }
        }
      }
      if (pass == 1) {
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < KTN / 32; ++ks) {
          const int row = ks * 32 + kq * 4 + tq;
          const char* pa = s_p + row * PROW + dt * 32 + tp * 8;
          const char* pb = s_v + row * PROW + et * 32 + tp * 8;
          const uint2 a1 = tr8(pa), a2 = tr8(pa + 16 * PROW);
          const uint2 b1 = tr8(pb), b2 = tr8(pb + 16 * PROW);
          mma16<T>(cacc, make_uint4(a1.x, a1.y, a2.x, a2.y), make_uint4(b1.x, b1.y, b2.x, b2.y));
        }
      }
    }
    if (pass == 0) {
      __syncthreads();
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float cm = wave16_max(cmax[m][r]);
          if (li == 0) s_m[wv * 32 + 16 * m + 4 * kq + r] = cm;
        }
      __syncthreads();
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int ch = 16 * m + 4 * kq + r;
          mloc[m][r] = fmaxf(fmaxf(s_m[ch], s_m[32 + ch]), fmaxf(s_m[64 + ch], s_m[96 + ch]));
        }
    }
  }
  float* dst = a.ctx_part + (((size_t)b * a.heads + h) * a.nchunks + ck) * CTX_STRIDE;
#pragma unroll
  for (int r = 0; r < 4; ++r) store_f32_out(&dst[(dt * 16 + kq * 4 + r) * 32 + et * 16 + li], cacc[r]);
  // Z: lanes -> waves -> block, fixed order (deterministic)
  __syncthreads();
  float* s_z = reinterpret_cast<float*>(s_p);           // [4][32]
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float z = wave16_sum(zs[m][r]);
      if (li == 0) s_z[wv * 32 + 16 * m + 4 * kq + r] = z;
    }
  __syncthreads();
  if (tid < 32) {
    store_f32_out(&dst[1024 + tid], s_z[tid] + s_z[32 + tid] + s_z[64 + tid] + s_z[96 + tid]);
    store_f32_out(&dst[1056 + tid], fmaxf(fmaxf(s_m[tid], s_m[32 + tid]), fmaxf(s_m[64 + tid], s_m[96 + tid])));
  }
}

// ------------------------------------------------------------------------------------------------
// kvctx, wave-per-head schedule (heads == 4): one workgroup = (batch, pixel chunk); the x tile is staged
// and RMS-normalised ONCE for all heads, and wave h computes k_h, v_h, P_h and the whole 32x32 context of
// head h privately.  The projection is issued TRANSPOSED -- pixels as the M side, W as the N side (the same two
// fragments, operands swapped) -- so an accumulator lane (li, kq) holds output channel li of pixels 4 kq + r: for the
// context product ctx[d][e] = sum_n P[n][d] V[n][e], whose contraction runs over PIXELS, the k accumulators of two
// 16-pixel groups ARE lane (d, kq)'s eight K-slots of the A operand and the v accumulators lane (e, kq)'s of the B
// operand (any pixel order serves a sum, as long as both operands use the same one).  P and V never touch LDS: no
// wave-private strip, no transposed reads, no LDS round trip between the softmax and the context MFMAs (round 5; the
// strip version spent two exposed round trips per 32 pixels).  Per-pixel factors (1 / rms) come as one broadcast
// 16-byte LDS read per group, per-channel ones (shift, Z, max) are one value per lane.  Emits the same partials as
// kvctx_kernel.

// SINGLE (caller-supplied shift): the max sweep, its registers and its code are compiled out.
// WS = 1: two-term weights (ld_pack_conv_weight_terms): 2 * NCH weight chunks, chunk v multiplies x chunk v >> 1.
template <typename T, int NCH, bool SINGLE, int WS = 0>
__global__ __launch_bounds__(256) void kvctx_wph_kernel(KVCTX_PARAMS) {
  const KvCtxArgs a{x_, wkv_, ctx_part_, n_, C_, heads_, nchunks_, kshift_, wsplit_};
  constexpr int NCW = NCH << WS;                        // weight chunks
  constexpr int PLANE = KTN * 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* s_x = smem;                                     // [NCH][4][KTN][16 B]
  float* s_rinv = reinterpret_cast<float*>(s_x + NCH * 4 * PLANE);           // [KTN]
  const int ck = blockIdx.x, b = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, li = lane & 15, kq = lane >> 4;
  const int h = wv;
  const int n = a.n, C = a.C;
  const int npc = (n + a.nchunks - 1) / a.nchunks;
  const int lo = ck * npc, hi = min(n, lo + npc);
  uint4 A[4][NCW];
#pragma unroll
  for (int c = 0; c < NCW; ++c)
#pragma unroll
    for (int m = 0; m < 4; ++m) A[m][c] = a.wkv[(((size_t)h * NCW + c) * 4 + m) * 64 + lane];
  // (Z_d = sum_n P_nd stays an fp32 sum of the UNROUNDED weights: taking it from the matrix pipe -- a B tile of ones in
  //  the context product -- was measured in round 4: no step time, and the fp16 context moved from 4e-3 to 6e-3 of its
  //  range against the oracle)
  float cmax[2], mloc[2], zs[2];                         // k-channel 16 m + li of this lane: running max, shift, Z
#pragma unroll
  for (int m = 0; m < 2; ++m) { cmax[m] = -INFINITY; mloc[m] = 0.f; zs[m] = 0.f; }
  f32x4 cacc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) cacc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const T* xb = reinterpret_cast<const T*>(a.x) + (size_t)b * n * C;

  uint4 xr[NCH][4];
  auto issue = [&](int p0) {
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int p = p0 + (it * 4 + wv) * 16 + li;
        xr[c][it] = make_uint4(0u, 0u, 0u, 0u);
        if (p < hi) xr[c][it] = *reinterpret_cast<const uint4*>(xb + (size_t)p * C + c * 32 + kq * 8);
      }
  };
  issue(lo);
  // softmax over n is shift-invariant: with a caller-supplied bound m_d >= max_n k_d (the Cauchy-Schwarz bound
  // ||W_k[d] g sqrt(C)||_2 of the RMS-normalised input, computed once at weight-packing time) the max sweep
  // (pass 0) is skipped and every chunk's partial is already on the same scale.
  constexpr int first_pass = SINGLE ? 1 : 0;
  if (SINGLE) {
#pragma unroll
    for (int m = 0; m < 2; ++m) mloc[m] = a.kshift[h * 32 + 16 * m + li];
  }
  constexpr float LOG2E = 1.4426950408889634f;
  float m2[2];                                           // shift * log2(e): P = exp2(k*log2e - m2), one FMA + v_exp
#pragma unroll
  for (int m = 0; m < 2; ++m) m2[m] = mloc[m] * LOG2E - DT<T>::pshift;   // P stored as P * 2^pshift (fp16 range)
  for (int pass = first_pass; pass < 2; ++pass) {
    for (int p0 = lo; p0 < hi; p0 += KTN) {
      __syncthreads();                                   // every wave is done with the previous x tile
      float rs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int qq = (it * 4 + wv) * 16 + li;
          const uint4 raw = xr[c][it];
          float v[8];
          unpack16<T>(raw, v);
#pragma unroll
          for (int e = 0; e < 8; ++e) rs[it] = fmaf(v[e], v[e], rs[it]);
          *reinterpret_cast<uint4*>(s_x + (c * 4 + kq) * PLANE + qq * 16) = raw;
        }
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        float r = rs[it];
        r = kq4_sum(r);
        if (kq == 0) s_rinv[(it * 4 + wv) * 16 + li] = rms_rinv<false>(r);
      }
      __syncthreads();
      {
        const int nxt = p0 + KTN;
        if (nxt < hi) issue(nxt);
        else if (!SINGLE && pass == 0) issue(lo);
      }
      const int ngrp = min(KTN, hi - p0 + 15) / 16;      // 16-pixel groups that contain valid pixels
      if (!SINGLE && pass == 0) {
        for (int g = 0; g < ngrp; ++g) {
          const int qq = g * 16 + li;
          f32x4 k0 = {0.f, 0.f, 0.f, 0.f}, k1 = k0;
#pragma unroll
          for (int c = 0; c < NCW; ++c) {
            const uint4 Xf = *reinterpret_cast<const uint4*>(s_x + ((c >> WS) * 4 + kq) * PLANE + qq * 16);
            mma16<T>(k0, Xf, A[0][c]);                   // (pixels x channels: lane = channel li, registers = pixels 4 kq + r)
            mma16<T>(k1, Xf, A[1][c]);
          }
          const float4 rv = *reinterpret_cast<const float4*>(s_rinv + g * 16 + 4 * kq);
          const float rr[4] = {rv.x, rv.y, rv.z, rv.w};
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (p0 + g * 16 + 4 * kq + r < hi) {
              cmax[0] = fmaxf(cmax[0], k0[r] * rr[r]);
              cmax[1] = fmaxf(cmax[1], k1[r] * rr[r]);
            }
          }
        }
      } else {
        for (int g2 = 0; g2 < (ngrp + 1) / 2; ++g2) {     // 32 pixels = one context K-step
          // groups that lie entirely inside the chunk (all of them at the model's sizes) skip the per-element
          // validity selects
          uint4 Pf[2], Vf[2];
          auto softmax_part = [&](auto full_tag) {
            constexpr bool FULL = decltype(full_tag)::value;
            unsigned pk[2][4], vk[2][4];
#pragma unroll
            for (int half = 0; half < 2; ++half) {
              const int g16 = (g2 * 2 + half) * 16;
              f32x4 acc[4];
#pragma unroll
              for (int m = 0; m < 4; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
              for (int c = 0; c < NCW; ++c) {
                const uint4 Xf = *reinterpret_cast<const uint4*>(s_x + ((c >> WS) * 4 + kq) * PLANE + (g16 + li) * 16);
#pragma unroll
                for (int m = 0; m < 4; ++m) mma16<T>(acc[m], Xf, A[m][c]);
              }
              const float4 rv = *reinterpret_cast<const float4*>(s_rinv + g16 + 4 * kq);
              const float rr[4] = {rv.x, rv.y, rv.z, rv.w};
#pragma unroll
              for (int m = 0; m < 2; ++m) {
                float pv[4], vv[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                  const bool valid = FULL || (p0 + g16 + 4 * kq + r) < hi;
                  pv[r] = valid ? __builtin_amdgcn_exp2f(fmaf(acc[m][r], rr[r] * LOG2E, -m2[m])) : 0.f;
                  vv[r] = valid ? acc[2 + m][r] * rr[r] : 0.f;
                  zs[m] += pv[r];
                }
                pk[m][half * 2 + 0] = pack2<T>(pv[0], pv[1]); pk[m][half * 2 + 1] = pack2<T>(pv[2], pv[3]);
                vk[m][half * 2 + 0] = pack2<T>(vv[0], vv[1]); vk[m][half * 2 + 1] = pack2<T>(vv[2], vv[3]);
              }
            }
#pragma unroll
            for (int m = 0; m < 2; ++m) {
              Pf[m] = make_uint4(pk[m][0], pk[m][1], pk[m][2], pk[m][3]);
              Vf[m] = make_uint4(vk[m][0], vk[m][1], vk[m][2], vk[m][3]);
            }
          };
          if (p0 + (g2 * 2 + 2) * 16 <= hi) softmax_part(std::true_type{});
          else softmax_part(std::false_type{});
#pragma unroll
          for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int et = 0; et < 2; ++et) mma16<T>(cacc[dt][et], Pf[dt], Vf[et]);
        }
      }
    }
    if (!SINGLE && pass == 0) {
#pragma unroll
      for (int m = 0; m < 2; ++m) { mloc[m] = kq4_max(cmax[m]); m2[m] = mloc[m] * LOG2E - DT<T>::pshift; }
    }
  }
  float* dst = a.ctx_part + (((size_t)b * a.heads + h) * a.nchunks + ck) * CTX_STRIDE;
  // (measured and not kept, finding 98: the context as 16-byte write-through stores, transposed through LDS --
  //  1.3283 -> 1.3374 ms, +0.7 %: the extra LDS round trip at the tail costs more than the 8.9 MB of dirty lines it avoids)
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int et = 0; et < 2; ++et)
#pragma unroll
      for (int r = 0; r < 4; ++r) store_f32_out(&dst[(dt * 16 + kq * 4 + r) * 32 + et * 16 + li], cacc[dt][et][r]);
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    const float z = kq4_sum(zs[m]);
    if (kq == 0) {
      store_f32_out(&dst[1024 + 16 * m + li], z);
      store_f32_out(&dst[1056 + 16 * m + li], mloc[m]);
    }
  }
}

// ------------------------------------------------------------------------------------------------
constexpr int LINOUT_TPB = 1;             // pixel tiles (of 128) per workgroup in linout_kernel (4 measured slower: fewer workgroups)

struct LinOutArgs {
  const void* x;
  const uint4* wq;       // [NCH][8][64] fragments of W_q (g*sqrt(C) folded in)
  const uint4* mfold;    // per batch: [4 k-steps][MT2][64] fragments of M_b in chained-operand order
  const float* bias;
  const float* g2;
  void* out;
  int n, C;
  float q_scale;
  const float* qshift;   // optional [4]: per head an upper bound of q (softmax_d shift): skips the max reduction
  int wsplit;            // 1: two-term W_q (C = 32 / 64)
};
// scalar kernel arguments, what the head's requests need first: 14 dwords are preloaded (finding 83); out / q_scale / wsplit
// arrive by an ordinary scalar load that nothing in the head waits for
#define LINOUT_PARAMS const void* x_, const uint4* wq_, const uint4* mfold_, const float* bias_, const float* g2_, const float* qshift_, \
                      int n_, int C_, void* out_, float q_scale_, int wsplit_
#define LINOUT_ARGS(a) (a).x, (a).wq, (a).mfold, (a).bias, (a).g2, (a).qshift, (a).n, (a).C, (a).out, (a).q_scale, (a).wsplit

template <typename T, int NCH, int WS = 0>
__global__ __launch_bounds__(256) void linout_kernel(LINOUT_PARAMS) {
  const LinOutArgs a{x_, wq_, mfold_, bias_, g2_, out_, n_, C_, q_scale_, qshift_, wsplit_};
  constexpr int NW = 2, NPT = 64 * NW, PLANE = NPT * 16, MT2 = 2 * NCH;
  constexpr int NCW = NCH << WS;                         // weight chunks of W_q (two per x chunk with two-term weights)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* s_x = smem;                                      // [NCH][4][NPT][16 B]
  char* s_wq = s_x + NCH * 4 * PLANE;                    // [NCW][8][1 KiB]
  char* s_mf = s_wq + NCW * 8 * 1024;                    // [4][MT2][1 KiB]
  float* s_rinv = reinterpret_cast<float*>(s_mf + 4 * MT2 * 1024);
  const int b = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, li = lane & 15, kq = lane >> 4;
  const int n = a.n, C = a.C;
  const T* xb = reinterpret_cast<const T*>(a.x) + (size_t)b * n * C;
  // W_q and M_b are staged once per workgroup and reused for LINOUT_TPB consecutive pixel tiles.  Every request of the
  // head -- both matrices, the first x tile, bias and gain -- leaves before the first wait: written as two
  // `lds[u] = global[u]` loops in front of the x loads, the head was three dependent round trips (load, wait,
  // ds_write per loop; tools/scan_head_chain.py)
  constexpr int WQ_IT = NCW * 8 * 64 / 256, MF_IT = 4 * MT2 * 64 / 256;
  const uint4* mf = a.mfold + (size_t)b * 4 * MT2 * 64;
  // (native vectors: a `uint4` copy is a memcpy in the IR, and with a scheduling barrier between the copy in and the
  // copy out the register array stays a private-memory array -- scratch traffic)
  u32x4 wq_r[WQ_IT], mf_r[MF_IT];
  u32x4 x_r[NCH][NW];
#pragma unroll
  for (int i = 0; i < WQ_IT; ++i) wq_r[i] = *reinterpret_cast<const u32x4*>(a.wq + tid + i * 256);
#pragma unroll
  for (int i = 0; i < MF_IT; ++i) mf_r[i] = *reinterpret_cast<const u32x4*>(mf + tid + i * 256);
  {
    const int p0 = blockIdx.x * LINOUT_TPB * NPT;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int it = 0; it < NW; ++it) {
        const int p = p0 + (it * 4 + wv) * 16 + li;
        x_r[c][it] = *reinterpret_cast<const u32x4*>(xb + (size_t)(p < n ? p : n - 1) * C + c * 32 + kq * 8);   // branch-free; masked at use
      }
  }
  // bias and the RMSNorm gain of this lane's output channels: loaded once, not inside the per-pixel epilogue
  float4 bvr[MT2], gvr[MT2];
#pragma unroll
  for (int m2 = 0; m2 < MT2; ++m2) {
    bvr[m2] = *reinterpret_cast<const float4*>(a.bias + m2 * 16 + kq * 4);
    gvr[m2] = *reinterpret_cast<const float4*>(a.g2 + m2 * 16 + kq * 4);
  }
  // the q bound as a (uniform) VECTOR load in the same batch: as scalar loads through the pointer it was one more
  // dependent round trip behind the kernel arguments, and hipcc put it behind the vector waits
  const float4 qsv = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(a.qshift ? a.qshift : a.bias) + (lane & 0));
  __builtin_amdgcn_sched_barrier(0);                     // every request above leaves before the first wait below
  const float qs2[4] = {qsv.x * 1.4426950408889634f, qsv.y * 1.4426950408889634f, qsv.z * 1.4426950408889634f, qsv.w * 1.4426950408889634f};
#pragma unroll
  for (int i = 0; i < WQ_IT; ++i) *reinterpret_cast<u32x4*>(s_wq + (tid + i * 256) * 16) = wq_r[i];
#pragma unroll
  for (int i = 0; i < MF_IT; ++i) *reinterpret_cast<u32x4*>(s_mf + (tid + i * 256) * 16) = mf_r[i];
  for (int tl = 0; tl < LINOUT_TPB; ++tl) {
  const int p0 = (blockIdx.x * LINOUT_TPB + tl) * NPT;
  if (p0 >= n) break;
  if (tl > 0) __syncthreads();                          // previous tile's x / rinv fully consumed
  float rs[NW];
#pragma unroll
  for (int i = 0; i < NW; ++i) rs[i] = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int it = 0; it < NW; ++it) {
      const int qq = (it * 4 + wv) * 16 + li, p = p0 + qq;
      uint4 raw = make_uint4(0u, 0u, 0u, 0u);
      if (p < n) {
        if (LINOUT_TPB == 1 || tl == 0) raw = make_uint4(x_r[c][it][0], x_r[c][it][1], x_r[c][it][2], x_r[c][it][3]);
        else raw = *reinterpret_cast<const uint4*>(xb + (size_t)p * C + c * 32 + kq * 8);
        float v[8];
        unpack16<T>(raw, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) rs[it] = fmaf(v[e], v[e], rs[it]);
      }
      *reinterpret_cast<uint4*>(s_x + (c * 4 + kq) * PLANE + qq * 16) = raw;
    }
#pragma unroll
  for (int it = 0; it < NW; ++it) {
    float r = rs[it];
    r = kq4_sum(r);
    if (kq == 0) s_rinv[(it * 4 + wv) * 16 + li] = rms_rinv<false>(r);
  }
  // bias / gain were requested with the head's batch and are first USED in the epilogue: retire them here (free: the
  // batch has landed), or hipcc's wait for them sits behind the first output store and drains it (the vector-memory
  // counter is in order and counts stores; tools/scan_store_waits.py)
  if (tl == 0) __builtin_amdgcn_s_waitcnt(0x0F70);        // vmcnt(0), expcnt / lgkmcnt untouched
  __syncthreads();
#pragma unroll
  for (int j = 0; j < NW; ++j) {
    const int qq = (wv * NW + j) * 16 + li, p = p0 + qq;
    const bool valid = p < n;
    f32x4 q[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) q[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < NCW; ++c) {
      const uint4 Bf = *reinterpret_cast<const uint4*>(s_x + ((c >> WS) * 4 + kq) * PLANE + qq * 16);
#pragma unroll
      for (int m = 0; m < 8; ++m)
        mma16<T>(q[m], *reinterpret_cast<const uint4*>(s_wq + (c * 8 + m) * 1024 + lane * 16), Bf);
    }
    const float rinv2 = s_rinv[qq] * 1.4426950408889634f;  // softmax_d(q) = exp2(q*log2e - shift*log2e) / sum
    uint4 B2[4];
#pragma unroll
    for (int hh = 0; hh < 4; ++hh) {                    // one head = channel tiles 2hh, 2hh+1
      float v0[4], v1[4];
      float mx;
      if (a.qshift) {
        mx = qs2[hh];                                   // data-independent bound (see ld_linattn_out)
      } else {
        mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 4; ++r) mx = fmaxf(mx, fmaxf(q[2 * hh][r], q[2 * hh + 1][r]) * rinv2);   // rinv2 > 0
        mx = kq4_max(mx);
      }
      float sum = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {                     // one fma + one v_exp per value (the file is built with -ffp-contract=off)
        v0[r] = __builtin_amdgcn_exp2f(fmaf(q[2 * hh][r], rinv2, -mx)); v1[r] = __builtin_amdgcn_exp2f(fmaf(q[2 * hh + 1][r], rinv2, -mx));
        sum += v0[r] + v1[r];
      }
      sum = kq4_sum(sum);
      const float sc = a.q_scale * __builtin_amdgcn_rcpf(sum);
      B2[hh] = make_uint4(pack2<T>(v0[0] * sc, v0[1] * sc), pack2<T>(v0[2] * sc, v0[3] * sc),
                          pack2<T>(v1[0] * sc, v1[1] * sc), pack2<T>(v1[2] * sc, v1[3] * sc));
    }
    f32x4 o[MT2];
#pragma unroll
    for (int m2 = 0; m2 < MT2; ++m2) o[m2] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int m2 = 0; m2 < MT2; ++m2)
        mma16<T>(o[m2], *reinterpret_cast<const uint4*>(s_mf + (s * MT2 + m2) * 1024 + lane * 16), B2[s]);
    float ss = 0.f;
    float y[MT2][4];
#pragma unroll
    for (int m2 = 0; m2 < MT2; ++m2) {
      const float4 bv = bvr[m2];
      y[m2][0] = o[m2][0] + bv.x; y[m2][1] = o[m2][1] + bv.y; y[m2][2] = o[m2][2] + bv.z; y[m2][3] = o[m2][3] + bv.w;
#pragma unroll
      for (int r = 0; r < 4; ++r) ss = fmaf(y[m2][r], y[m2][r], ss);
    }
    ss = kq4_sum(ss);
    const float inv = rms_rinv<false>(ss);
    {
      // two adjacent m-tiles leave as ONE 16-byte store per lane (pair_frag16: the exchange runs on every lane, only
      // the store is predicated; the residual comes from LDS, which holds zeros for pixels past the image)
      static_assert(MT2 % 2 == 0, "m-tiles are stored in pairs");
      char* opix = reinterpret_cast<char*>(reinterpret_cast<T*>(a.out) + ((size_t)b * n + (valid ? p : 0)) * C);
#pragma unroll
      for (int m2 = 0; m2 < MT2; m2 += 2) {
        float r4[2][4];
#pragma unroll
        for (int mm = 0; mm < 2; ++mm) {
          const int mq = m2 + mm;
          const float4 gv = gvr[mq];
          // the residual x is already in the staging tile: channels co..co+3 = chunk mq/2, fragment 2*(mq&1) + kq/2
          float xr[4];
          load4<T>(reinterpret_cast<const T*>(s_x + ((mq >> 1) * 4 + (mq & 1) * 2 + (kq >> 1)) * PLANE + qq * 16 + (kq & 1) * 8), xr);
          r4[mm][0] = y[mq][0] * inv * gv.x + xr[0]; r4[mm][1] = y[mq][1] * inv * gv.y + xr[1];
          r4[mm][2] = y[mq][2] * inv * gv.z + xr[2]; r4[mm][3] = y[mq][3] * inv * gv.w + xr[3];
        }
        const uint4 w16 = pair_frag16<T>(r4[0], r4[1]);
        if (valid) store16_out(opix + m2 * 16 * sizeof(T) + pair_frag16_off(kq), w16);
      }
    }
  }
  }  // tiles of this workgroup
}

template <typename T>
int kvctx_launch(const KvCtxArgs& a, int B, hipStream_t st) {
  const int nch = a.C / 32, heads = a.heads, nchunks = a.nchunks;
  const float* kshift = a.kshift;
#ifdef LD_DEBUG_VARIANTS          // experiment-only: the first (workgroup-per-head) schedule, for reproducing DESIGN section 5
  static const int no_wph = getenv("LD_KVCTX_V1") ? 1 : 0;
#else
  constexpr int no_wph = 0;
#endif
  if (heads == 4 && (!no_wph || kshift)) {               // wave-per-head schedule
    const size_t lds2 = (size_t)nch * 4 * KTN * 16 + KTN * sizeof(float);
    dim3 grid2(nchunks, B);
    if (a.wsplit) {                                        // two-term weights: the full- and half-resolution blocks
      LD_REQUIRE(nch <= 2, "ld_linattn_kvctx: two-term weights are built for C = 32 / 64 (got %d)", a.C);
      if (nch == 1) {
        if (kshift) LD_LAUNCH((kvctx_wph_kernel<T, 1, true, 1>), grid2, dim3(256), lds2, st, KVCTX_ARGS(a));
        else LD_LAUNCH((kvctx_wph_kernel<T, 1, false, 1>), grid2, dim3(256), lds2, st, KVCTX_ARGS(a));
      } else {
        if (kshift) LD_LAUNCH((kvctx_wph_kernel<T, 2, true, 1>), grid2, dim3(256), lds2, st, KVCTX_ARGS(a));
        else LD_LAUNCH((kvctx_wph_kernel<T, 2, false, 1>), grid2, dim3(256), lds2, st, KVCTX_ARGS(a));
      }
    } else if (nch == 1) {
      if (kshift) LD_LAUNCH((kvctx_wph_kernel<T, 1, true>), grid2, dim3(256), lds2, st, KVCTX_ARGS(a));
      else LD_LAUNCH((kvctx_wph_kernel<T, 1, false>), grid2, dim3(256), lds2, st, KVCTX_ARGS(a));
    } else if (nch == 2) {
      if (kshift) LD_LAUNCH((kvctx_wph_kernel<T, 2, true>), grid2, dim3(256), lds2, st, KVCTX_ARGS(a));
      else LD_LAUNCH((kvctx_wph_kernel<T, 2, false>), grid2, dim3(256), lds2, st, KVCTX_ARGS(a));
    } else {
      if (kshift) LD_HIP(ld_allow_lds((kvctx_wph_kernel<T, 4, true>), lds2));      // cached per device
      else LD_HIP(ld_allow_lds((kvctx_wph_kernel<T, 4, false>), lds2));
      if (kshift) LD_LAUNCH((kvctx_wph_kernel<T, 4, true>), grid2, dim3(256), lds2, st, KVCTX_ARGS(a));
      else LD_LAUNCH((kvctx_wph_kernel<T, 4, false>), grid2, dim3(256), lds2, st, KVCTX_ARGS(a));
    }
    LD_LAUNCH_CHECK("linattn_kvctx(wave-per-head)");
    return LD_OK;
  }
  LD_REQUIRE(!a.wsplit, "ld_linattn_kvctx: two-term weights need heads == 4 (the wave-per-head kernel)");
  dim3 grid(nchunks, heads, B);
  const size_t lds = (size_t)nch * 4 * KTN * 16 + 2 * KTN * PROW + KTN * sizeof(float) + 128 * sizeof(float);
  if (nch == 1) {
    LD_LAUNCH((kvctx_kernel<T, 1>), grid, dim3(256), lds, st, KVCTX_ARGS(a));
  } else if (nch == 2) {
    LD_HIP(ld_allow_lds((kvctx_kernel<T, 2>), lds));
    LD_LAUNCH((kvctx_kernel<T, 2>), grid, dim3(256), lds, st, KVCTX_ARGS(a));
  } else {
    LD_HIP(ld_allow_lds((kvctx_kernel<T, 4>), lds));
    LD_LAUNCH((kvctx_kernel<T, 4>), grid, dim3(256), lds, st, KVCTX_ARGS(a));
  }
  LD_LAUNCH_CHECK("linattn_kvctx");
  return LD_OK;
}

template <typename T>
int linout_launch(const LinOutArgs& a, int B, hipStream_t st) {
  const int n = a.n, nch = a.C / 32;
  dim3 grid((n + 128 * LINOUT_TPB - 1) / (128 * LINOUT_TPB), B);
  const size_t lds = (size_t)nch * 4 * 128 * 16 + ((size_t)nch << a.wsplit) * 8 * 1024 + (size_t)4 * 2 * nch * 1024 + 128 * sizeof(float);
  if (a.wsplit) {
    LD_REQUIRE(nch <= 2, "ld_linattn_out: two-term weights are built for C = 32 / 64 (got %d)", a.C);
    if (nch == 1) LD_LAUNCH((linout_kernel<T, 1, 1>), grid, dim3(256), lds, st, LINOUT_ARGS(a));
    else LD_LAUNCH((linout_kernel<T, 2, 1>), grid, dim3(256), lds, st, LINOUT_ARGS(a));
  } else if (nch == 1) {
    LD_LAUNCH((linout_kernel<T, 1>), grid, dim3(256), lds, st, LINOUT_ARGS(a));
  } else if (nch == 2) {
    LD_LAUNCH((linout_kernel<T, 2>), grid, dim3(256), lds, st, LINOUT_ARGS(a));
  } else {
    LD_HIP(ld_allow_lds((linout_kernel<T, 4>), lds));
    LD_LAUNCH((linout_kernel<T, 4>), grid, dim3(256), lds, st, LINOUT_ARGS(a));
  }
  LD_LAUNCH_CHECK("linattn_out");
  return LD_OK;
}
}  // namespace

extern "C" int ld_linattn_kvctx(const void* x, const void* wkv_packed, const float* kshift, float* ctx_part, int B,
                                int n, int C, int heads, int dim_head, int nchunks, int dtype, void* stream) {
  return ld_linattn_kvctx_terms(x, wkv_packed, kshift, ctx_part, B, n, C, heads, dim_head, nchunks, dtype, 1, stream);
}

extern "C" int ld_linattn_kvctx_terms(const void* x, const void* wkv_packed, const float* kshift, float* ctx_part, int B,
                                      int n, int C, int heads, int dim_head, int nchunks, int dtype, int weight_terms,
                                      void* stream) {
  LD_REQUIRE(weight_terms == 1 || weight_terms == 2, "ld_linattn_kvctx: weight_terms %d", weight_terms);
  LD_REQUIRE(x && wkv_packed && ctx_part && B > 0 && n > 0 && nchunks > 0, "ld_linattn_kvctx: bad args");
  LD_REQUIRE(ld_dtype_16(dtype), "ld_linattn_kvctx: 16-bit storage only (fp32 uses the unfused path)");
  LD_REQUIRE(dim_head == 32 && (C == 32 || C == 64 || C == 128), "ld_linattn_kvctx: dim_head 32, C in {32,64,128}");
  LD_REQUIRE(kshift == nullptr || heads == 4, "ld_linattn_kvctx: the single-sweep mode (kshift) needs heads == 4");
  KvCtxArgs a{x, (const uint4*)wkv_packed, ctx_part, n, C, heads, nchunks, kshift, weight_terms == 2 ? 1 : 0};
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  return LD_DISPATCH16(dtype, kvctx_launch<T>(a, B, st));
}

extern "C" int ld_linattn_out(const void* x, const void* wq_packed, const float* qshift, const void* mfold,
                              const float* bias, const float* g2, void* out, int B, int n, int C, float q_scale,
                              int dtype, void* stream) {
  return ld_linattn_out_terms(x, wq_packed, qshift, mfold, bias, g2, out, B, n, C, q_scale, dtype, 1, stream);
}

extern "C" int ld_linattn_out_terms(const void* x, const void* wq_packed, const float* qshift, const void* mfold,
                                    const float* bias, const float* g2, void* out, int B, int n, int C, float q_scale,
                                    int dtype, int weight_terms, void* stream) {
  LD_REQUIRE(weight_terms == 1 || weight_terms == 2, "ld_linattn_out: weight_terms %d", weight_terms);
  LD_REQUIRE(x && wq_packed && mfold && bias && g2 && out && B > 0 && n > 0, "ld_linattn_out: bad args");
  LD_REQUIRE(ld_dtype_16(dtype), "ld_linattn_out: 16-bit storage only (fp32 uses the unfused path)");
  LD_REQUIRE(C == 32 || C == 64 || C == 128, "ld_linattn_out: C in {32,64,128}");
  LinOutArgs a{x, (const uint4*)wq_packed, (const uint4*)mfold, bias, g2, out, n, C, q_scale, qshift, weight_terms == 2 ? 1 : 0};
  return LD_DISPATCH16(dtype, linout_launch<T>(a, B, reinterpret_cast<hipStream_t>(stream)));
}
