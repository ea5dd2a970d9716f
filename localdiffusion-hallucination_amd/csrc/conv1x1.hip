// 1x1 convolution = GEMM over channels on MFMA, NHWC, gfx950, with the attention plumbing of the
// reference fused in as prologues / epilogues:
//   - input: channel concat of two tensors (res_conv over torch.cat, ddpm.py:198,439-443),
//            pixel-unshuffle gather (Downsample, ddpm.py:120-124),
//            RMSNorm on the input (ddpm.py:131-132,237,274): the per-pixel 1/max(||x||,1e-12) is
//            accumulated while the tile is staged and applied to the output column; g*sqrt(C) is
//            folded into the packed weight;
//   - output: bias; q-softmax over each head's 32 channels * dim_head^-0.5 (ddpm.py:242,245);
//            RMSNorm over the output channels + residual (ddpm.py:229-232,425,444);
//            residual add (ddpm.py:425,431,444);
//   - per-batch weights (weight_bstride) so that linear attention's  to_out(ctx^T q)  is one GEMM
//     with M_b = W_out (ctx_b/Z_b)^T  (see linattn.hip).
//
// Tiling: 256 threads = 4 waves; tile = 64*NW consecutive pixels of one image x 16*MT output
// channels; wave w owns pixel tiles [w*NW,(w+1)*NW) and all MT channel tiles.
#include "common.hip.h"
#include <stdlib.h>

namespace {

constexpr int EPI_GN_TAIL_RES = 6;   // internal: LD_EPI_GN_TAIL with the `residual` operand set (conv_fusion's invariant half of res_conv)

struct Conv1Dev {       // what the setup reads first sits together (pointers, then scalars): few, wide s_loads at the head (finding 82)
  SrcDev s[2];
  const void* w;
  long w_bstride;
  const void* res;
  void* out;
  int nsrc, unshuffle, rms_in;
  int B, H, W, Cout;
  int wsplit;           // 1: two-term weights (pack.hip): 2*nch virtual chunks, source chunk v >> 1, weight chunk v
  int group;            // host decision: K-chunks staged per barrier pair (0 = 1; KG on small maps, 2 on mid-size ones)
  int epi, hidden;
  float q_scale;
  const float* bias;
  const float* g2;
  unsigned* kmax;
  SrcDev tail;          // LD_EPI_GN_TAIL operand
};

// EPI (the epilogue kind) is a template parameter: as a runtime switch inside the store loop it kept every
// epilogue's registers and branches alive in every launch.
// G (small maps): the K loop of the plain variant (G = 1) is one dependent global round trip per 64-byte channel
// chunk (load -> LDS -> barrier -> MT*NW MFMAs: 12-16 round trips for the 384- and 512-channel inputs of the 32^2 /
// 64^2 stages, 15-20 us for a megabyte of data).  With G = KG the loads of KG chunks are in flight together and
// share one barrier pair: nch / KG round trips, KG staging buffers of LDS (48 KB at 128-pixel x 64-channel tiles, so
// three workgroups still fit a CU; staging the whole K extent at once -- the shelved tools/experiments/ variant of
// finding 28 -- left one workgroup per CU and lost on the launches with several workgroups per CU).
constexpr int KG = 4;
typedef const Conv1Dev __attribute__((address_space(4)))* Conv1KernargPtr;   // the block in kernel-argument (constant) memory
// TLEAD (conv1x1_taillead_kernel, the GroupNorm-tail launches): `head` holds only the tail operand -- preloaded scalar
// kernel arguments, so the coefficient requests and the second-operand requests leave without a scalar round trip -- and
// everything else is read from `rest` under the statistics' round trip (build_gn_coef's after_issue hook; finding 84).
template <typename T, int MT, int NW, int EPI, int G, bool TLEAD, bool KLEAD = false>
__device__ __forceinline__ void conv1x1_body(const Conv1Dev& head, Conv1KernargPtr rest) {
  constexpr int E = DT<T>::E, CK = DT<T>::CK;
  constexpr int NPT = 64 * NW, PLANE = NPT * 16;
  constexpr int XCH = 4 * PLANE, WCHB = MT * 1024;                    // bytes of one staged chunk: tile, weights
  static_assert(!TLEAD || EPI == LD_EPI_GN_TAIL || EPI == EPI_GN_TAIL_RES, "tail-lead variant: GroupNorm-tail epilogues only");
  Conv1Dev a = head;
  if constexpr (KLEAD) {
    // conv1x1_klead_kernel: sources, weights, second operand and geometry are preloaded arguments; what the epilogue
    // needs is read from the block (invariant scalar loads that nothing in the head waits for)
    a.out = rest->out; a.bias = rest->bias; a.g2 = rest->g2; a.kmax = rest->kmax; a.hidden = rest->hidden; a.q_scale = rest->q_scale;
  }
  if constexpr (!TLEAD && !KLEAD) {
    // every scalar argument the setup needs, requested in ONE batch (left alone hipcc fetches the block in four dependent ones)
    asm volatile("" ::"s"(a.H), "s"(a.W), "s"(a.Cout), "s"(a.nsrc), "s"(a.unshuffle), "s"(a.rms_in), "s"(a.wsplit), "s"(a.s[0].C),
                 "s"(a.s[0].ld), "s"(a.s[0].data), "s"(a.s[1].C), "s"(a.s[1].ld), "s"(a.s[1].data), "s"(a.w), "s"(a.w_bstride),
                 "s"(a.res), "s"(a.out));
    if constexpr (EPI == LD_EPI_GN_TAIL || EPI == EPI_GN_TAIL_RES)
      asm volatile("" ::"s"(a.tail.data), "s"(a.tail.stats), "s"(a.tail.gamma), "s"(a.tail.beta), "s"(a.tail.film), "s"(a.tail.C),
                   "s"(a.tail.groups), "s"(a.tail.act), "s"(a.tail.film_bstride));
  }

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* s_x = smem;                                                  // [G][4][PLANE]
  char* s_w = smem + G * XCH;                                        // [G][MT][1 KiB]
  float* s_rinv = reinterpret_cast<float*>(s_w + G * WCHB);
  float* s_tcoef = s_rinv + NPT;                                   // [2*Cout] GN_TAIL coefficients, then 64 doubles scratch

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, px = lane & 15, kq = lane >> 4;
  const int b = blockIdx.z, m0 = blockIdx.y * MT;
  const int HW = a.H * a.W, p0 = blockIdx.x * NPT;      // (TLEAD: head.H = pixels, head.W = 1, head.Cout set)
  const int mt_total = a.Cout / 16;

  f32x4 acc[MT][NW];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int j = 0; j < NW; ++j) acc[m][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  float rs[NW];
#pragma unroll
  for (int i = 0; i < NW; ++i) rs[i] = 0.f;

  // The epilogue's second operand (residual / GroupNorm-tail input) does not depend on the GEMM: request it now,
  // so its global round trip overlaps the K loop instead of following it.
  constexpr bool TAIL = EPI == LD_EPI_GN_TAIL || EPI == EPI_GN_TAIL_RES;
  constexpr bool HAS_OP2 = EPI == LD_EPI_RMS_RES || EPI == LD_EPI_RES || TAIL;
  // Raw and from a clamped address (pixels past the image are never stored): as `if (p < HW) load4(...)` every one of
  // the MT * NW loads was waited for inside its own branch -- 8-16 dependent round trips (3-6 us) in front of the K loop
  // of every launch with a second operand (tools/scan_serial_loads.py).
  // 16-bit storage: the fragments of two adjacent m-tiles arrive as ONE 16-byte load per lane (W2 = the raw piece; it is
  // regrouped into the lane's two fragments by unpair_frag16 where it is used, with the whole wave active)
  typedef typename Raw4<T>::type R4;
  constexpr bool W2 = sizeof(T) == 2;
  constexpr int MP = W2 ? MT / 2 : MT;                                // second-operand pieces per pixel row
  typedef typename std::conditional<W2, uint4, R4>::type RP;
  RP op2[HAS_OP2 ? MP : 1][HAS_OP2 ? NW : 1];
  auto load_op = [&](const T* src, RP (&dst)[MP][NW]) {
#pragma unroll
    for (int j = 0; j < NW; ++j) {
      const int p = min(p0 + (wv * NW + j) * 16 + px, HW - 1);
      const T* pix = src + ((size_t)b * HW + p) * a.Cout + m0 * 16;
#pragma unroll
      for (int m = 0; m < MP; ++m) {
        if constexpr (W2) dst[m][j] = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(pix + m * 32) + pair_frag16_off(kq));
        else dst[m][j] = load4_raw<T>(pix + m * 16 + kq * 4);
      }
    }
  };
  auto pin_op = [&](const RP& r) {
    if constexpr (W2) asm volatile("" ::"v"(r.x), "v"(r.y), "v"(r.z), "v"(r.w));
    else pin_raw4(r);
  };
  // fragment of m-tile m of pixel row j (call with the whole wave active)
  auto frag_op = [&](const RP (&src)[MP][NW], int m, int j, float* v) {
    if constexpr (W2) {
      uint2 f0, f1;
      unpair_frag16(src[m >> 1][j], f0, f1);
      unpack4<T>((m & 1) ? f1 : f0, v);
    } else {
      unpack4<T>(src[m][j], v);
    }
  };
  if constexpr (HAS_OP2) load_op(reinterpret_cast<const T*>(TAIL ? a.tail.data : a.res), op2);
  if constexpr (TAIL) {
    double* red = reinterpret_cast<double*>(s_tcoef + 2 * a.Cout);
    if constexpr (TLEAD) {
      build_gn_coef<DT<T>::precise>(a.tail, b, 0, (long)HW, s_tcoef, red, tid, 256, [&]() {
        Conv1KernargPtr pr = rest;                       // laundered here: read under the statistics' round trip, not in front of it
        asm volatile("" : "+s"(pr));
        a.s[0].data = pr->s[0].data; a.s[0].C = pr->s[0].C; a.s[0].ld = pr->s[0].ld;
        a.s[1].data = pr->s[1].data; a.s[1].C = pr->s[1].C; a.s[1].ld = pr->s[1].ld;
        a.w = pr->w; a.w_bstride = pr->w_bstride; a.res = pr->res; a.out = pr->out;
        a.nsrc = pr->nsrc; a.unshuffle = pr->unshuffle; a.rms_in = pr->rms_in; a.H = pr->H; a.W = pr->W;
        a.wsplit = pr->wsplit; a.hidden = pr->hidden; a.q_scale = pr->q_scale; a.bias = pr->bias; a.g2 = pr->g2; a.kmax = pr->kmax;
        a.tail.act = pr->tail.act;
      });
    } else {
      build_gn_coef<DT<T>::precise>(a.tail, b, 0, (long)HW, s_tcoef, red, tid, 256);
    }
  }
  // K-chunk bookkeeping: plain = chunks of src0 then src1; unshuffle = 4 sub-pixels x chunks of src0
  const int nc0 = a.s[0].C / CK;
  const int ws = a.wsplit;
  const int nch = (a.unshuffle ? 4 * nc0 : nc0 + (a.nsrc > 1 ? a.s[1].C / CK : 0)) << ws;     // virtual chunks
  const uint4* wg = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(a.w) + (size_t)b * a.w_bstride);

  // global address of this thread's fragment of chunk `ch` for tile pixel slot `it` (nullptr: past the image)
  auto frag_ptr = [&](int chv, int it) -> const uint4* {
    const int ch = chv >> ws;                                          // source chunk of the virtual chunk
    int si = 0, c0, p1 = 0, p2 = 0;
    if (a.unshuffle) {
      const int pp = ch / nc0;
      c0 = (ch - pp * nc0) * CK;
      p1 = pp >> 1; p2 = pp & 1;
    } else {
      si = ch >= nc0 ? 1 : 0;
      c0 = (ch - si * nc0) * CK;
    }
    const T* sdata = reinterpret_cast<const T*>(si ? a.s[1].data : a.s[0].data);
    const int Cs = si ? a.s[1].ld : a.s[0].ld;   // pixel stride in elements
    const int p = p0 + (it * 4 + wv) * 16 + px;
    if (p >= HW) return nullptr;
    size_t pix;
    if (a.unshuffle) {
      const int y = p / a.W, x = p - y * a.W;
      pix = ((size_t)b * (2 * a.H) + 2 * y + p1) * (2 * a.W) + 2 * x + p2;
    } else {
      pix = (size_t)b * HW + p;
    }
    return reinterpret_cast<const uint4*>(sdata + pix * Cs + c0 + kq * E);
  };
  auto stage = [&](const uint4& raw, int it, char* xdst) {             // tile fragment -> LDS (+ RMSNorm sum of squares)
    if (a.rms_in) {
      float v[E];
      unpack16<T>(raw, v);
#pragma unroll
      for (int e = 0; e < E; ++e) rs[it] = fmaf(v[e], v[e], rs[it]);
    }
    *reinterpret_cast<uint4*>(xdst + kq * PLANE + ((it * 4 + wv) * 16 + px) * 16) = raw;
  };
  auto mfma_chunk = [&](const char* xsrc, const char* wsrc) {
    uint4 A[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) A[m] = *reinterpret_cast<const uint4*>(wsrc + m * 1024 + lane * 16);
#pragma unroll
    for (int j = 0; j < NW; ++j) {
      const uint4 Bf = *reinterpret_cast<const uint4*>(xsrc + kq * PLANE + ((wv * NW + j) * 16 + px) * 16);
#pragma unroll
      for (int m = 0; m < MT; ++m) mma16<T>(acc[m][j], A[m], Bf);
    }
  };
  constexpr int WPT = (MT * 64 + 255) / 256;                           // weight fragments per thread and chunk
  if constexpr (G > 1) {
    for (int c0 = 0; c0 < nch; c0 += G) {
      uint4 raw[G][NW], wr[G][WPT];
#pragma unroll
      for (int g = 0; g < G; ++g) {
        if (c0 + g < nch) {                                            // uniform
#pragma unroll
          for (int it = 0; it < NW; ++it) {
            const uint4* gp = frag_ptr(c0 + g, it);
            raw[g][it] = gp ? *gp : make_uint4(0u, 0u, 0u, 0u);
          }
#pragma unroll
          for (int k = 0; k < WPT; ++k) {
            const int u = k * 256 + tid;
            wr[g][k] = u < MT * 64 ? wg[((size_t)(c0 + g) * mt_total + m0) * 64 + u] : make_uint4(0u, 0u, 0u, 0u);
          }
        }
      }
      __syncthreads();                                                 // the previous group's fragments are consumed
#pragma unroll
      for (int g = 0; g < G; ++g) {
        if (c0 + g < nch) {
#pragma unroll
          for (int it = 0; it < NW; ++it) stage(raw[g][it], it, s_x + (size_t)g * XCH);
#pragma unroll
          for (int k = 0; k < WPT; ++k) {
            const int u = k * 256 + tid;
            if (u < MT * 64) *reinterpret_cast<uint4*>(s_w + (size_t)g * WCHB + u * 16) = wr[g][k];
          }
        }
      }
      __syncthreads();
#pragma unroll
      for (int g = 0; g < G; ++g)
        if (c0 + g < nch) mfma_chunk(s_x + (size_t)g * XCH, s_w + (size_t)g * WCHB);
    }
  } else {
    for (int ch = 0; ch < nch; ++ch) {
      __syncthreads();
#pragma unroll
      for (int it = 0; it < NW; ++it) {
        const uint4* gp = frag_ptr(ch, it);
        stage(gp ? *gp : make_uint4(0u, 0u, 0u, 0u), it, s_x);
      }
#pragma unroll
      for (int k = 0; k < WPT; ++k) {
        const int u = k * 256 + tid;
        if (u < MT * 64) *reinterpret_cast<uint4*>(s_w + u * 16) = wg[((size_t)ch * mt_total + m0) * 64 + u];
      }
      __syncthreads();
      mfma_chunk(s_x, s_w);
    }
  }

  if (a.rms_in) {
    // the 4 kq lanes of a staged pixel sit 16 lanes apart in the same wave
#pragma unroll
    for (int it = 0; it < NW; ++it) {
      float r = rs[it];
      r = kq4_sum(r);
      if (kq == 0) s_rinv[(it * 4 + wv) * 16 + px] = rms_rinv<DT<T>::precise>(r);
    }
    __syncthreads();
  }

  // ---- epilogue: lane holds channels (m0+m)*16 + 4kq + r of pixel p0 + (wv*NW+j)*16 + px
  // The other operands of the epilogue -- bias, the RMSNorm gain g2, the step-invariant half of res_conv (conv_fusion):
  // loaded inside the store loop, each of them put a wait BEHIND the stores of the previous pixel row (the
  // vector-memory counter is in order and counts stores: a store round trip exposed per row,
  // tools/scan_store_waits.py).  They are requested together here -- after the K loop, whose staging registers are
  // dead by now: held across the loop they cost 8-100 registers -- and retired before the first store.
  float bv[MT][4], g2v[EPI == LD_EPI_RMS_RES ? MT : 1][4];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int co = (m0 + m) * 16 + kq * 4;
    const float4 t = a.bias ? *reinterpret_cast<const float4*>(a.bias + co) : make_float4(0.f, 0.f, 0.f, 0.f);
    bv[m][0] = t.x; bv[m][1] = t.y; bv[m][2] = t.z; bv[m][3] = t.w;
    if constexpr (EPI == LD_EPI_RMS_RES) {
      const float4 g = *reinterpret_cast<const float4*>(a.g2 + co);
      g2v[m][0] = g.x; g2v[m][1] = g.y; g2v[m][2] = g.z; g2v[m][3] = g.w;
    }
  }
  constexpr bool HAS_OP3 = EPI == EPI_GN_TAIL_RES;      // (its own instantiation: 4*MT*NW registers only conv_fusion needs)
  RP op3[HAS_OP3 ? MP : 1][HAS_OP3 ? NW : 1];
  if constexpr (HAS_OP3) load_op(reinterpret_cast<const T*>(a.res), op3);

  T* out = reinterpret_cast<T*>(a.out);
  const bool q_part = (m0 * 16) < a.hidden;
  const bool k_part = a.kmax != nullptr && (m0 * 16) >= a.hidden && (m0 * 16) < 2 * a.hidden;
  float cmax[MT][4];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) cmax[m][r] = -INFINITY;
#pragma unroll
  for (int m = 0; m < MT; ++m) {                       // every epilogue operand has landed before the first store
    asm volatile("" ::"v"(bv[m][0]), "v"(bv[m][1]), "v"(bv[m][2]), "v"(bv[m][3]));
    if constexpr (EPI == LD_EPI_RMS_RES) asm volatile("" ::"v"(g2v[m][0]), "v"(g2v[m][1]), "v"(g2v[m][2]), "v"(g2v[m][3]));
  }
#pragma unroll
  for (int m = 0; m < MP; ++m) {
    if constexpr (HAS_OP2) {
#pragma unroll
      for (int j = 0; j < NW; ++j) pin_op(op2[m][j]);
    }
    if constexpr (HAS_OP3) {
#pragma unroll
      for (int j = 0; j < NW; ++j) pin_op(op3[m][j]);
    }
  }
#pragma unroll
  for (int j = 0; j < NW; ++j) {
    const int qq = (wv * NW + j) * 16 + px;
    const int p = p0 + qq;
    const bool valid = p < HW;
    const float rinv = a.rms_in ? s_rinv[qq] : 1.0f;
    float v[MT][4];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
#pragma unroll
      for (int r = 0; r < 4; ++r) v[m][r] = acc[m][j][r] * rinv + bv[m][r];
    }
    if (EPI == LD_EPI_QKV_LINEAR && q_part) {
      // softmax over the 32 channels of each head = 2 channel tiles x 4 regs x 4 kq lanes
#pragma unroll
      for (int m = 0; m < MT; m += 2) {   // MT is even; (m, m+1) = one head
        float mx = v[m][0];
#pragma unroll
        for (int r = 0; r < 4; ++r) { mx = fmaxf(mx, v[m][r]); mx = fmaxf(mx, v[m + 1][r]); }
        mx = kq4_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v[m][r] = DT<T>::precise ? expf(v[m][r] - mx) : __expf(v[m][r] - mx);
          v[m + 1][r] = DT<T>::precise ? expf(v[m + 1][r] - mx) : __expf(v[m + 1][r] - mx);
          sum += v[m][r] + v[m + 1][r];
        }
        sum = kq4_sum(sum);
        const float sc = DT<T>::precise ? a.q_scale / sum : a.q_scale * __builtin_amdgcn_rcpf(sum);
#pragma unroll
        for (int r = 0; r < 4; ++r) { v[m][r] *= sc; v[m + 1][r] *= sc; }
      }
    } else if (EPI == LD_EPI_QKV_FULL && q_part) {
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) v[m][r] *= a.q_scale;
    } else if (EPI == LD_EPI_RMS_RES) {
      float ss = 0.f;
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) ss = fmaf(v[m][r], v[m][r], ss);
      ss = kq4_sum(ss);
      const float inv = rms_rinv<DT<T>::precise>(ss);
#pragma unroll
      for (int m = 0; m < MT; ++m) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[m][r] = v[m][r] * inv * g2v[m][r];
      }
    }
    if (k_part && valid) {
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) cmax[m][r] = fmaxf(cmax[m][r], v[m][r]);
    }
    // (the second operands come from clamped, always-valid addresses: evaluated for every lane so that the 16-byte
    //  pairing below runs with the whole wave active; only the store is predicated)
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int co = (m0 + m) * 16 + kq * 4;
      if constexpr (EPI == LD_EPI_RMS_RES || EPI == LD_EPI_RES) {
        float o2[4];
        frag_op(op2, m, j, o2);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[m][r] += o2[r];
      } else if constexpr (TAIL) {
        float rv[4];
        frag_op(op2, m, j, rv);
        affine_act_n<DT<T>::precise, 4>(rv, s_tcoef + co, s_tcoef + a.Cout + co, a.tail.act);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[m][r] += rv[r];
        if constexpr (HAS_OP3) {                       // step-invariant half of res_conv (conv_fusion)
          float o3[4];
          frag_op(op3, m, j, o3);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[m][r] += o3[r];
        }
      }
    }
    T* opix = out + ((size_t)b * HW + (valid ? p : 0)) * a.Cout + m0 * 16;
    if constexpr (sizeof(T) == 2) {
      // the fragments of two adjacent m-tiles leave as ONE 16-byte store per lane (pair_frag16, common.hip.h)
      static_assert(MT % 2 == 0, "m-tiles are stored in pairs");
#pragma unroll
      for (int m = 0; m < MT; m += 2) {
        const uint4 w16 = pair_frag16<T>(v[m], v[m + 1]);
        if (valid) store16_out(reinterpret_cast<char*>(opix + m * 16) + pair_frag16_off(kq), w16);
      }
    } else if (valid) {
#pragma unroll
      for (int m = 0; m < MT; ++m) store4<T>(opix + m * 16 + kq * 4, v[m]);
    }
  }
  if (k_part) {
    // softmax_n(k) needs max_n k per (batch, channel) (ddpm.py:243): row-reduce over the 16 pixel
    // lanes (DPP), combine the 4 waves through LDS, then ONE 64-lane integer
    // atomicMax per workgroup into its stripe of the [B, stripes, hidden] buffer.
    float* s_cm = reinterpret_cast<float*>(s_x);          // the staging tile is dead after the K loop
    __syncthreads();
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float cm = wave16_max(cmax[m][r]);
        if (px == 0) s_cm[wv * 16 * MT + m * 16 + kq * 4 + r] = cm;
      }
    __syncthreads();
    if (tid < 16 * MT) {
      const float cm = fmaxf(fmaxf(s_cm[tid], s_cm[16 * MT + tid]), fmaxf(s_cm[32 * MT + tid], s_cm[48 * MT + tid]));
      const int stripe = blockIdx.x % LD_STAT_STRIPES;    // spread same-address atomics (see LD_STAT_STRIPES)
      if (cm > -INFINITY)
        atomicMax(a.kmax + ((size_t)b * LD_STAT_STRIPES + stripe) * a.hidden + m0 * 16 + tid - a.hidden, enc_max(cm));
    }
  }
}

template <typename T, int MT, int NW, int EPI, int G = 1>
__global__ __launch_bounds__(256) void conv1x1_kernel(Conv1Dev a) {
  conv1x1_body<T, MT, NW, EPI, G, false>(a, nullptr);
}
// the GroupNorm-tail launches: the tail operand as 12 preloaded dwords (cout_groups: Cout | groups << 16)
template <typename T, int MT, int NW, int EPI, int G>
__global__ __launch_bounds__(256) void conv1x1_taillead_kernel(const void* t_data, const double* t_stats, const float* t_gamma, const float* t_beta,
                                                               const float* t_film, int cout_groups, int hw, Conv1Dev rest) {
  Conv1Dev a{};
  a.tail.data = t_data; a.tail.stats = t_stats; a.tail.gamma = t_gamma; a.tail.beta = t_beta; a.tail.film = t_film;
  a.Cout = cout_groups & 0xffff; a.tail.C = a.Cout; a.tail.ld = a.Cout; a.tail.groups = (int)((unsigned)cout_groups >> 16);
  a.H = hw; a.W = 1;
  constexpr unsigned REST_OFF = ld_kernarg_offset<const void*, const double*, const float*, const float*, const float*, int, int>(alignof(Conv1Dev));
  static_assert(REST_OFF == 48, "conv1x1_taillead_kernel: leading arguments changed");
  typedef const char __attribute__((address_space(4)))* KChar;
  conv1x1_body<T, MT, NW, EPI, G, true>(a, (Conv1KernargPtr)((KChar)__builtin_amdgcn_kernarg_segment_ptr() + REST_OFF));
}

// the other grouped launches: both sources, the weights, the second operand and the geometry as 14 preloaded dwords
// (c0_flags: s[0].C | nsrc << 16 | unshuffle << 18 | rms_in << 19 | wsplit << 20; c1_ld1: s[1].C | s[1].ld << 16); code order unchanged
template <typename T, int MT, int NW, int EPI, int G>
__global__ __launch_bounds__(256) void conv1x1_klead_kernel(const void* data0, const void* w, const void* data1, const void* res, int H, int W,
                                                            int Cout, int c0_flags, int ld0, int c1_ld1, Conv1Dev rest) {
  Conv1Dev a{};
  a.s[0].data = data0; a.s[0].C = c0_flags & 0xffff; a.s[0].ld = ld0;
  a.s[1].data = data1; a.s[1].C = c1_ld1 & 0xffff; a.s[1].ld = (int)((unsigned)c1_ld1 >> 16);
  a.w = w; a.res = res; a.H = H; a.W = W; a.Cout = Cout;
  a.nsrc = (c0_flags >> 16) & 3; a.unshuffle = (c0_flags >> 18) & 1; a.rms_in = (c0_flags >> 19) & 1; a.wsplit = (c0_flags >> 20) & 1;
  constexpr unsigned REST_OFF = ld_kernarg_offset<const void*, const void*, const void*, const void*, int, int, int, int, int, int>(alignof(Conv1Dev));
  static_assert(REST_OFF == 56, "conv1x1_klead_kernel: leading arguments changed");
  typedef const char __attribute__((address_space(4)))* KChar;
  conv1x1_body<T, MT, NW, EPI, G, false, true>(a, (Conv1KernargPtr)((KChar)__builtin_amdgcn_kernarg_segment_ptr() + REST_OFF));
}

template <typename T, int MT, int NW, int EPI>
int launch_epi(const Conv1Dev& a, hipStream_t st) {
  constexpr int NPT = 64 * NW;
  const size_t tail = NPT * sizeof(float) + (EPI == LD_EPI_GN_TAIL || EPI == EPI_GN_TAIL_RES ? 2 * a.Cout * sizeof(float) + 64 * sizeof(double) : 0);
  const size_t chunk = 4 * NPT * 16 + MT * 1024;
  const int HW = a.H * a.W;
  dim3 grid((HW + NPT - 1) / NPT, a.Cout / (16 * MT), a.B);
  // the GroupNorm-tail launches of the sampling loop (FiLM shared by the batch): preloaded tail operand
  constexpr bool CAN_TLEAD = EPI == LD_EPI_GN_TAIL || EPI == EPI_GN_TAIL_RES;
  const bool tlead = CAN_TLEAD && ld_tuning().lead_args && a.tail.film_bstride == 0 && a.tail.film_tstride == 0 && a.Cout < 65536 && a.tail.groups < 32768 &&
                     a.tail.C == a.Cout && a.tail.ld == a.Cout && (long)HW < (1L << 31);
#define LD_C1_TLEAD_ARGS a.tail.data, a.tail.stats, a.tail.gamma, a.tail.beta, a.tail.film, a.Cout | (a.tail.groups << 16), HW, a
  if constexpr (CAN_TLEAD) {
    if (tlead && (a.group == KG || a.group == 2)) {
      const size_t lds = (size_t)a.group * chunk + tail;
      if (a.group == KG) {
        if (lds > 65536) LD_HIP(ld_allow_lds((conv1x1_taillead_kernel<T, MT, NW, EPI, KG>), lds));
        LD_LAUNCH((conv1x1_taillead_kernel<T, MT, NW, EPI, KG>), grid, dim3(256), lds, st, LD_C1_TLEAD_ARGS);
      } else {
        if (lds > 65536) LD_HIP(ld_allow_lds((conv1x1_taillead_kernel<T, MT, NW, EPI, 2>), lds));
        LD_LAUNCH((conv1x1_taillead_kernel<T, MT, NW, EPI, 2>), grid, dim3(256), lds, st, LD_C1_TLEAD_ARGS);
      }
      LD_LAUNCH_CHECK("conv1x1(GroupNorm tail, lead)");
      return LD_OK;
    }
  }
#undef LD_C1_TLEAD_ARGS
  if constexpr (!CAN_TLEAD) {
    const bool klead = ld_tuning().lead_args && (a.group == KG || a.group == 2) && a.w_bstride == 0 && a.s[0].C < 65536 && a.s[1].C < 65536 && a.s[1].ld < 65536;
    if (klead) {
      const int c0f = a.s[0].C | (a.nsrc << 16) | (a.unshuffle ? 1 << 18 : 0) | (a.rms_in ? 1 << 19 : 0) | (a.wsplit ? 1 << 20 : 0);
      const int c1l = (a.nsrc > 1 ? a.s[1].C : 0) | ((a.nsrc > 1 ? a.s[1].ld : 0) << 16);
      const size_t lds = (size_t)a.group * chunk + tail;
#define LD_C1_KLEAD_ARGS a.s[0].data, a.w, a.s[1].data, a.res, a.H, a.W, a.Cout, c0f, a.s[0].ld, c1l, a
      if (a.group == KG) {
        if (lds > 65536) LD_HIP(ld_allow_lds((conv1x1_klead_kernel<T, MT, NW, EPI, KG>), lds));
        LD_LAUNCH((conv1x1_klead_kernel<T, MT, NW, EPI, KG>), grid, dim3(256), lds, st, LD_C1_KLEAD_ARGS);
      } else {
        if (lds > 65536) LD_HIP(ld_allow_lds((conv1x1_klead_kernel<T, MT, NW, EPI, 2>), lds));
        LD_LAUNCH((conv1x1_klead_kernel<T, MT, NW, EPI, 2>), grid, dim3(256), lds, st, LD_C1_KLEAD_ARGS);
      }
#undef LD_C1_KLEAD_ARGS
      LD_LAUNCH_CHECK("conv1x1(grouped K, lead)");
      return LD_OK;
    }
  }
  if (a.group == KG) {
    const size_t lds = KG * chunk + tail;
    if (lds > 65536) LD_HIP(ld_allow_lds((conv1x1_kernel<T, MT, NW, EPI, KG>), lds));   // cached per device
    LD_LAUNCH((conv1x1_kernel<T, MT, NW, EPI, KG>), grid, dim3(256), lds, st, a);
    LD_LAUNCH_CHECK("conv1x1(grouped K)");
    return LD_OK;
  }
  if (a.group == 2) {
    const size_t lds = 2 * chunk + tail;
    if (lds > 65536) LD_HIP(ld_allow_lds((conv1x1_kernel<T, MT, NW, EPI, 2>), lds));
    LD_LAUNCH((conv1x1_kernel<T, MT, NW, EPI, 2>), grid, dim3(256), lds, st, a);
    LD_LAUNCH_CHECK("conv1x1(grouped K, pairs)");
    return LD_OK;
  }
  LD_LAUNCH((conv1x1_kernel<T, MT, NW, EPI>), grid, dim3(256), chunk + tail, st, a);
  LD_LAUNCH_CHECK("conv1x1");
  return LD_OK;
}

template <typename T, int MT, int NW>
int launch(const Conv1Dev& a, hipStream_t st) {
  switch (a.epi) {
    case LD_EPI_PLAIN: return launch_epi<T, MT, NW, LD_EPI_PLAIN>(a, st);
    case LD_EPI_QKV_LINEAR: return launch_epi<T, MT, NW, LD_EPI_QKV_LINEAR>(a, st);
    case LD_EPI_QKV_FULL: return launch_epi<T, MT, NW, LD_EPI_QKV_FULL>(a, st);
    case LD_EPI_RMS_RES: return launch_epi<T, MT, NW, LD_EPI_RMS_RES>(a, st);
    case LD_EPI_RES: return launch_epi<T, MT, NW, LD_EPI_RES>(a, st);
    default: return a.res ? launch_epi<T, MT, NW, EPI_GN_TAIL_RES>(a, st) : launch_epi<T, MT, NW, LD_EPI_GN_TAIL>(a, st);
  }
}

template <typename T>
int dispatch(const Conv1Dev& a0, hipStream_t st) {
  Conv1Dev a = a0;
  const int HW = a.H * a.W;
  {
    // grouped staging where the launch is a chain of dependent round trips over a small map: at most
    // c1_group_max_px pixels in the whole launch (default 64^2 x 8) and at least 4 K-chunks; c1_group = 0: off
    // (entries of the tuning table, include/localdiff_hip.h)
    const LdTuning& tn = ld_tuning();
    const long group_max_px = tn.c1_group_max_px;
    const int group_on = (int)tn.c1_group;
    const int ck = DT<T>::CK;
    const int nc0 = a.s[0].C / ck;
    const int nch = (a.unshuffle ? 4 * nc0 : nc0 + (a.nsrc > 1 ? a.s[1].C / ck : 0)) << a.wsplit;
    const int group_min_ch = (int)tn.c1_group_min_ch;
    a.group = (group_on && (long)HW * a.B <= group_max_px && nch >= group_min_ch) ? KG : 0;
    // every other launch with at least two chunks: pairs (one round trip per two chunks).  Until round 3 only up to
    // 65,536 pixels per launch (the 128^2 stage at 4 patches) -- "at 256^2 pairs change nothing" was measured while the
    // second-operand loads of those launches were still waited for one by one (finding 63); with that gone the 64->32 @256^2
    // tail launches gain: step 1.4293 -> 1.3978 ms (-2.2 %), cfg5 -2.0 %, 64 patches per GPU and one batch of 8 unchanged.
    const long pair_max_px = tn.c1_pair_max_px;
    if (group_on && !a.group && (long)HW * a.B <= pair_max_px && nch >= 2) a.group = 2;
  }
  if (a.epi == LD_EPI_RMS_RES) {
    switch (a.Cout) {
      case 32: return launch<T, 2, 2>(a, st);
      case 64: return launch<T, 4, 2>(a, st);
      case 128: return launch<T, 8, 2>(a, st);
      default: return ld_fail(LD_EINVAL, "ld_conv1x1: RMS_RES epilogue supports Cout 32/64/128 (got %d)", a.Cout);
    }
  }
  bool mt4 = (a.Cout % 64) == 0;
  // small maps: prefer 32-channel tiles while the 64-channel grid (128-pixel tiles) would leave CUs with at most
  // one workgroup -- the K loop is a chain of synchronous chunk loads and a second resident workgroup hides it
  // (256 since round 3: at 512 the two 192->128 @64^2 tail launches of a 4-patch sub-batch -- exactly 256 workgroups of
  //  64-channel tiles -- fell back to 512 workgroups of 32-channel tiles; with 64-channel tiles the step is 0.65 % faster,
  //  six alternating pairs; 128 and 384 measure like 512)
  const long small_min = ld_tuning().c1_small_min;
  if (mt4 && (long)((HW + 127) / 128) * (a.Cout / 64) * a.B < small_min) mt4 = false;
  const long blocks4 = (long)((HW + 255) / 256) * (a.Cout / (mt4 ? 64 : 32)) * a.B;
  const bool big = blocks4 >= 512;
  if (mt4) return big ? launch<T, 4, 4>(a, st) : launch<T, 4, 2>(a, st);
  return big ? launch<T, 2, 4>(a, st) : launch<T, 2, 2>(a, st);
}

}  // namespace

extern "C" int ld_conv1x1(const ld_conv1x1_args* p, void* stream) {
  LD_REQUIRE(p != nullptr, "ld_conv1x1: null args");
  LD_REQUIRE(p->nsrc == 1 || p->nsrc == 2, "ld_conv1x1: nsrc must be 1 or 2");
  LD_REQUIRE(ld_dtype_ok(p->dtype), "ld_conv1x1: bad dtype %d", p->dtype);
  LD_REQUIRE(p->Cout > 0 && p->Cout % 32 == 0, "ld_conv1x1: Cout %d must be a multiple of 32", p->Cout);
  LD_REQUIRE(p->B > 0 && p->H > 0 && p->W > 0 && p->weight && p->out, "ld_conv1x1: bad shape/null");
  LD_REQUIRE(!(p->unshuffle && p->nsrc != 1), "ld_conv1x1: unshuffle takes one source");
  LD_REQUIRE(p->epilogue >= LD_EPI_PLAIN && p->epilogue <= LD_EPI_GN_TAIL, "ld_conv1x1: bad epilogue");
  if (p->epilogue == LD_EPI_QKV_LINEAR || p->epilogue == LD_EPI_QKV_FULL)
    LD_REQUIRE(p->hidden > 0 && p->hidden % 64 == 0 && p->Cout == 3 * p->hidden,
               "ld_conv1x1: QKV epilogue needs Cout == 3*hidden, hidden %% 64 == 0");
  if (p->epilogue == LD_EPI_RMS_RES) LD_REQUIRE(p->g2 && p->residual, "ld_conv1x1: RMS_RES needs g2/residual");
  LD_REQUIRE((reinterpret_cast<uintptr_t>(p->bias) & 15) == 0 && (reinterpret_cast<uintptr_t>(p->g2) & 15) == 0,
             "ld_conv1x1: bias / g2 must be 16-byte aligned (read as float4)");
  if (p->epilogue == LD_EPI_RES) LD_REQUIRE(p->residual, "ld_conv1x1: RES needs residual");
  Conv1Dev a;
  for (int s = 0; s < p->nsrc; ++s) {
    LD_REQUIRE(p->src[s].data && p->src[s].C > 0 && p->src[s].C % 32 == 0,
               "ld_conv1x1: src[%d] null or C %% 32 != 0", s);
    LD_REQUIRE(p->src[s].gn_stats == nullptr && !p->src[s].upsample,
               "ld_conv1x1: GroupNorm/upsample prologues are 3x3-only");
    a.s[s] = to_dev(p->src[s]);
  }
  if (p->nsrc == 1) a.s[1] = a.s[0];
  a.nsrc = p->nsrc; a.unshuffle = p->unshuffle; a.rms_in = p->rms_in;
  a.w = p->weight; a.w_bstride = p->weight_bstride; a.bias = p->bias;
  a.epi = p->epilogue;
  a.hidden = (p->epilogue == LD_EPI_QKV_LINEAR || p->epilogue == LD_EPI_QKV_FULL) ? p->hidden : 0;
  a.q_scale = p->q_scale; a.g2 = p->g2; a.res = p->residual; a.out = p->out;
  a.kmax = (p->epilogue == LD_EPI_QKV_LINEAR) ? p->kmax_out : nullptr;
  if (p->epilogue == LD_EPI_GN_TAIL) {
    const ld_src& t = p->gn_tail;
    LD_REQUIRE(t.gn_groups <= 16, "ld_conv1x1: gn_tail groups %d > 16 (the stripe reduction uses 16 lanes per group)", t.gn_groups);
    LD_REQUIRE(t.data && t.gn_stats && t.gn_gamma && t.gn_beta && t.gn_groups > 0 && t.C == p->Cout &&
               t.C % t.gn_groups == 0 && (t.pix_stride == 0 || t.pix_stride == t.C) && !t.film,
               "ld_conv1x1: GN_TAIL operand incomplete");
    a.tail = to_dev(t);
  } else {
    a.tail = a.s[0];
  }
  a.B = p->B; a.H = p->H; a.W = p->W; a.Cout = p->Cout;
  a.group = 0;
  LD_REQUIRE(p->weight_terms >= 0 && p->weight_terms <= 2 && !(p->weight_terms == 2 && (!ld_dtype_16(p->dtype) || p->rms_in || p->weight_bstride)),
             "ld_conv1x1: weight_terms %d (2 needs 16-bit storage, no rms_in, no per-batch weights)", p->weight_terms);
  a.wsplit = p->weight_terms == 2 ? 1 : 0;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  return LD_DISPATCH(p->dtype, dispatch<T>(a, st));
}
