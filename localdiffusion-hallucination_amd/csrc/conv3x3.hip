// 3x3 convolution (pad 1) as an implicit GEMM on MFMA, NHWC, gfx950.
//
// Replaces nn.Conv2d(k=3,p=1) in Block.proj (ddpm.py:173), Upsample (:117), the last-stage
// convs (:372,:391) and BasicBlock (unet_model.py:20,24,30), with
//   - prologue fused into the input staging: channel concat of two sources (torch.cat,
//     ddpm.py:435-448), nearest x2 upsample (:116), GroupNorm-apply + FiLM + SiLU/ReLU of the
//     producer (:179-185, unet_model.py:21-22);
//   - epilogue: bias, GroupNorm statistics of the result (sum, sum^2 per (batch, group), fp64
//     atomics -- one per group per workgroup), NHWC store.
//
// Tiling.  Workgroup = 256 threads = 4 waves; output tile = (4*NW rows) x 16 cols of pixels x
// (16*MT) output channels.  Wave w owns rows [w*NW, (w+1)*NW) (each row = one 16-pixel MFMA
// column tile) and all MT channel tiles: acc[MT][NW] fragments of 16x16.
// K loop over 64-byte channel chunks (32 bf16 / 16 fp32 channels):
//   LDS input image  [kq 0..3][halo pixel q][16 B]   -- plane stride = multiple of 256 B, so the
//       16 lanes of a ds_read_b128 group (consecutive q, same kq) hit 16 distinct 16-B slots for
//       every tap offset: conflict-free without a swizzle;
//   LDS weights      [tap][m][lane][16 B]            -- already in fragment order in HBM
//       (ld_pack_conv_weight), read linearly.
// Inner order: dx outer (3*MT weight fragments live), then halo rows rr: one activation fragment
// feeds the (up to) 3 taps dy that touch it => 9*MT + 3*(NW+2) LDS reads per 9*MT*NW MFMAs.
#include "common.hip.h"
#include <stdlib.h>

int ld_conv3x3_c32_try(const ld_conv3x3_args* p, hipStream_t st);   // conv3x3_c32.hip
int ld_conv3x3_s32_try(const ld_conv3x3_args* p, hipStream_t st);   // conv3x3_s32.hip
#ifdef LD_DEBUG_VARIANTS
int ld_conv3x3_ksplit_try(const ld_conv3x3_args* p, hipStream_t st);   // tools/experiments/conv3x3_ksplit.hip (shelved, finding 52)
int ld_conv3x3_ring_try(const ld_conv3x3_args* p, hipStream_t st);     // tools/experiments/conv3x3_ring.hip (finding 58)
#endif

#include "conv3x3_body.hip.h"

namespace {

#ifdef LD_DEBUG_VARIANTS
int g_span_shape[1024][5];
#endif

template <typename T, int MT, int NW, bool DEEP, int DBG, bool SK = false, bool RAW = false, bool SIDE = false>
int launch_dbg(const Conv3Dev& a, hipStream_t st) {
  constexpr int TR = 4 * NW, HR = TR + 2, HC = 18;
  constexpr int NPIXP = (HR * HC + 15) / 16 * 16;
  const int ctot = a.s[0].C + (a.nsrc > 1 ? a.s[1].C : 0);
  const size_t lds = (SK ? 2 : 1) * (4 * NPIXP * 16 + (9 + (SIDE ? 1 : 0)) * MT * 1024) + 2 * ctot * sizeof(float) + 4 * 2 * 16 * MT * sizeof(double);
  Conv3Dev d = a;
  d.tiles_x = (a.W + 15) / 16;
  const int tiles_y = (a.H + TR - 1) / TR;
  dim3 grid(d.tiles_x * tiles_y, a.Cout / (16 * MT), a.B);
  if (lds > 65536) LD_HIP(ld_allow_lds(conv3x3_kernel<T, MT, NW, DEEP, DBG, SK, RAW, SIDE>, lds));   // cached per device
  LD_LAUNCH((conv3x3_kernel<T, MT, NW, DEEP, DBG, SK, RAW, SIDE>), grid, dim3(SK ? 512 : 256), lds, st, d.s[0].data, d.w, d.H, d.W, d.tiles_x,
            d.s[0].C, d.s[0].ld, d.s[0].ups, d.nsrc, d.wsplit, d.Cout, d);
  LD_LAUNCH_CHECK("conv3x3");
  return LD_OK;
}

// The LD_CONV_DEBUG ablation variants are separate instantiations (bf16, non-DEEP only) that exist only in a
// library built with -DLD_DEBUG_VARIANTS (build.sh --debug-variants): the production library carries neither their
// branches nor their code objects.
template <typename T, int MT, int NW, bool DEEP>
int launch(const Conv3Dev& a, hipStream_t st) {
#ifdef LD_DEBUG_VARIANTS
  if constexpr (std::is_same<T, bf16>::value && !DEEP) {
    switch (a.dbg & 255) {
      case 1: return launch_dbg<T, MT, NW, DEEP, 1>(a, st);
      case 2: return launch_dbg<T, MT, NW, DEEP, 2>(a, st);
      case 3: return launch_dbg<T, MT, NW, DEEP, 3>(a, st);
      case 4: return launch_dbg<T, MT, NW, DEEP, 4>(a, st);
      case 7: return launch_dbg<T, MT, NW, DEEP, 7>(a, st);
      case 8: return launch_dbg<T, MT, NW, DEEP, 8>(a, st);
      case 12: return launch_dbg<T, MT, NW, DEEP, 12>(a, st);
      case 15: return launch_dbg<T, MT, NW, DEEP, 15>(a, st);
      case 64: return launch_dbg<T, MT, NW, DEEP, 64>(a, st);
      case 128:                                        // launch spans: the same RAW / general split as production
        if (!a.s[0].stats && !(a.nsrc > 1 && a.s[1].stats)) return launch_dbg<T, MT, NW, DEEP, 128, false, true>(a, st);
        return launch_dbg<T, MT, NW, DEEP, 128>(a, st);
      default: break;
    }
  }
#endif
  if constexpr (!DEEP) {
    const bool raw = !a.s[0].stats && !(a.nsrc > 1 && a.s[1].stats);
    if (raw && ld_tuning().conv_raw) return launch_dbg<T, MT, NW, DEEP, 0, false, true>(a, st);
  }
  return launch_dbg<T, MT, NW, DEEP, 0>(a, st);
}

template <typename T>
int dispatch(const Conv3Dev& a, hipStream_t st) {
#ifdef LD_DEBUG_VARIANTS          // experiment-only: force a tile variant
  static const int force_mt = getenv("LD_CONV_MT") ? atoi(getenv("LD_CONV_MT")) : 0;
  static const int force_nw = getenv("LD_CONV_NW") ? atoi(getenv("LD_CONV_NW")) : 0;
#else
  constexpr int force_mt = 0, force_nw = 0;
#endif
  const LdTuning& tn = ld_tuning();
  bool mt4 = (a.Cout % 64) == 0;
  // a launch that would not even give every CU one 64-channel-tile workgroup uses 32-channel tiles instead
  // (128->128 @32^2, B=8: 128 workgroups -> 256; rocprofv3: 10.3 -> 8.3 us)
  const long mt4_min = tn.conv_mt4_min_wgs;
  if (mt4 && (long)((a.W + 15) / 16) * ((a.H + 7) / 8) * (a.Cout / 64) * a.B < mt4_min) mt4 = false;
  // enough workgroups to fill 256 CUs a couple of times over with the big tile?
  const long blocks16 = (long)((a.W + 15) / 16) * ((a.H + 15) / 16) * (a.Cout / (mt4 ? 64 : 32)) * a.B;
  // 64-channel tiles keep 2 pixel rows per wave: with the prefetch registers the 4-row variant
  // drops to one wave per SIMD and measured slower (64->64@128^2: 23.3 vs 19.7 us)
  const long big_min = tn.conv_big_min;
  bool big = blocks16 >= big_min && a.H >= 16 && !mt4;
  // 64 channels x 16 rows (<4,4>, two workgroups per CU): 0.375 KB of LDS fragment reads per MFMA instead of <4,2>'s 0.667 --
  // the throughput regime (dozens of patches per launch), where the LDS pipe and not a dependent chain is the bound
  if (mt4 && a.H >= 16 && blocks16 >= tn.conv_big4_min) big = true;
  if (force_mt == 2) mt4 = false;
  if (force_mt == 4 && (a.Cout % 64) == 0) mt4 = true;
  if (force_nw == 2) big = false;
  if (force_nw == 4) big = true;
#ifdef LD_DEBUG_VARIANTS          // experiment-only (finding 15): exists in a library built with --debug-variants
  static const int force_deep = getenv("LD_CONV_DEEP") ? atoi(getenv("LD_CONV_DEEP")) : -1;
#else
  constexpr int force_deep = -1;
#endif
  const int ck = sizeof(T) == 4 ? 16 : 32;
  const int nch = (a.s[0].C + (a.nsrc > 1 ? a.s[1].C : 0)) / ck;
  const long wg = (long)((a.W + 15) / 16) * ((a.H + 7) / 8) * (a.Cout / (mt4 ? 64 : 32)) * a.B;
  // measured: distance-2 prefetch is within noise of distance 1 on every small-map shape (the waits there are
  // barrier skew, not load latency), so it stays an opt-in experiment (LD_CONV_DEEP=1)
  bool deep = false;
  if (force_deep >= 0) deep = force_deep != 0;
  // split-K halves (opt-in, LD_CONV_SK=1: launches with >= 8 chunks and at most LD_CONV_SK_MAX_WGS workgroups;
  // LD_CONV_SK=2: every eligible launch).  Measured: 256->256 @32^2 18.0 -> 17.1 us, with the GroupNorm prologue
  // 25.5 -> 22.5, 512->256 46.0 -> 37.6; four-chunk launches and grids beyond one workgroup per CU lose.  Over a
  // step: +0.6 % for one batch of 8 on one stream, -1 % with two concurrent sub-batches (the second stream already
  // fills the gaps this variant closes), hence off by default.
  if constexpr (sizeof(T) == 2) {
    // SIDE: the block's res_conv as a second output (ld_conv3x3_args.side_*): RAW launches on the register tiles below 16 fragments
    // (<2,2>, <4,2>): a launch that would take a 16-row tile takes the 8-row tile of its channel width instead
    if (a.w2) {       // (<2,4> with the side accumulators does not fit 168 registers: 156 bytes of scratch -- not instantiated)
      if (mt4) return launch_dbg<T, 4, 2, false, 0, false, true, true>(a, st);
      return launch_dbg<T, 2, 2, false, 0, false, true, true>(a, st);
    }
  }
  const int force_sk = (int)tn.conv_sk;
  const long sk_max_wgs = tn.conv_sk_max_wgs;
  bool sk = force_sk >= 1 && sizeof(T) == 2 && !big && !deep && ((nch >= 8 && wg <= sk_max_wgs) || force_sk == 2);
  if constexpr (sizeof(T) == 2) {
    if (sk) return mt4 ? launch_dbg<T, 4, 2, false, 0, true>(a, st) : launch_dbg<T, 2, 2, false, 0, true>(a, st);
  }
#ifdef LD_DEBUG_VARIANTS
  if (deep && !big) return mt4 ? launch<T, 4, 2, true>(a, st) : launch<T, 2, 2, true>(a, st);
#endif
  (void)deep;
  if (mt4) return big ? launch<T, 4, 4, false>(a, st) : launch<T, 4, 2, false>(a, st);
  return big ? launch<T, 2, 4, false>(a, st) : launch<T, 2, 2, false>(a, st);
}

}  // namespace

#ifdef LD_DEBUG_VARIANTS
// Debug hook (not part of the public ABI): LD_CONV_DEBUG=128 launch spans.  spans[slot] = {start, end} on the 100 MHz
// real-time clock, shapes[slot] = {B, H, W, Cin, Cout} of the call that owns the slot (1,024 slots, round robin).
extern "C" int ld_debug_conv_spans(unsigned long long* spans, int* shapes) {
  LD_HIP(hipDeviceSynchronize());
  LD_HIP(hipMemcpyFromSymbol(spans, HIP_SYMBOL(g_conv_span), sizeof(unsigned long long) * 2048));
  for (int i = 0; i < 1024 * 5; ++i) shapes[i] = g_span_shape[i / 5][i % 5];
  return LD_OK;
}
#endif

// Debug hook (not part of the public ABI): cycle stamps of the last LD_CONV_DEBUG=64 launch (24 uint64).
extern "C" int ld_debug_conv_trace(unsigned long long* host) {
  LD_HIP(hipDeviceSynchronize());
  LD_HIP(hipMemcpyFromSymbol(host, HIP_SYMBOL(g_conv_trace), sizeof(unsigned long long) * 24));
  return LD_OK;
}

extern "C" int ld_conv3x3(const ld_conv3x3_args* p, void* stream) {
  LD_REQUIRE(p != nullptr, "ld_conv3x3: null args");
  LD_REQUIRE(p->nsrc == 1 || p->nsrc == 2, "ld_conv3x3: nsrc must be 1 or 2 (got %d)", p->nsrc);
  LD_REQUIRE(ld_dtype_ok(p->dtype), "ld_conv3x3: bad dtype %d", p->dtype);
  LD_REQUIRE(p->Cout > 0 && p->Cout % 32 == 0, "ld_conv3x3: Cout %d must be a multiple of 32", p->Cout);
  LD_REQUIRE(p->B > 0 && p->H > 0 && p->W > 0, "ld_conv3x3: bad shape");
  LD_REQUIRE(p->weight && p->bias && p->out, "ld_conv3x3: null weight/bias/out");
  Conv3Dev a;
  a.nsrc = p->nsrc;
  for (int s = 0; s < p->nsrc; ++s) {
    const ld_src& S = p->src[s];
    LD_REQUIRE(S.data != nullptr, "ld_conv3x3: src[%d] null", s);
    LD_REQUIRE(S.C > 0 && S.C % 32 == 0, "ld_conv3x3: src[%d].C=%d must be a multiple of 32", s, S.C);
    if (S.upsample) LD_REQUIRE(p->H % 2 == 0 && p->W % 2 == 0, "ld_conv3x3: upsample needs even H,W");
    if (S.gn_stats) {
      LD_REQUIRE(S.gn_groups <= 16, "ld_conv3x3: src[%d] gn_groups %d > 16 (the stripe reduction uses 16 lanes per group)", s, S.gn_groups);
      LD_REQUIRE(S.gn_gamma && S.gn_beta && S.gn_groups > 0 && S.C % S.gn_groups == 0,
                 "ld_conv3x3: src[%d] GroupNorm prologue incomplete", s);
    }
    a.s[s] = to_dev(S);
  }
  if (p->nsrc == 1) a.s[1] = a.s[0];
  if (p->out_stats) {
    LD_REQUIRE(p->out_groups > 0 && p->Cout % p->out_groups == 0 && (p->Cout / p->out_groups) <= 32,
               "ld_conv3x3: out_groups %d incompatible with Cout %d", p->out_groups, p->Cout);
  }
  a.w = p->weight; a.bias = p->bias; a.out = p->out; a.ostats = p->out_stats;
  a.ogroups = p->out_groups > 0 ? p->out_groups : 1;
  a.B = p->B; a.H = p->H; a.W = p->W; a.Cout = p->Cout; a.t_ptr = p->t_ptr; a.tiles_x = 0;
  a.addend = p->addend;
  a.w2 = p->side_weight; a.bias2 = p->side_bias; a.out2 = p->side_out;
  if (p->side_weight) {
    LD_REQUIRE(ld_dtype_16(p->dtype), "ld_conv3x3: the side (res_conv) output needs 16-bit storage");
    LD_REQUIRE(p->side_bias && p->side_out, "ld_conv3x3: side_weight without side_bias / side_out");
    LD_REQUIRE(p->weight_terms != 2 && !p->addend, "ld_conv3x3: the side output does not combine with two-term weights or an addend");
    for (int s = 0; s < p->nsrc; ++s)
      LD_REQUIRE(!p->src[s].gn_stats && !p->src[s].upsample, "ld_conv3x3: the side output needs raw, un-resampled sources (a ResnetBlock's block1)");
  }
  LD_REQUIRE(p->weight_terms >= 0 && p->weight_terms <= 2 && !(p->weight_terms == 2 && !ld_dtype_16(p->dtype)),
             "ld_conv3x3: weight_terms %d (2 needs 16-bit storage)", p->weight_terms);
  a.wsplit = p->weight_terms == 2 ? 1 : 0;
  a.dbg = 0;
#ifdef LD_DEBUG_VARIANTS          // ablation / trace variants exist only in a library built with --debug-variants
  static const int dbg = getenv("LD_CONV_DEBUG") ? atoi(getenv("LD_CONV_DEBUG")) : 0;
  a.dbg = dbg;
  if (dbg == 128) {                                    // launch spans: a slot per call, its shape kept for the reader
    static int calls = 0;
    const int slot = calls++ & 1023;
    g_span_shape[slot][0] = p->B; g_span_shape[slot][1] = p->H; g_span_shape[slot][2] = p->W;
    g_span_shape[slot][3] = p->src[0].C + (p->nsrc > 1 ? p->src[1].C : 0); g_span_shape[slot][4] = p->Cout;
    a.dbg = 128 | (slot << 8);
  }
#endif
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (p->side_weight) {         // (only the generic kernel has the side output)
    ld_count(LD_COUNTER_CONV3X3_GENERIC);
    return LD_DISPATCH16(p->dtype, dispatch<T>(a, st));
  }
  if (p->weight_terms != 2) {   // (the persistent kernel keeps ONE chunk of weights in registers)
    const int rc = ld_conv3x3_c32_try(p, st);        // persistent LDS-DMA kernel for the Cout=32 stages (>= 2,048 tiles)
    if (rc != 0) return rc < 0 ? rc : LD_OK;
  }
  {
    const int rc = ld_conv3x3_s32_try(p, st);        // lean kernel for the Cout=32 stages (the ResBlock conv path)
    if (rc != 0) return rc < 0 ? rc : LD_OK;
  }
#ifdef LD_DEBUG_VARIANTS
  {                                          // experiment: both operands by LDS-DMA, one barrier per chunk
    const int rc = ld_conv3x3_ring_try(p, st);
    if (rc != 0) return rc < 0 ? rc : LD_OK;
  }
  static const int ksplit = getenv("LD_CONV_KSPLIT") ? atoi(getenv("LD_CONV_KSPLIT")) : 0;
  if (ksplit && p->weight_terms != 2) {                                       // shelved experiment: weights in registers, K split over waves
    const int rc = ld_conv3x3_ksplit_try(p, st);
    if (rc != 0) return rc < 0 ? rc : LD_OK;
  }
#endif
  ld_count(LD_COUNTER_CONV3X3_GENERIC);
  return LD_DISPATCH(p->dtype, dispatch<T>(a, st));
}
