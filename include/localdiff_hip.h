/*
 * localdiff_hip.h -- C-ABI of the MI355X (gfx950) local-diffusion sampling hot path.
 *
 * The reference (edshkim98/LocalDiffusion-Hallucination) has no FFI boundary of its own: its hot
 * path sits behind two Python nn.Module surfaces, Unet.forward (ddpm.py:404-451) and
 * GaussianDiffusion.sample / p_sample_loop / ddim_sample / p_sample (ddpm.py:841-1125), whose
 * bodies are stock ATen ops.  This library is what a ctypes binding on the reference side would
 * call instead of those ATen ops; every entry point names the reference lines
 * whose arithmetic it replaces.
 *
 * Conventions
 *   - plain pointers and sizes only; all pointers are DEVICE pointers unless marked "host".
 *   - every launch function takes the HIP stream to enqueue on (void*, a hipStream_t) and returns
 *     0 on success or a negative LD_E* code; ld_last_error() gives a thread-local message.
 *   - no allocation and no synchronisation inside launch functions, so they may be captured into a HIP graph
 *     (ld_graph_*).  The library's only process-wide mutable state is what this header documents further down: the
 *     launch-routing table (ld_tuning_set / ld_tuning_get: read by the launch functions, written only by an explicit
 *     call or once from the environment), the launch-routing counters (ld_counter: diagnostics, relaxed atomics), the
 *     per-(kernel, device) LDS-limit cache and an open timing session (ld_timing_*); none of it changes what a launch
 *     computes, only which kernel variant computes it.
 *   - internal activations are NHWC (channels-last) in the storage dtype (LD_F32, LD_BF16 or LD_F16),
 *     accumulation is always fp32 (the 16-bit types run v_mfma_f32_16x16x32_bf16 / _f16 at the same rate; fp16
 *     keeps 10 mantissa bits against bf16's 7, at a range of 6e-8 .. 65504); tensors that cross the reference's API (x_t, cond, mask, model
 *     output) are NCHW fp32 exactly as the reference holds them.
 *   - "t_ptr" arguments are device pointers to the current timestep index (int32).  Kernels read
 *     the step through them so that one captured graph can be replayed for every timestep; pass
 *     NULL to mean row 0 of the table argument.
 */
#ifndef LOCALDIFF_HIP_H
#define LOCALDIFF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LD_OK 0
#define LD_EINVAL (-1)   /* bad argument / unsupported shape */
#define LD_EHIP (-2)     /* HIP runtime error (message has hipGetErrorString) */
#define LD_ETIMEOUT (-3) /* a deadline passed (ld_comm_init_timeout and the calls on the communicator it made) */

#define LD_F32 0
#define LD_BF16 1
#define LD_F16 2

/* GroupNorm statistics buffers are [B, LD_STAT_STRIPES, groups, 2] fp64 (sum, sum of squares):
 * a producer workgroup adds into stripe (workgroup index % LD_STAT_STRIPES) so that the 256
 * workgroups of one image do not serialise on 16 addresses (measured: 23 us per launch at
 * 256x256 with one stripe); the consumer sums the stripes while building its coefficients. */
#define LD_STAT_STRIPES 16

#define LD_ACT_NONE 0
#define LD_ACT_SILU 1
#define LD_ACT_RELU 2

/* ---- runtime ------------------------------------------------------------------------------ */
const char* ld_last_error(void);
int ld_version(void);
/* host out-params: device name (buf, n), compute units, bytes of global memory */
int ld_device_info(char* name, int name_len, int* compute_units, int64_t* global_mem_bytes);
/* HIP-graph capture of everything enqueued on `stream` between begin and end (no tracing compiler:
 * the host issues the step once, the graph replays it T times). */
int ld_graph_begin(void* stream);
int ld_graph_end(void* stream, void** graph_exec_out);
int ld_graph_launch(void* graph_exec, void* stream);
int ld_graph_destroy(void* graph_exec);
/* hipMemsetAsync(ptr, 0, bytes) on the stream (statistics arenas are zeroed once per forward) */
int ld_memset_zero(void* ptr, size_t bytes, void* stream);
/* hipMemsetAsync(ptr, value & 0xff, bytes): the activation pool's verify mode poisons dead buffers with 0xFF (NaN in every storage type) */
int ld_memset_bytes(void* ptr, int value, size_t bytes, void* stream);
/* event timing on the stream the kernels run on (bench.py's roofline leg) */
/* Per-launch timing session: between begin and end every kernel launched by this library (from the calling
 * process, any stream) carries its own start/stop events; ld_timing_count() = launches so far, ld_timing_end fills
 * ms[i] with launch i's execution time (dispatch begin -> end, what rocprofv3 --kernel-trace reports). */
int ld_timing_begin(int max_launches);
int ld_timing_count(void);
int ld_timing_end(float* ms, int cap, int* count);
/* the same, with the launches' positions on the device clock: start_ms[i] / stop_ms[i] = begin / end of launch i relative
 * to the begin of launch 0 (one clock for every stream: the overlap of launches on different streams can be read off) */
int ld_timing_end_abs(float* start_ms, float* stop_ms, int cap, int* count);
int ld_event_create(void** ev_out);
int ld_event_record(void* ev, void* stream);
int ld_event_elapsed_ms(void* ev_start, void* ev_stop, float* ms_out); /* synchronises on stop */
int ld_event_destroy(void* ev);
/* make `stream` wait for `ev` (fork/join of independent branches on two streams, e.g. a ResnetBlock's
 * res_conv beside its 3x3 convs; also captured as graph edges) */
int ld_stream_wait_event(void* stream, void* ev);
/* ---- tuning: the launch-routing thresholds of the library, ONE table per process ------------------------------------
 * Every threshold a dispatcher consults (which tile variant, grouped K staging, the persistent C=32 convolution, ...)
 * is an entry of this table: defaults compiled in, an environment override LD_<NAME IN CAPITALS> read ONCE when the
 * table is first used, and explicit control through ld_tuning_set (what localdiffusion_hallucination_amd.tuning.Tuning
 * applies).  Names (defaults): c1_group (1), c1_group_max_px (32768), c1_group_min_ch (4), c1_pair_max_px (2^40),
 * c1_small_min (256), conv_raw (1), conv_mt4_min_wgs (256), conv_big_min (512), conv_sk (0), conv_sk_max_wgs (256),
 * conv_c32 (0: the persistent LDS-DMA kernel is retired from the default routing, finding 99), conv_c32_min_tiles (2048), gn_frags_per_block (512), fold_split_min (32), attn_split_max_wgs (256), attn_split_min_n
 * (2048; and the two-key-group kernel is only taken when the second group owns a key: n > tile size), lead_args (1: gn_apply /
 * conv1x1 launches that qualify use the kernels with preloaded leading arguments), conv_s32 (3: bit mask of the launches the lean
 * Cout = 32 kernel of conv3x3_s32.hip takes -- 1 single-chunk without prologue, 2 with the GroupNorm prologue, 4 two-chunk),
 * conv_s32_min_tiles (1024: 16 x 16 tiles per launch from which it is used), attn_xcd_map (1: ld_attention's XCD-aware workgroup
 * order), conv_big4_min (256: workgroups of the 64-channel x 16-row tile from which ld_conv3x3 uses it instead of 64 x 8 rows).
 * Values change routing, never
 * results beyond the summation order of a tile variant.  Unknown name: LD_EINVAL.  Not thread-safe against concurrent
 * launches (set it before launching).  The reference has no counterpart (its tuning is cuDNN's). */
int ld_tuning_set(const char* name, long long value);
int ld_tuning_get(const char* name, long long* value_out);
int ld_tuning_count(void);
const char* ld_tuning_name(int index);              /* NULL past the end */
/* Profiler-visible phase markers: nested roctx ranges (host side) around the phases of a sample -- "encoder",
 * "step", "exchange" -- so that a rocprofv3 --marker-trace of sample() is readable.  The reference's only hook is the
 * wall-clock timer around sample() (test.py:392-415).  No-ops when no roctx library can be loaded (LD_NO_ROCTX=1:
 * never try). */
/* Host-side counters of how the dispatchers routed launch calls since the library was loaded (a call made during
 * graph capture counts once, its replays do not): lets a test assert WHICH kernel a shape ran on. */
#define LD_COUNTER_CONV3X3_C32 0      /* ld_conv3x3 calls taken by the persistent LDS-DMA kernel (conv3x3_c32.hip) */
#define LD_COUNTER_CONV3X3_GENERIC 1  /* ... by the register-staged generic kernel (conv3x3.hip) */
#define LD_COUNTER_CONV3X3_S32 2      /* ... by the lean Cout = 32 large-map kernel (conv3x3_s32.hip) */
#define LD_COUNTER_MAX 8
long long ld_counter(int which);
int ld_range_push(const char* name /* host string */);
int ld_range_pop(void);

/* ---- one input of a convolution, with an optional normalise-on-load prologue --------------- */
/* Replaces the separate GroupNorm / FiLM / SiLU / ReLU / concat / nearest-upsample passes of
 * Block.forward (ddpm.py:177-186), ResnetBlock.forward (:200-212), Upsample (:114-118),
 * torch.cat (:435,439,442,448) and BasicBlock (unet_model.py:19-26): the consumer convolution
 * applies  y = act(x * a[b,c] + s[b,c])  while staging its input tile, where a,s come from the
 * producer's GroupNorm statistics (sum, sum of squares per (batch, group), fp64), gamma/beta and
 * the FiLM (scale+1, shift) row of this timestep. */
typedef struct ld_src {
  const void* data;        /* NHWC [B, Hs, Ws, C], storage dtype */
  int32_t C;               /* channels (multiple of 32) */
  int32_t pix_stride;      /* elements between consecutive pixels; 0 = C (dense). Lets a conv read a
                              channel slice, e.g. q = channels [0,hidden) of a qkv tensor */
  int32_t upsample;        /* 1: source is [B, H/2, W/2, C], read with nearest x2 */
  const double* gn_stats;  /* [B, LD_STAT_STRIPES, groups, 2] or NULL = no prologue */
  const float* gn_gamma;   /* [C] */
  const float* gn_beta;    /* [C] */
  int32_t gn_groups;
  int32_t act;             /* LD_ACT_* applied after the affine */
  const float* film;       /* [rows, 2*C] (scale | shift) or NULL */
  int32_t film_tstride;    /* floats between timestep rows (0 if per-batch rows only) */
  int32_t film_bstride;    /* floats between batch rows (0 if shared by the batch) */
} ld_src;

/* ---- 3x3 convolution, pad 1, implicit GEMM on MFMA ----------------------------------------- */
/* nn.Conv2d(k=3,p=1): Block.proj (ddpm.py:173), Upsample conv (:117), stage convs (:372,:391),
 * BasicBlock convs (unet_model.py:20,24,30).  Epilogue adds bias and (optionally) accumulates the
 * GroupNorm statistics of the result for the next layer (ddpm.py:174 / unet_model.py:21,25,31). */
typedef struct ld_conv3x3_args {
  ld_src src[2];           /* channel-concatenated inputs (torch.cat order) */
  int32_t nsrc;
  const void* weight;      /* packed by ld_pack_conv_weight, storage dtype */
  const float* bias;       /* [Cout] fp32 */
  void* out;               /* NHWC [B,H,W,Cout] */
  double* out_stats;       /* [B, LD_STAT_STRIPES, out_groups, 2] accumulated (caller zeroes) or NULL */
  int32_t out_groups;
  int32_t B, H, W, Cout;   /* Cout multiple of 32 */
  const int32_t* t_ptr;
  int32_t dtype;
  const void* addend;      /* optional NHWC [B,H,W,Cout] tensor (storage dtype) added to conv + bias BEFORE the     */
                           /* statistics: the step-invariant half of a convolution over a concatenation whose     */
                           /* second operand does not change between reverse steps (conv_fusion, ddpm.py:434-436) */
  int32_t weight_terms;    /* 0 / 1: `weight` from ld_pack_conv_weight; 2: two-term weights from                    */
                           /* ld_pack_conv_weight_terms(..., 2) (16-bit storage): x*hi + x*lo, weights exact to 2^-17 */
  const void* side_weight; /* optional second output of the launch (16-bit storage, raw sources): a 1x1 convolution of the  */
  const float* side_bias;  /* SAME concatenated input -- the ResnetBlock's res_conv (ddpm.py:198, 212) beside its block1     */
  void* side_out;          /* convolution.  side_weight [Cout, Cin] packed by ld_pack_conv_weight(ksize 1), side_bias [Cout],  */
                           /* side_out NHWC [B,H,W,Cout] = W_side x + side_bias (no statistics).  NULL = none.              */
} ld_conv3x3_args;
int ld_conv3x3(const ld_conv3x3_args* args, void* stream);

/* ---- 1x1 convolution (GEMM over channels) -------------------------------------------------- */
#define LD_EPI_PLAIN 0      /* out = W x + b                                                    */
#define LD_EPI_QKV_LINEAR 1 /* LinearAttention.to_qkv (ddpm.py:239-245): first `hidden` outputs   */
                            /* (q) get softmax over each head's 32 channels times dim_head^-0.5  */
#define LD_EPI_QKV_FULL 2   /* Attention.to_qkv (ddpm.py:276) with attend.py:98's scale folded   */
                            /* into q                                                           */
#define LD_EPI_RMS_RES 3    /* to_out conv + RMSNorm + residual (ddpm.py:229-232,251,425,444)    */
#define LD_EPI_RES 4        /* to_out conv + residual (ddpm.py:269,282,425,431)                  */
#define LD_EPI_GN_TAIL 5    /* ResnetBlock tail fused into its res_conv (ddpm.py:198,210-212);    */
                            /* `residual`, if given, is added too (step-invariant half of res_conv) */
                            /* out = res_conv(x) + act(GroupNorm(gn_tail))  -- gn_tail is block2's */
                            /* raw conv output with its statistics                               */
typedef struct ld_conv1x1_args {
  ld_src src[2];
  int32_t nsrc;
  int32_t unshuffle;       /* 1: Downsample (ddpm.py:120-124): src[0] is [B,2H,2W,C], K = 4C in  */
                           /*    (p1,p2,c) order (weights repacked accordingly)                  */
  int32_t rms_in;          /* 1: RMSNorm on the input (ddpm.py:131-132): columns scaled by        */
                           /*    1/max(||x_p||,1e-12); g*sqrt(C) is folded into `weight`          */
  const void* weight;
  int64_t weight_bstride;  /* bytes between per-batch weight sets (linear attention) or 0        */
  const float* bias;       /* [Cout] or NULL */
  int32_t epilogue;        /* LD_EPI_* */
  int32_t hidden;          /* heads*dim_head for the QKV epilogues */
  float q_scale;           /* dim_head^-0.5 */
  const float* g2;         /* [Cout] g*sqrt(Cout) for LD_EPI_RMS_RES */
  const void* residual;    /* NHWC [B,H,W,Cout] for *_RES */
  ld_src gn_tail;          /* LD_EPI_GN_TAIL: NHWC [B,H,W,Cout] + GroupNorm prologue fields (no FiLM) */
  void* out;
  uint32_t* kmax_out;      /* LD_EPI_QKV_LINEAR only, optional: [B, LD_STAT_STRIPES, hidden] order-encoded running max
                              of the k channels over pixels (integer atomicMax; caller zeroes) -- see ld_linattn_kmax */
  int32_t B, H, W, Cout;
  int32_t dtype;
  int32_t weight_terms;    /* as in ld_conv3x3_args (not with rms_in or per-batch weights) */
} ld_conv1x1_args;
int ld_conv1x1(const ld_conv1x1_args* args, void* stream);

/* Repack an OIHW fp32 convolution weight (device) into the MFMA fragment order the kernels read.
 * ksize 1 or 3.  `scale_in` (optional, [Cin]) multiplies input channel c (RMSNorm g*sqrt(C)).
 * unshuffle=1 reorders K from (c,p1,p2) to (p1,p2,c).  out must hold Cout*Cin*k*k elements. */
int ld_pack_conv_weight(const float* w_oihw, const float* scale_in, void* out, int cout, int cin,
                        int ksize, int unshuffle, int dtype, void* stream);
/* terms = 2 (16-bit storage): two-term weights W = hi + lo, hi = round(W), lo = round(W - hi); out holds twice the
 * elements, every K-chunk of hi followed by its chunk of lo.  Consumed by ld_conv3x3 / ld_conv1x1 with
 * weight_terms = 2 at twice the matrix work: rounding the WEIGHTS to 16 bits is what separates a 16-bit sampling chain
 * from the reference's fp32 weights (the same perturbation at every reverse step; activation roundings average out). */
int ld_pack_conv_weight_terms(const float* w_oihw, const float* scale_in, void* out, int cout, int cin,
                              int ksize, int unshuffle, int dtype, int terms, void* stream);

/* ---- small-Cin direct convolution from an NCHW fp32 image --------------------------------- */
/* init_conv 7x7 (ddpm.py:319,413) and the first BasicBlock convs (unet_model.py:20,30) whose
 * Cin is 1 or 3.  Cout must be 32.  Output NHWC storage dtype (+ optional GN statistics). */
int ld_conv_image(const float* x_nchw, const float* w_oihw, const float* bias, void* out,
                  double* out_stats, int out_groups, int B, int Cin, int H, int W, int ksize,
                  int dtype, void* stream);

/* init_conv 7x7 (ddpm.py:319,413) for 16-bit storage as an implicit GEMM on MFMA: x NCHW fp32 [B,Cin<=3,H,W] ->
 * out NHWC bf16 / fp16 [B,H,W,32].  Image and weights are split into bf16 hi+lo parts (three MFMA products, fp32
 * accumulate), so the result equals the fp32-FMA kernel of ld_conv_image up to the rounding of the stored bf16.
 * w_packed: ld_stem_packed_bytes() bytes written once per model by ld_pack_stem_weight from the OIHW fp32 weight. */
size_t ld_stem_packed_bytes(void);
int ld_pack_stem_weight(const float* w_oihw /*[32,Cin,7,7]*/, void* out_packed, int Cin, void* stream);
int ld_conv_stem(const float* x, const void* w_packed, const float* bias, void* out, int B, int Cin, int H, int W,
                 int dtype /* LD_BF16 or LD_F16: type of the stored output */, void* stream);
/* ld_step_begin_film's arguments (below) as a struct, and ld_conv_stem with that head-of-step work folded into the SAME
 * launch: a few extra workgroups zero the arenas, move the step counter and copy the timestep's FiLM row beside the
 * convolution, which reads none of them -- init_conv is the first operation of Unet.forward (ddpm.py:413) and no
 * launch before the third of an evaluation consumes statistics or FiLM, so an evaluation loses its first launch. */
typedef struct ld_step_begin_args {
  void* zero_a; size_t bytes_a; void* zero_b; size_t bytes_b;
  int32_t* t_ptr; int delta; int32_t* idx_ptr; const int32_t* t_table;
  const float* film_rows; int row_floats; float* film_cur;
} ld_step_begin_args;
int ld_conv_stem_begin(const float* x, const void* w_packed, const float* bias, void* out, int B, int Cin, int H, int W,
                       int dtype, const ld_step_begin_args* begin, void* stream);


/* ---- GroupNorm apply (+FiLM) + activation + residual, optional second normalised input ----- */
/* ResnetBlock tail  h = SiLU(GN(conv2)) + res(x)  (ddpm.py:210-212) and BasicBlock tail
 * ReLU(GN(conv2) + GN(conv_id)) followed by MaxPool2d(2) (unet_model.py:38-51,120,123,129). */
typedef struct ld_gn_apply_args {
  ld_src a;                /* first input, prologue required */
  ld_src b;                /* optional second input (data NULL = none); gn_stats NULL = raw add */
  int32_t final_act;       /* LD_ACT_* after the sum */
  int32_t pool;            /* 1: 2x2 max-pool the result (out is [B,H/2,W/2,C]) */
  void* out;
  int32_t B, H, W;         /* input spatial size */
  const int32_t* t_ptr;
  int32_t dtype;
} ld_gn_apply_args;
int ld_gn_apply(const ld_gn_apply_args* args, void* stream);

/* ---- attention ---------------------------------------------------------------------------- */
/* Linear attention core (ddpm.py:243,247,249) on a qkv tensor [B, n, 3*hidden] whose q part was
 * already soft-maxed by ld_conv1x1(LD_EPI_QKV_LINEAR):
 *   1. ld_linattn_kmax:  kmax[b, c] = max_n k[b, n, c] as an order-preserving uint32 code, combined with
 *                        integer atomicMax into a zeroed [B, LD_STAT_STRIPES, hidden] buffer (softmax over n; the
 *                        consumer takes the max over the stripes).
 *                        The product path gets the same buffer for free from ld_conv1x1's kmax_out.
 *   2. ld_linattn_ctx:   partial  ctx[d,e] = sum_n exp(k-max) v,  Z[d] = sum_n exp(k-max) per pixel chunk
 *   3. ld_linattn_ctx_reduce: ctxn[b,h,d,e] = sum_chunks ctx / sum_chunks Z[d]   [B,heads,32,32]
 *   4. ld_linattn_fold:  M_b = W_out . ctxn^T  packed as a per-batch 1x1 weight, so that
 *                        to_out(ctx^T q) becomes ONE 1x1 convolution over q (ld_conv1x1). */
int ld_linattn_kmax(const void* qkv, uint32_t* kmax_enc, int B, int n, int heads, int dim_head,
                    int dtype, void* stream);
int ld_linattn_ctx(const void* qkv, const uint32_t* kmax_enc, float* ctx_part,
                   int B, int n, int heads, int dim_head, int nchunks, int dtype, void* stream);
int ld_linattn_ctx_reduce(const float* ctx_part, int nchunks, float* ctxn, int B, int heads,
                          int dim_head, void* stream);
/* perm=0: standard k=1 packing (consumed by ld_conv1x1); perm=1 (bf16 / fp16): chained-MFMA operand order
 * consumed by ld_linattn_out. */
int ld_linattn_fold(const float* ctxn, const float* w_out /*[C,hidden] fp32*/,
                    void* w_packed /*[B] packed C x hidden*/, int B, int C, int heads, int dim_head,
                    int perm, int dtype, void* stream);
/* Steps 3 + 4 in one launch (grid heads x B x 4): chunk partials -> 8 rows of the normalised context (kept in
 * LDS) -> the matching 8 columns of the packed M_b.  Same arithmetic and output layout as the two calls above. */
int ld_linattn_ctxfold(const float* ctx_part, int nchunks, const float* w_out /*[C,hidden] fp32*/,
                       void* w_packed /*[B] packed C x hidden*/, int B, int C, int heads, int dim_head,
                       int perm, int dtype, void* stream);
/* Fused 16-bit (bf16 / fp16) path: q, k, v never reach HBM (both kernels recompute their slice of to_qkv from x).
 *   ld_linattn_kvctx: x [B,n,C] -> ctx partials (same layout/consumers as ld_linattn_ctx), with the
 *        RMSNorm (ddpm.py:237), the k/v rows of to_qkv (:239) and softmax_n(k) (:243) inside;
 *        wkv_packed = per head the 64 rows (k_h | v_h) of to_qkv, packed k=1 with g*sqrt(C) folded.
 *        kshift (optional, [heads*32] fp32, heads == 4): an upper bound m_d >= max_n k[d] per k channel.
 *        softmax over n is shift-invariant, so exp(k - m_d) / sum_n exp(k - m_d) equals the reference's
 *        exp(k - max) form; with it the kernel makes ONE sweep over x instead of two.  The RMS-normalised
 *        input has unit 2-norm per pixel, hence |k_d| <= ||W_k[d,:] * g * sqrt(C)||_2 (Cauchy-Schwarz): the
 *        host passes that norm, and passes NULL (exact two-sweep maximum) when it exceeds 40, where
 *        exp(k - m_d) could underflow.
 *   ld_linattn_out:   x -> out = RMSNorm(to_out(ctx^T softmax_d(q)*scale)) + x  (:242,245,249,251,425);
 *        wq_packed = the 128 q rows packed the same way, mfold from ld_linattn_fold / _ctxfold (perm=1);
 *        qshift (optional, [4] fp32): per head an upper bound of q over its 32 channels (max_d of the same
 *        Cauchy-Schwarz norm, <= 40), used as the softmax_d shift instead of the per-pixel maximum. */
int ld_linattn_kvctx(const void* x, const void* wkv_packed, const float* kshift, float* ctx_part, int B,
                     int n, int C, int heads, int dim_head, int nchunks, int dtype, void* stream);
int ld_linattn_out(const void* x, const void* wq_packed, const float* qshift, const void* mfold,
                   const float* bias, const float* g2, void* out, int B, int n, int C, float q_scale,
                   int dtype, void* stream);
/* Two-term weights (ld_pack_conv_weight_terms(..., 2) of the same rows, with the RMSNorm scale) for the fused linear
 * attention of the full- and half-resolution blocks (C = 32 / 64, heads = 4): W_kv / W_q enter the matrix pipe as hi + lo.
 * weight_terms = 1 is ld_linattn_kvctx / ld_linattn_out. */
int ld_linattn_kvctx_terms(const void* x, const void* wkv_packed, const float* kshift, float* ctx_part, int B, int n, int C,
                           int heads, int dim_head, int nchunks, int dtype, int weight_terms, void* stream);
int ld_linattn_out_terms(const void* x, const void* wq_packed, const float* qshift, const void* mfold, const float* bias,
                         const float* g2, void* out, int B, int n, int C, float q_scale, int dtype, int weight_terms,
                         void* stream);
size_t ld_linattn_ctx_part_floats(int B, int heads, int dim_head, int nchunks);

/* Full softmax attention (attend.py:84-113) on qkv [B, n, 3*hidden] with q pre-scaled;
 * out [B, n, hidden].  Flash-style (the n x n similarity matrix is never materialised). */
int ld_attention(const void* qkv, void* out, int B, int n, int heads, int dim_head, int dtype,
                 void* stream);

/* ---- timestep embedding (ddpm.py:136-149, 339-344, 191-206) -------------------------------- */
/* temb[i] = Linear(GELU(Linear(sincos(times[i]))))  for i < n;  freqs [dim/2] fp32 (host-made). */
int ld_time_mlp(const int32_t* times, int n, const float* freqs, int dim, const float* w1,
                const float* b1, const float* w2, const float* b2, int time_dim, float* temb,
                void* stream);
/* The same with RandomOrLearnedSinusoidalPosEmb in front (ddpm.py:151-165; learned_sinusoidal_cond / random_fourier_features):
 * emb = [t | sin(2 pi w_k t) | cos(2 pi w_k t)] with the module's `weights` [learned_dim / 2]; w1 is [time_dim, learned_dim + 1]. */
int ld_time_mlp_fourier(const int32_t* times, int n, const float* weights, int learned_dim, const float* w1,
                        const float* b1, const float* w2, const float* b2, int time_dim, float* temb, void* stream);
/* film[i] = Linear(SiLU(temb[i]))  -> [n, 2*C] */
int ld_film(const float* temb, int n, int time_dim, const float* w, const float* b, int two_c,
            float* film, void* stream);

/* ---- final 1x1 conv to the image (ddpm.py:398,451): NHWC storage -> NCHW fp32 -------------- */
int ld_final_conv(const void* x, const float* w /*[Cout,Cin]*/, const float* b, float* out_nchw,
                  int B, int H, int W, int Cin, int Cout, int dtype, void* stream);

/* ld_final_conv + ld_ddpm_step (+ ld_randn for the step's noise, stream index `noise_stream`) of the same pixels
 * in ONE launch: model_out = final_conv(x) is still written (fp32 NCHW), x_t is updated in place to x_{t-1}
 * (ddpm.py:451, 775-776, 659-666, 853-858).  Bitwise the same result as the three separate calls. */
int ld_final_step(const void* x, const float* w, const float* b, float* model_out, float* x_t, float* x0_out,
                  const float* sched, const int32_t* t_ptr, float lo, float hi, int objective, uint64_t seed,
                  int64_t noise_stream, int B, int H, int W, int Cin, int Cout, int dtype, void* stream);


/* ld_final_step for a sub-batch inside a replayed HIP graph: the noise stream index is
 * noise_base + noise_tmul * (*t_ptr) (the sampler's draw counter as a function of the device step counter) and
 * the sub-batch's elements are [noise_first, noise_first + B*Cout*H*W) of that stream's sequence.
 * keep_mask (optional, [B, H*W]): ld_mask_out folded in -- where keep_mask < 1 the prediction is replaced by `lo`
 * before the update (the OOD branch with mask_x, ddpm.py:693-696). */
int ld_final_step_at(const void* x, const float* w, const float* b, float* model_out, float* x_t, float* x0_out,
                     const float* sched, const int32_t* t_ptr, float lo, float hi, int objective, uint64_t seed,
                     int64_t noise_base, int64_t noise_tmul, int64_t noise_first, const float* keep_mask,
                     int B, int H, int W, int Cin, int Cout, int dtype, void* stream);

/* ---- reverse-process pointwise kernels (NCHW fp32) ---------------------------------------- */
#define LD_OBJ_X0 0
#define LD_OBJ_NOISE 1
#define LD_OBJ_V 2
/* schedule table row layout (floats): see LD_SCHED_* ; built by the host from the fp32 buffers */
#define LD_SCHED_COLS 8
#define LD_SCHED_COEF1 0      /* posterior_mean_coef1                (ddpm.py:592) */
#define LD_SCHED_COEF2 1      /* posterior_mean_coef2                (ddpm.py:593) */
#define LD_SCHED_SIGMA 2      /* exp(0.5*posterior_log_variance_clipped) (ddpm.py:591,853) */
#define LD_SCHED_SQRT_RECIP 3 /* sqrt(1/abar)                        (ddpm.py:578) */
#define LD_SCHED_SQRT_RECIPM1 4 /* sqrt(1/abar-1)                    (ddpm.py:579) */
#define LD_SCHED_SQRT_AB 5    /* sqrt(abar)                          (ddpm.py:575) */
#define LD_SCHED_SQRT_1MAB 6  /* sqrt(1-abar)                        (ddpm.py:576) */
#define LD_SCHED_ABAR 7       /* abar                                (ddpm.py:570) */

/* portable counter-based normals (rng.py): stream = stream_base + stream_tmul * (*t_ptr) */
int ld_randn(float* out, int64_t n, uint64_t seed, int64_t stream_base, int64_t stream_tmul,
             const int32_t* t_ptr, void* stream);
/* the slice [first, first + n) of that stream's sequence: sub-batches of one draw generated separately */
int ld_randn_at(float* out, int64_t n, int64_t first, uint64_t seed, int64_t stream_base, int64_t stream_tmul,
                const int32_t* t_ptr, void* stream);
/* *t_ptr += delta  (graph-replayable step counter) */
int ld_step_add(int32_t* t_ptr, int delta, void* stream);
/* head of one denoiser evaluation in ONE launch: zero up to two arenas (GroupNorm statistics, k-max codes; 16-byte
 * aligned and sized; either may be NULL / 0) and move the device step counter BEFORE any kernel of the evaluation
 * reads it: *t_ptr += delta (ancestral loop), or -- when idx_ptr and t_table are given (strided DDIM loop,
 * ddpm.py:984-986) -- *idx_ptr += 1; *t_ptr = t_table[*idx_ptr].  Replaces two hipMemsetAsync nodes + ld_step_add
 * at the head / tail of every replayed step. */
int ld_step_begin(void* zero_a, size_t bytes_a, void* zero_b, size_t bytes_b, int32_t* t_ptr, int delta,
                  int32_t* idx_ptr, const int32_t* t_table, void* stream);
/* the same + the step's FiLM row: film_cur[0 .. row_floats) = film_rows[t_new * row_floats ...] with t_new the counter's
 * value AFTER the move (t_ptr required; delta may be 0) -- the (scale, shift) vectors of every ResnetBlock for this
 * timestep (ddpm.py:191-206 evaluated for all t at setup, ld_film) land at a fixed address, so the launches that apply
 * FiLM need no dependent load of the step counter.  row_floats % 4 == 0, 16-byte aligned rows. */
int ld_step_begin_film(void* zero_a, size_t bytes_a, void* zero_b, size_t bytes_b, int32_t* t_ptr, int delta,
                       int32_t* idx_ptr, const int32_t* t_table, const float* film_rows, int row_floats,
                       float* film_cur, void* stream);

/* p_sample, single branch (ddpm.py:631-666, 739-761, 817-838, 857-858):
 *   x0 = clamp(to_x0(model_out)), x_prev = c1*x0 + c2*x_t + (t>0 ? sigma*z : 0).
 * x0_out may be NULL.  `noise` may be NULL: then no draw is added, in either mode (round 6: table mode used to dereference it
 * at t > 0).  Table mode (`t_ptr` non-NULL) adds sigma*z iff *t_ptr > 0 and `noise` is non-NULL.  Row mode (`t_ptr` NULL:
 * `sched` points at the step's own row, the kernel does not know t) adds it iff `noise` is non-NULL -- the CALLER carries
 * the reference's `t > 0` test (ddpm.py:857): pass NULL at t == 0, or the last step gets noise the reference does not add. */
int ld_ddpm_step(const float* x_t, const float* model_out, const float* noise, float* x_prev,
                 float* x0_out, const float* sched, const int32_t* t_ptr, float lo, float hi,
                 int objective, int64_t n, void* stream);
/* DDIM update (ddpm.py:1046-1068): x0 = clamp(to_x0(model_out)); eps re-derived from x0;
 * x_next = x0*sqrt(abar_next) + c*eps + sigma*z.  Scalars are host-computed per pair.
 * last=1 writes x0 (time_next < 0, :1053-1056). */
int ld_ddim_step(const float* x_t, const float* model_out, const float* noise, float* x_next,
                 float sqrt_recip, float sqrt_recipm1, float sqrt_ab, float sqrt_1mab,
                 float sqrt_abar_next, float c, float sigma, float lo, float hi, int objective,
                 int last, int64_t n, void* stream);
/* ld_ddim_step with the pair's scalars in a device table [pairs, 8] = {sqrt_recip, sqrt_recipm1, sqrt_ab, sqrt_1mab,
 * sqrt_abar_next, c, sigma, last} and the row selected by *idx_ptr: the form a replayed HIP graph needs. */
int ld_ddim_step_at(const float* x_t, const float* model_out, const float* noise, float* x_next,
                    const float* pair_table, const int32_t* idx_ptr, float lo, float hi, int objective, int64_t n,
                    void* stream);
/* branch conditioning (ddpm.py:672-690): binary=(mask>=1); cond_out=cond*binary;
 * cond_in=cond*clip(1-binary, lo_clip, 1).  mask [B,1,H,W], cond [B,C,H,W]. */
int ld_branch_conditions(const float* cond, const float* mask, float* cond_out, float* cond_in,
                         float lo_clip, int B, int C, int HW, void* stream);
/* mask_x on the OOD-branch prediction (ddpm.py:700-703): where(binary==0, min_val, out*binary) */
int ld_mask_out(float* model_out, const float* mask, float min_val, int B, int C, int HW,
                void* stream);
/* DDPM fusion step recomposition (ddpm.py:775-776, 784-804) from per-branch x_t and per-branch
 * x0 predictions (clamped here):  x0 = clamp(clamp(x0_in)*(1-m) + clamp(x0_out));
 *   x = where(x_out*m == 0, x_in*(1-m), x_out*m) */
int ld_fuse_ddpm(const float* x_out, const float* x_in, const float* x0_out, const float* x0_in,
                 const float* mask, float* x, float* x0, float lo, float hi, int B, int C, int HW,
                 void* stream);
/* K-mask generalisation of the two calls above (the reference has K = 2).  masks [B,K,HW]; branch 0
 * is the OOD-style branch, branches 1..K-1 IND-style:
 *   cond_k[0] = cond*(m_0>=1),  cond_k[k] = cond*clip((m_k>=1), lo_clip, 1)            -> cond_k [K,B,C,HW]
 *   x0 = clamp(sum_{k>=1} clamp(x0_k)*(m_k>=1) + clamp(x0_0)),  x = first non-zero of x_k*(m_k>=1), k = 0..K-1
 * x_rest / x0_rest hold branches 1..K-1 contiguously ([K-1,B,C,HW]); branch 0 has its own pointers (its state and
 * prediction may live outside the denoiser's batch, ddpm.py:704-708).  With K = 2 and m_1 = 1-(m_0>=1) the results
 * are bitwise those of ld_branch_conditions / ld_fuse_ddpm. */
int ld_branch_conditions_k(const float* cond, const float* masks, float* cond_k, float lo_clip, int B, int C, int K,
                           int HW, void* stream);
int ld_fuse_ddpm_k(const float* x_first, const float* x_rest, const float* x0_first, const float* x0_rest,
                   const float* masks, float* x, float* x0, float lo, float hi, int B, int C, int K, int HW,
                   void* stream);
/* posterior step from an already-formed x0 (fusion step, ddpm.py:809 + :858) */
int ld_posterior_step(const float* x_t, const float* x0, const float* noise, float* x_prev,
                      const float* sched, const int32_t* t_ptr, int64_t n, void* stream);
/* DDIM fusion (ddpm.py:1025-1041): x0 = clamp(where(x0o==0, x0i, x0o)); eps = where(eo*m==0,
 * ei*(1-m), eo*m) with e* re-derived from the clamped per-branch x0;  x_next as ld_ddim_step. */
int ld_fuse_ddim(const float* x_out, const float* x_in, const float* x0_out, const float* x0_in,
                 const float* mask, const float* noise, float* x_next, float sqrt_recip,
                 float sqrt_recipm1, float sqrt_abar_next, float c, float sigma, float lo,
                 float hi, int B, int C, int HW, void* stream);
/* K-branch form of ld_fuse_ddim (the reference has K = 2): masks [B,K,HW], branch 0 OOD-style with its
 * own pointers, branches 1..K-1 in x_rest / x0_rest ([K-1,B,C,HW]).  Per element, with a_k = clamp(x0_k) and
 * e_k = (sqrt_recip x_k - a_k) / sqrt_recipm1:
 *   x0  = clamp(a_0 if a_0 != 0 else a_j),  j = the first k >= 1 with m_k >= 1, K-1 if there is none
 *   eps = first non-zero of e_k*(m_k>=1), k = 0..K-1 (the last one if all are zero)
 * With K = 2 and m_1 = 1-(m_0>=1) the result is bitwise that of ld_fuse_ddim (ddpm.py:1025-1041). */
int ld_fuse_ddim_k(const float* x_first, const float* x_rest, const float* x0_first, const float* x0_rest,
                   const float* masks, const float* noise, float* x_next, float sqrt_recip, float sqrt_recipm1,
                   float sqrt_abar_next, float c, float sigma, float lo, float hi, int B, int C, int K, int HW,
                   void* stream);
/* q_sample (ddpm.py:1148-1154) for the use_gt start (:937-944) */
int ld_q_sample(const float* x0, const float* noise, float* out, float sqrt_ab, float sqrt_1mab,
                int64_t n, void* stream);
/* ---- training-side forward (forward half): GaussianDiffusion.forward / p_losses, ddpm.py:1147-1214 -----
 * q_sample with one timestep per sample: out_b = sqrt_ab[t_b] x0_b + sqrt_1mab[t_b] noise_b  (t: int32 [B] on the
 * device, sqrt_ab / sqrt_1mab: the fp32 schedule buffers [T] on the device). */
int ld_q_sample_t(const float* x0, const float* noise, float* out, const int* t, const float* sqrt_ab,
                  const float* sqrt_1mab, int B, int64_t elems_per_sample, void* stream);
/* per-sample loss (:1186-1201): loss_out[b] = loss_weight[t_b] * mean_{c,h,w} (model_out - target)^2, target = noise
 * (LD_OBJ_NOISE) | x_start (LD_OBJ_X0) | sqrt_ab[t_b] noise - sqrt_1mab[t_b] x_start (LD_OBJ_V, predict_v :643-647).
 * fp32 NCHW operands, fp64 accumulation in a fixed order; the batch mean (:1201) is the caller's. */
int ld_p_losses(const float* model_out, const float* x_start, const float* noise, const int* t, const float* sqrt_ab,
                const float* sqrt_1mab, const float* loss_weight, float* loss_out, int B, int64_t elems_per_sample,
                int objective, void* stream);
/* recomposition of K gathered local patches by their masks (north-star multi-GPU path,
 * SURVEY.md 8e): out[b] = sum_k patches[b,k]*m_k  with m_k = (masks[k] >= 1) */
int ld_recompose(const float* patches /*[B,K,C,HW]*/, const float* masks /*[K,HW]*/, float* out,
                 int B, int K, int C, int HW, void* stream);

/* ---- the one collective of the path: all-gather of every rank's finished samples, RCCL over xGMI ---- */
/* RCCL is dlopen'ed on first use (the copy the process already mapped, e.g. torch's, is preferred; LD_RCCL_PATH
 * overrides), so the library loads without it.  ld_comm_unique_id on one rank -> hand the 128 bytes to every rank ->
 * ld_comm_init on each (uses the calling thread's current HIP device) -> ld_allgather enqueued on `stream`:
 * recv[r*bytes_per_rank ...] = rank r's send buffer -> ld_comm_destroy.  The product's default keeps this collective
 * in torch.distributed (same RCCL); dist.gather_patches uses these entry points when asked to. */
int ld_comm_unique_id(void* id_out_128 /* host, 128 bytes */);
int ld_comm_init(void** comm_out, const void* id_128 /* host */, int world, int rank);
/* The same with a deadline: the communicator comes up non-blocking (ncclCommInitRankConfig + ncclCommGetAsyncError polled
 * against timeout_s); if it is not up in time -- a peer that never arrives, a stale unique id -- it is aborted
 * (ncclCommAbort) and the call returns LD_ETIMEOUT instead of hanging.  timeout_s <= 0 = ld_comm_init (blocking).
 * ld_allgather / ld_comm_destroy on such a communicator poll with the same deadline where RCCL answers ncclInProgress. */
int ld_comm_init_timeout(void** comm_out, const void* id_128 /* host */, int world, int rank, double timeout_s);
int ld_allgather(const void* send, void* recv, size_t bytes_per_rank, void* comm, void* stream);
int ld_comm_destroy(void* comm);

#ifdef __cplusplus
}
#endif
#endif /* LOCALDIFF_HIP_H */
