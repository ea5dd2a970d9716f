// Shared device/host helpers for the gfx950 kernels (wave64, MFMA 16x16 tiles, 64-byte K-chunks).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <type_traits>

#include "../../include/localdiff_hip.h"

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;   // a 16-byte register value that is NOT a struct (uint4 copies are memcpys in the IR)
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __bf16 bf16;
typedef _Float16 f16;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;

// ---------------------------------------------------------------- host-side error plumbing
int ld_fail(int code, const char* fmt, ...);   // runtime.hip
void ld_count(int which);                        // runtime.hip: LD_COUNTER_* launch-routing counters
#define LD_HIP(call)                                                                   \
  do {                                                                                 \
    hipError_t e_ = (call);                                                            \
    if (e_ != hipSuccess) return ld_fail(LD_EHIP, "%s: %s", #call, hipGetErrorString(e_)); \
  } while (0)
// Every kernel launch goes through LD_LAUNCH: while a timing session is open (ld_timing_begin, runtime.hip) the launch
// carries its own start/stop events (hipExtLaunchKernelGGL: the dispatch packet's begin/end timestamps, i.e. the
// kernel's execution time as rocprofv3 reports it, without the barrier packets of events recorded between launches).
bool ld_timing_next(hipEvent_t* start, hipEvent_t* stop);
#define LD_LAUNCH(kernel, grid, block, lds, st, ...)                                                              \
  do {                                                                                                            \
    hipEvent_t ld_e0_, ld_e1_;                                                                                    \
    if (ld_timing_next(&ld_e0_, &ld_e1_))                                                                         \
      hipExtLaunchKernelGGL(kernel, grid, block, lds, st, ld_e0_, ld_e1_, 0, __VA_ARGS__);                        \
    else                                                                                                          \
      hipLaunchKernelGGL(kernel, grid, block, lds, st, __VA_ARGS__);                                              \
  } while (0)

#define LD_LAUNCH_CHECK(name)                                                          \
  do {                                                                                 \
    hipError_t e_ = hipGetLastError();                                                 \
    if (e_ != hipSuccess) return ld_fail(LD_EHIP, "%s launch: %s", name, hipGetErrorString(e_)); \
  } while (0)
#define LD_REQUIRE(cond, ...)                                                          \
  do {                                                                                 \
    if (!(cond)) return ld_fail(LD_EINVAL, __VA_ARGS__);                               \
  } while (0)

// Opt a kernel into > 64 KiB of dynamic LDS once per process.
// Raise a kernel's dynamic-LDS limit to at least `bytes` on the CURRENT device.  Cached per (kernel, device) under a
// mutex in runtime.hip, so call sites call it unconditionally before a launch that needs more than 64 KB: a process
// that drives several devices, or launches from several host threads, gets the attribute on each of them.
hipError_t ld_allow_lds_ptr(const void* kernel, size_t bytes);
template <typename K>
static inline hipError_t ld_allow_lds(K kernel, size_t bytes) {
  if (bytes <= 65536) return hipSuccess;           // within the default dynamic-LDS limit: no attribute, no lookup
  return ld_allow_lds_ptr(reinterpret_cast<const void*>(kernel), bytes);
}

// ---------------------------------------------------------------- tuning table (runtime.hip; include/localdiff_hip.h)
struct LdTuning {
  long long c1_group, c1_group_max_px, c1_group_min_ch, c1_pair_max_px, c1_small_min;
  long long conv_raw, conv_mt4_min_wgs, conv_big_min, conv_sk, conv_sk_max_wgs;
  long long conv_c32, conv_c32_min_tiles;
  long long conv_s32, conv_s32_min_tiles;
  long long conv_big4_min;
  long long gn_frags_per_block, fold_split_min, attn_split_max_wgs, attn_split_min_n, lead_args, attn_xcd_map;
};
const LdTuning& ld_tuning();

// ---------------------------------------------------------------- dtype traits
// One "fragment" is 16 bytes per lane for every storage type: 8 bf16 / 8 fp16 or 4 fp32 consecutive
// channels.  A K-chunk is 4 fragments = 64 bytes of channels per pixel (32 bf16 or fp16 / 16 fp32).
// The two 16-bit types share every tiling constant and every kernel template; they differ in the
// unpack / pack conversions and in the MFMA opcode (v_mfma_f32_16x16x32_bf16 / _f16, same rate).
template <typename T> struct DT;
template <> struct DT<float> {
  static constexpr int E = 4;       // elements per fragment
  static constexpr int CK = 16;     // channels per K-chunk
  static constexpr bool precise = true;
  static constexpr float pshift = 0.0f;
};
template <> struct DT<bf16> {
  static constexpr int E = 8;
  static constexpr int CK = 32;
  static constexpr bool precise = false;
  static constexpr float pshift = 0.0f;
};
template <> struct DT<f16> {
  static constexpr int E = 8;
  static constexpr int CK = 32;
  static constexpr bool precise = false;
  // softmax_n(k) weights P = exp(k - m) <= 1 are stored as P * 2^pshift (their normaliser Z carries the same
  // factor): fp16's normal range ends at 6e-5, so unscaled small weights would be rounded as subnormals
  static constexpr float pshift = 10.0f;
};
// dtype code (LD_F32 / LD_BF16 / LD_F16) -> storage type: LD_DISPATCH(dtype, expr using T) evaluates the expression
// for the matching type (the entry points validate the code first)
#define LD_DISPATCH(dtype, ...)                                              \
  ((dtype) == LD_F32 ? [&] { using T = float; return __VA_ARGS__; }()        \
   : (dtype) == LD_BF16 ? [&] { using T = bf16; return __VA_ARGS__; }()      \
                        : [&] { using T = f16; return __VA_ARGS__; }())
#define LD_DISPATCH16(dtype, ...)                                            \
  ((dtype) == LD_BF16 ? [&] { using T = bf16; return __VA_ARGS__; }()        \
                      : [&] { using T = f16; return __VA_ARGS__; }())
static inline bool ld_dtype_ok(int dtype) { return dtype == LD_F32 || dtype == LD_BF16 || dtype == LD_F16; }
static inline bool ld_dtype_16(int dtype) { return dtype == LD_BF16 || dtype == LD_F16; }

template <typename T> __device__ __forceinline__ void unpack16(const uint4& r, float* v);
template <> __device__ __forceinline__ void unpack16<float>(const uint4& r, float* v) {
  v[0] = __uint_as_float(r.x); v[1] = __uint_as_float(r.y);
  v[2] = __uint_as_float(r.z); v[3] = __uint_as_float(r.w);
}
template <> __device__ __forceinline__ void unpack16<bf16>(const uint4& r, float* v) {
  v[0] = __uint_as_float(r.x << 16); v[1] = __uint_as_float(r.x & 0xffff0000u);
  v[2] = __uint_as_float(r.y << 16); v[3] = __uint_as_float(r.y & 0xffff0000u);
  v[4] = __uint_as_float(r.z << 16); v[5] = __uint_as_float(r.z & 0xffff0000u);
  v[6] = __uint_as_float(r.w << 16); v[7] = __uint_as_float(r.w & 0xffff0000u);
}
__device__ __forceinline__ void unpack_f16x2(unsigned r, float& a, float& b) {
  const f32x2 f = __builtin_convertvector(__builtin_bit_cast(f16x2, r), f32x2);   // v_cvt_f32_f16 (+ SDWA high half)
  a = f[0]; b = f[1];
}
template <> __device__ __forceinline__ void unpack16<f16>(const uint4& r, float* v) {
  unpack_f16x2(r.x, v[0], v[1]); unpack_f16x2(r.y, v[2], v[3]);
  unpack_f16x2(r.z, v[4], v[5]); unpack_f16x2(r.w, v[6], v[7]);
}
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
  // ONE v_cvt_pk_bf16_f32 (round-to-nearest-even, NaN preserved) for the pair: converting the two values
  // separately costs a cvt each plus shift/or to merge them (4 VALU instructions per pair in every epilogue)
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  const f32x2_t v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ unsigned pack_f16x2(float lo, float hi) {
  const f32x2 v = {lo, hi};                                  // ONE v_cvt_pk_f16_f32 (round-to-nearest-even)
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2));
}
// two fp32 values -> one packed pair of the 16-bit storage type T
template <typename T> __device__ __forceinline__ unsigned pack2(float lo, float hi);
template <> __device__ __forceinline__ unsigned pack2<bf16>(float lo, float hi) { return pack_bf16x2(lo, hi); }
template <> __device__ __forceinline__ unsigned pack2<f16>(float lo, float hi) { return pack_f16x2(lo, hi); }
template <typename T> __device__ __forceinline__ uint4 pack16(const float* v);
template <> __device__ __forceinline__ uint4 pack16<float>(const float* v) {
  return make_uint4(__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3]));
}
template <> __device__ __forceinline__ uint4 pack16<bf16>(const float* v) {
  return make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]),
                    pack_bf16x2(v[6], v[7]));
}

template <> __device__ __forceinline__ uint4 pack16<f16>(const float* v) {
  return make_uint4(pack_f16x2(v[0], v[1]), pack_f16x2(v[2], v[3]), pack_f16x2(v[4], v[5]), pack_f16x2(v[6], v[7]));
}

// 4 consecutive output channels (one accumulator fragment) -> 8 or 16 bytes
template <typename T> __device__ __forceinline__ void store4(T* p, const float* v);
__device__ __forceinline__ void store16_out(void* p, const uint4& v);       // (below: write-through output stores)
template <> __device__ __forceinline__ void store4<float>(float* p, const float* v) {
  store16_out(p, make_uint4(__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])));
}
template <> __device__ __forceinline__ void store4<bf16>(bf16* p, const float* v) {
  *reinterpret_cast<uint2*>(p) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
}
template <> __device__ __forceinline__ void store4<f16>(f16* p, const float* v) {
  *reinterpret_cast<uint2*>(p) = make_uint2(pack_f16x2(v[0], v[1]), pack_f16x2(v[2], v[3]));
}
// ---------------------------------------------------------------- wide epilogue stores (16-bit storage)
// An accumulator fragment is 4 consecutive channels of one pixel = 8 bytes in 16-bit storage, and the 4 kq lanes of a
// pixel sit 16 lanes apart: per m-tile a wave instruction stores 64 x 8 B in 32-byte runs.  For two ADJACENT m-tiles
// (32 consecutive channels = 64 B per pixel) one v_permlane16_swap per dword regroups the fragments so that every
// lane holds 8 consecutive channels: ONE 16-byte store instead of two 8-byte ones -- half the store instructions for
// the same bytes, and with 32 output channels a wave instruction writes 1 KiB contiguous (cdna_hip_programming.md
// T21 in its 16-lane-row form; the store tails were issue-bound: 10 of the 22.5 us of a 32->32 @256^2 launch).
//   swap(vdst = p0, src = p1) exchanges rows 1 / 3 of p0 with rows 0 / 2 of p1 (row = 16 lanes = one kq):
//   kq 0: (own p0 [ch 0-3 of m],        p0 of kq 1 [ch 4-7 of m])        -> m-tile m,     bytes  0..15
//   kq 1: (p1 of kq 0 [ch 0-3 of m+1],  own p1 [ch 4-7 of m+1])          -> m-tile m + 1, bytes  0..15
//   kq 2: (own p0 [ch 8-11 of m],       p0 of kq 3 [ch 12-15 of m])      -> m-tile m,     bytes 16..31
//   kq 3: (p1 of kq 2 [ch 8-11 of m+1], own p1 [ch 12-15 of m+1])        -> m-tile m + 1, bytes 16..31
// Every lane of the wave must execute the exchange (callers predicate only the store; the lanes of one pixel share
// their predicate).  The values and their roundings are those of two store4 calls: results are bit-identical.
template <typename T>
__device__ __forceinline__ uint4 pair_frag16(const float* v0, const float* v1) {
  static_assert(sizeof(T) == 2, "pair_frag16: 16-bit storage only (an fp32 fragment is 16 bytes already)");
  const unsigned a0 = pack2<T>(v0[0], v0[1]), a1 = pack2<T>(v0[2], v0[3]);
  const unsigned b0 = pack2<T>(v1[0], v1[1]), b1 = pack2<T>(v1[2], v1[3]);
  const auto r0 = __builtin_amdgcn_permlane16_swap(a0, b0, false, false);
  const auto r1 = __builtin_amdgcn_permlane16_swap(a1, b1, false, false);
  return make_uint4(r0[0], r1[0], r0[1], r1[1]);
}
// The same regrouping for LOADS of an operand in accumulator-fragment layout (residual / GroupNorm-tail input of the
// 1x1 epilogues): ONE 16-byte load at pair_frag16_off(kq) instead of two 8-byte ones; the exchange is an involution, so
// applying it to the loaded piece returns this lane's fragments of m-tile m (f0) and m + 1 (f1).  Whole wave active.
__device__ __forceinline__ void unpair_frag16(const uint4& w, uint2& f0, uint2& f1) {
  const auto r0 = __builtin_amdgcn_permlane16_swap(w.x, w.z, false, false);
  const auto r1 = __builtin_amdgcn_permlane16_swap(w.y, w.w, false, false);
  f0 = make_uint2(r0[0], r1[0]);
  f1 = make_uint2(r0[1], r1[1]);
}
// byte offset of the lane's 16-byte piece from channel 0 of m-tile m of its pixel
__device__ __forceinline__ unsigned pair_frag16_off(int kq) { return (kq & 1) * 32u + (kq >> 1) * 16u; }

template <typename T> __device__ __forceinline__ void load4(const T* p, float* v);
template <> __device__ __forceinline__ void load4<float>(const float* p, float* v) {
  float4 r = *reinterpret_cast<const float4*>(p);
  v[0] = r.x; v[1] = r.y; v[2] = r.z; v[3] = r.w;
}
template <> __device__ __forceinline__ void load4<bf16>(const bf16* p, float* v) {
  uint2 r = *reinterpret_cast<const uint2*>(p);
  v[0] = __uint_as_float(r.x << 16); v[1] = __uint_as_float(r.x & 0xffff0000u);
  v[2] = __uint_as_float(r.y << 16); v[3] = __uint_as_float(r.y & 0xffff0000u);
}
template <> __device__ __forceinline__ void load4<f16>(const f16* p, float* v) {
  uint2 r = *reinterpret_cast<const uint2*>(p);
  unpack_f16x2(r.x, v[0], v[1]); unpack_f16x2(r.y, v[2], v[3]);
}
// Four consecutive elements as they sit in memory (no conversion): a load whose conversion is written next to it
// inside an `if (in range)` branch is waited for INSIDE that branch -- one dependent round trip per load.  Load raw
// (from a clamped, always-valid address), convert where the value is used.
template <typename T> struct Raw4 { typedef uint2 type; };
template <> struct Raw4<float> { typedef float4 type; };
template <typename T> __device__ __forceinline__ typename Raw4<T>::type load4_raw(const T* p) {
  return *reinterpret_cast<const typename Raw4<T>::type*>(p);
}
template <typename T> __device__ __forceinline__ void unpack4(const typename Raw4<T>::type& r, float* v);
template <> __device__ __forceinline__ void unpack4<float>(const float4& r, float* v) { v[0] = r.x; v[1] = r.y; v[2] = r.z; v[3] = r.w; }
template <> __device__ __forceinline__ void unpack4<bf16>(const uint2& r, float* v) {
  v[0] = __uint_as_float(r.x << 16); v[1] = __uint_as_float(r.x & 0xffff0000u);
  v[2] = __uint_as_float(r.y << 16); v[3] = __uint_as_float(r.y & 0xffff0000u);
}
template <> __device__ __forceinline__ void unpack4<f16>(const uint2& r, float* v) {
  unpack_f16x2(r.x, v[0], v[1]); unpack_f16x2(r.y, v[2], v[3]);
}
template <typename R> __device__ __forceinline__ void pin_raw4(const R& r);
template <> __device__ __forceinline__ void pin_raw4<uint2>(const uint2& r) { asm volatile("" ::"v"(r.x), "v"(r.y)); }
template <> __device__ __forceinline__ void pin_raw4<float4>(const float4& r) { asm volatile("" ::"v"(r.x), "v"(r.y), "v"(r.z), "v"(r.w)); }
template <typename T> __device__ __forceinline__ float to_f(T v);
template <> __device__ __forceinline__ float to_f<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f<bf16>(bf16 v) { return (float)v; }
template <> __device__ __forceinline__ float to_f<f16>(f16 v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f(float v);
template <> __device__ __forceinline__ float from_f<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16 from_f<bf16>(float v) { return (bf16)v; }
template <> __device__ __forceinline__ f16 from_f<f16>(float v) { return (f16)v; }

// ---------------------------------------------------------------- MFMA, D[cout 16][pixel 16]
// A = weights fragment (row = output channel, lane&15), B = activation fragment (col = pixel,
// lane&15); both hold the K-slice (lane>>4).  D: col = lane&15, row = 4*(lane>>4) + reg.
// fp32 uses v_mfma_f32_16x16x4_f32 (bitwise an fmaf chain, exact f32) four times per fragment:
// step e consumes element e of both fragments, i.e. K index 4*(lane>>4)+e of the chunk.
template <typename T> __device__ __forceinline__ void mma16(f32x4& acc, const uint4& a, const uint4& b);
template <> __device__ __forceinline__ void mma16<float>(f32x4& acc, const uint4& a, const uint4& b) {
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
}
template <> __device__ __forceinline__ void mma16<bf16>(f32x4& acc, const uint4& a, const uint4& b) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a),
                                                 __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
}

template <> __device__ __forceinline__ void mma16<f16>(f32x4& acc, const uint4& a, const uint4& b) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc, 0, 0, 0);
}

// ---------------------------------------------------------------- LDS-DMA
// global_load_lds_dwordx4 as inline asm: 64 lanes x 16 B land at lds_dst + lane*16 (lds_dst wave-uniform).
// Issued through the builtin, hipcc orders every later LDS access of the wave behind the DMA with
// s_waitcnt vmcnt(0) (it cannot prove the ds_read does not alias the DMA destination), which serialises the
// ring; the asm form is invisible to that bookkeeping, so completion is tracked by hand: the issuing wave's
// counted s_waitcnt vmcnt(N), then a barrier, then the reads (cdna_hip_programming.md "What hipcc does not do").
__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return (unsigned)(size_t)p;            // low 32 bits of a flat LDS address = byte offset in the workgroup's LDS
}
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// ---------------------------------------------------------------- output stores that do not stay dirty in L2
// The eight L2s are not coherent with each other, so the release at the end of a kernel writes every dirty line back before
// the next kernel of the stream may start: a launch that leaves its plain-stored output in L2 pays for that write-back at its
// boundary (MI355X_MICROARCH.md, price list "boundary": + B / 6 TB/s behind B dirty bytes; fence table: a release is ~1.7 us
// clean and ~6.5 us behind freshly dirtied lines).  An `sc1` store is written THROUGH as it is issued -- the write-back
// overlaps the kernel instead of trailing it -- and leaves no line behind; the consumer is the NEXT kernel, on any XCD, and
// would read through the fabric anyway.  Measured (finding 98): 32->32 @256^2 alone 11.1 -> 9.6 us (x 4 patches, with
// statistics), 20.9 -> 16.4 (x 8).  Every 16-byte output store of the activations goes through here (LD_STORE_WT=0: plain
// stores, `build.sh --plain-stores`, the A/B build).  The asm store is invisible to hipcc's s_waitcnt bookkeeping (nothing later
// in a kernel depends on its output) and ends in `s_nop 1`: the next instruction must not overwrite its data registers before
// the store has read them (cdna_hip_programming.md 5.7 item 1).  Other policy bits on the same store, in the sampler (three
// alternating runs, sc1 = 1.244 ms): `sc0 sc1` 1.242 (the same), `sc1 nt` 1.298 (+4.4 %), `nt` alone 1.311 (+5.3 %: worse than
// plain stores, although one kernel alone on the chip ran as fast with `nt` as with `sc1`).
// PRECONDITION: the data and address registers must be VALU results, never raw MFMA accumulators.  hipcc's hazard
// recogniser does not see inside the asm: the `s_nop 1` covers the overwrite of the data registers BEHIND the store, nothing covers
// an MFMA write -> vector-memory read hazard (up to ~18 wait states) IN FRONT of it.  Every call site packs / converts / adds a bias
// between the accumulator and the store, and the compiler-handled MFMA -> VALU wait protects that VALU instruction.  The shipped
// machine code is checked for it: tools/scan_asm_store_sources.py (tests/test_codegen.py) finds the last writer of every sc1
// store's registers in the library's code objects and fails if one is a matrix instruction.
#ifndef LD_STORE_WT
#define LD_STORE_WT 1
#endif
__device__ __forceinline__ void store16_out(void* p, const uint4& v) {
#if LD_STORE_WT
  const u32x4 d = {v.x, v.y, v.z, v.w};
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(d) : "memory");
#else
  *reinterpret_cast<uint4*>(p) = v;
#endif
}
// The same for the narrow stores of small outputs (kvctx's partial sums, ctxfold's folded weights, the final step's NCHW
// images): a relaxed agent-scope atomic store IS a `global_store_dword / _short ... sc1`.  Narrow write-through stores cost
// ~6-12x a 16-byte one per byte (each 64-byte segment is its own fabric write), so this level is only worth it where the
// bytes are few and what matters is that the kernel ends with clean L2s (LD_STORE_WT >= 2).  MEASURED AND NOT KEPT: with the
// narrow stores written through as well the step is 1.263 -> 1.281 ms (+1.4 %, three alternating pairs, `build.sh
// --wt-level=2`): level 1 stays the default and these compile to plain stores.
__device__ __forceinline__ void store_f32_out(float* p, float v) {
#if LD_STORE_WT >= 2
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
  *p = v;
#endif
}
template <typename T> __device__ __forceinline__ void store_elem_out(T* p, T v) {
#if LD_STORE_WT >= 2
  if constexpr (sizeof(T) == 2) __hip_atomic_store(reinterpret_cast<unsigned short*>(p), __builtin_bit_cast(unsigned short, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else __hip_atomic_store(reinterpret_cast<unsigned*>(p), __builtin_bit_cast(unsigned, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
  *p = v;
#endif
}

// ---------------------------------------------------------------- activation loads (experiment switch, finding 101)
// LD_LOAD_NT = 1: the read-once activation tiles of the convolutions are requested with the non-temporal policy.
#ifndef LD_LOAD_NT
#define LD_LOAD_NT 0
#endif
__device__ __forceinline__ u32x4 load16_act(const void* p) {
#if LD_LOAD_NT
  return __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
#else
  return *reinterpret_cast<const u32x4*>(p);
#endif
}

// ---------------------------------------------------------------- tile -> XCD placement (experiment switch, finding 102)
// Workgroups are dealt round-robin over the 8 XCDs (id % 8: observed, speed only).  With tiles numbered row by row, the
// horizontal neighbours of a tile run on other XCDs and the halo columns they share are fetched through the fabric twice.
// LD_TILE_XCD = 1: XCD x owns one of 2 x 4 rectangular regions of the tile grid, so that only region borders do.
#ifndef LD_TILE_XCD
#define LD_TILE_XCD 0
#endif
__device__ __forceinline__ void tile_of(int id, int tiles_x, int tiles_y, int& ty, int& tx) {
  ty = id / tiles_x;
  tx = id - ty * tiles_x;
#if LD_TILE_XCD
  if ((tiles_x & 1) == 0 && (tiles_y & 3) == 0) {
    const int rw = tiles_x >> 1, rh = tiles_y >> 2, x = id & 7, s = id >> 3;
    ty = (x >> 1) * rh + s / rw;
    tx = (x & 1) * rw + s % rw;
  }
#endif
}

// ---------------------------------------------------------------- kernel-argument layout
// Byte offset, in the kernel-argument segment, of the argument that follows leading arguments of types Lead... and has
// alignment `align` (each argument sits at its natural alignment, in order): where a kernel finds its trailing by-value
// struct when it reads it through __builtin_amdgcn_kernarg_segment_ptr() (docs/findings.md 83).
template <typename... Lead>
constexpr unsigned ld_kernarg_offset(unsigned align) {
  unsigned off = 0;
  const unsigned sizes[] = {(unsigned)sizeof(Lead)...}, aligns[] = {(unsigned)alignof(Lead)...};
  for (unsigned i = 0; i < sizeof...(Lead); ++i) off = ((off + aligns[i] - 1) / aligns[i]) * aligns[i] + sizes[i];
  return ((off + align - 1) / align) * align;
}

// ---------------------------------------------------------------- head-of-step work (pointwise.hip: step_begin_kernel;
// conv_image.hip: the stem convolution's extra workgroups, ld_conv_stem_begin)
struct StepBeginDev {
  uint4* a; long na; uint4* b; long nb;          // arenas to zero, in 16-byte units
  int* t; int delta; int* idx; const int* t_table;
  const float* film_rows; int row_floats; float* film_cur;
};
// Part `e` of `ne` (whole workgroups): zero the arenas; part 0 also moves the step counter and copies the new
// timestep's FiLM row to its fixed address.  s_t: one int of shared memory.
__device__ __forceinline__ void step_begin_work(const StepBeginDev& s, long e, long ne, int* s_t) {
  const uint4 z = make_uint4(0u, 0u, 0u, 0u);
  const long stride = ne * blockDim.x;
  for (long i = e * blockDim.x + threadIdx.x; i < s.na; i += stride) store16_out(&s.a[i], z);     // (write-through: finding 98)
  for (long i = e * blockDim.x + threadIdx.x; i < s.nb; i += stride) store16_out(&s.b[i], z);
  if (e == 0) {
    if (threadIdx.x == 0) {
      int tn = s.t ? *s.t : 0;
      if (s.idx && s.t_table) {           // strided (DDIM) sampling: advance the pair counter, look the timestep up
        const int k = *s.idx + 1;
        *s.idx = k;
        tn = s.t_table[k];
        if (s.t) *s.t = tn;
      } else if (s.t && s.delta != 0) {
        tn += s.delta;
        *s.t = tn;
      }
      *s_t = tn;
    }
    if (s.film_rows) {                    // (uniform)
      __syncthreads();
      const float4* src = reinterpret_cast<const float4*>(s.film_rows + (size_t)*s_t * s.row_floats);
      float4* dst = reinterpret_cast<float4*>(s.film_cur);
      for (int i = threadIdx.x; i < s.row_floats / 4; i += blockDim.x) {
        const float4 v = src[i];
        store16_out(&dst[i], make_uint4(__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)));
      }
    }
  }
}
int ld_step_begin_check(const void* zero_a, size_t bytes_a, const void* zero_b, size_t bytes_b, const void* t_ptr, const void* idx_ptr,
                        const void* t_table, const void* film_rows, int row_floats, const void* film_cur);

// ---------------------------------------------------------------- activations
template <bool PRECISE> __device__ __forceinline__ float silu_f(float v) {
  // parity mode: libm expf + IEEE divide.  Storage-bf16 mode: v_exp_f32 / v_rcp_f32 (1 ulp each, far below the
  // bf16 rounding of the result); __frcp_rn would expand to the 11-instruction IEEE division sequence.
  if constexpr (PRECISE) return v / (1.0f + expf(-v));
  else return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * v));
}
template <bool PRECISE> __device__ __forceinline__ float act_f(float v, int act) {
  if (act == LD_ACT_SILU) return silu_f<PRECISE>(v);
  if (act == LD_ACT_RELU) return fmaxf(v, 0.0f);
  return v;
}
// v[e] = act(v[e]) for N values with ONE (wave-uniform) switch on `act`: called per element, the switch is not
// hoisted by hipcc and every element pays scalar compares and branches that also split the VALU schedule.
template <bool PRECISE, int N> __device__ __forceinline__ void act_n(float* v, int act) {
  if (act == LD_ACT_SILU) {
#pragma unroll
    for (int e = 0; e < N; ++e) v[e] = silu_f<PRECISE>(v[e]);
  } else if (act == LD_ACT_RELU) {
#pragma unroll
    for (int e = 0; e < N; ++e) v[e] = fmaxf(v[e], 0.0f);
  }
}
// v[e] = act(v[e]*a[e] + s[e])
template <bool PRECISE, int N> __device__ __forceinline__ void affine_act_n(float* v, const float* a, const float* s, int act) {
#pragma unroll
  for (int e = 0; e < N; ++e) v[e] = fmaf(v[e], a[e], s[e]);
  act_n<PRECISE, N>(v, act);
}

// ---------------------------------------------------------------- normalise-on-load coefficients
// Device mirror of ld_src (kernel-argument POD).
struct SrcDev {
  const void* data;
  const double* stats;
  const float* gamma;
  const float* beta;
  const float* film;
  int C, ld, ups, groups, act, film_tstride, film_bstride;   // ld = elements between pixels
};
static inline SrcDev to_dev(const ld_src& s) {
  SrcDev d;
  d.data = s.data; d.stats = s.gn_stats; d.gamma = s.gn_gamma; d.beta = s.gn_beta; d.film = s.film;
  d.C = s.C; d.ld = s.pix_stride > 0 ? s.pix_stride : s.C; d.ups = s.upsample; d.groups = s.gn_groups; d.act = s.act;
  d.film_tstride = s.film_tstride; d.film_bstride = s.film_bstride;
  return d;
}

// Build y = x*a + s coefficients for batch element b into LDS: coef[0..C) = a, coef[C..2C) = s.
//   a = rstd*gamma*(scale+1),  s = (beta - mean*rstd*gamma)*(scale+1) + shift
// (GroupNorm eps 1e-5, biased variance: ddpm.py:174 / unet_model.py:21; FiLM: ddpm.py:181-183).
// npix = pixels the statistics were accumulated over (the producer's H*W).
// `red` is LDS scratch for 2*groups doubles.  Contains two __syncthreads(): call from all threads.
// PRECISE (fp32 storage, the parity mode): IEEE fp64 divide and square root for 1/n and 1/sqrt(var + eps).  16-bit
// storage: v_rcp_f32 / v_rsq_f32 (1 ulp of fp32 each, four orders below the storage rounding) -- the two fp64 divisions
// and the square root are ~100 dependent instructions at the head of every consumer workgroup.
// after_issue: called once every request of the head has left and before the first wait -- a caller whose remaining
// kernel arguments are read late (finding 84) requests them there, under the statistics' round trip.
struct GnNoHook { __device__ __forceinline__ void operator()() const {} };
__device__ __forceinline__ double row_group_sum_d(double v, int n);       // (below)
template <bool PRECISE = true, typename F = GnNoHook>
__device__ __forceinline__ void build_gn_coef(const SrcDev& S, int b, int trow, long npix, float* coef,
                                              double* red, int tid, int nthreads, F after_issue = F()) {
  const int C = S.C, G = S.groups, gs = C / G;
  float* gstat = reinterpret_cast<float*>(red);          // after the stripe sum: [G] mean, [G] rstd (floats)
  // ONE global round trip in front of the arithmetic: gamma / beta / FiLM of this thread's first channel (their addresses
  // depend on nothing: in table mode the step's FiLM row sits at a fixed address, ld_step_begin_film) and the 2 * stripes
  // statistics words of its group are all requested before anything waits.  hipcc otherwise places the first
  // `s2 += st2[0]` INSIDE the block of statistics loads (`s_waitcnt vmcnt(6)` behind the seventh of sixteen requests:
  // the first load's whole round trip in front of the other nine and of gamma / beta) -- seen with hipcc -S in every
  // kernel that builds coefficients (~47 launches per step); the scheduling barrier pins the requests together.
  const float* film = S.film ? S.film + (long)trow * S.film_tstride + (long)b * S.film_bstride : nullptr;
  float g0 = 0.f, b0 = 0.f, f0 = 0.f, f1 = 0.f;
  if (tid < C) {
    // (the FiLM words come from an always-valid address and are selected afterwards: inside an `if (film)` branch hipcc
    //  waits for them before leaving it, i.e. in front of the statistics requests)
    const float* fp = film ? film : S.gamma;
    g0 = S.gamma[tid]; b0 = S.beta[tid];
    f0 = fp[tid]; f1 = fp[(film ? C : 0) + tid];
  }
  // Stripe sums: 16 lanes per group (thread g * 16 + s takes stripes s, s + 16, ... of group g: LD_STAT_STRIPES / 16 pairs
  // of doubles each), the 16 partial sums meet by DPP inside the 16-lane row.  (Until round 5: G threads x 2 * 16 loads and a
  // sequential fp64 chain; with 64 stripes that would be 128 loads per thread.)
  constexpr int SPL = LD_STAT_STRIPES / 16;
  static_assert(LD_STAT_STRIPES % 16 == 0, "stripes are reduced 16 lanes at a time");
  double st1[SPL], st2[SPL];
  const int sg = tid >> 4, sl = tid & 15;
  const bool lead = tid < 16 * G;                        // (whole 16-lane rows: the DPP row sums need every lane of a row;
                                                         //  16 * G <= nthreads: G <= 16 at 256 threads, checked on the host)
  if (lead) {
    const double* p = S.stats + ((size_t)b * LD_STAT_STRIPES + sl) * G * 2 + 2 * sg;
#pragma unroll
    for (int k = 0; k < SPL; ++k) { st1[k] = p[(size_t)k * 16 * G * 2]; st2[k] = p[(size_t)k * 16 * G * 2 + 1]; }
    __builtin_amdgcn_sched_barrier(0);
  }
  after_issue();
  // (opaque from here on: hipcc otherwise computes `f0 + 1` right behind its load, in front of the statistics requests)
  asm volatile("" : "+v"(g0), "+v"(b0), "+v"(f0), "+v"(f1));
  if (lead) {
    double s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int k = 0; k < SPL; ++k) { s1 += st1[k]; s2 += st2[k]; }
    s1 = row_group_sum_d(s1, 16);
    s2 = row_group_sum_d(s2, 16);
    // ONE double divide/sqrt per group (not per channel)
    const double inv_n = PRECISE ? 1.0 / ((double)npix * gs) : (double)__builtin_amdgcn_rcpf((float)npix * (float)gs);
    const double mean = s1 * inv_n;
    double var = s2 * inv_n - mean * mean;
    var = var > 0.0 ? var : 0.0;
    const float rstd = PRECISE ? (float)(1.0 / sqrt(var + 1e-5)) : __builtin_amdgcn_rsqf((float)(var + 1e-5));
    if (sl == 0) {
      gstat[sg] = (float)mean;
      gstat[G + sg] = rstd;
    }
  }
  __syncthreads();
  for (int c = tid; c < C; c += nthreads) {
    const int g = c / gs;
    const bool first = c == tid;
    float a = gstat[G + g] * (first ? g0 : S.gamma[c]);
    float s = (first ? b0 : S.beta[c]) - gstat[g] * a;
    if (film) {
      const float sc = (first ? f0 : film[c]) + 1.0f, sh = first ? f1 : film[C + c];
      a *= sc;
      s = s * sc + sh;
    }
    coef[c] = a;
    coef[C + c] = s;
  }
  __syncthreads();
}

// RMSNorm's 1 / max(||x||, 1e-12) from the sum of squares (ddpm.py:131, F.normalize): exact IEEE sqrt + divide for fp32
// storage (parity mode), one v_rsq_f32 for bf16 storage (the pair costs ~25 VALU instructions).
template <bool PRECISE> __device__ __forceinline__ float rms_rinv(float sumsq) {
  if (PRECISE) return 1.0f / fmaxf(sqrtf(sumsq), 1e-12f);
  return __builtin_amdgcn_rsqf(fmaxf(sumsq, 1e-24f));
}

// DPP row rotate inside each 16-lane row (one VALU op, no LDS crossbar): dpp_ctrl 0x120+n = row_ror:n
template <int N> __device__ __forceinline__ float row_ror(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + N, 0xf, 0xf, false));
}
__device__ __forceinline__ float wave16_sum(float v) {
  // sum over the 16 lanes that share (lane >> 4); every lane of the row ends with the total
  v += row_ror<8>(v); v += row_ror<4>(v); v += row_ror<2>(v); v += row_ror<1>(v);
  return v;
}

// Reductions over the 4 lanes that hold one pixel's K-slices / channel quarters (lane & 15 equal, 16 lanes apart): the
// fragment layouts keep a pixel's (or a query's) partial sums there.  v_permlane16_swap / v_permlane32_swap of a value
// with itself leave (row 0 | row 0 | row 2 | row 2) and (row 1 | row 1 | row 3 | row 3), resp. (lower | lower) and
// (upper | upper), in the two results: two VALU exchanges + two ops, where __shfl_xor(x, 16) / (x, 32) are two dependent
// trips through the LDS crossbar (ds_bpermute, ~120 cycles each behind an s_waitcnt).  Every lane ends with the
// reduction; the operand order of each lane's two additions is that of the shuffle form (bit-identical sums).
// Whole wave active.
__device__ __forceinline__ float kq4_sum(float x) {
  const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  const float y = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(y), __float_as_uint(y), false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
__device__ __forceinline__ float kq4_max(float x) {
  const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  const float y = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
  const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(y), __float_as_uint(y), false, false);
  return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}

// fp64 value of another lane of the same 16-lane row (DPP on the two halves: VALU speed, no LDS crossbar)
template <int CTRL> __device__ __forceinline__ double dpp_mov_d(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
// Sum over aligned groups of n = 1, 2, 4, 8 or 16 consecutive lanes of a 16-lane row; every lane of a group ends with
// the group's total.  All 16 lanes of the row must be active.
__device__ __forceinline__ double row_group_sum_d(double v, int n) {
  if (n >= 2) v += dpp_mov_d<0xB1>(v);     // quad_perm [1,0,3,2]: lane ^ 1
  if (n >= 4) v += dpp_mov_d<0x4E>(v);     // quad_perm [2,3,0,1]: lane ^ 2
  if (n >= 8) v += dpp_mov_d<0x141>(v);    // row_half_mirror: the other quad of the 8 (whose lanes all hold its total)
  if (n >= 16) v += dpp_mov_d<0x128>(v);   // row_ror:8: the other half of the row
  return v;
}

__device__ __forceinline__ double wave16_sum_d(double v) {
  v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
  return v;
}

// Order-preserving float <-> uint32 map for integer atomicMax on floats; 0 (a memset buffer) is the
// identity of max.  enc is monotone: a < b  <=>  enc(a) < enc(b).
__device__ __forceinline__ unsigned enc_max(float f) {
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float dec_max(unsigned u) {
  return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}
__device__ __forceinline__ float wave16_max(float v) {
  v = fmaxf(v, row_ror<8>(v)); v = fmaxf(v, row_ror<4>(v)); v = fmaxf(v, row_ror<2>(v)); v = fmaxf(v, row_ror<1>(v));
  return v;
}
// atomicMax that skips the atomic when a (possibly stale) plain load already shows a larger value:
// after the first few workgroups almost every call is a load only, so same-address contention vanishes.
__device__ __forceinline__ void atomic_max_enc(unsigned* p, float v) {
  const unsigned e = enc_max(v);
  if (*reinterpret_cast<volatile unsigned*>(p) < e) atomicMax(p, e);
}
