// Full softmax attention (Attention.forward, ddpm.py:271-282 + Attend.forward, attend.py:84-113)
// on the NHWC qkv tensor [B, n, 3*hidden] with q already scaled by dim_head^-0.5
// (ld_conv1x1 LD_EPI_QKV_FULL);  out [B, n, hidden].
//
// Flash-style: the n x n similarity matrix (attend.py:102) is never materialised.  One workgroup
// = 64 queries of one (batch, head); K/V tiles of 64 keys go through LDS as fp32; wave w scores
// keys 16w..16w+15 of every tile for all 64 queries (lane = query) with an online softmax
// (running max m, normaliser l), and the four partial (m, l, o) are merged at the end.
// fp32 storage: VALU arithmetic below (the parity path, exact fp32 products).
// bf16 / fp16 storage: attention_mfma_kernel further down (QK^T and PV on v_mfma_f32_16x16x32_bf16 / _f16).
#include "common.hip.h"

namespace {
constexpr int D = 32;     // dim_head
constexpr int TK = 64;    // keys per tile
constexpr int KW = 16;    // keys per wave per tile

template <typename T>
__global__ __launch_bounds__(256) void attention_kernel(const T* __restrict__ qkv, T* __restrict__ out, int n,
                                                        int heads) {
  __shared__ __attribute__((aligned(16))) float s_k[TK][D];
  __shared__ __attribute__((aligned(16))) float s_v[TK][D];
  __shared__ float s_m[4][64], s_l[4][64];
  __shared__ float s_o[4][D][64];                       // [wave][d][query] -> conflict-free
  const int hidden = heads * D;
  const int b = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * 64;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const size_t rowstride = (size_t)3 * hidden;
  const T* base = qkv + (size_t)b * n * rowstride;
  const int qi = min(q0 + lane, n - 1);
  float q[D], o[D];
  {
    const T* qp = base + (size_t)qi * rowstride + h * D;
#pragma unroll
    for (int d = 0; d < D; d += 4) load4<T>(qp + d, q + d);
#pragma unroll
    for (int d = 0; d < D; ++d) o[d] = 0.f;
  }
  float m = -1e30f, l = 0.f;
  for (int j0 = 0; j0 < n; j0 += TK) {
    __syncthreads();
    // stage K and V tiles: 64 keys x 32 dims x 2 = 4096 floats, 256 threads x 4 float4... (4 floats each x4)
    for (int i = tid; i < TK * D / 4; i += 256) {
      const int key = i / (D / 4), d4 = (i - key * (D / 4)) * 4;
      const int j = j0 + key;
      float kv[4] = {0.f, 0.f, 0.f, 0.f}, vv[4] = {0.f, 0.f, 0.f, 0.f};
      if (j < n) {
        load4<T>(base + (size_t)j * rowstride + hidden + h * D + d4, kv);
        load4<T>(base + (size_t)j * rowstride + 2 * hidden + h * D + d4, vv);
      }
      *reinterpret_cast<float4*>(&s_k[key][d4]) = make_float4(kv[0], kv[1], kv[2], kv[3]);
      *reinterpret_cast<float4*>(&s_v[key][d4]) = make_float4(vv[0], vv[1], vv[2], vv[3]);
    }
    __syncthreads();
    float s[KW];
    float tmax = -1e30f;
#pragma unroll
    for (int kk = 0; kk < KW; ++kk) {
      const int key = wv * KW + kk;
      float acc = 0.f;
#pragma unroll
      for (int d = 0; d < D; d += 4) {
        const float4 kvv = *reinterpret_cast<const float4*>(&s_k[key][d]);   // broadcast read
        acc = fmaf(q[d], kvv.x, acc); acc = fmaf(q[d + 1], kvv.y, acc);
        acc = fmaf(q[d + 2], kvv.z, acc); acc = fmaf(q[d + 3], kvv.w, acc);
      }
      const bool ok = (j0 + key) < n;
      s[kk] = ok ? acc : -1e30f;
      tmax = fmaxf(tmax, s[kk]);
    }
    const float mn = fmaxf(m, tmax);
    const float alpha = expf(m - mn);
    l *= alpha;
#pragma unroll
    for (int d = 0; d < D; ++d) o[d] *= alpha;
    m = mn;
#pragma unroll
    for (int kk = 0; kk < KW; ++kk) {
      const int key = wv * KW + kk;
      const bool ok = (j0 + key) < n;
      const float p = ok ? expf(s[kk] - m) : 0.f;
      l += p;
#pragma unroll
      for (int d = 0; d < D; d += 4) {
        const float4 vvv = *reinterpret_cast<const float4*>(&s_v[key][d]);
        o[d] = fmaf(p, vvv.x, o[d]); o[d + 1] = fmaf(p, vvv.y, o[d + 1]);
        o[d + 2] = fmaf(p, vvv.z, o[d + 2]); o[d + 3] = fmaf(p, vvv.w, o[d + 3]);
      }
    }
  }
  s_m[wv][lane] = m;
  s_l[wv][lane] = l;
#pragma unroll
  for (int d = 0; d < D; ++d) s_o[wv][d][lane] = o[d];
  __syncthreads();
  if (wv == 0) {
    float M = fmaxf(fmaxf(s_m[0][lane], s_m[1][lane]), fmaxf(s_m[2][lane], s_m[3][lane]));
    float f[4], L = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) { f[w] = expf(s_m[w][lane] - M); L += s_l[w][lane] * f[w]; }
    const float inv = 1.0f / L;
    float r[D];
#pragma unroll
    for (int d = 0; d < D; ++d)
      r[d] = (s_o[0][d][lane] * f[0] + s_o[1][d][lane] * f[1] + s_o[2][d][lane] * f[2] + s_o[3][d][lane] * f[3]) * inv;
    if (q0 + lane < n) {
      T* op = out + ((size_t)b * n + q0 + lane) * hidden + h * D;
#pragma unroll
      for (int d = 0; d < D; d += 4) store4<T>(op + d, r + d);
    }
  }
}

// ------------------------------------------------------------------ bf16 MFMA flash attention
// One workgroup = 64 queries (wave w: queries 16w..16w+15) of one (batch, head); key/value tiles of
// 64 keys stream through LDS.  Orientation chosen so that NOTHING is shuffled between the two
// products (cdna_hip_programming.md section 3, "an accumulator tile as the next MFMA's operand"):
//   S^T[j][i] = K Q^T   A = K rows (keys) read as 16-B fragments, B = Q held in registers;
//                       D: lane = query i (lane&15), regs = keys 4*(lane>>4)+r  -> the softmax row
//                       state (m, l) of a query lives in ONE lane (replicated over lane>>4);
//   O^T[d][i] = V^T P^T A = V^T fragments via ds_read_b64_tr_b16 (transposed LDS read of the
//                       row-major [key][d] tile), B = P^T = the exponentiated S^T registers as they
//                       stand: k-slot (lane>>4, e) <-> key 4*(lane>>4)+e of tile e<4 ? 0 : 1.
// LDS: K as [kgrp][key][16 B] planes (conflict-free b128 reads), V as 96-byte rows (64 B data +
// 32 B pad: the 8 rows a half-wave's transposed read touches land on 8 disjoint bank octets).
typedef __attribute__((ext_vector_type(4))) short s16x4;
constexpr int VROW = 96;                 // (TKV = keys per tile, a template parameter: one barrier pair, one row maximum and one rescale per tile)

__device__ __forceinline__ uint2 tr_read(const char* p) {
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) s16x4*)(const_cast<char*>(p)));
  return __builtin_bit_cast(uint2, v);
}

// KS = 2 (launches with at most one 256-thread workgroup per CU -- every launch of the two-sub-batch regime and of
// cfg5): the workgroup is TWO groups of 4 waves, group g walks the g-th half of the keys with its own LDS tiles for the
// SAME 64 queries, and group 0 merges the two (m, l, O) states at the end (flash-decoding's split, inside a
// workgroup).  One wave per SIMD left the softmax VALU and the MFMAs of a tile strictly serial; with two waves per
// SIMD one group's exponentials run under the other's matrix products, and each wave's chain of tiles is half as long.
template <typename T, int TKV, int KS>
__global__ __launch_bounds__(256 * KS) void attention_mfma_kernel(const T* __restrict__ qkv, T* __restrict__ out,
                                                                  int n, int heads, int xcd_map) {
  constexpr int NKT = TKV / 16, NST = TKV / 64;          // 16-key MFMA tiles per tile; staging rows per thread
  constexpr int KBYTES = 4 * (TKV + 1) * 16, VBYTES = TKV * VROW;
  extern __shared__ __attribute__((aligned(16))) char smem[];         // per group: K planes [4][TKV + 1][16 B], V rows [TKV][VROW]
  const int grp = KS == 1 ? 0 : (int)(threadIdx.x >> 8);
  uint4 (*s_k)[TKV + 1] = reinterpret_cast<uint4 (*)[TKV + 1]>(smem + grp * (KBYTES + VBYTES));   // +1: the 4 chunk lanes of a key write 4 distinct bank quads
  char* s_v = smem + grp * (KBYTES + VBYTES) + KBYTES;
  const int hidden = heads * D;
  // Workgroup -> (image, head, query tile).  The grid is ONE-dimensional and workgroups are dealt round-robin over the 8
  // XCDs (id % 8 says which workgroups share an XCD's L2: observed, for speed only -- any placement computes the same).
  // Every query tile of an (image, head) re-reads that head's whole K and V; with the natural order the 16-64 tiles of a
  // head sat on all eight XCDs and each L2 fetched its own copy (rocprofv3 FETCH_SIZE: 4.9x the algorithmic bytes at
  // n = 1,024).  Here XCD x owns whole (image, head) pairs -- adjacent heads of one image, whose 64-byte K / V slices
  // share 128-byte lines -- or, with fewer pairs than XCDs (cfg5: one image), an interleaved share of one pair's tiles.
  int b, h, q0;
  {
    const int nq = (n + 63) / 64, npair = (int)gridDim.x / nq, id = blockIdx.x;
    const int x = id & 7, s = id >> 3;
    int pair = id / nq, qt = id % nq;                      // natural order (any grid)
    if (!xcd_map) {
    } else if ((npair & 7) == 0) {                         // XCD x: pairs [x G, (x + 1) G), all their tiles
      const int G = npair >> 3;
      pair = x * G + s / nq;
      qt = s % nq;
    } else if (npair < 8 && 8 % npair == 0 && nq % (8 / npair) == 0) {   // a pair spans R XCDs, tile q on XCD (q mod R)
      const int R = 8 / npair;
      pair = x / R;
      qt = s * R + x % R;
    }
    b = pair / heads;
    h = pair - b * heads;
    q0 = qt * 64;
  }
  const int tid = threadIdx.x & 255, lane = tid & 63, wv = tid >> 6, li = lane & 15, kg = lane >> 4;
  const size_t rowstride = (size_t)3 * hidden;
  const T* base = qkv + (size_t)b * n * rowstride;
  // keys of this group: [kbeg, kend); every group runs the same number of tiles (the barriers are workgroup-wide)
  const int span = ((n + KS * TKV - 1) / (KS * TKV)) * TKV;
  const int kbeg = grp * span, kend = min(n, kbeg + span);
  const int qi = q0 + wv * 16 + li;
  uint4 qf = make_uint4(0u, 0u, 0u, 0u);
  if (qi < n) qf = *reinterpret_cast<const uint4*>(base + (size_t)qi * rowstride + h * D + kg * 8);
  f32x4 o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};
  float m = -1e30f;
  // staging role: 4 consecutive lanes fetch the 64 contiguous bytes of one key's head slice (one line request per
  // key; with one lane per key every lane touched its own 128-B line for 16 bytes: 8x the L2->L1 traffic)
  const int skey = tid >> 2, schunk = tid & 3;
  const int tq = (lane >> 2) & 3, tp = lane & 3;         // transposed-read role inside the 16-lane group
  // register-staged prefetch (T14): the K/V rows of tile j+1 are requested before tile j's MFMAs, so the
  // tiles of a 1024-key sequence cost one exposed global round trip instead of one each
  uint4 kk[NST], vv[NST];
  auto request = [&](int j0) {
#pragma unroll
    for (int u = 0; u < NST; ++u) {
      const int j = j0 + u * 64 + skey;
      kk[u] = make_uint4(0u, 0u, 0u, 0u);
      vv[u] = kk[u];
      if (j < kend) {
        const T* rp = base + (size_t)j * rowstride + h * D + schunk * 8;
        kk[u] = *reinterpret_cast<const uint4*>(rp + hidden);
        vv[u] = *reinterpret_cast<const uint4*>(rp + 2 * hidden);
      }
    }
  };
  request(kbeg);
  // The normaliser l = sum_j P_ij comes out of the matrix pipe: a third A tile of ONES beside the two V^T tiles makes
  // every row of its product the column sum of P^T (of the P that is actually multiplied: the storage-rounded one, so
  // the output is an exactly normalised average of V rows).  32 dependent v_add per 128 keys become 4 MFMAs on a pipe
  // that is a quarter busy.
  const unsigned one2 = pack2<T>(1.0f, 1.0f);
  const uint4 ones = make_uint4(one2, one2, one2, one2);
  f32x4 lacc = {0.f, 0.f, 0.f, 0.f};
  // One TKV-key tile.  FULL: no key of the tile is past the group's range (every tile at the model's sizes): the four
  // instructions per score that masked such keys (index, compare, mask merge, select: 128 of the ~400 instructions of
  // a tile) exist only in the ragged variant.
  auto tile = [&](int j0, auto full_tag) {
    constexpr bool FULL = decltype(full_tag)::value;
    f32x4 s[NKT];
    float mx = -1e30f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      const uint4 kf = s_k[kg][kt * 16 + li];
      s[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
      mma16<T>(s[kt], kf, qf);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if constexpr (!FULL) {
          if ((j0 + kt * 16 + kg * 4 + r) >= kend) s[kt][r] = -1e30f;
        }
        mx = fmaxf(mx, s[kt][r]);
      }
    }
    mx = kq4_max(mx);                                    // over the 4 kg lanes of the query (VALU exchanges, common.hip.h)
    const float mn = fmaxf(m, mx);
    const float alpha = __builtin_amdgcn_exp2f((m - mn) * 1.4426950408889634f);
    m = mn;
#pragma unroll
    for (int r = 0; r < 4; ++r) { o0[r] *= alpha; o1[r] *= alpha; lacc[r] *= alpha; }
    unsigned pk[NKT][2];
    const float mn2 = mn * 1.4426950408889634f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      float pv[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) pv[r] = __builtin_amdgcn_exp2f(fmaf(s[kt][r], 1.4426950408889634f, -mn2));   // masked keys: exp2(-huge) = 0
      pk[kt][0] = pack2<T>(pv[0], pv[1]);
      pk[kt][1] = pack2<T>(pv[2], pv[3]);
    }
#pragma unroll
    for (int ks = 0; ks < NKT / 2; ++ks) {
      const uint4 pf = make_uint4(pk[2 * ks][0], pk[2 * ks][1], pk[2 * ks + 1][0], pk[2 * ks + 1][1]);
      const char* vrow = s_v + (ks * 32 + kg * 4 + tq) * VROW + tp * 8;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        const uint2 v1 = tr_read(vrow + dt * 32);
        const uint2 v2 = tr_read(vrow + 16 * VROW + dt * 32);
        const uint4 vf = make_uint4(v1.x, v1.y, v2.x, v2.y);
        if (dt == 0) mma16<T>(o0, vf, pf);
        else mma16<T>(o1, vf, pf);
      }
      mma16<T>(lacc, ones, pf);
    }
  };
  for (int j0 = kbeg; j0 < kbeg + span; j0 += TKV) {
    __syncthreads();
#pragma unroll
    for (int u = 0; u < NST; ++u) {
      s_k[schunk][u * 64 + skey] = kk[u];
      *reinterpret_cast<uint4*>(s_v + (u * 64 + skey) * VROW + schunk * 16) = vv[u];
    }
    __syncthreads();
    if (j0 + TKV < kend) request(j0 + TKV);
    if (j0 + TKV <= kend) tile(j0, std::true_type{});
    else tile(j0, std::false_type{});
  }
  float lsum = lacc[0];
  if constexpr (KS == 2) {
    // group 1 hands (m, l, O^T fragment) over through LDS (its tiles are dead); group 0 merges.  A group without a
    // single key in range (short sequences) has m = -1e30: its weight underflows to exactly 0.
    __syncthreads();
    float* s_mrg = reinterpret_cast<float*>(smem);       // [10][256]
    if (grp == 1) {
      s_mrg[0 * 256 + tid] = m;
      s_mrg[1 * 256 + tid] = lsum;
#pragma unroll
      for (int r = 0; r < 4; ++r) { s_mrg[(2 + r) * 256 + tid] = o0[r]; s_mrg[(6 + r) * 256 + tid] = o1[r]; }
    }
    __syncthreads();
    if (grp == 1) return;
    const float m1 = s_mrg[0 * 256 + tid], l1 = s_mrg[1 * 256 + tid];
    const float M = fmaxf(m, m1);
    const float f0 = __builtin_amdgcn_exp2f((m - M) * 1.4426950408889634f), f1 = __builtin_amdgcn_exp2f((m1 - M) * 1.4426950408889634f);
    lsum = lsum * f0 + l1 * f1;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      o0[r] = o0[r] * f0 + s_mrg[(2 + r) * 256 + tid] * f1;
      o1[r] = o1[r] * f0 + s_mrg[(6 + r) * 256 + tid] * f1;
    }
  }
  {
    // the head's two 16-channel halves leave as ONE 16-byte store per lane (pair_frag16: the exchange runs on every
    // lane, only the store is predicated)
    const float inv = 1.0f / lsum;
    char* op = reinterpret_cast<char*>(out + ((size_t)b * n + (qi < n ? qi : 0)) * hidden + h * D);
    float r0[4] = {o0[0] * inv, o0[1] * inv, o0[2] * inv, o0[3] * inv};
    float r1[4] = {o1[0] * inv, o1[1] * inv, o1[2] * inv, o1[3] * inv};
    const uint4 w16 = pair_frag16<T>(r0, r1);
    if (qi < n) store16_out(op + pair_frag16_off(kg), w16);
  }
}

template <typename T, int TKV, int KS>
int launch_mfma(dim3 grid, hipStream_t st, const void* qkv, void* out, int n, int heads) {
  const size_t lds = (size_t)KS * (4 * (TKV + 1) * 16 + TKV * VROW);
  if (ld_allow_lds(attention_mfma_kernel<T, TKV, KS>, lds) != hipSuccess) return ld_fail(LD_EHIP, "ld_attention: %zu bytes of LDS refused", lds);
  LD_LAUNCH((attention_mfma_kernel<T, TKV, KS>), dim3(grid.x * grid.y * grid.z), dim3(256 * KS), lds, st, (const T*)qkv, (T*)out, n, heads,
            (int)ld_tuning().attn_xcd_map);
  return LD_OK;
}
}  // namespace

extern "C" int ld_attention(const void* qkv, void* out, int B, int n, int heads, int dim_head, int dtype,
                            void* stream) {
  LD_REQUIRE(qkv && out && B > 0 && n > 0 && heads > 0, "ld_attention: bad args");
  LD_REQUIRE(dim_head == D, "ld_attention: dim_head must be 32 (got %d)", dim_head);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  dim3 grid((n + 63) / 64, heads, B);
  if (dtype == LD_F32)
    LD_LAUNCH(attention_kernel<float>, grid, dim3(256), 0, st, (const float*)qkv, (float*)out, n, heads);
  else if (ld_dtype_16(dtype))
    return LD_DISPATCH16(dtype, [&] {
      // long sequences (cfg5: n = 4,096): 256-key tiles halve the barriers, row reductions and rescales per key;
      // launches of at most attn_split_max_wgs workgroups (tuning table; default 256 = one per CU): two key groups
      // (measured: n = 4,096 at B = 1 alone 27.6 -> 23.8 us, cfg5 +0.3 %; n = 1,024 at B = 4 alone 9.2 -> 8.5 us but in
      //  the two-sub-batch step +0.2 % -- the other stream's launch already is the second wave per SIMD: long sequences only)
      // (and only when the second group owns at least one key, n > tile size: a group whose every score is the masking
      //  constant has exp2(rounding error of -1e30 * log2 e) = +inf weights and its inf * 0 products are NaN for every query)
      const int tkv = n >= 2048 ? 256 : 128;
      const bool split = (long)grid.x * grid.y * grid.z <= ld_tuning().attn_split_max_wgs && n >= ld_tuning().attn_split_min_n && n > tkv;
      int rc;
      if (n >= 2048) rc = split ? launch_mfma<T, 256, 2>(grid, st, qkv, out, n, heads) : launch_mfma<T, 256, 1>(grid, st, qkv, out, n, heads);
      else rc = split ? launch_mfma<T, 128, 2>(grid, st, qkv, out, n, heads) : launch_mfma<T, 128, 1>(grid, st, qkv, out, n, heads);
      if (rc != LD_OK) return rc;
      LD_LAUNCH_CHECK("attention");
      return (int)LD_OK;
    }());
  else
    return ld_fail(LD_EINVAL, "ld_attention: bad dtype %d", dtype);
  LD_LAUNCH_CHECK("attention");
  return LD_OK;
}
