// Weight repacking into MFMA fragment order (run once at model load).
//
// Packed layout, storage dtype, E = 16/sizeof(T) elements per fragment, CK = 4E channels/chunk:
//   k=3:  [chunk][tap 0..8][mtile][kq 0..3][i 0..15][e]   = W[16*mtile+i][chunk*CK+kq*E+e][tap/3][tap%3]
//   k=1:  [chunk][mtile][kq][i][e]                         = W[16*mtile+i][chunk*CK+kq*E+e]
// Two-term weights (terms = 2, 16-bit storage): every chunk is followed by a second chunk of the same shape holding
// the rounding REMAINDERS, W = hi + lo with hi = round(W), lo = round(W - hi): [chunk][term 0..1][...].  The
// convolution kernels walk 2*nch "virtual" chunks whose source chunk is v/2 and whose weight chunk is v, so the
// products are x*hi + x*lo = x*W to ~2^-17 (bf16) instead of 2^-9 -- rounding the weights is what separates a 16-bit
// sampling chain from the fp32 reference (tools/exp_error_budget.py).
// so that lane l = kq*16+i of a wave reads its A fragment (16 B) at byte offset 16*l of a 1 KiB
// block: global->LDS staging is a linear copy and LDS fragment reads are conflict-free.
#include "common.hip.h"

namespace {
template <typename T>
__global__ void pack_kernel(const float* __restrict__ w, const float* __restrict__ scale_in, T* out,
                            int cout, int cin, int ks, int unshuffle, int terms) {
  constexpr int E = DT<T>::E, CK = DT<T>::CK;
  const long total = (long)cout * cin * ks * ks * terms;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int mt_total = cout / 16, taps = ks * ks;
  long r = idx;
  const int e = r % E; r /= E;
  const int i = r % 16; r /= 16;
  const int kq = r % 4; r /= 4;
  const int m = r % mt_total; r /= mt_total;
  const int tap = r % taps; r /= taps;
  const int term = (int)(r % terms); r /= terms;
  const int ch = (int)r;
  const int co = m * 16 + i;
  int ci = ch * CK + kq * E + e;
  if (unshuffle) {            // packed K order (p1,p2,c) <- reference order (c,p1,p2)
    const int c4 = cin / 4, pp = ci / c4, c = ci - pp * c4;
    ci = c * 4 + pp;
  }
  float v = w[((long)co * cin + ci) * taps + tap];
  if (scale_in) v *= scale_in[ci];
  const T hi = from_f<T>(v);
  out[idx] = term == 0 ? hi : from_f<T>(v - to_f<T>(hi));
}
}  // namespace

extern "C" int ld_pack_conv_weight(const float* w, const float* scale_in, void* out, int cout, int cin,
                                   int ksize, int unshuffle, int dtype, void* stream) {
  return ld_pack_conv_weight_terms(w, scale_in, out, cout, cin, ksize, unshuffle, dtype, 1, stream);
}

extern "C" int ld_pack_conv_weight_terms(const float* w, const float* scale_in, void* out, int cout, int cin,
                                         int ksize, int unshuffle, int dtype, int terms, void* stream) {
  LD_REQUIRE(w && out, "ld_pack_conv_weight: null pointer");
  LD_REQUIRE(terms == 1 || (terms == 2 && ld_dtype_16(dtype)), "ld_pack_conv_weight: terms %d (2 needs 16-bit storage)", terms);
  LD_REQUIRE(ksize == 1 || ksize == 3, "ld_pack_conv_weight: ksize %d", ksize);
  LD_REQUIRE(cout % 16 == 0 && cin % 32 == 0, "ld_pack_conv_weight: cout %% 16 / cin %% 32 (%d,%d)", cout, cin);
  LD_REQUIRE(!(unshuffle && (ksize != 1 || cin % 128 != 0)), "ld_pack_conv_weight: unshuffle needs k=1, cin %% 128 == 0");
  const long total = (long)cout * cin * ksize * ksize * terms;
  const int bs = 256;
  const unsigned grid = (unsigned)((total + bs - 1) / bs);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  LD_REQUIRE(ld_dtype_ok(dtype), "ld_pack_conv_weight: bad dtype %d", dtype);
  LD_DISPATCH(dtype, [&] {
    LD_LAUNCH(pack_kernel<T>, dim3(grid), dim3(bs), 0, st, w, scale_in, (T*)out, cout, cin, ksize, unshuffle, terms);
    return 0;
  }());
  LD_LAUNCH_CHECK("pack_conv_weight");
  return LD_OK;
}
