// Probe of ds_read_b64_tr_b16 lane semantics (gfx950).  LDS holds M[r][c] = r*100 + c (shorts),
// 32 columns per row.  Lane l: group g = l>>4, q = (l>>2)&3, p = l&3 supplies the address of
// row (4g+q), columns 4p..4p+3.  Prints the 4 shorts each lane receives.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) short s16x4;
__global__ void k(short* out) {
  __shared__ __attribute__((aligned(16))) short lds[64 * 32];
  for (int i = threadIdx.x; i < 64 * 32; i += 64) lds[i] = (short)((i / 32) * 100 + (i % 32));
  __syncthreads();
  const int l = threadIdx.x, g = l >> 4, q = (l >> 2) & 3, p = l & 3;
  auto* ptr = (__attribute__((address_space(3))) s16x4*)(lds + (4 * g + q) * 32 + 4 * p);
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(ptr);
  *(s16x4*)(out + l * 4) = v;
}
int main() {
  short* d; short h[256];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) printf("lane %2d: %5d %5d %5d %5d\n", l, h[4*l], h[4*l+1], h[4*l+2], h[4*l+3]);
  return 0;
}
