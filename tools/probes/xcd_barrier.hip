// Probe (gfx950): what does a phase boundary cost INSIDE one persistent launch, compared with a kernel boundary?
// ("one persistent kernel per sub-batch for the <= 64^2 section, one XCD per patch, phases
// separated by an XCD-local counter barrier, budget <= 2 us per barrier".)
//
// Work model of one phase = what a small-map layer does to memory: every workgroup (one per CU, 256 threads)
// writes a 16-KiB slab, and after the boundary reads the slab another workgroup of ITS group wrote (checking every
// word, so a stale read is counted, not assumed away), plus `spin` cycles of ALU work.
// Boundary variants:
//   0  kernel boundary: one launch per phase (same stream, back to back)
//   1  agent-scope barrier over all 256 workgroups: lane-0 release fence -> counter -> poll -> acquire fence
//      (the placement-independent form: correct whatever XCD a workgroup landed on)
//   2  XCD-local barrier: groups are DEFINED by s_getreg(HW_REG_XCC_ID) (so "same XCD" is a fact, not an assumption
//      about round-robin placement; a group has however many workgroups landed there), producers store plainly (the
//      line stays in that XCD's L2), drain with s_waitcnt vmcnt(0), add to the group's counter; consumers poll it
//      with sc1 loads and read the slab with sc1 loads (L1 bypass, L2-served): no buffer_wbl2, no buffer_inv
//   3  as 2, but plain slab loads behind a buffer_inv sc1 (L1 invalidate) after the poll
// Reports microseconds per phase (wall clock of the whole run / phases) and the number of stale words seen.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define SLAB_WORDS 4096   // 16 KiB per workgroup

__device__ __forceinline__ unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xf;
}
__device__ __forceinline__ unsigned load_sc1(const unsigned* p) {
  unsigned v;
  asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ uint4 load4_plain(const uint4* p) {
  uint4 v;
  asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ uint4 load4_sc1(const uint4* p) {
  uint4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}

struct Ctl {
  unsigned ticket[8];        // workgroups seen per XCD (group membership)
  unsigned pad0[24];
  unsigned arrive[8][32];    // per-XCD arrival counters, one 128-B line each
  unsigned all[32];          // chip-wide arrival counter
  unsigned stale;
  unsigned members[8][64];   // blockIdx of the k-th member of XCD g
};

__device__ __forceinline__ void work(unsigned* slab, unsigned tag, int spin) {
  for (int i = threadIdx.x; i < SLAB_WORDS; i += 256) slab[i] = tag + i;
  unsigned a = threadIdx.x;
  for (int i = 0; i < spin; ++i) a = a * 1664525u + 1013904223u;
  if (a == 0x12345u) slab[0] = a;
}

__global__ __launch_bounds__(256) void phase_kernel(unsigned* slabs, Ctl* c, int phase, int spin, int check) {
  extern __shared__ unsigned pad[];        // 80 KB of dynamic LDS: one workgroup per CU
  pad[threadIdx.x] = 0;
  unsigned* mine = slabs + ((size_t)(phase & 1) * gridDim.x + blockIdx.x) * SLAB_WORDS;      // slabs alternate by phase parity
  if (check && phase > 0) {
    const unsigned* other = slabs + ((size_t)((phase - 1) & 1) * gridDim.x + (blockIdx.x + 8) % gridDim.x) * SLAB_WORDS;
    unsigned bad = 0;
    const unsigned want = (phase - 1) * 1000003u + ((blockIdx.x + 8) % gridDim.x) * 7919u;
    for (int i = threadIdx.x; i < SLAB_WORDS; i += 256) bad += other[i] != want + i;
    if (bad) atomicAdd(&c->stale, bad);
  }
  __syncthreads();
  work(mine, phase * 1000003u + blockIdx.x * 7919u, spin);
}

template <int MODE>
__global__ __launch_bounds__(256) void persistent_kernel(unsigned* slabs, Ctl* c, int phases, int spin, unsigned long long* cyc) {
  extern __shared__ unsigned pad[];
  __shared__ unsigned s_g, s_k, s_n;
  pad[threadIdx.x] = 0;
  const int tid = threadIdx.x;
  if (tid == 0) {
    const unsigned g = MODE == 1 ? 0 : xcc_id();
    s_g = g;
    s_k = atomicAdd(&c->ticket[g], 1u);
    c->members[g][s_k] = blockIdx.x;
  }
  __syncthreads();
  const unsigned g = s_g, k = s_k;
  // everyone has taken a ticket once the chip-wide counter reaches gridDim.x (one-off, agent scope)
  if (tid == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    atomicAdd(&c->all[0], 1u);
    while (load_sc1(&c->all[0]) < gridDim.x) __builtin_amdgcn_s_sleep(2);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    s_n = MODE == 1 ? gridDim.x : load_sc1(&c->ticket[g]);
  }
  __syncthreads();
  const unsigned n = s_n;                                   // members of my group
  const unsigned partner = MODE == 1 ? (blockIdx.x + 8) % gridDim.x : c->members[g][(k + 1) % n];
  unsigned* mine = slabs + (size_t)blockIdx.x * SLAB_WORDS;
  const unsigned* other = slabs + (size_t)partner * SLAB_WORDS;
  unsigned long long t0 = __builtin_readcyclecounter(), tb = 0;
  for (int ph = 0; ph < phases; ++ph) {
    work(mine, ph * 1000003u + blockIdx.x * 7919u, spin);
    // ---- boundary
    const unsigned long long b0 = __builtin_readcyclecounter();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's stores have reached L2
    __syncthreads();
    if (MODE == 1) {
      if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");  // buffer_wbl2 sc1
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        atomicAdd(&c->all[1], 1u);
        while (load_sc1(&c->all[1]) < (unsigned)(ph + 1) * gridDim.x) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // buffer_inv sc1
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    } else {
      if (tid == 0) {
        atomicAdd(&c->arrive[g][0], 1u);
        while (load_sc1(&c->arrive[g][0]) < (unsigned)(ph + 1) * n) __builtin_amdgcn_s_sleep(1);
        if (MODE == 3) {
          asm volatile("buffer_inv sc1\n\ts_waitcnt vmcnt(0)" ::: "memory");
        }
      }
    }
    __syncthreads();
    tb += __builtin_readcyclecounter() - b0;
    // ---- consume the partner's slab of this phase
    unsigned bad = 0;
    const unsigned want = ph * 1000003u + partner * 7919u;
    if (MODE == 2) {
      for (int i = tid * 4; i < SLAB_WORDS; i += 1024) {
        const uint4 v = load4_sc1(reinterpret_cast<const uint4*>(other + i));
        bad += (v.x != want + i) + (v.y != want + i + 1) + (v.z != want + i + 2) + (v.w != want + i + 3);
      }
    } else {
      for (int i = tid * 4; i < SLAB_WORDS; i += 1024) {
        const uint4 v = load4_plain(reinterpret_cast<const uint4*>(other + i));
        bad += (v.x != want + i) + (v.y != want + i + 1) + (v.z != want + i + 2) + (v.w != want + i + 3);
      }
    }
    if (bad) atomicAdd(&c->stale, bad);
    // (the reader of a slab is another workgroup: slabs alternate by phase parity so that phase ph+1's writes cannot
    // race with a slow reader of phase ph; the boundary of phase ph+1 orders them against phase ph+2's)
    mine = slabs + ((size_t)((ph + 1) & 1) * gridDim.x + blockIdx.x) * SLAB_WORDS;
    other = slabs + ((size_t)((ph + 1) & 1) * gridDim.x + partner) * SLAB_WORDS;
  }
  if (tid == 0 && blockIdx.x == 0) { cyc[0] = __builtin_readcyclecounter() - t0; cyc[1] = tb; }
}

int main(int argc, char** argv) {
  const int phases = argc > 1 ? atoi(argv[1]) : 200;
  const int nwg = 256;
  unsigned* slabs; Ctl* c; unsigned long long* cyc;
  (void)hipMalloc(&slabs, (size_t)2 * nwg * SLAB_WORDS * 4);
  (void)hipMalloc(&c, sizeof(Ctl));
  (void)hipMalloc(&cyc, 16);
  const size_t lds = 80 * 1024;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(phase_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(persistent_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(persistent_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(persistent_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int spin : {0, 2000}) {
    for (int mode = 0; mode < 4; ++mode) {
      float best = 1e30f; unsigned stale = 0; unsigned long long hc[2] = {0, 0};
      unsigned tick[8] = {0};
      for (int rep = 0; rep < 3; ++rep) {
        (void)hipMemset(c, 0, sizeof(Ctl));
        (void)hipMemset(slabs, 0xff, (size_t)2 * nwg * SLAB_WORDS * 4);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0, 0);
        if (mode == 0) {
          for (int ph = 0; ph < phases; ++ph) hipLaunchKernelGGL(phase_kernel, dim3(nwg), dim3(256), lds, 0, slabs, c, ph, spin, 1);
        } else if (mode == 1) {
          hipLaunchKernelGGL(persistent_kernel<1>, dim3(nwg), dim3(256), lds, 0, slabs, c, phases, spin, cyc);
        } else if (mode == 2) {
          hipLaunchKernelGGL(persistent_kernel<2>, dim3(nwg), dim3(256), lds, 0, slabs, c, phases, spin, cyc);
        } else {
          hipLaunchKernelGGL(persistent_kernel<3>, dim3(nwg), dim3(256), lds, 0, slabs, c, phases, spin, cyc);
        }
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
        Ctl h; (void)hipMemcpy(&h, c, sizeof(Ctl), hipMemcpyDeviceToHost);
        stale += h.stale;
        for (int g = 0; g < 8; ++g) tick[g] = h.ticket[g];
        if (mode) (void)hipMemcpy(hc, cyc, 16, hipMemcpyDeviceToHost);
      }
      const char* names[4] = {"kernel boundary (launch per phase)", "agent-scope barrier, 256 workgroups", "XCD-local barrier, sc1 slab loads", "XCD-local barrier, buffer_inv + plain loads"};
      printf("spin %4d  %-46s %7.2f us per phase", spin, names[mode], 1e3 * best / phases);
      if (mode) printf("   (in-kernel: %.0f cycles per phase, %.0f of them in the boundary)", (double)hc[0] / phases, (double)hc[1] / phases);
      printf("   stale words %u", stale);
      if (mode >= 2) { printf("   workgroups per XCD:"); for (int g = 0; g < 8; ++g) printf(" %u", tick[g]); }
      printf("\n");
    }
  }
  return 0;
}
