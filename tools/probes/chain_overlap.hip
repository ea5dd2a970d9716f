// Probe (gfx950): can the HEAD of a dependent launch run under the TAIL of its predecessor?
//
// A reverse step is ~104 dependent launches per sub-batch; at <= 64^2 each is a latency chain of 4-14 us whose first
// 1.5-2 us (dispatch after the predecessor drained, kernel arguments, the first weight chunk into LDS) does not depend on
// the predecessor's output.  HIP orders the launches of a stream with the AQL barrier bit; this probe measures what a chain
// costs when the ordering moves INTO the kernels instead -- a completion flag per launch that the successor polls right in
// front of its first dependent load -- and the launches themselves are free to start early:
//   mode 0  plain launches, one stream (the product's structure; no flags)
//   mode 1  hipExtLaunchKernelGGL(..., hipExtAnyOrderLaunch) on one stream + flags  (barrier bit cleared, if the runtime does it)
//   mode 2  launches alternate between two streams + flags (each stream in order: at most one successor spins)
// each eagerly and as replayed HIP graphs, for S = 1 and S = 2 independent chains ("sub-batches") at a time.
// Work model of a link (256 workgroups x 256 threads, 40 KB of LDS): head = `hspin` cycles of ALU + a 16-KB weight chunk
// into LDS; body = 8 dependent rounds of (16-byte load of the partner workgroup's slab written by the PREVIOUS link, checked
// word by word, + `bspin` cycles of ALU, + barrier); tail = write-through store of its own 32-KB slab, drain, completion count.
// Slabs ping-pong between two sets, so a stale line (L2 of another XCD, last written two links ago) is COUNTED.
// Consumer loads after the flag: variant 0 = sc1 loads, variant 1 = plain loads behind buffer_inv sc1.
// Every poll is bounded (timeouts are counted, never hang).
// build: hipcc --offload-arch=gfx950 -O2 -o tools/probes/chain_overlap tools/probes/chain_overlap.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int NWG = 256, SLAB_WORDS = 8192, ROUNDS = 8;      // 32 KiB per workgroup, 8 MiB per slab set

struct Ctl {                 // one per chain
  unsigned done[128][NWG];   // passes completed by workgroup b of link n: written by that workgroup only (no atomics anywhere)
  unsigned stale, timeouts;
  unsigned long long t_first[128][NWG], t_go[128][NWG], t_last[128][NWG];      // 100 MHz clock per workgroup: start, past the flags, end
};

__device__ __forceinline__ unsigned load_sc1(const unsigned* p) {
  unsigned v;
  asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ u32x4 load4_sc1(const u32x4* p) {
  u32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ void store4_sc1(u32x4* p, u32x4 v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(p), "v"(v) : "memory");    // s_nop: hipcc does not guard an asm store's data registers
}
__device__ __forceinline__ unsigned spin(unsigned a, int n) {
  for (int i = 0; i < n; ++i) a = a * 1664525u + 1013904223u;
  return a;
}

// FLAGS: 0 = ordered by the kernel boundary, 1 = ordered by the predecessor's flag.  LOADS: 0 = sc1, 1 = buffer_inv + plain
template <int FLAGS, int LOADS>
__global__ __launch_bounds__(256) void link_kernel(unsigned* slabs, const u32x4* wts, Ctl* c, int n, int prev, int prev_off, int pass_base,
                                                   int hspin, int bspin, int trace) {
  extern __shared__ u32x4 lds[];
  const int tid = threadIdx.x, b = blockIdx.x;
  unsigned long long t0 = 0;
  if (trace && tid == 0) { t0 = wall_clock64(); c->t_first[n][b] = t0; }
  // ---- head: nothing here depends on the predecessor
  u32x4 w[4];
  for (int i = 0; i < 4; ++i) w[i] = wts[(n & 7) * 1024 + i * 256 + tid];
  unsigned a = spin(tid, hspin);
  for (int i = 0; i < 4; ++i) lds[i * 256 + tid] = w[i];
  __syncthreads();
  // ---- the ordering point
  const unsigned* in = slabs + ((size_t)((n + 1) & 1) * NWG + (b + 8) % NWG) * SLAB_WORDS;       // written by link n-1
  unsigned* out = slabs + ((size_t)(n & 1) * NWG + b) * SLAB_WORDS;
  unsigned mine = 0;
  if (FLAGS) {
    if (tid < 64) {
      // my own slot says how many passes I have completed; every producer workgroup must have completed one more
      // (the same number for link 0, whose producer is the previous pass's last link).  One wave reads all 256 flags.
      mine = load_sc1(&c->done[n][b]);
      const unsigned want = mine + 1 - prev_off;
      int it = load_sc1(&c->timeouts) ? 20000 : 0;                          // after the first timeout nobody waits any more
      for (;;) {
        bool ok = true;
        for (int j = 0; j < NWG / 64; ++j) ok = ok && (int)(load_sc1(&c->done[prev][j * 64 + tid]) - want) >= 0;
        if (__all(ok)) break;
        __builtin_amdgcn_s_sleep(1);
        if (++it > 20000) { if (tid == 0) atomicAdd(&c->timeouts, 1u); break; }
      }
      if (LOADS == 1) asm volatile("buffer_inv sc1\n\ts_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
  }
  if (trace && tid == 0) c->t_go[n][b] = wall_clock64();
  // ---- body: 8 dependent rounds
  const unsigned tag = (unsigned)(pass_base + n - 1) * 1000003u + ((b + 8) % NWG) * 7919u;
  unsigned bad = 0;
  const bool check = pass_base + n > 0;
  u32x4 v;
  if (FLAGS && LOADS == 0) v = load4_sc1(reinterpret_cast<const u32x4*>(in) + tid);
  else v = reinterpret_cast<const u32x4*>(in)[tid];
  for (int r = 0; r < ROUNDS; ++r) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const u32x4 cur = v;
    if (r + 1 < ROUNDS) {
      if (FLAGS && LOADS == 0) v = load4_sc1(reinterpret_cast<const u32x4*>(in) + (r + 1) * 256 + tid);
      else v = reinterpret_cast<const u32x4*>(in)[(r + 1) * 256 + tid];
    }
    const unsigned base = tag + (r * 256 + tid) * 4;
    bad += (cur.x != base) + (cur.y != base + 1) + (cur.z != base + 2) + (cur.w != base + 3);
    a = spin(a + cur.x + lds[(r * 64 + tid) & 1023].x, bspin);
    __syncthreads();
  }
  if (check && bad) atomicAdd(&c->stale, bad);
  // ---- tail: the output leaves as write-through stores, then the completion count
  const unsigned mytag = (unsigned)(pass_base + n) * 1000003u + b * 7919u;
  for (int r = 0; r < ROUNDS; ++r) {
    const unsigned base = mytag + (r * 256 + tid) * 4;
    u32x4 o = {base, base + 1, base + 2, base + 3};
    if (a == 0x12345u) o.x = a;
    store4_sc1(reinterpret_cast<u32x4*>(out) + r * 256 + tid, o);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    if (FLAGS) asm volatile("global_store_dword %0, %1, off sc1" :: "v"(&c->done[n][b]), "v"(mine + 1) : "memory");
    if (trace) c->t_last[n][b] = wall_clock64();
  }
}

template <int FLAGS, int LOADS>
static void launch(hipStream_t st, bool any_order, unsigned* slabs, const u32x4* wts, Ctl* c, int n, int prev, int prev_off, int pass_base,
                   int hspin, int bspin, int trace) {
  if (any_order)
    hipExtLaunchKernelGGL((link_kernel<FLAGS, LOADS>), dim3(NWG), dim3(256), 40960, st, nullptr, nullptr, hipExtAnyOrderLaunch, slabs, wts, c, n, prev,
                          prev_off, pass_base, hspin, bspin, trace);
  else
    hipLaunchKernelGGL((link_kernel<FLAGS, LOADS>), dim3(NWG), dim3(256), 40960, st, slabs, wts, c, n, prev, prev_off, pass_base, hspin, bspin, trace);
}

int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 64;             // links per pass (<= 128)
  const int REP = argc > 2 ? atoi(argv[2]) : 40;           // passes timed
  const int hspin = argc > 3 ? atoi(argv[3]) : 80, bspin = argc > 4 ? atoi(argv[4]) : 100;
  u32x4* wts; CK(hipMalloc(&wts, 8 * 1024 * 16)); CK(hipMemset(wts, 1, 8 * 1024 * 16));
  unsigned* slabs[2]; Ctl* ctl[2];
  for (int s = 0; s < 2; ++s) { CK(hipMalloc(&slabs[s], (size_t)2 * NWG * SLAB_WORDS * 4)); CK(hipMalloc(&ctl[s], sizeof(Ctl))); }
  hipStream_t st[2][2];
  for (int s = 0; s < 2; ++s) for (int k = 0; k < 2; ++k) CK(hipStreamCreateWithFlags(&st[s][k], hipStreamNonBlocking));
  const char* names[3] = {"plain launches, one stream        ", "AnyOrder launches + flags, 1 stream", "alternating two streams + flags    "};
  for (int loads = 0; loads < 2; ++loads)
  for (int graph = 0; graph < 2; ++graph)
  for (int S = 1; S <= 2; ++S)
  for (int mode = 0; mode < 3; ++mode) [&] {
    if (mode == 0 && loads == 1) return;
    // one pass = N links; pass p's link 0 follows pass p-1's link N-1 (prev_off = 1: that flag is one ahead of mine)
    auto enqueue_pass = [&](int s, int pass, int trace, bool into_graph) {
      for (int n = 0; n < N; ++n) {
        hipStream_t q = st[s][mode == 2 ? (n & 1) : 0];
        const int prev = n == 0 ? N - 1 : n - 1, off = n == 0 ? 1 : 0;
        // pass_base makes the expected tags unique per pass when launched eagerly; a replayed graph bakes pass 1's in, so the
        // check only compares links of ONE pass there (link 0 is not checked in graphs: pass_base + n > 0 holds, so skip by tag reuse)
        const int pb = into_graph ? 0 : pass * N;
        if (mode == 0) launch<0, 0>(q, false, slabs[s], wts, ctl[s], n, prev, off, pb, hspin, bspin, trace);
        else if (loads == 0) launch<1, 0>(q, mode == 1, slabs[s], wts, ctl[s], n, prev, off, pb, hspin, bspin, trace);
        else launch<1, 1>(q, mode == 1, slabs[s], wts, ctl[s], n, prev, off, pb, hspin, bspin, trace);
      }
    };
    for (int s = 0; s < 2; ++s) {
      CK(hipMemset(ctl[s], 0, sizeof(Ctl)));
      CK(hipMemset(slabs[s], 0xff, (size_t)2 * NWG * SLAB_WORDS * 4));
    }
    CK(hipDeviceSynchronize());
    std::vector<hipGraphExec_t> ex;
    if (graph) {
      for (int s = 0; s < S; ++s) for (int k = 0; k < (mode == 2 ? 2 : 1); ++k) {
        // capture one stream's share of a pass
        hipGraph_t g; hipGraphExec_t e;
        CK(hipStreamBeginCapture(st[s][k], hipStreamCaptureModeThreadLocal));
        for (int n = 0; n < N; ++n) {
          if (mode == 2 && (n & 1) != k) continue;
          const int prev = n == 0 ? N - 1 : n - 1, off = n == 0 ? 1 : 0;
          if (mode == 0) launch<0, 0>(st[s][k], false, slabs[s], wts, ctl[s], n, prev, off, 0, hspin, bspin, 0);
          else if (loads == 0) launch<1, 0>(st[s][k], mode == 1, slabs[s], wts, ctl[s], n, prev, off, 0, hspin, bspin, 0);
          else launch<1, 1>(st[s][k], mode == 1, slabs[s], wts, ctl[s], n, prev, off, 0, hspin, bspin, 0);
        }
        if (hipStreamEndCapture(st[s][k], &g) != hipSuccess) { printf("%s graph: capture failed\n", names[mode]); (void)hipGetLastError(); return; }
        CK(hipGraphInstantiate(&e, g, nullptr, nullptr, 0));
        CK(hipGraphDestroy(g));
        ex.push_back(e);
      }
    }
    auto run = [&](int pass0, int passes) {
      for (int p = 0; p < passes; ++p) for (int s = 0; s < S; ++s) {
        if (graph) {
          if (mode == 2) { CK(hipGraphLaunch(ex[2 * s], st[s][0])); CK(hipGraphLaunch(ex[2 * s + 1], st[s][1])); }
          else CK(hipGraphLaunch(ex[s], st[s][0]));
        } else enqueue_pass(s, pass0 + p, 0, false);
      }
    };
    run(0, 5); CK(hipDeviceSynchronize());
    auto t0 = std::chrono::steady_clock::now();
    run(5, REP); CK(hipDeviceSynchronize());
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    // one traced pass (eager), stream 0 only: gaps between a link's last end and the next link's first start / go
    double gap_start = 0, gap_go = 0, dur = 0;
    static Ctl h;
    if (!graph) {
      enqueue_pass(0, 5 + REP, 1, false); CK(hipDeviceSynchronize());
      CK(hipMemcpy(&h, ctl[0], sizeof(Ctl), hipMemcpyDeviceToHost));
      auto mn = [&](unsigned long long* a) { unsigned long long m = ~0ull; for (int b = 0; b < NWG; ++b) m = a[b] < m ? a[b] : m; return (double)m; };
      auto mx = [&](unsigned long long* a) { unsigned long long m = 0; for (int b = 0; b < NWG; ++b) m = a[b] > m ? a[b] : m; return (double)m; };
      for (int n = 8; n < N; ++n) {
        gap_start += (mn(h.t_first[n]) - mx(h.t_last[n - 1])) * 0.01;
        gap_go += (mn(h.t_go[n]) - mx(h.t_last[n - 1])) * 0.01;
        dur += (mx(h.t_last[n]) - mx(h.t_last[n - 1])) * 0.01;
      }
      gap_start /= N - 8; gap_go /= N - 8; dur /= N - 8;
    }
    unsigned stale = 0, timeouts = 0;
    for (int s = 0; s < S; ++s) { CK(hipMemcpy(&h, ctl[s], sizeof(Ctl), hipMemcpyDeviceToHost)); stale += h.stale; timeouts += h.timeouts; }
    printf("%s %s %s S=%d: %6.2f us per link per chain", names[mode], mode == 0 ? "           " : loads ? "inv + plain" : "sc1 loads  ",
           graph ? "graph" : "eager", S, us / (REP * N));
    if (!graph) printf("   [traced pass: end-to-end %.2f us per link, next first-start %+.2f us, next go %+.2f us after the last end]", dur, gap_start, gap_go);
    printf("   stale %u timeouts %u\n", stale, timeouts);
    fflush(stdout);
    for (auto e : ex) (void)hipGraphExecDestroy(e);
  }();
  return 0;
}
