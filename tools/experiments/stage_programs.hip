// Stage programs: a run of consecutive launches of one denoiser evaluation executed by ONE persistent launch.
//
// Why: the <= 64^2 half of a reverse step is ~60 dependent launches of 10-15 us
// whose cost is mostly fixed per launch; tools/probes/xcd_barrier.hip measures a kernel boundary between two small
// phases at 3.6 us, an agent-scope grid barrier at 12 us, and a barrier among the workgroups of ONE XCD at 0.9-1.7 us.
// A patch's 32^2 / 64^2 tensors (<= 1 MB in 16-bit storage) fit the 4 MiB L2 of one XCD, so:
//
//   * one persistent launch runs the recorded phases (the SAME tile functions the stand-alone kernels run:
//     conv3x3_tile, gn_apply_tile) for every image of the batch, ONE IMAGE PER XCD;
//   * a workgroup's group is DEFINED by the XCD it finds itself on (s_getreg HW_REG_XCC_ID) -- "same XCD" is a fact
//     the kernel reads, never an assumption about the dispatcher.  The first workgroup of an XCD claims an image for
//     it (preferring image (xcc - xcd_base) mod 8, so that the two sub-batch streams of the sampler settle on disjoint
//     XCDs; any XCD may claim what is left after a grace period, so every image is processed whatever the placement);
//     XCDs without an image exit at once;
//   * inside a phase the group's workgroups take tiles from a per-(XCD, phase) counter (no workgroup needs to know how
//     many others there are, or that they are resident: a tile is only ever claimed by a running workgroup);
//   * the phase boundary is XCD-local: every wave drains its stores and statistics atomics (s_waitcnt vmcnt(0)), one
//     lane adds the workgroup's tile count to the (XCD, phase) done-counter, polls it with sc1 loads, and invalidates
//     the CU's L1 (buffer_inv sc1).  Producers and consumers share the XCD's L2, so no L2 write-back is needed
//     (MI355X_MICROARCH.md "Workgroup dispatch, XCD placement & inter-workgroup visibility").
//
// Recording: between ld_stage_begin() and ld_stage_end() the launch functions of this thread (ld_conv3x3, ld_gn_apply)
// validate and dispatch as usual but hand the chosen kernel variant and its argument block to ld_stage_record instead
// of launching; a call that has no tile function here (another dtype, a debug variant, the persistent C=32 kernel)
// fails the recording and the caller keeps its ordinary launches.
#include "stage.hip.h"

#include <string.h>
#include <vector>

#include "conv3x3_body.hip.h"
#include "gn_apply_body.hip.h"

namespace {

constexpr int MAXPH = 48;

struct Phase {
  int kind, variant;
  int gx, gy, gz;          // grid of the stand-alone launch: gz images (conv3x3: z, gn_apply: y), gx*gy tiles per image
  unsigned lds;
  int pad[2];
  alignas(16) unsigned char args[LD_STAGE_ARG_BYTES];
};

struct PhaseCtl { unsigned next, done, pad[14]; };       // one 64-B line per (XCD, phase)
struct StageCtl {                                       // zeroed before every launch (the caller's ld_step_begin)
  unsigned claim[8];       // image -> XCD + 1
  unsigned xcd_image[8];   // XCD -> image + 1, or 0xffffffff: none
  unsigned lead[8];        // workgroups seen per XCD
  unsigned nclaimed, pad[7];
  PhaseCtl ph[8][MAXPH];
};

__device__ __forceinline__ unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 7u;
}
__device__ __forceinline__ unsigned load_sc1(const unsigned* p) {
  unsigned v;
  asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ void store_sc1(unsigned* p, unsigned v) {
  asm volatile("global_store_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" ::"v"(p), "v"(v) : "memory");
}

template <typename T>
__device__ __forceinline__ void run_tile(const Phase& P, int t, int img, char* smem) {
  if (P.kind == LD_STAGE_CONV3) {
    const Conv3Dev a = *reinterpret_cast<const Conv3Dev*>(P.args);
    const int bx = t % P.gx, by = t / P.gx;
    switch (P.variant) {
      case 0: conv3x3_tile<T, 2, 2, false, 0, false, false>(a, bx, by, img, P.gx, P.gz, smem); break;
      default: conv3x3_tile<T, 2, 2, false, 0, false, true>(a, bx, by, img, P.gx, P.gz, smem); break;
    }
  } else {
    const GnDev g = *reinterpret_cast<const GnDev*>(P.args);
    float* s_coef = reinterpret_cast<float*>(smem);
    if (P.variant == 0) gn_apply_tile<T, false, false>(g, t, P.gx, img, s_coef);
    else gn_apply_tile<T, true, false>(g, t, P.gx, img, s_coef);
  }
}

// Per-phase cycle stamps of ONE workgroup (the first one of the XCD that took image 0), last launch: for each phase
// {phase start, first tile taken (0 if none), tiles done, arrived at the boundary (stores drained), boundary passed}.
// Always compiled in (five scalar stores per phase of one workgroup); read with ld_debug_stage_trace.
__device__ unsigned long long g_stage_trace[MAXPH * 5 + 2];

template <typename T>
__global__ __launch_bounds__(256) void stage_kernel(const Phase* __restrict__ prog, int nphase, StageCtl* ctl, int B, int xcd_base) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ int s_img;
  __shared__ unsigned s_tile;
  const int tid = threadIdx.x;
  unsigned x = 0;
  if (tid == 0) {
    x = xcc_id();
    int img = -1;
    if (atomicAdd(&ctl->lead[x], 1u) == 0) {             // first workgroup of this XCD: claim an image for it
      const unsigned pref = (x - (unsigned)xcd_base) & 7u;
      if (pref < (unsigned)B && atomicCAS(&ctl->claim[pref], 0u, x + 1) == 0u) img = (int)pref;
      for (int it = 0; img < 0; ++it) {                  // not a preferred XCD: give the preferred ones ~20 us, then help
        if (load_sc1(&ctl->nclaimed) >= (unsigned)B) break;
        if (it >= 64) {
          for (int p = 0; p < B && img < 0; ++p)
            if (atomicCAS(&ctl->claim[p], 0u, x + 1) == 0u) img = p;
          if (img < 0) break;                            // everything is claimed, the counter is about to say so
        }
        __builtin_amdgcn_s_sleep(32);
      }
      if (img >= 0) atomicAdd(&ctl->nclaimed, 1u);
      store_sc1(&ctl->xcd_image[x], img >= 0 ? (unsigned)img + 1u : 0xffffffffu);
    } else {
      unsigned v;
      while ((v = load_sc1(&ctl->xcd_image[x])) == 0u) __builtin_amdgcn_s_sleep(4);
      img = v == 0xffffffffu ? -1 : (int)v - 1;
    }
    s_img = img;
    s_tile = x;
  }
  __syncthreads();
  const int img = s_img;
  x = s_tile;
  __syncthreads();
  if (img < 0) return;

  // Tiles are claimed one ticket AHEAD: the ticket of a workgroup's next tile (or of its first tile of the next phase)
  // is requested before the current tile runs / before the phase boundary, so the counter's round trip (1-2 us with
  // 64 workgroups on it) is never waited for in the open.  A ticket taken early is only USED behind the boundary.
  unsigned ahead = 0;
  if (tid == 0 && nphase > 0) ahead = atomicAdd(&ctl->ph[x][0].next, 1u);
  const bool tracing = tid == 0 && img == 0 && ahead == 0;          // the workgroup that got ticket 0 of image 0's first phase
  if (tracing) { g_stage_trace[MAXPH * 5] = __builtin_readcyclecounter(); g_stage_trace[MAXPH * 5 + 1] = (unsigned long long)nphase; }
  for (int ph = 0; ph < nphase; ++ph) {
    const Phase& P = prog[ph];
    const unsigned ntile = (unsigned)(P.gx * P.gy);
    PhaseCtl* pc = &ctl->ph[x][ph];
    unsigned mine = 0;
    // pull the NEXT phase's descriptor into this XCD's L2 now: read cold behind the boundary it cost 5,000 cycles per phase
    if (tid < (int)(sizeof(Phase) / 64) && ph + 1 < nphase) {
      unsigned sink;
      asm volatile("global_load_dword %0, %1, off" : "=v"(sink) : "v"(reinterpret_cast<const char*>(prog + ph + 1) + tid * 64) : "memory");
    }
    if (tracing) { g_stage_trace[ph * 5] = __builtin_readcyclecounter(); g_stage_trace[ph * 5 + 1] = 0; }
    for (;;) {
      if (tid == 0) {
        s_tile = ahead;
        if (tracing && mine == 0 && ahead < ntile) g_stage_trace[ph * 5 + 1] = __builtin_readcyclecounter();
        if (ahead < ntile) ahead = atomicAdd(&pc->next, 1u);                                   // next tile of this phase
        else if (ph + 1 < nphase) ahead = atomicAdd(&ctl->ph[x][ph + 1].next, 1u);          // first tile of the next phase
      }
      __syncthreads();
      const unsigned t = s_tile;
      if (t >= ntile) break;
      run_tile<T>(P, (int)t, img, smem);
      ++mine;
      __syncthreads();                                   // LDS and s_tile are reused by the next tile
    }
    // ---- XCD-local phase boundary
    if (tracing) g_stage_trace[ph * 5 + 2] = __builtin_readcyclecounter();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's stores and statistics atomics have reached L2 / memory
    __syncthreads();
    if (tracing) g_stage_trace[ph * 5 + 3] = __builtin_readcyclecounter();
    if (tid == 0) {
      if (mine) atomicAdd(&pc->done, mine);
      while (load_sc1(&pc->done) < ntile) {}
      asm volatile("buffer_inv sc1\n\ts_waitcnt vmcnt(0)" ::: "memory");   // this CU's L1 holds nothing of the previous phases
    }
    __syncthreads();
    if (tracing) g_stage_trace[ph * 5 + 4] = __builtin_readcyclecounter();
  }
}

struct Rec {
  std::vector<Phase> ph;
  bool failed = false;
  char why[160] = "";
};
thread_local Rec* g_rec = nullptr;

struct Program {
  Phase* dev = nullptr;
  int nphase = 0;
  size_t lds = 0;
  int images = 0;
};

}  // namespace

bool ld_stage_recording() { return g_rec != nullptr; }

int ld_stage_record(int kind, int variant, const void* args, size_t bytes, int gx, int gy, int gz, size_t lds) {
  if (!g_rec) return ld_fail(LD_EINVAL, "ld_stage_record: no recording");
  if (bytes > LD_STAGE_ARG_BYTES || g_rec->ph.size() >= (size_t)MAXPH) {
    g_rec->failed = true;
    snprintf(g_rec->why, sizeof(g_rec->why), "argument block of %zu bytes / more than %d phases", bytes, MAXPH);
    return LD_OK;
  }
  Phase p;
  memset(&p, 0, sizeof(p));
  p.kind = kind; p.variant = variant; p.gx = gx; p.gy = gy; p.gz = gz; p.lds = (unsigned)lds;
  memcpy(p.args, args, bytes);
  g_rec->ph.push_back(p);
  return LD_OK;
}

int ld_stage_unsupported(const char* what) {
  if (g_rec) {
    g_rec->failed = true;
    snprintf(g_rec->why, sizeof(g_rec->why), "%s has no tile function in stage.hip", what);
  }
  return LD_OK;
}

extern "C" int ld_stage_begin(void) {
  LD_REQUIRE(!g_rec, "ld_stage_begin: a recording is already open on this thread");
  g_rec = new Rec();
  return LD_OK;
}

extern "C" size_t ld_stage_ctl_bytes(void) { return sizeof(StageCtl); }

extern "C" int ld_stage_end(void** program_out, int* nphase_out) {
  LD_REQUIRE(g_rec, "ld_stage_end: no recording");
  Rec* r = g_rec;
  g_rec = nullptr;
  if (program_out) *program_out = nullptr;
  if (nphase_out) *nphase_out = 0;
  if (r->failed || r->ph.empty() || !program_out) {
    const int rc = r->failed ? ld_fail(LD_EINVAL, "ld_stage_end: %s", r->why) : LD_OK;
    delete r;
    return rc;
  }
  Program* pg = new Program();
  pg->nphase = (int)r->ph.size();
  pg->images = r->ph[0].gz;
  for (const Phase& p : r->ph) {
    if (p.lds > pg->lds) pg->lds = p.lds;
    if (p.gz != pg->images) {
      delete pg; delete r;
      return ld_fail(LD_EINVAL, "ld_stage_end: phases with different batch sizes");
    }
  }
  hipError_t e = hipMalloc(&pg->dev, sizeof(Phase) * r->ph.size());
  if (e == hipSuccess) e = hipMemcpy(pg->dev, r->ph.data(), sizeof(Phase) * r->ph.size(), hipMemcpyHostToDevice);
  delete r;
  if (e != hipSuccess) { delete pg; return ld_fail(LD_EHIP, "ld_stage_end: %s", hipGetErrorString(e)); }
  *program_out = pg;
  if (nphase_out) *nphase_out = pg->nphase;
  return LD_OK;
}

extern "C" int ld_stage_launch(void* program, void* ctl_zeroed, int xcd_base, int workgroups, int dtype, void* stream) {
  Program* pg = reinterpret_cast<Program*>(program);
  LD_REQUIRE(pg && pg->dev && ctl_zeroed, "ld_stage_launch: null program / control block");
  LD_REQUIRE(ld_dtype_16(dtype), "ld_stage_launch: 16-bit storage only");
  LD_REQUIRE(pg->images >= 1 && pg->images <= 8, "ld_stage_launch: one image per XCD: 1..8 images (got %d)", pg->images);
  LD_REQUIRE(workgroups >= 8 && workgroups <= 4096, "ld_stage_launch: workgroups %d", workgroups);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const size_t lds = pg->lds + 64;
  return LD_DISPATCH16(dtype, [&] {
    if (lds > 65536) LD_HIP(ld_allow_lds(stage_kernel<T>, lds));
    LD_LAUNCH(stage_kernel<T>, dim3(workgroups), dim3(256), lds, st, pg->dev, pg->nphase, reinterpret_cast<StageCtl*>(ctl_zeroed),
              pg->images, xcd_base);
    LD_LAUNCH_CHECK("stage");
    return (int)LD_OK;
  }());
}

extern "C" int ld_stage_destroy(void* program) {
  Program* pg = reinterpret_cast<Program*>(program);
  if (pg) {
    if (pg->dev) (void)hipFree(pg->dev);
    delete pg;
  }
  return LD_OK;
}

// Debug hook (not part of the public ABI): cycle stamps of the last stage launch, MAXPH * 5 + 2 uint64 (see g_stage_trace).
extern "C" int ld_debug_stage_trace(unsigned long long* host, int n) {
  LD_HIP(hipDeviceSynchronize());
  LD_HIP(hipMemcpyFromSymbol(host, HIP_SYMBOL(g_stage_trace), sizeof(unsigned long long) * (n < MAXPH * 5 + 2 ? n : MAXPH * 5 + 2)));
  return LD_OK;
}
