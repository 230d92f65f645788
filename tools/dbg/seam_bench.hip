// What does a grid-wide seam cost on this chip — as a KERNEL BOUNDARY inside a replayed hipGraph, or as an XCD-hierarchical GRID
// BARRIER inside one persistent launch?  (the small-layer section of the training step as one persistent launch.)
//
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/seam_bench tools/dbg/seam_bench.hip && /tmp/seam_bench
//
// Both forms run PHASES phases; in each, every workgroup writes a record of `rec` floats (the per-workgroup BatchNorm partials a
// layer leaves) and, after the seam, reads back `nread` records of OTHER workgroups (what a consumer-side finalize does).
//   (A) launches: PHASES kernels captured into a hipGraph (linear chain), replayed;
//   (B) persistent: ONE kernel, the phases separated by a grid barrier: per-XCD arrival counter -> the XCD's last arriver bumps
//       a top counter and waits for all 8 -> releases its XCD's generation word; release fence before arriving, acquire fence
//       after (the workgroups of one XCD = linear ids congruent mod 8).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ float phase_work(float *records, int rec, int nread, int phase, int nwg) {
  // write my record, return a value that depends on it (the read-back happens after the seam)
  float *mine = records + (size_t)blockIdx.x * rec;
  for (int i = threadIdx.x; i < rec; i += blockDim.x) mine[i] = (float)(phase + i);
  return 0.f;
}
__device__ __forceinline__ float read_back(const float *records, int rec, int nread, int nwg) {
  float s = 0.f;
  for (int r = 0; r < nread; ++r) {
    const float *o = records + (size_t)((blockIdx.x + 1 + r * 37) % nwg) * rec;
    for (int i = threadIdx.x; i < rec; i += blockDim.x) s += o[i];
  }
  return s;
}

__global__ void phase_kernel(float *records, int rec, int nread, int phase, int nwg, float *sink) {
  float s = read_back(records, rec, nread, nwg);          // consume the previous phase's records (made visible by the boundary)
  __syncthreads();
  s += phase_work(records, rec, nread, phase, nwg);
  if (s == -1.f) sink[0] = s;
}

__device__ __forceinline__ void grid_barrier(unsigned *xcc_cnt, unsigned *top, unsigned *xcc_gen, unsigned phase, unsigned per_xcc) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");                         // my records become visible to the other XCDs
    const int x = blockIdx.x & 7;
    const unsigned a = __hip_atomic_fetch_add(&xcc_cnt[x], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
    if (a == per_xcc * phase) {
      __hip_atomic_fetch_add(top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      for (unsigned spin = 0; __hip_atomic_load(top, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 8u * phase && spin < (1u << 22); ++spin)
        __builtin_amdgcn_s_sleep(1);                                            // (bounded: a stranded workgroup must not hang the box)
      __hip_atomic_store(&xcc_gen[x], phase, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      for (unsigned spin = 0; __hip_atomic_load(&xcc_gen[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < phase && spin < (1u << 22); ++spin)
        __builtin_amdgcn_s_sleep(1);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
}

__global__ void persistent_kernel(float *records, int rec, int nread, int phases, int nwg, unsigned *sync, float *sink) {
  unsigned *xcc_cnt = sync, *top = sync + 8, *xcc_gen = sync + 16;
  float s = 0.f;
  for (int p = 1; p <= phases; ++p) {
    s += phase_work(records, rec, nread, p, nwg);
    grid_barrier(xcc_cnt, top, xcc_gen, (unsigned)p, (unsigned)(nwg / 8));
    s += read_back(records, rec, nread, nwg);
    __syncthreads();
  }
  if (s == -1.f) sink[0] = s;
}

int main() {
  const int PHASES = 64, REPS = 200;
  hipStream_t st;
  CK(hipStreamCreate(&st));
  float *records, *sink;
  unsigned *sync;
  CK(hipMalloc(&records, 1024 * 4096 * sizeof(float)));
  CK(hipMalloc(&sink, 16));
  CK(hipMalloc(&sync, 32 * sizeof(unsigned)));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  printf("%6s %6s %6s | %12s %12s\n", "WGs", "rec", "nread", "launches us", "barrier us");
  for (int nwg : {256, 512}) {
    for (int cfg = 0; cfg < 3; ++cfg) {
      const int rec = cfg == 0 ? 32 : 512, nread = cfg == 2 ? 64 : (cfg == 1 ? 1 : 0);     // bytes written / read back per workgroup
      // (A) a linear graph of PHASES kernels
      hipGraph_t g;
      hipGraphExec_t ge;
      CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
      for (int p = 1; p <= PHASES; ++p) phase_kernel<<<nwg, 256, 0, st>>>(records, rec, nread, p, nwg, sink);
      CK(hipStreamEndCapture(st, &g));
      CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      for (int i = 0; i < 5; ++i) CK(hipGraphLaunch(ge, st));
      CK(hipStreamSynchronize(st));
      CK(hipEventRecord(e0, st));
      for (int i = 0; i < REPS; ++i) CK(hipGraphLaunch(ge, st));
      CK(hipEventRecord(e1, st));
      CK(hipStreamSynchronize(st));
      float msA = 0.f;
      CK(hipEventElapsedTime(&msA, e0, e1));
      // (B) one persistent launch
      float msB = 0.f;
      for (int i = 0; i < 5 + REPS; ++i) {
        if (i == 5) { CK(hipStreamSynchronize(st)); CK(hipEventRecord(e0, st)); }
        CK(hipMemsetAsync(sync, 0, 32 * sizeof(unsigned), st));
        persistent_kernel<<<nwg, 256, 0, st>>>(records, rec, nread, PHASES, nwg, sync, sink);
      }
      CK(hipEventRecord(e1, st));
      CK(hipStreamSynchronize(st));
      CK(hipEventElapsedTime(&msB, e0, e1));
      printf("%6d %6d %6d | %12.2f %12.2f   (per phase)\n", nwg, rec * 4, nread, 1e3 * msA / REPS / PHASES, 1e3 * msB / REPS / PHASES);
      CK(hipGraphExecDestroy(ge));
      CK(hipGraphDestroy(g));
    }
  }
  return 0;
}
