// What does a BatchNorm seam cost when the per-channel sums leave the producing kernel as NO-RETURN 64-bit fixed-point atomics
//  instead of per-workgroup partial rows + a [C]-sized finalize launch?
//
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/atomic_seam tools/dbg/atomic_seam_bench.hip && /tmp/atomic_seam
//
// A "layer" is a streaming pass over X [P, 128] bf16 -> Y [P, 128] bf16 (what a 128-channel GEMM launch moves) that takes the
// per-channel sums of y and y^2 of its rows and applies the PREVIOUS layer's scale / shift on the operand load.  64 layers are
// captured as one linear hipGraph and replayed; reported: microseconds per layer.
//   A  today:  producer writes partial[workgroup][2][C] fp32  ->  finalize launch (C/16 workgroups x 1024 lanes: 16 channels x 64
//              row subsets, fixed order) writes scale / shift  ->  next producer reads scale / shift [2][C]
//   B  atomics: producer adds llrint(partial * 2^S) to acc[replica][2][C] (int64, no return; replica = workgroup % R)  ->  next
//              producer's prologue sums the R replicas and forms scale / shift itself.  No finalize launch.  Integer addition is
//              associative: the totals are bit-reproducible whatever the order.  Three accumulator sets rotate; the set two layers
//              ahead is zeroed by workgroup 0 (in the network: by a launch that exists anyway).
//   C  floor:  A without the finalize launch (the next producer reads constant scale / shift): the layer's own cost.
//   D  resident finalizer: A's partial rows, but the finalize work is done by ONE kernel (C/16 workgroups) that stays resident on a
//              second stream for the whole chain: every producer workgroup counts itself done (release) after its partial row, the
//              finalizer waits for the count, sums in A's fixed order, publishes scale / shift and a ready count; the next producer —
//              the next launch on the main stream, no finalize launch between them — waits for the ready count INSIDE the kernel.
//              All spins are bounded by the wall clock (20 ms) and report through an error word.
//              D1 orders payload and counts with agent-scope fences (buffer_wbl2 / buffer_inv sc1 per workgroup): 28-56 us per seam.
//              D2 uses no fence: every payload word is 8 bytes = value | tag (single-copy atomic, like the multi-workgroup sampling
//              kernel's keys), written and read with agent-scope relaxed atomics; the counts are only hints for when to look.
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <cstring>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int C = 128, NT = 256, RL = NT / 16;         // 16 lanes per row (8 channels = 16 B each), 16 rows per trip
constexpr float FX_SCALE = 16777216.0f;                 // 2^24

typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float bf2f(unsigned short u) { return __uint_as_float((unsigned)u << 16); }
__device__ __forceinline__ unsigned short f2bf(float f) {
  unsigned u = __float_as_uint(f);
  return (unsigned short)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}

// MODE 0: partial rows (A / C), 1: atomics (B)
template <int MODE>
__global__ __launch_bounds__(NT) void layer_kernel(const unsigned short *__restrict__ X, unsigned short *__restrict__ Y, int P,
                                                   const float *__restrict__ ss_in /* [2][C] (MODE 0) */, float *__restrict__ partial,
                                                   const long long *__restrict__ acc_in, long long *__restrict__ acc_out,
                                                   long long *__restrict__ acc_zero, int R,
                                                   unsigned *__restrict__ done = nullptr, const unsigned *__restrict__ ready = nullptr,
                                                   unsigned ready_target = 0, unsigned *__restrict__ err = nullptr,
                                                   const unsigned *__restrict__ epoch = nullptr, int layer = 0,
                                                   unsigned long long *__restrict__ tpartial = nullptr,
                                                   const unsigned long long *__restrict__ tss_in = nullptr) {
  __shared__ float s_scale[C], s_shift[C];
  __shared__ float s_red[RL][2 * C];
  __shared__ double s_tot[2 * C];
  const int t = threadIdx.x, wg = blockIdx.x, nwg = gridDim.x;
  if (MODE == 2 && ready) {
    if (t == 0) {
      const unsigned long long t0 = wall_clock64();
      while (__hip_atomic_load(ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < ready_target) {
        __builtin_amdgcn_s_sleep(1);
        if (wall_clock64() - t0 > 2000000ull) { *err = 1u; break; }
      }
    }
    __syncthreads();
    __atomic_thread_fence(__ATOMIC_ACQUIRE);        // (agent scope: buffer_inv sc1)
  }
  unsigned tag = 0;
  if (MODE == 3) {
    tag = *epoch * 64u + (unsigned)layer + 1u;
    if (ready) {
      if (t == 0) {
        const unsigned long long t0 = wall_clock64();
        while (__hip_atomic_load(ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < ready_target) {
          __builtin_amdgcn_s_sleep(1);
          if (wall_clock64() - t0 > 2000000ull) { *err = 1u; break; }
        }
      }
      __syncthreads();
      // t < 256: scale[t] / shift[t - C] of the copy for this workgroup's XCD; valid when the tag is the PREVIOUS layer's
      const unsigned long long *src = tss_in + (size_t)(wg & 7) * 2 * C + t;
      unsigned long long w = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned long long t0 = wall_clock64();
      while ((unsigned)(w >> 32) != tag - 1u) {
        if (wall_clock64() - t0 > 2000000ull) { *err = 3u; break; }
        __builtin_amdgcn_s_sleep(1);
        w = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (t < C) s_scale[t] = __uint_as_float((unsigned)w); else s_shift[t - C] = __uint_as_float((unsigned)w);
    } else {
      if (t < C) { s_scale[t] = ss_in[t]; s_shift[t] = ss_in[C + t]; }
    }
  } else if (MODE == 0 || MODE == 2) {
    if (t < C) { s_scale[t] = ss_in[t]; s_shift[t] = ss_in[C + t]; }
  } else {
    long long s = 0;
    for (int r = 0; r < R; ++r) s += acc_in[(size_t)r * 2 * C + t];
    s_tot[t] = (double)s / (double)FX_SCALE;
    if (wg == 0)
      for (int r = 0; r < R; ++r) acc_zero[(size_t)r * 2 * C + t] = 0;
    __syncthreads();
    if (t < C) {
      const double mean = s_tot[t] / P, var = s_tot[C + t] / P - mean * mean;
      const float rstd = rsqrtf((float)var + 1e-5f);
      s_scale[t] = rstd; s_shift[t] = (float)(-mean) * rstd;
    }
  }
  __syncthreads();
  const int rows_per = (P + nwg - 1) / nwg, r0 = wg * rows_per, r1 = min(P, r0 + rows_per);
  const int rl = t >> 4, ch = (t & 15) * 8;
  float sc[8], sh[8], s1[8], s2[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { sc[i] = s_scale[ch + i]; sh[i] = s_shift[ch + i]; s1[i] = 0.f; s2[i] = 0.f; }
  for (int r = r0 + rl; r < r1; r += 4 * RL) {
    u16x8 v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (r + q * RL < r1) v[q] = *(const u16x8 *)(X + (size_t)(r + q * RL) * C + ch);
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (r + q * RL < r1) {
        u16x8 o;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float y = fmaxf(bf2f(v[q][i]) * sc[i] + sh[i], -4.f) * 0.999f + 0.01f;      // (bounded: 64 layers must not blow up)
          s1[i] += y; s2[i] += y * y;
          o[i] = f2bf(y);
        }
        *(u16x8 *)(Y + (size_t)(r + q * RL) * C + ch) = o;
      }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) { s_red[rl][ch + i] = s1[i]; s_red[rl][C + ch + i] = s2[i]; }
  __syncthreads();
  float v = 0.f;
#pragma unroll
  for (int q = 0; q < RL; ++q) v += s_red[q][t];
  if (MODE == 0) {
    partial[(size_t)wg * 2 * C + t] = v;
  } else if (MODE == 3) {
    __hip_atomic_store(&tpartial[(size_t)wg * 2 * C + t], (unsigned long long)__float_as_uint(v) | ((unsigned long long)tag << 32), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t == 0) (void)__hip_atomic_fetch_add(done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else if (MODE == 2) {
    partial[(size_t)wg * 2 * C + t] = v;
    __atomic_thread_fence(__ATOMIC_RELEASE);        // (buffer_wbl2 sc1 + waitcnt)
    __syncthreads();
    if (t == 0) (void)__hip_atomic_fetch_add(done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else {
    const long long q = (long long)llrintf(v * FX_SCALE);
    (void)__hip_atomic_fetch_add((unsigned long long *)&acc_out[(size_t)(wg % R) * 2 * C + t], (unsigned long long)q, __ATOMIC_RELAXED,
                                 __HIP_MEMORY_SCOPE_AGENT);
  }
}

// 16 channels x 64 row subsets per workgroup, like cpfn_bn_finalize
__global__ __launch_bounds__(1024) void finalize_kernel(const float *__restrict__ partial, int nblk, int P, float *__restrict__ ss) {
  __shared__ float s_a[64][16], s_b[64][16];
  const int t = threadIdx.x, c = blockIdx.x * 16 + (t & 15), sub = t >> 4;
  float a = 0.f, b = 0.f;
  for (int r = sub; r < nblk; r += 64) { a += partial[(size_t)r * 2 * C + c]; b += partial[(size_t)r * 2 * C + C + c]; }
  s_a[sub][t & 15] = a; s_b[sub][t & 15] = b;
  __syncthreads();
  if (t < 16) {
    double A = 0, B = 0;
    for (int q = 0; q < 64; ++q) { A += s_a[q][t]; B += s_b[q][t]; }
    const double mean = A / P, var = B / P - mean * mean;
    const float rstd = rsqrtf((float)var + 1e-5f);
    ss[c] = rstd; ss[C + c] = (float)(-mean) * rstd;
  }
}

// D: the resident finalizer.  partial2 / ss2: two buffers each, layer l uses [l & 1].
__global__ __launch_bounds__(1024) void resident_finalizer_kernel(const float *__restrict__ partial2, int nblk, int P, float *__restrict__ ss2,
                                                                  const unsigned *__restrict__ done, unsigned *__restrict__ ready,
                                                                  int layers, unsigned *__restrict__ err) {
  __shared__ float s_a[64][16], s_b[64][16];
  const int t = threadIdx.x, c = blockIdx.x * 16 + (t & 15), sub = t >> 4;
  for (int l = 0; l < layers; ++l) {
    if (t == 0) {
      const unsigned long long t0 = wall_clock64();
      while (__hip_atomic_load(done + l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)nblk) {
        __builtin_amdgcn_s_sleep(1);
        if (wall_clock64() - t0 > 2000000ull) { *err = 2u; break; }
      }
    }
    __syncthreads();
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    const float *partial = partial2 + (size_t)(l & 1) * 1024 * 2 * C;
    float *ss = ss2 + (size_t)(l & 1) * 2 * C;
    float a = 0.f, b = 0.f;
    for (int r = sub; r < nblk; r += 64) { a += partial[(size_t)r * 2 * C + c]; b += partial[(size_t)r * 2 * C + C + c]; }
    s_a[sub][t & 15] = a; s_b[sub][t & 15] = b;
    __syncthreads();
    if (t < 16) {
      double A = 0, B = 0;
      for (int q = 0; q < 64; ++q) { A += s_a[q][t]; B += s_b[q][t]; }
      const double mean = A / P, var = B / P - mean * mean;
      const float rstd = rsqrtf((float)var + 1e-5f);
      ss[c] = rstd; ss[C + c] = (float)(-mean) * rstd;
      __atomic_thread_fence(__ATOMIC_RELEASE);
    }
    __syncthreads();
    if (t == 0) (void)__hip_atomic_fetch_add(ready + l, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

__global__ void epoch_inc_kernel(unsigned *epoch) { ++*epoch; }

// D2: tagged words, no fences.  tpartial2: two buffers [1024][2C] u64, tss2: two buffers [8 copies][2C] u64.
__global__ __launch_bounds__(1024) void resident_finalizer_tagged_kernel(const unsigned long long *__restrict__ tpartial2, int nblk, int P,
                                                                         unsigned long long *__restrict__ tss2, const unsigned *__restrict__ done,
                                                                         unsigned *__restrict__ ready, int layers, unsigned *__restrict__ err,
                                                                         const unsigned *__restrict__ epoch) {
  __shared__ float s_a[64][16], s_b[64][16];
  const int t = threadIdx.x, c = blockIdx.x * 16 + (t & 15), sub = t >> 4;
  const unsigned tag0 = *epoch * 64u;
  for (int l = 0; l < layers; ++l) {
    const unsigned tag = tag0 + (unsigned)l + 1u;
    if (t == 0) {
      const unsigned long long t0 = wall_clock64();
      while (__hip_atomic_load(done + l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)nblk) {
        __builtin_amdgcn_s_sleep(1);
        if (wall_clock64() - t0 > 2000000ull) { *err = 2u; break; }
      }
    }
    __syncthreads();
    const unsigned long long *partial = tpartial2 + (size_t)(l & 1) * 1024 * 2 * C;
    unsigned long long *ss = tss2 + (size_t)(l & 1) * 8 * 2 * C;
    float a = 0.f, b = 0.f;
    for (int r = sub; r < nblk; r += 64) {
      unsigned long long wa = __hip_atomic_load(&partial[(size_t)r * 2 * C + c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      unsigned long long wb = __hip_atomic_load(&partial[(size_t)r * 2 * C + C + c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned long long t0 = wall_clock64();
      while ((unsigned)(wa >> 32) != tag || (unsigned)(wb >> 32) != tag) {
        if (wall_clock64() - t0 > 2000000ull) { *err = 4u; break; }
        __builtin_amdgcn_s_sleep(1);
        wa = __hip_atomic_load(&partial[(size_t)r * 2 * C + c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        wb = __hip_atomic_load(&partial[(size_t)r * 2 * C + C + c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      a += __uint_as_float((unsigned)wa); b += __uint_as_float((unsigned)wb);
    }
    s_a[sub][t & 15] = a; s_b[sub][t & 15] = b;
    __syncthreads();
    if (t < 16) {
      double A = 0, B = 0;
      for (int q = 0; q < 64; ++q) { A += s_a[q][t]; B += s_b[q][t]; }
      const double mean = A / P, var = B / P - mean * mean;
      const float rstd = rsqrtf((float)var + 1e-5f);
      s_a[0][t] = rstd; s_b[0][t] = (float)(-mean) * rstd;
    }
    __syncthreads();
    if (t < 256) {        // 8 copies x (16 scales + 16 shifts)
      const int copy = t >> 5, which = (t >> 4) & 1, j = t & 15;
      const float v = which ? s_b[0][j] : s_a[0][j];
      __hip_atomic_store(&ss[(size_t)copy * 2 * C + which * C + blockIdx.x * 16 + j], (unsigned long long)__float_as_uint(v) | ((unsigned long long)tag << 32),
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (t == 0) (void)__hip_atomic_fetch_add(ready + l, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

static float replay_us(hipGraphExec_t ge, hipStream_t st, int layers) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 5; ++i) CK(hipGraphLaunch(ge, st));
  std::vector<float> ms;
  for (int rep = 0; rep < 7; ++rep) {
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < 10; ++i) CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float m; CK(hipEventElapsedTime(&m, e0, e1));
    ms.push_back(m);
  }
  std::sort(ms.begin(), ms.end());
  return ms[ms.size() / 2] * 1000.f / 10.f / layers;
}

int main() {
  const int LAYERS = 64;
  hipStream_t st;
  CK(hipStreamCreate(&st));
  const int Pmax = 524288;
  unsigned short *X, *Y;
  float *partial, *ss;
  long long *acc;
  CK(hipMalloc(&X, (size_t)Pmax * C * 2)); CK(hipMalloc(&Y, (size_t)Pmax * C * 2));
  CK(hipMalloc(&partial, 1024 * 2 * C * 4)); CK(hipMalloc(&ss, 2 * C * 4));
  CK(hipMalloc(&acc, 3 * 64 * 2 * C * 8));
  {
    std::vector<unsigned short> h((size_t)Pmax * C);
    unsigned s = 12345;
    for (auto &v : h) { s = s * 1664525u + 1013904223u; float f = ((s >> 8) & 0xFFFF) / 32768.f - 1.f; unsigned u; std::memcpy(&u, &f, 4); v = (unsigned short)(u >> 16); }
    CK(hipMemcpy(X, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    std::vector<float> one(2 * C, 0.f);
    for (int i = 0; i < C; ++i) one[i] = 1.f;
    CK(hipMemcpy(ss, one.data(), 2 * C * 4, hipMemcpyHostToDevice));
  }
  printf("%-10s %-6s %-34s %s\n", "rows", "wgs", "variant", "us per layer");
  for (int P : {131072, 524288, 8192}) {
    for (int nwg : {448, 256}) {
      if (P == 8192 && nwg == 448) continue;
      auto capture = [&](int variant, int R) {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipMemsetAsync(acc, 0, 3 * 64 * 2 * C * 8, st));
        CK(hipStreamSynchronize(st));
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int l = 0; l < LAYERS; ++l) {
          const unsigned short *in = (l & 1) ? Y : X;
          unsigned short *out = (l & 1) ? X : Y;
          if (variant == 1) {
            long long *a_in = acc + (size_t)((l + 2) % 3) * 64 * 2 * C, *a_out = acc + (size_t)(l % 3) * 64 * 2 * C,
                      *a_zero = acc + (size_t)((l + 1) % 3) * 64 * 2 * C;
            layer_kernel<1><<<nwg, NT, 0, st>>>(in, out, P, nullptr, nullptr, a_in, a_out, a_zero, R);
          } else {
            layer_kernel<0><<<nwg, NT, 0, st>>>(in, out, P, ss, partial, nullptr, nullptr, nullptr, 1);
            if (variant == 0) finalize_kernel<<<C / 16, 1024, 0, st>>>(partial, nwg, P, ss);
          }
        }
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        const float us = replay_us(ge, st, LAYERS);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
        return us;
      };
      const float a = capture(0, 1), c = capture(2, 1);
      printf("%-10d %-6d %-34s %.2f\n", P, nwg, "A partial rows + finalize launch", a);
      printf("%-10d %-6d %-34s %.2f   (seam A costs %.2f)\n", P, nwg, "C no finalize (floor)", c, a - c);
      for (int dmode = 1; dmode <= 2; ++dmode) {   // D: main-stream graph of the 64 layer launches + the resident finalizer on a second stream, per replay
        static unsigned *cnt = nullptr;     // done[64] | ready[64] | err | epoch
        static float *partial2 = nullptr, *ss2 = nullptr;
        static unsigned long long *tpartial2 = nullptr, *tss2 = nullptr;
        static hipStream_t st2;
        static hipEvent_t ev_a, ev_b;
        if (!cnt) {
          CK(hipMalloc(&cnt, 130 * 4)); CK(hipMalloc(&partial2, 2 * 1024 * 2 * C * 4)); CK(hipMalloc(&ss2, 2 * 2 * C * 4));
          CK(hipMalloc(&tpartial2, 2 * 1024 * 2 * C * 8)); CK(hipMalloc(&tss2, 2 * 8 * 2 * C * 8));
          CK(hipMemset(cnt, 0, 130 * 4)); CK(hipMemset(tpartial2, 0, 2 * 1024 * 2 * C * 8)); CK(hipMemset(tss2, 0, 2 * 8 * 2 * C * 8));
          CK(hipStreamCreate(&st2)); CK(hipEventCreateWithFlags(&ev_a, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ev_b, hipEventDisableTiming));
        }
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int l = 0; l < LAYERS; ++l) {
          const unsigned short *in = (l & 1) ? Y : X;
          unsigned short *out = (l & 1) ? X : Y;
          if (dmode == 1)
            layer_kernel<2><<<nwg, NT, 0, st>>>(in, out, P, l ? ss2 + (size_t)((l - 1) & 1) * 2 * C : ss, partial2 + (size_t)(l & 1) * 1024 * 2 * C, nullptr,
                                                nullptr, nullptr, 1, cnt + l, l ? cnt + 64 + (l - 1) : nullptr, C / 16, cnt + 128);
          else
            layer_kernel<3><<<nwg, NT, 0, st>>>(in, out, P, ss, nullptr, nullptr, nullptr, nullptr, 1, cnt + l, l ? cnt + 64 + (l - 1) : nullptr, C / 16,
                                                cnt + 128, cnt + 129, l, tpartial2 + (size_t)(l & 1) * 1024 * 2 * C,
                                                l ? tss2 + (size_t)((l - 1) & 1) * 8 * 2 * C : nullptr);
        }
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        auto one = [&]() {
          CK(hipMemsetAsync(cnt, 0, 128 * 4, st));
          epoch_inc_kernel<<<1, 1, 0, st>>>(cnt + 129);
          CK(hipEventRecord(ev_a, st));
          CK(hipStreamWaitEvent(st2, ev_a, 0));
          if (dmode == 1) resident_finalizer_kernel<<<C / 16, 1024, 0, st2>>>(partial2, nwg, P, ss2, cnt, cnt + 64, LAYERS, cnt + 128);
          else resident_finalizer_tagged_kernel<<<C / 16, 1024, 0, st2>>>(tpartial2, nwg, P, tss2, cnt, cnt + 64, LAYERS, cnt + 128, cnt + 129);
          CK(hipGraphLaunch(ge, st));
          CK(hipEventRecord(ev_b, st2));
          CK(hipStreamWaitEvent(st, ev_b, 0));
        };
        CK(hipMemsetAsync(cnt, 0, 129 * 4, st));
        for (int i = 0; i < 5; ++i) one();
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        std::vector<float> ms;
        for (int rep = 0; rep < 7; ++rep) {
          CK(hipEventRecord(e0, st));
          for (int i = 0; i < 10; ++i) one();
          CK(hipEventRecord(e1, st));
          CK(hipEventSynchronize(e1));
          float m; CK(hipEventElapsedTime(&m, e0, e1));
          ms.push_back(m);
        }
        std::sort(ms.begin(), ms.end());
        const float d = ms[ms.size() / 2] * 1000.f / 10.f / LAYERS;
        unsigned herr = 0;
        CK(hipMemcpy(&herr, cnt + 128, 4, hipMemcpyDeviceToHost));
        double cs = 0;
        if (dmode == 1) {
          std::vector<float> ssd(2 * C);
          CK(hipMemcpy(ssd.data(), ss2 + (size_t)((LAYERS - 1) & 1) * 2 * C, 2 * C * 4, hipMemcpyDeviceToHost));
          for (float v : ssd) cs += v;
        } else {
          std::vector<unsigned long long> ssd(2 * C);
          CK(hipMemcpy(ssd.data(), tss2 + (size_t)((LAYERS - 1) & 1) * 8 * 2 * C, 2 * C * 8, hipMemcpyDeviceToHost));
          for (unsigned long long w : ssd) { unsigned u = (unsigned)w; float f; std::memcpy(&f, &u, 4); cs += f; }
        }
        printf("%-10d %-6d D%d resident finalizer, %-14s %.2f   (seam D%d costs %.2f; D - A = %+.2f; err word %u; checksum %.6g)\n", P, nwg, dmode,
               dmode == 1 ? "fences" : "tagged words", d, dmode, d - c, d - a, herr, cs);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
      }
      for (int R : {1, 8, 32}) {
        const float b = capture(1, R);
        printf("%-10d %-6d B atomics, %2d replica(s)%11s %.2f   (seam B costs %.2f; B - A = %+.2f)\n", P, nwg, R, "", b, b - c, b - a);
      }
    }
  }
  // reproducibility of the fixed-point totals: two runs of one layer, bitwise
  {
    long long h0[2 * C], h1[2 * C];
    for (int run = 0; run < 2; ++run) {
      CK(hipMemsetAsync(acc, 0, 3 * 64 * 2 * C * 8, st));
      layer_kernel<1><<<448, NT, 0, st>>>(X, Y, 131072, nullptr, nullptr, acc + 2 * 64 * 2 * C, acc, acc + 64 * 2 * C, 1);
      CK(hipMemcpyAsync(run ? h1 : h0, acc, sizeof(h0), hipMemcpyDeviceToHost, st));
      CK(hipStreamSynchronize(st));
    }
    int same = 1;
    for (int i = 0; i < 2 * C; ++i) same &= h0[i] == h1[i];
    printf("fixed-point totals of two runs bit-identical: %s\n", same ? "yes" : "NO");
  }
  return 0;
}
