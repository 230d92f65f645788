// Second micro-benchmark of the packed-fp32 interaction: WHICH ingredient of the sampling loop makes it a victim?
// The aggressor is the real thing — cpfn_mlp_wgrad (64 x 64: the shape that fits beside every victim here) from the built library,
// looping on a second stream; the victim is a stripped sampling loop that grows ingredient by ingredient (template flags):
//   base    min-distances of 8 points per lane in registers, updated against a moving sample with the PACKED distance arithmetic
//           (v_pk_add_f32 with negation and op_sel broadcast, v_pk_mul_f32) and v_min_f32; the sample is a point of the cloud chosen by
//           a fixed schedule (no arg-max), so after the update its owner's min-distance MUST be 0: violations are counted, by lane
//   +1      the wave maximum by six DPP-modified v_max_f32 (row-restricted writes: rows 1,3 and 2,3) and a v_readlane, every sample
//   +2      a 64-bit key per wave through LDS and one workgroup barrier per sample
//   +4      the sample's coordinates read back from an LDS mirror (three broadcast ds_read_b32)
//   +8      (control) the same with one float per instruction instead of packed arithmetic
// Build on the GPU box from the repository root (the library must be built):
//     hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/dbg/pk_micro2.hip -Iinclude -Lcpfn_amd -lcpfn_hip -Wl,-rpath,$PWD/cpfn_amd -o /tmp/pk_micro2
//     /tmp/pk_micro2 [seconds per case]
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "cpfn_hip.h"

typedef float f32x2 __attribute__((ext_vector_type(2)));

#define CHECK(x)                                                                          \
  do {                                                                                    \
    hipError_t e_ = (x);                                                                  \
    if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); }     \
  } while (0)

__device__ __forceinline__ float v_min(float a, float b) {
  float r;
  asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float wave_max_f32(float v) {
  asm("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf"
      : "+v"(v));
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// point k of cloud c (what every lane can recompute: the sample's coordinates need no memory unless flag 4 asks for the mirror)
__device__ __forceinline__ void point(int c, int k, float &x, float &y, float &z) {
  const unsigned h = (unsigned)(k * 2654435761u) ^ (unsigned)(c * 40503u);
  x = (float)(h & 1023) * (1.f / 1024.f);
  y = (float)((h >> 10) & 1023) * (1.f / 1024.f);
  z = (float)((h >> 20) & 1023) * (1.f / 1024.f);
}

constexpr int NT = 256, PPT = 8, NPTS = NT * PPT;

template <int F>
__global__ __launch_bounds__(NT) void victim(int samples, unsigned *lane_hist /*[64]*/, unsigned *total, float *sink) {
  __shared__ float s_x[(F & 4) ? NPTS : 1], s_y[(F & 4) ? NPTS : 1], s_z[(F & 4) ? NPTS : 1];
  __shared__ unsigned long long s_key[2][NT / 64];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, c = blockIdx.x;
  f32x2 px[PPT / 2], py[PPT / 2], pz[PPT / 2], md[PPT / 2];
#pragma unroll
  for (int j = 0; j < PPT; ++j) {
    float x, y, z;
    point(c, t + j * NT, x, y, z);
    px[j / 2][j & 1] = x; py[j / 2][j & 1] = y; pz[j / 2][j & 1] = z; md[j / 2][j & 1] = 1e10f;
    if (F & 4) { s_x[t + j * NT] = x; s_y[t + j * NT] = y; s_z[t + j * NT] = z; }
  }
  __syncthreads();
  unsigned bad = 0;
  float acc = 0.f;
  for (int i = 0; i < samples; ++i) {
    const unsigned far = (unsigned)(i * 1103 + 7 * c) % NPTS;            // wave-uniform schedule: a different owner lane every sample
    float fx, fy, fz;
    if (F & 4) { fx = s_x[far]; fy = s_y[far]; fz = s_z[far]; }
    else point(c, (int)far, fx, fy, fz);
    float lm = -1.f;
    if (F & 8) {
#pragma unroll
      for (int j = 0; j < PPT; ++j) {
        float dx, dy, dz, xx, yy, zz, d;
        asm("v_sub_f32 %0, %1, %2" : "=v"(dx) : "v"(px[j / 2][j & 1]), "v"(fx));
        asm("v_sub_f32 %0, %1, %2" : "=v"(dy) : "v"(py[j / 2][j & 1]), "v"(fy));
        asm("v_sub_f32 %0, %1, %2" : "=v"(dz) : "v"(pz[j / 2][j & 1]), "v"(fz));
        asm("v_mul_f32 %0, %1, %1" : "=v"(xx) : "v"(dx));
        asm("v_mul_f32 %0, %1, %1" : "=v"(yy) : "v"(dy));
        asm("v_mul_f32 %0, %1, %1" : "=v"(zz) : "v"(dz));
        asm("v_add_f32 %0, %1, %2" : "=v"(d) : "v"(xx), "v"(yy));
        asm("v_add_f32 %0, %1, %2" : "=v"(d) : "v"(d), "v"(zz));
        const float m = v_min(md[j / 2][j & 1], d);
        md[j / 2][j & 1] = m;
        lm = fmaxf(lm, m);
      }
    } else {
      const f32x2 f2x = {fx, fx}, f2y = {fy, fy}, f2z = {fz, fz};
#pragma unroll
      for (int j = 0; j < PPT / 2; ++j) {
        const f32x2 dx = px[j] - f2x, dy = py[j] - f2y, dz = pz[j] - f2z;
        const f32x2 d = (dx * dx + dy * dy) + dz * dz;
        f32x2 m = md[j];
        m.x = v_min(m.x, d.x);
        m.y = v_min(m.y, d.y);
        md[j] = m;
        lm = fmaxf(lm, fmaxf(m.x, m.y));
      }
    }
    if ((unsigned)t == far % NT) {               // the invariant: the sample's own min-distance is 0 now
      float m = -2.f;
#pragma unroll
      for (int j = 0; j < PPT; ++j) if ((unsigned)j == far / NT) m = md[j / 2][j & 1];
      bad += m != 0.f;
    }
    if (F & 1) acc += wave_max_f32(lm);
    else acc += lm;
    if (F & 2) {
      if (lane == 0) s_key[i & 1][wave] = ((unsigned long long)__float_as_uint(lm) << 32) | (unsigned)i;
      __syncthreads();
      acc += (float)(unsigned)(s_key[i & 1][lane & 3] & 0xff);
    }
  }
  if (bad) { atomicAdd(&lane_hist[lane], bad); atomicAdd(total, bad); }
  if (sink) sink[blockIdx.x * NT + t] = acc + md[0].x;
}

template <int F>
static void run_case(const char *name, double secs, int clouds, hipStream_t sv, hipStream_t sa, unsigned *hist, unsigned *total, float *sink,
                     const void *Y, float *ws, bool aggress) {
  CHECK(hipMemset(hist, 0, 64 * sizeof(unsigned)));
  CHECK(hipMemset(total, 0, sizeof(unsigned)));
  CHECK(hipDeviceSynchronize());
  const auto t0 = std::chrono::steady_clock::now();
  unsigned long long launches = 0;
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs) {
    for (int r = 0; r < 4; ++r) victim<F><<<clouds, NT, 0, sv>>>(512, hist, total, sink);
    if (aggress)
      for (int r = 0; r < 60; ++r)
        if (cpfn_mlp_wgrad(Y, 64, Y, 64, nullptr, 131072, 64, 64, nullptr, nullptr, ws, nullptr, (void *)sa) != 0) { printf("cpfn_mlp_wgrad failed\n"); exit(1); }
    CHECK(hipStreamSynchronize(sv));
    CHECK(hipStreamSynchronize(sa));
    launches += 4;
  }
  std::vector<unsigned> h(64);
  unsigned tot = 0;
  CHECK(hipMemcpy(h.data(), hist, 64 * sizeof(unsigned), hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(&tot, total, sizeof(unsigned), hipMemcpyDeviceToHost));
  unsigned rows[4] = {0, 0, 0, 0};
  for (int l = 0; l < 64; ++l) rows[l >> 4] += h[l];
  printf("%-66s %s  %6llu launches, %7u violations; by 16-lane row: %u %u %u %u\n", name, aggress ? "beside mlp_wgrad" : "alone           ", launches, tot,
         rows[0], rows[1], rows[2], rows[3]);
  fflush(stdout);
}

int main(int argc, char **argv) {
  const double secs = argc > 1 ? atof(argv[1]) : 3.0;
  const int clouds = 16;
  hipStream_t sv, sa;
  CHECK(hipStreamCreate(&sv));
  CHECK(hipStreamCreate(&sa));
  unsigned *hist, *total;
  float *sink, *ws;
  void *Y;
  CHECK(hipMalloc(&hist, 64 * sizeof(unsigned)));
  CHECK(hipMalloc(&total, sizeof(unsigned)));
  CHECK(hipMalloc(&sink, (size_t)clouds * NT * sizeof(float)));
  CHECK(hipMalloc(&Y, (size_t)131072 * 64 * 2));
  CHECK(hipMemset(Y, 0x3c, (size_t)131072 * 64 * 2));
  const int splits = cpfn_mlp_wgrad_splits(131072, 64, 64);
  CHECK(hipMalloc(&ws, (size_t)splits * 64 * 64 * sizeof(float)));
  run_case<0>("packed update + invariant", secs, clouds, sv, sa, hist, total, sink, Y, ws, false);
  run_case<0>("packed update + invariant", secs, clouds, sv, sa, hist, total, sink, Y, ws, true);
  run_case<1>("... + DPP wave maximum (row-restricted writes, v_readlane)", secs, clouds, sv, sa, hist, total, sink, Y, ws, true);
  run_case<2>("... + LDS key slot and a barrier per sample", secs, clouds, sv, sa, hist, total, sink, Y, ws, true);
  run_case<4>("... + sample read back from the LDS mirror", secs, clouds, sv, sa, hist, total, sink, Y, ws, true);
  run_case<7>("all of them (the sampling loop without its arg-max)", secs, clouds, sv, sa, hist, total, sink, Y, ws, true);
  run_case<15>("all of them, one float per instruction (control)", secs, clouds, sv, sa, hist, total, sink, Y, ws, true);
  return 0;
}
