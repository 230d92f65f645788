// Which instruction of a weight-gradient kernel makes packed fp32 in ANOTHER wave lose a row?  
// A self-contained program, not part of the library:
//     hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/dbg/pk_micro.hip -o /tmp/pk_micro && /tmp/pk_micro [seconds per case]
// VICTIM: every wave keeps eight pairs of points in registers and, per iteration, forms the squared distance to a moving sample twice
// — packed (v_pk_add_f32 with negation + op_sel broadcast, v_pk_mul_f32: the sampling kernel's form) and one float at a time (inline
// assembly, so the vectoriser cannot pair it up) — and counts the lanes whose two results differ in their bits, by lane.  One
// 256-lane workgroup per compute unit, little LDS and few registers, so that the aggressor's workgroups can be resident beside it.
// AGGRESSORS (one at a time on a second stream, two workgroups per compute unit): nothing; transposed LDS reads only
// (ds_read_b64_tr_b16); plain 64-bit LDS reads; LDS writes; MFMA only; transposed reads + MFMA (the weight-gradient inner loop).
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define CHECK(x)                                                                          \
  do {                                                                                    \
    hipError_t e_ = (x);                                                                  \
    if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); }     \
  } while (0)

__device__ __forceinline__ float scalar_d2(float px, float py, float pz, float fx, float fy, float fz) {
  float dx, dy, dz, xx, yy, zz, d;
  asm volatile("v_sub_f32 %0, %1, %2" : "=v"(dx) : "v"(px), "v"(fx));
  asm volatile("v_sub_f32 %0, %1, %2" : "=v"(dy) : "v"(py), "v"(fy));
  asm volatile("v_sub_f32 %0, %1, %2" : "=v"(dz) : "v"(pz), "v"(fz));
  asm volatile("v_mul_f32 %0, %1, %1" : "=v"(xx) : "v"(dx));
  asm volatile("v_mul_f32 %0, %1, %1" : "=v"(yy) : "v"(dy));
  asm volatile("v_mul_f32 %0, %1, %1" : "=v"(zz) : "v"(dz));
  asm volatile("v_add_f32 %0, %1, %2" : "=v"(d) : "v"(xx), "v"(yy));
  asm volatile("v_add_f32 %0, %1, %2" : "=v"(d) : "v"(d), "v"(zz));
  return d;
}

__global__ __launch_bounds__(256) void victim(unsigned long long iterations, unsigned *lane_hist /*[64]*/, unsigned *total) {
  const int t = threadIdx.x, lane = t & 63;
  f32x2 px[8], py[8], pz[8];
  for (int j = 0; j < 8; ++j) {
    const float b = 0.001f * (float)(t + 256 * j + 17 * blockIdx.x);
    px[j] = (f32x2){b, b + 0.37f};
    py[j] = (f32x2){0.5f - b, 0.25f + b};
    pz[j] = (f32x2){b * 0.7f, 1.0f - b};
  }
  unsigned bad = 0;
  float fx = 0.1f, fy = 0.2f, fz = 0.3f;
  for (unsigned long long it = 0; it < iterations; ++it) {
    const f32x2 f2x = {fx, fx}, f2y = {fy, fy}, f2z = {fz, fz};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const f32x2 dx = px[j] - f2x, dy = py[j] - f2y, dz = pz[j] - f2z;
      const f32x2 d = (dx * dx + dy * dy) + dz * dz;                 // packed
      const float s0 = scalar_d2(px[j].x, py[j].x, pz[j].x, fx, fy, fz), s1 = scalar_d2(px[j].y, py[j].y, pz[j].y, fx, fy, fz);
      bad += (__float_as_uint(d.x) != __float_as_uint(s0)) + (__float_as_uint(d.y) != __float_as_uint(s1));
    }
    fx += 0.001f; fy -= 0.0007f; fz += 0.0003f;                      // (wave-uniform: the sample moves)
    if (fx > 1.f) { fx = 0.1f; fy = 0.2f; fz = 0.3f; }
  }
  if (bad) { atomicAdd(&lane_hist[lane], bad); atomicAdd(total, bad); }
}

template <int KIND>   // 1 transposed LDS reads, 2 plain 64-bit LDS reads, 3 LDS writes, 4 MFMA only, 5 transposed reads + MFMA
__global__ __launch_bounds__(256) void aggressor(unsigned long long iterations, float *sink) {
  __shared__ __attribute__((aligned(16))) unsigned short tile[64 * 72];
  const int t = threadIdx.x, lane = t & 63;
  for (int e = t; e < 64 * 72; e += 256) tile[e] = (unsigned short)(0x3c00 + (e & 255));
  __syncthreads();
  typedef s16x4 __attribute__((address_space(3))) * lds_p;
  typedef unsigned long long __attribute__((address_space(3))) * lds_u64;
  const int grp = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
  f32x4 acc = {0, 0, 0, 0};
  long long sum = 0;
  for (unsigned long long it = 0; it < iterations; ++it) {
    const int col0 = (int)(it & 3) * 16;
    const unsigned short *a0 = tile + (8 * grp + q) * 72 + col0 + 4 * pp;
    if (KIND == 1 || KIND == 5) {
      const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)a0);
      const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(a0 + 4 * 72));
      if (KIND == 5) {
        typedef __attribute__((ext_vector_type(8))) short s16x8;
        const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        const bf16x8 f = __builtin_bit_cast(bf16x8, v);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f, f, acc, 0, 0, 0);
      } else {
        sum += lo.x + hi.w;
      }
    } else if (KIND == 2) {
      sum += (long long)*(volatile lds_u64)(a0);
    } else if (KIND == 3) {
      *(volatile lds_u64)(a0) = (unsigned long long)it;
    } else if (KIND == 4) {
      bf16x8 f;
      for (int j = 0; j < 8; ++j) f[j] = (__bf16)(float)(lane + j);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f, f, acc, 0, 0, 0);
    }
  }
  if (sink) sink[blockIdx.x * 256 + t] = acc[0] + acc[1] + (float)sum;
}

int main(int argc, char **argv) {
  const double secs = argc > 1 ? atof(argv[1]) : 3.0;
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  hipStream_t sv, sa;
  CHECK(hipStreamCreate(&sv));
  CHECK(hipStreamCreate(&sa));
  unsigned *hist, *total;
  float *sink;
  CHECK(hipMalloc(&hist, 64 * sizeof(unsigned)));
  CHECK(hipMalloc(&total, sizeof(unsigned)));
  CHECK(hipMalloc(&sink, (size_t)2 * cus * 256 * sizeof(float)));
  const char *names[6] = {"nothing", "transposed LDS reads (ds_read_b64_tr_b16)", "plain 64-bit LDS reads", "64-bit LDS writes", "MFMA only",
                          "transposed reads + MFMA"};
  for (int kind = 0; kind < 6; ++kind) {
    CHECK(hipMemset(hist, 0, 64 * sizeof(unsigned)));
    CHECK(hipMemset(total, 0, sizeof(unsigned)));
    CHECK(hipDeviceSynchronize());
    const auto t0 = std::chrono::steady_clock::now();
    unsigned long long launches = 0;
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs) {
      victim<<<cus, 256, 0, sv>>>(20000ull, hist, total);            // ~ms-long launches on both streams, re-issued until the time is up
      const unsigned long long n = 200000ull;
      switch (kind) {
        case 1: aggressor<1><<<2 * cus, 256, 0, sa>>>(n, sink); break;
        case 2: aggressor<2><<<2 * cus, 256, 0, sa>>>(n, sink); break;
        case 3: aggressor<3><<<2 * cus, 256, 0, sa>>>(n, sink); break;
        case 4: aggressor<4><<<2 * cus, 256, 0, sa>>>(n, sink); break;
        case 5: aggressor<5><<<2 * cus, 256, 0, sa>>>(n, sink); break;
        default: break;
      }
      CHECK(hipStreamSynchronize(sv));
      CHECK(hipStreamSynchronize(sa));
      ++launches;
    }
    std::vector<unsigned> h(64);
    unsigned tot = 0;
    CHECK(hipMemcpy(h.data(), hist, 64 * sizeof(unsigned), hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(&tot, total, sizeof(unsigned), hipMemcpyDeviceToHost));
    unsigned rows[4] = {0, 0, 0, 0};
    for (int l = 0; l < 64; ++l) rows[l >> 4] += h[l];
    printf("%-44s %6llu victim launches (%.2e packed instructions), %8u mismatching results; by 16-lane row: %u %u %u %u\n", names[kind],
           launches, (double)launches * cus * 4 * 20000.0 * 8 * 8, tot, rows[0], rows[1], rows[2], rows[3]);
    fflush(stdout);
  }
  return 0;
}
