// Shared helpers for the gfx950 kernels of libcpfn_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/cpfn_hip.h"

#define CPFN_WAVE 64

static inline int cpfn_launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

static inline int cpfn_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

bool cpfn_background_geometry();      // abi.hip: cpfn_set_background_geometry

// ‖p‖² rounded like torch.sum(p**2, dim=1) on CPU: ((x²+y²)+z²), no contraction.
// 1 if v is NaN or +-inf (exponent all ones): the test of the gradient scans (optim.hip, cpfn_multi_copy_checked)
__device__ __forceinline__ unsigned cpfn_nonfinite(float v) { return (__float_as_uint(v) & 0x7f800000u) == 0x7f800000u; }

__device__ __forceinline__ float cpfn_sqnorm3(float x, float y, float z) {
  return __fadd_rn(__fadd_rn(__fmul_rn(x, x), __fmul_rn(y, y)), __fmul_rn(z, z));
}

// One entry of the reference's pairwise_squared_distance on CPU
// (modules/geometry_utils.py:20-22): -2*(fma chain x,y,z) + |src|² + |dst|².
__device__ __forceinline__ float cpfn_pair_sqdist(float sx, float sy, float sz, float sn,
                                                  float dx, float dy, float dz, float dn) {
  float dot = __fmul_rn(sx, dx);
  dot = __fmaf_rn(sy, dy, dot);
  dot = __fmaf_rn(sz, dz, dot);
  float d = __fmul_rn(-2.0f, dot);
  d = __fadd_rn(d, sn);
  return __fadd_rn(d, dn);
}

// 16-byte LDS read that the compiler may neither narrow nor merge.  Why it exists: when only three of the four
// floats are used the compiler narrows the access to ds_read_b96, and on gfx950 that instruction was measured to
// return wrong data now and then while a workgroup of ANOTHER kernel on the same CU is writing LDS heavily
// (furthest-point sampling next to the weight-gradient kernel: 195 of 200 runs picked a spurious point; three
// ds_read_b32 or one ds_read_b128: 0 of 200 — tools/debug_fps_eager.py).  build.py rejects any object that
// contains a 96-bit DS instruction.
typedef float cpfn_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ cpfn_f32x4 cpfn_lds_read4(const float *p) {
  cpfn_f32x4 v = *(const cpfn_f32x4 *)p;
  asm volatile("" : "+v"(v));   // all four lanes "used": the load cannot be narrowed to 96 bits (it can still be scheduled freely)
  return v;
}

// Row tiles between global memory and LDS with fully coalesced accesses: `rows` consecutive rows of `width`
// floats are contiguous in memory; in LDS they sit at row stride `ld` (odd: a lane-per-row reader is
// conflict-free).  A lane that walks its own row in GLOBAL memory instead (width*4-byte stride between lanes)
// touches 64 cache lines per load instruction and ran the per-point loss / fitter kernels at ~1 TB/s.
template <int THREADS>
__device__ __forceinline__ void cpfn_rows_to_lds(float *s, int ld, const float *__restrict__ src, int rows, int width, int t) {
  for (int e = t; e < rows * width; e += THREADS) {
    const int r = e / width;
    s[r * ld + (e - r * width)] = src[e];
  }
}
template <int THREADS>
__device__ __forceinline__ void cpfn_rows_from_lds(const float *s, int ld, float *__restrict__ dst, int rows, int width, int t) {
  for (int e = t; e < rows * width; e += THREADS) {
    const int r = e / width;
    dst[e] = s[r * ld + (e - r * width)];
  }
}

__device__ __forceinline__ unsigned long long cpfn_shfl_xor_u64(unsigned long long v, int m) {
  unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
  lo = __shfl_xor(lo, m, CPFN_WAVE);
  hi = __shfl_xor(hi, m, CPFN_WAVE);
  return ((unsigned long long)hi << 32) | lo;
}
