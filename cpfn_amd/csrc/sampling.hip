// Furthest-point sampling for gfx950.
//
// One workgroup per cloud.  The running min-distance of every point lives in
// registers (PPT points per lane), the cloud itself is mirrored in LDS (x, y, z planes) so
// that the coordinates of the newly selected point are three broadcast ds_read_b32
// away, and the arg-max is a 64-bit key max-reduction:
//     key = (fp32 bits of distance) << 32 | ~index
// (distances are >= 0 so their bit patterns order like the floats; ~index makes the
// LOWEST index win a tie, which is what torch.max returns on CPU).  Per sample: one
// wave-level butterfly, one LDS slot per wave, ONE workgroup barrier (slots are
// double-buffered by sample parity), one 16-lane butterfly.
//
// Arithmetic follows the reference's CPU route (modules/geometry_utils.py:88-101)
// op for op — ((dx*dx + dy*dy) + dz*dz) with every op rounded — so that the selected
// indices are identical to it, not merely close.
#include "common.h"

namespace {

__device__ __forceinline__ unsigned long long wave_max_key(unsigned long long k) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    unsigned long long o = cpfn_shfl_xor_u64(k, m);
    k = o > k ? o : k;
  }
  return k;
}

template <int NT, int PPT>
__global__ __launch_bounds__(NT) void fps_resident_kernel(const float *__restrict__ xyz, int N, int S,
                                                          const int *__restrict__ start, int flags,
                                                          int *__restrict__ idx_out) {
  constexpr int NW = NT / CPFN_WAVE;
  __shared__ float s_x[NT * PPT], s_y[NT * PPT], s_z[NT * PPT];   // three b32 broadcasts per sample, NOT one float4:
                                                                   // see cpfn_lds_read4 in common.h (ds_read_b96)
  __shared__ unsigned long long s_key[2][NW > 1 ? NW : 1];

  const int b = blockIdx.x;
  const int t = threadIdx.x;
  const int lane = t & (CPFN_WAVE - 1);
  const int wave = t / CPFN_WAVE;
  const float *p = xyz + (size_t)b * N * 3;
  int *out = idx_out + (size_t)b * S;

  float px[PPT], py[PPT], pz[PPT], md[PPT];
#pragma unroll
  for (int j = 0; j < PPT; ++j) {
    const int k = t + j * NT;
    float x = 0.f, y = 0.f, z = 0.f, m = -1.0f;  // m < 0 marks "not a candidate"
    if (k < N) {
      x = p[3 * k];
      y = p[3 * k + 1];
      z = p[3 * k + 2];
      m = 1e10f;
      if ((flags & CPFN_FPS_SKIP_NEAR_ORIGIN) && cpfn_sqnorm3(x, y, z) <= 1e-3f) m = -1.0f;
    }
    px[j] = x; py[j] = y; pz[j] = z; md[j] = m;
    s_x[k] = x; s_y[k] = y; s_z[k] = z;
  }
  __syncthreads();

  unsigned far = start ? (unsigned)start[b] : 0u;
  for (int i = 0; i < S; ++i) {
    if (t == 0) out[i] = (int)far;
    const float fx = s_x[far], fy = s_y[far], fz = s_z[far];
    float best = -1.0f;
    unsigned besti = 0xFFFFFFFFu;
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
      const float dx = __fsub_rn(px[j], fx), dy = __fsub_rn(py[j], fy), dz = __fsub_rn(pz[j], fz);
      const float d = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
      float m = md[j];
      m = d < m ? d : m;  // never true for m = -1 (d >= 0)
      md[j] = m;
      if (m > best) {
        best = m;
        besti = (unsigned)(t + j * NT);
      }
    }
    unsigned long long key =
        best < 0.f ? 0ull : (((unsigned long long)__float_as_uint(best) << 32) | (unsigned)(~besti));
    key = wave_max_key(key);
    if (NW > 1) {
      if (lane == 0) s_key[i & 1][wave] = key;
      __syncthreads();
      key = s_key[i & 1][lane & (NW - 1)];
#pragma unroll
      for (int m = NW / 2; m >= 1; m >>= 1) {
        unsigned long long o = cpfn_shfl_xor_u64(key, m);
        key = o > key ? o : key;
      }
    }
    far = key ? ~(unsigned)(key & 0xFFFFFFFFull) : 0u;
  }
}

// Any N: min-distances in a global scratch row (L2-resident), coordinates re-read from
// global memory.  Used for clouds that do not fit the resident kernel (e.g. the 128k-pt
// evaluation clouds); same arithmetic, same tie-break.
template <int NT>
__global__ __launch_bounds__(NT) void fps_streaming_kernel(const float *__restrict__ xyz, int N, int S,
                                                           const int *__restrict__ start, int flags,
                                                           int *__restrict__ idx_out,
                                                           float *__restrict__ scratch) {
  constexpr int NW = NT / CPFN_WAVE;
  __shared__ unsigned long long s_key[2][NW];
  const int b = blockIdx.x;
  const int t = threadIdx.x;
  const int lane = t & (CPFN_WAVE - 1);
  const int wave = t / CPFN_WAVE;
  const float *p = xyz + (size_t)b * N * 3;
  float *md = scratch + (size_t)b * N;
  int *out = idx_out + (size_t)b * S;

  for (int k = t; k < N; k += NT) {
    float m = 1e10f;
    if ((flags & CPFN_FPS_SKIP_NEAR_ORIGIN) && cpfn_sqnorm3(p[3 * k], p[3 * k + 1], p[3 * k + 2]) <= 1e-3f)
      m = -1.0f;
    md[k] = m;
  }
  unsigned far = start ? (unsigned)start[b] : 0u;
  for (int i = 0; i < S; ++i) {
    if (t == 0) out[i] = (int)far;
    const float fx = p[3 * far], fy = p[3 * far + 1], fz = p[3 * far + 2];
    float best = -1.0f;
    unsigned besti = 0xFFFFFFFFu;
    for (int k = t; k < N; k += NT) {
      const float dx = __fsub_rn(p[3 * k], fx), dy = __fsub_rn(p[3 * k + 1], fy), dz = __fsub_rn(p[3 * k + 2], fz);
      const float d = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
      float m = md[k];
      m = d < m ? d : m;
      md[k] = m;
      if (m > best) {
        best = m;
        besti = (unsigned)k;
      }
    }
    unsigned long long key =
        best < 0.f ? 0ull : (((unsigned long long)__float_as_uint(best) << 32) | (unsigned)(~besti));
    key = wave_max_key(key);
    if (lane == 0) s_key[i & 1][wave] = key;
    __syncthreads();
    key = s_key[i & 1][lane & (NW - 1)];
#pragma unroll
    for (int m = NW / 2; m >= 1; m >>= 1) {
      unsigned long long o = cpfn_shfl_xor_u64(key, m);
      key = o > key ? o : key;
    }
    far = key ? ~(unsigned)(key & 0xFFFFFFFFull) : 0u;
  }
}

}  // namespace

extern "C" int cpfn_fps(const float *xyz, int B, int N, int S, const int *start, int flags, int *idx_out,
                        float *scratch, void *stream) {
  if (B < 0 || N <= 0 || S < 0 || !xyz || (!idx_out && B * S > 0)) return CPFN_EINVAL;
  if (B == 0 || S == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  if (N <= 512) {
    fps_resident_kernel<64, 8><<<B, 64, 0, st>>>(xyz, N, S, start, flags, idx_out);
  } else if (N <= 2048) {
    fps_resident_kernel<256, 8><<<B, 256, 0, st>>>(xyz, N, S, start, flags, idx_out);
  } else if (N <= CPFN_FPS_MAX_RESIDENT) {
    fps_resident_kernel<1024, 8><<<B, 1024, 0, st>>>(xyz, N, S, start, flags, idx_out);
  } else {
    if (!scratch) return CPFN_EINVAL;
    fps_streaming_kernel<1024><<<B, 1024, 0, st>>>(xyz, N, S, start, flags, idx_out, scratch);
  }
  return cpfn_launch_status();
}
