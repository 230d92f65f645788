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

// max of a 64-bit key over lanes by DPP row operations (VALU moves, ~8 cycles each) instead of ds_bpermute butterflies
// (an LDS-pipeline round trip per step: the two key reductions of a sample were ~0.5 us of its ~1.2 us).
//   quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror: every lane of a 16-lane row holds the row's maximum;
//   row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3: lane 63 holds the wave's maximum.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned long long key_max_dpp_step(unsigned long long k) {
  const int lo = (int)(unsigned)k, hi = (int)(unsigned)(k >> 32);
  const unsigned olo = (unsigned)__builtin_amdgcn_update_dpp(lo, lo, CTRL, ROW_MASK, 0xF, false);
  const unsigned ohi = (unsigned)__builtin_amdgcn_update_dpp(hi, hi, CTRL, ROW_MASK, 0xF, false);
  const unsigned long long o = ((unsigned long long)ohi << 32) | olo;
  return o > k ? o : k;
}
// every lane of a group of G consecutive lanes (G = 4, 8, 16: aligned) gets the group's maximum
template <int G>
__device__ __forceinline__ unsigned long long group_max_key(unsigned long long k) {
  k = key_max_dpp_step<0xB1, 0xF>(k);
  k = key_max_dpp_step<0x4E, 0xF>(k);
  if (G >= 8) k = key_max_dpp_step<0x141, 0xF>(k);
  if (G >= 16) k = key_max_dpp_step<0x140, 0xF>(k);
  return k;
}
// the wave's maximum, wave-uniform (read back from lane 63)
__device__ __forceinline__ unsigned long long wave_max_key(unsigned long long k) {
  k = group_max_key<16>(k);
  k = key_max_dpp_step<0x142, 0xA>(k);
  k = key_max_dpp_step<0x143, 0xC>(k);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)k, 63);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(k >> 32), 63);
  return ((unsigned long long)hi << 32) | lo;
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
    // the lane's own arg-max over its PPT points: NCH interleaved running maxima (slot j feeds chain j % NCH), combined at
    // the end — one chain of PPT dependent compare / select pairs is what a wave waits for when it has a SIMD to itself
    // (16 or 32 points per lane); ties: the lowest index, as the single chain's strict '>' gave
    constexpr int NCH = PPT >= 16 ? 4 : 1;
    float bestc[NCH];
    unsigned bestic[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) { bestc[c] = -1.0f; bestic[c] = 0xFFFFFFFFu; }
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
      const float dx = __fsub_rn(px[j], fx), dy = __fsub_rn(py[j], fy), dz = __fsub_rn(pz[j], fz);
      const float d = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
      float m = md[j];
      m = d < m ? d : m;  // never true for m = -1 (d >= 0)
      md[j] = m;
      if (m > bestc[j % NCH]) {
        bestc[j % NCH] = m;
        bestic[j % NCH] = (unsigned)(t + j * NT);
      }
    }
    float best = bestc[0];
    unsigned besti = bestic[0];
#pragma unroll
    for (int c = 1; c < NCH; ++c) {
      if (bestc[c] > best || (bestc[c] == best && bestic[c] < besti)) { best = bestc[c]; besti = bestic[c]; }
    }
    unsigned long long key =
        best < 0.f ? 0ull : (((unsigned long long)__float_as_uint(best) << 32) | (unsigned)(~besti));
    key = wave_max_key(key);
    if (NW > 1) {
      if (lane == 0) s_key[i & 1][wave] = key;
      __syncthreads();
      key = s_key[i & 1][lane & (NW - 1)];
      if (NW >= 4) {
        key = group_max_key<(NW >= 4 ? NW : 4)>(key);
      } else {
#pragma unroll
        for (int m = NW / 2; m >= 1; m >>= 1) {
          unsigned long long o = cpfn_shfl_xor_u64(key, m);
          key = o > key ? o : key;
        }
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

// Large clouds (8192 < N <= 524288: the 131072-point evaluation clouds of the cascade), SEVERAL workgroups per cloud.
// The streaming kernel above runs a cloud on ONE compute unit: 131072 points cost ~24 us per sample there (12.4 ms for
// 512 samples — 71 % of the GlobalSPFN evaluation forward).  Here G = N / (256 * PPT) workgroups share a cloud: every
// lane keeps PPT points and their min-distances in registers exactly like the resident kernel, a workgroup reduces to one
// 64-bit key, and the G keys of a sample are exchanged through G 8-byte slots in global memory (double-buffered by
// sample parity): one agent-scope 8-byte store per workgroup and sample, G lanes of every workgroup poll the slots
// (agent-scope 8-byte loads) until each carries the sample's tag.  The whole hand-off is the 8-byte word itself —
//     key = dist bits << 32 | (0xFFFFF - index) << 12 | ((sample + 1) & 0xFFF)
// — so no payload has to be ordered behind a flag (8-byte stores / loads are single-copy atomic), ties still go to the
// lowest index, and a stale slot (tag of another sample) can never be taken for a fresh one: S <= 4094, slots zeroed by
// a memset node in front of the launch.  All workgroups of a cloud must be resident together (they spin on each other):
// the launcher only takes this path while B * G fits the device — hipOccupancyMaxActiveBlocksPerMultiprocessor of the
// instantiation x the number of compute units, queried once per device — and otherwise falls back to the one-workgroup
// streaming kernel.  Other kernels on the same CUs only delay a sibling; should one never arrive, a bounded spin
// (~1 s, once: the sample loop ends there) leaves index 0 (a valid point) in the remaining outputs and counts the cloud in
// a device-side fault counter that cpfn_fps_faults() reads: a hung GPU and out-of-range indices are both worse.
// Same arithmetic, same tie-break as the other two kernels: bit-identical selections.
__device__ unsigned g_fps_faults = 0;
template <int PPT>
__global__ __launch_bounds__(256) void fps_shared_kernel(const float *__restrict__ xyz, int N, int S,
                                                         const int *__restrict__ start, int flags,
                                                         int *__restrict__ idx_out, unsigned long long *__restrict__ slots) {
  constexpr int NT = 256, NW = 4;
  __shared__ unsigned long long s_key[2][NW];
  __shared__ unsigned s_far;
  const int G = gridDim.x, wg = blockIdx.x, b = blockIdx.y;
  const int t = threadIdx.x, lane = t & (CPFN_WAVE - 1), wave = t / CPFN_WAVE;
  const float *p = xyz + (size_t)b * N * 3;
  int *out = idx_out + (size_t)b * S;
  unsigned long long *sl = slots + (size_t)b * 2 * G;
  const int base = wg * NT * PPT;
  float px[PPT], py[PPT], pz[PPT], md[PPT];
#pragma unroll
  for (int j = 0; j < PPT; ++j) {
    const int k = base + t + j * NT;
    float x = 0.f, y = 0.f, z = 0.f, m = -1.0f;      // m < 0 marks "not a candidate"
    if (k < N) {
      x = p[3 * k]; y = p[3 * k + 1]; z = p[3 * k + 2];
      m = 1e10f;
      if ((flags & CPFN_FPS_SKIP_NEAR_ORIGIN) && cpfn_sqnorm3(x, y, z) <= 1e-3f) m = -1.0f;
    }
    px[j] = x; py[j] = y; pz[j] = z; md[j] = m;
  }
  unsigned far = start ? (unsigned)start[b] : 0u;
  bool dead = false;
  for (int i = 0; i < S; ++i) {
    if (wg == 0 && t == 0) out[i] = (int)far;
    const float fx = p[3 * far], fy = p[3 * far + 1], fz = p[3 * far + 2];
    float best = -1.0f;
    unsigned besti = 0xFFFFFu;
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
      const float dx = __fsub_rn(px[j], fx), dy = __fsub_rn(py[j], fy), dz = __fsub_rn(pz[j], fz);
      const float d = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
      float m = md[j];
      m = d < m ? d : m;
      md[j] = m;
      if (m > best) { best = m; besti = (unsigned)(base + t + j * NT); }
    }
    // key without the tag: candidates compare by (distance, lowest index); "no candidate" = 0
    unsigned long long key = best < 0.f ? 0ull : (((unsigned long long)__float_as_uint(best) << 32) |
                                                   ((unsigned long long)(0xFFFFFu - besti) << 12));
    key = wave_max_key(key);
    if (lane == 0) s_key[i & 1][wave] = key;
    __syncthreads();
    const unsigned tag = (unsigned)(i + 1) & 0xFFFu;
    if (t == 0) {
      unsigned long long k4 = s_key[i & 1][0];
#pragma unroll
      for (int w = 1; w < NW; ++w) k4 = s_key[i & 1][w] > k4 ? s_key[i & 1][w] : k4;
      __hip_atomic_store(&sl[(i & 1) * G + wg], k4 | tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (wave == 0) {       // G <= 64 lanes poll one slot each until it carries this sample's tag
      unsigned long long k = 0ull;
      if (lane < G) {
        unsigned spins = 0;
        do {
          k = __hip_atomic_load(&sl[(i & 1) * G + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (++spins > (1u << 24)) { k = ~0ull; break; }          // ~1 s: a sibling workgroup never arrived
        } while ((unsigned)(k & 0xFFFull) != tag);
      }
      const bool timeout = __ballot(k == ~0ull) != 0ull;
      k = (lane < G && !timeout) ? (k & ~0xFFFull) : 0ull;
      k = wave_max_key(k);
      if (lane == 0) s_far = timeout ? 0xFFFFFFFFu : (k ? 0xFFFFFu - (unsigned)((k >> 12) & 0xFFFFFull) : 0u);
    }
    __syncthreads();
    const unsigned nf = s_far;
    if (nf == 0xFFFFFFFFu) {          // a sibling workgroup never arrived: give up for this cloud (every workgroup of it
      dead = true;                    // takes this branch at most one spin period later)
      if (wg == 0 && t == 0) {
        for (int r = i + 1; r < S; ++r) out[r] = 0;
        atomicAdd(&g_fps_faults, 1u);
      }
      break;
    }
    far = nf;
  }
  (void)dead;
}

}  // namespace

// Workgroups of fps_shared_kernel<PPT> that can be resident at once on the current device (0: unknown -> do not use it).
template <int PPT>
static int fps_shared_capacity() {
  static int cached[64] = {0};                  // per device ordinal; 0 = not queried yet, -1 = query failed
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
  if (cached[dev] == 0) {
    int per_cu = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fps_shared_kernel<PPT>, 256, 0) == hipSuccess &&
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && per_cu > 0 && cus > 0)
      cached[dev] = per_cu * cus;
    else
      cached[dev] = -1;
  }
  return cached[dev] > 0 ? cached[dev] : 0;
}

// Clouds whose several-workgroups FPS gave up on a sibling since the library was loaded (synchronises the device).
extern "C" int cpfn_fps_faults(void) {
  unsigned n = 0;
  if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_fps_faults), sizeof(n)) != hipSuccess) return -1;
  return (int)n;
}

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
    // 8192 points on ONE CU either way (a sample is a VALU-throughput phase over the cloud plus two key reductions):
    // (with ds_bpermute key reductions) 16 waves x 8 points per lane took 670 us for 512 samples, 8 waves x 16 points 572 us,
    // 4 waves x 32 points 636 us (one wave per SIMD: no other wave hides a wave's dependent chains), 2 x 64 1003 us
    // (registers spill to AGPRs); with the DPP reductions 4 x 32 takes 522 us (the step beside it: 1.898 -> 1.868 ms).
    // Beside a training step (one workgroup per cloud on 16 CUs for the whole forward pass) the FEWER waves the better
    // for the step: 1.871 ms with 16 waves, 1.855 with 8, 1.849 with 4 (interleaved A/B on one box each).
    if (cpfn_background_geometry())
      fps_resident_kernel<256, 32><<<B, 256, 0, st>>>(xyz, N, S, start, flags, idx_out);
    else
      fps_resident_kernel<512, 16><<<B, 512, 0, st>>>(xyz, N, S, start, flags, idx_out);
  } else {
    if (!scratch) return CPFN_EINVAL;
    // several workgroups per cloud while all of them can be resident together and the key layout holds
    // (index < 2^20, sample tag < 4095); slots = the first B * 2 * G 8-byte words of the scratch row buffer
    int ppt = 8;
    while (ppt < 32 && (N + 256 * ppt - 1) / (256 * ppt) > 64) ppt *= 2;
    const int G = (N + 256 * ppt - 1) / (256 * ppt);
    const int capacity = ppt == 8 ? fps_shared_capacity<8>() : ppt == 16 ? fps_shared_capacity<16>() : fps_shared_capacity<32>();
    if (G <= 64 && (long long)B * G <= capacity && B <= 65535 && S <= 4094 && N <= (1 << 20) &&
        (size_t)B * 2 * G * 8 <= (size_t)B * N * 4 && ((uintptr_t)scratch & 7) == 0) {
      hipError_t e = hipMemsetAsync(scratch, 0, (size_t)B * 2 * G * 8, st);
      if (e != hipSuccess) return (int)e;
      unsigned long long *slots = (unsigned long long *)scratch;
      const dim3 grid(G, B);
      if (ppt == 8) fps_shared_kernel<8><<<grid, 256, 0, st>>>(xyz, N, S, start, flags, idx_out, slots);
      else if (ppt == 16) fps_shared_kernel<16><<<grid, 256, 0, st>>>(xyz, N, S, start, flags, idx_out, slots);
      else fps_shared_kernel<32><<<grid, 256, 0, st>>>(xyz, N, S, start, flags, idx_out, slots);
    } else {
      fps_streaming_kernel<1024><<<B, 1024, 0, st>>>(xyz, N, S, start, flags, idx_out, scratch);
    }
  }
  return cpfn_launch_status();
}
