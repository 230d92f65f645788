// Shared pieces of the per-point MLP translation units (mlp_fwd.hip, mlp_small.hip, mlp_bwd_fused.hip, bn.hip; one file,
// mlp.hip, until round 4): fragment types, the in-kernel timing probe, the LDS weight-panel fill, the transposing LDS
// fragment read, the counter-based dropout mask.
#pragma once
#include "common.h"
#include <cstdlib>
#include <type_traits>

struct GemmProbeSlot {
  unsigned long long nwg, pad;                 // header: workgroups of the launch that wrote this slot
  unsigned long long t[1][2];                  // [workgroup][start, end] (max_wg entries)
};
struct GemmProbeState {
  unsigned long long *buf = nullptr;           // slots x (2 + 2 max_wg) u64
  int slots = 0, max_wg = 0;
  unsigned next = 0;                           // launches handed a slot so far
};
extern GemmProbeState g_probe_state;          // (defined in mlp_fwd.hip: ONE slot counter for every family)

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;

__device__ __forceinline__ float bf2f(unsigned short u) { return __uint_as_float((unsigned)u << 16); }
__device__ __forceinline__ unsigned short f2bf(float f) {
  return __builtin_bit_cast(unsigned short, (__bf16)f);  // v_cvt_pk_bf16_f32: RNE, NaN-preserving
}

// a' = relu(scale·a + shift) on eight consecutive channels of one point, rounded to bf16 exactly like
// bn_relu_apply_kernel: lets a GEMM / weight-gradient kernel consume the PREVIOUS layer's pre-BN output directly
// (the activated tensor is then never written or read).
__device__ __forceinline__ bf16x8 bn_relu_frag(bf16x8 v, const float *sc, const float *sh) {
  typedef __attribute__((ext_vector_type(8))) unsigned short u16x8;
  const u16x8 u = __builtin_bit_cast(u16x8, v);
  u16x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = f2bf(fmaxf(fmaf(sc[j], bf2f(u[j]), sh[j]), 0.f));
  return __builtin_bit_cast(bf16x8, o);
}

// ---- in-kernel timing probe of the GEMM family (bench.py's roofline leg) -----------------------------------------
// A replayed hipGraph cannot be bracketed per kernel with host-side events, so when a probe is installed
// (cpfn_mlp_gemm_set_probe) every launch of the family times ITSELF with the device's constant-rate wall clock
// (s_memrealtime, 100 MHz): every workgroup stores its own (start, end) pair into the launch's slot of the probe
// buffer — plain 16-byte stores, no atomics (a first version folded the times into one counter with two device-scope
// atomics per workgroup: ~18 us per launch of pure atomic latency, the step went from 2.4 to 3.0 ms) — and the host
// takes max(end) - min(start) per slot afterwards.  A launch captured into a graph keeps the slot it was given at
// capture time, so after a run of replays the buffer holds the LAST replay's launches.  No probe (the default): the
// argument is NULL and the cost is one uniform branch.
// slot of the launch being issued (NULL: probe off or grid too large for a slot)
static inline unsigned long long *probe_slot(dim3 grid) {
  GemmProbeState &p = g_probe_state;
  if (!p.buf || (long long)grid.x * grid.y * grid.z > p.max_wg) return nullptr;
  return p.buf + (size_t)(p.next++ % (unsigned)p.slots) * (2 + 2 * (size_t)p.max_wg);
}
__device__ __forceinline__ unsigned long long probe_begin(const unsigned long long *slot) {
  return slot ? (unsigned long long)wall_clock64() : 0ull;
}
__device__ __forceinline__ void probe_end(unsigned long long *slot, unsigned long long t0, unsigned kind = 1) {
  if (!slot) return;
  __syncthreads();                // every wave of the workgroup has issued its last store
  if (threadIdx.x == 0) {
    const unsigned wg = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    typedef __attribute__((ext_vector_type(2))) unsigned long long u64x2;
    *(u64x2 *)&slot[2 + 2 * (size_t)wg] = (u64x2){t0, (unsigned long long)wall_clock64()};
    if (wg == 0) {
      slot[0] = (unsigned long long)gridDim.x * gridDim.y * gridDim.z;
      slot[1] = kind;            // 1 streaming GEMM, 2 generic GEMM, 3 small-P GEMM, 4 weight gradient, 5 one-pass backward, 6 small layer: weight + data gradient
    }
  }
}
// (the weight-gradient and one-pass backward kernels take probe slots too — kinds 4 and 5 — so that bench.py can name the
//  family with the most time in the step and tools/dbg/probe_timeline.py has ~50 anchors inside a replayed step)
static inline unsigned long long *probe_slot_all(dim3 grid) { return probe_slot(grid); }
constexpr int G_THREADS = 256;
constexpr int G_ROWS = 128;   // points per workgroup tile (4 waves x 32)
constexpr int G_KC = 128;     // K chunk staged in LDS
constexpr int G_LDW = G_KC + 8;

// Y[P,N] (bf16 or fp32) = A[P,K] (bf16, optional row gather) · W[N,K]ᵀ (bf16) (+ bias)
// stats_partial[gridDim.x][2][N]: per-workgroup Σy, Σy² over its valid rows (fp32 accumulators).
//
// Two kernels share the MFMA tile shape (128 points x BN channels per workgroup, 32 points per wave):
//  * mlp_gemm_stream_kernel<BN,KS,STATS>: K = 32·KS in {64,128}, bf16 output, N % BN == 0.  These are
//    the layers that carry the bytes (P >= 131072 rows).  HBM-bound streaming structure: W panel
//    loaded into LDS once per workgroup; A fragments of tile t+1 are requested BEFORE the MFMAs of
//    tile t (two register buffers, straight-line loop body so the compiler emits counted
//    s_waitcnt vmcnt(N) instead of vmcnt(0)); the output tile leaves through a wave-private LDS patch
//    as 16-byte row-contiguous stores (4 rows x 256 B per wave instruction).
//  * mlp_gemm_kernel<BN,STATS>: any K % 32 == 0 (chunked through LDS), optional row gather, bias,
//    fp32 / ragged-N output: the small-P layers (sa3, sfp1, sfp2), K > 128, and the fc2 heads.
constexpr int G_LDO = 128 + 8;  // output staging row stride (elements): 272 B

// eight 16-bit elements rotated left by r positions (result[i] = v[(i + r) & 7]) without register indexing
__device__ __forceinline__ uint4 rot_u16x8(uint4 v, int r) {
  if (r & 4) v = (uint4){v.z, v.w, v.x, v.y};
  if (r & 2) v = (uint4){v.y, v.z, v.w, v.x};
  if (r & 1) v = (uint4){(v.x >> 16) | (v.y << 16), (v.y >> 16) | (v.z << 16), (v.z >> 16) | (v.w << 16), (v.w >> 16) | (v.x << 16)};
  return v;
}

// Fill the LDS weight panel s_w[r][k] (r = output channel n0+r, k in [kc, kc+kcn)).
// w_trans == 0: W is [N,K] row-major (16-byte loads along k).
// w_trans == 1: W is [K,N] row-major (the FORWARD layer's weight used for the data gradient): 16-byte
//               loads along n, transposed on the way into LDS (8-byte pieces, see below), so no transposed
//               weight copy has to be materialised per step.
template <int BN, int LDW = G_LDW, int NT = G_THREADS>
__device__ __forceinline__ void fill_w_panel(unsigned short *s_w, const unsigned short *__restrict__ W, int K,
                                             int N, int n0, int kc, int kcn, int w_trans, int t) {
  if (!w_trans) {
    const int cpr = kcn / 8;  // 16-byte chunks per row
    for (int e = t; e < BN * cpr; e += NT) {
      const int r = e / cpr, c = e - r * cpr;
      *(uint4 *)&s_w[r * LDW + c * 8] = *(const uint4 *)&W[(size_t)(n0 + r) * K + kc + c * 8];
    }
  } else {
    // W is [K, N] (a forward weight used for the data gradient): transpose on the way into LDS.  A lane takes
    // FOUR consecutive k rows of one 8-column chunk and writes 8-byte pieces (4 k values of one column); the
    // column order is rotated by the chunk index, otherwise the 16 lanes of a k-row hit one LDS bank (a panel row
    // is 272 B, so 8 rows apart = 2176 B = 17 x 128 B).  The first version wrote single bf16s with that 16-way
    // conflict: the transposed launches ran 25 us against 17.5 us for the plain ones.
    constexpr int cpn = BN / 8;  // 16-byte chunks along n
    for (int e = t; e < (kcn / 4) * cpn; e += NT) {
      const int k4 = e / cpn, c = e - k4 * cpn;
      uint4 v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = *(const uint4 *)&W[(size_t)(kc + 4 * k4 + r) * N + n0 + c * 8];
      // rotate the eight columns of every row vector by (c & 7) positions with whole-register selects — indexing
      // the registers with a lane-dependent j would push them to scratch memory
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = rot_u16x8(v[r], c & 7);
      const unsigned short *h0 = (const unsigned short *)&v[0], *h1 = (const unsigned short *)&v[1],
                           *h2 = (const unsigned short *)&v[2], *h3 = (const unsigned short *)&v[3];
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) {
        const int j = (jj + c) & 7;      // rotated position jj holds column j
        uint2 o;
        o.x = (unsigned)h0[jj] | ((unsigned)h1[jj] << 16);
        o.y = (unsigned)h2[jj] | ((unsigned)h3[jj] << 16);
        *(uint2 *)&s_w[(c * 8 + j) * LDW + 4 * k4] = o;
      }
    }
  }
}

template <int LD>
__device__ __forceinline__ bf16x8 tr_frag(const unsigned short *tile, int col0, int lane) {
  // fragment F[x = lane&15][k = 8(lane>>4)+j] = tile[row k][col0 + x]  (tile rows = contraction index)
  const int grp = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
  const unsigned short *a0 = tile + (8 * grp + q) * LD + col0 + 4 * pp;
  const unsigned short *a1 = a0 + 4 * LD;
  typedef s16x4 __attribute__((address_space(3))) * lds_p;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)a0);
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)a1);
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, v);  // one whole-vector cast: element-wise casts of the tr-read result miscompile
}

constexpr int SP_MAX_ROWS = 16384;
constexpr int SP_SS_MAX = 512;
constexpr int SP_DEPTH = 2;   // slots in flight per wave (4 measured slower on every shape: registers -> occupancy)
// 64-row tiles halve the weight re-reads; 32-row tiles when that would leave fewer than 256 workgroups
static inline int sp_rows(long long P, int N) { return ((P + 63) / 64) * (N / 64 > 0 ? N / 64 : 1) >= 256 ? 64 : 32; }

// ---- dropout fused into the BatchNorm apply passes (the reference applies F.dropout(p = 0.5) to the fc1
// features in every mode, PointNet2/pn2_network.py:63; as PyTorch ops that is a mask-producing kernel in the forward
// pass and a masked-scale kernel in the backward pass over [B*N, 128]).  Counter-based: element chunk e (8 consecutive
// channels of one point) keeps element j iff the j-th 16-bit field of splitmix64(seed, e) is >= p * 65536, so the
// backward passes recompute the mask from the 8-byte seed instead of reading a stored one.
__device__ __forceinline__ unsigned long long splitmix64(unsigned long long x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
// keep[j] for the 8 elements of chunk e: scale factor (1/(1-p)) or 0
__device__ __forceinline__ void dropout_factors(unsigned long long seed, unsigned long long e, unsigned thresh16,
                                                float inv_keep, float (&f)[8]) {
  const unsigned long long h0 = splitmix64(seed ^ (2 * e)), h1 = splitmix64(seed ^ (2 * e + 1));
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    f[j] = ((unsigned)(h0 >> (16 * j)) & 0xffffu) >= thresh16 ? inv_keep : 0.f;
    f[4 + j] = ((unsigned)(h1 >> (16 * j)) & 0xffffu) >= thresh16 ? inv_keep : 0.f;
  }
}
__host__ __device__ inline unsigned dropout_thresh16(float p) {
  const float t = p * 65536.f + 0.5f;
  return t <= 0.f ? 0u : (t >= 65536.f ? 65536u : (unsigned)t);
}


// g_z = g_a·[z>0]; per-workgroup partial Σ g_z and Σ g_z·y per channel.  Gz may alias Ga.
// Rows per workgroup of the column-reduction passes: ~1024 workgroups when P allows (a 4-workgroup launch
// on the 2048-row sa4 layers was a 70 us latency chain), between 16 and 512 rows each.
static inline int bn_rows_per_block(long long P) {
  int r = 16;
  while (r < 512 && (P + r - 1) / r > 1024) r *= 2;
  return r;
}

// Sum over the 16 lanes of a DPP row (lanes that share lane>>4): every lane ends up with the total.
// quad_perm xor-1, quad_perm xor-2, row_half_mirror, row_mirror — four VALU adds, no LDS traffic.
__device__ __forceinline__ float row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
  return v;
}

constexpr int WG_STEP = 32;       // rows per MFMA step
constexpr int WG_DEPTH = 4;       // steps in flight

// launcher of the small-P GEMM kernel (mlp_small.hip) for cpfn_mlp_gemm's P <= SP_MAX_ROWS branch (mlp_fwd.hip)
}  // namespace

int cpfn_smallp_gemm_launch(const unsigned short *a, int lda, const unsigned short *w, int w_trans, long long P, int K, int N,
                            unsigned short *y, int ldy, float *stats_partial, const float *a_scale, const float *a_shift, int gx,
                            hipStream_t st, const cpfn_seam_out *seam_out = nullptr, const cpfn_seam_in *seam_in = nullptr);
// out[i] = sum over splits of ws[s][i], fixed order (split_reduce_kernel, bn.hip)
void cpfn_launch_split_reduce(const float *ws, int splits, long long n, float *out, hipStream_t st);
