// Per-point MLP stacks, small layers (P <= 16384 rows: sa3, sfp1, sfp2) and the stand-alone weight gradient: the split-K
// small-P GEMM, mlp_wgrad_kernel, and the two of them as ONE launch (mlp_bwd_small_kernel).  See mlp_fwd.hip for the data layout.
#include "mlp_common.h"
#include "seam.h"

namespace {

// ---- small-P kernel (P <= 16384 rows: sa3, sfp1, sfp2 and their data gradients).
// These layers move a few MB and are pure latency: with 128-row tiles they fill 16-128 workgroups and walk K
// (up to 1280) chunk by chunk, 2-3 us of exposed memory latency per chunk.  Here a workgroup owns RT (32|64) rows x
// 64 channels and its four waves SPLIT K (wave w takes the 32-wide k-steps w, w+4, ...): 4x-16x more workgroups,
// a 4x shorter serial chain per wave, no LDS panel and no barrier inside the K loop.  A and W fragments go straight
// from global memory (L2-resident after the first touch) into MFMA operand registers through a 2-slot register
// pipeline of bounds-checked buffer loads (counted vmcnt waits).  For the data gradient (W stored [K,N]) a wave
// bounces its 32 x 64 weight slice through a wave-private LDS tile and reads it back with ds_read_b64_tr_b16.
// The four K-partial accumulators are summed through LDS in a fixed order; wave w finishes channels 16w..16w+15:
// BatchNorm statistics by DPP row sums straight into the partial buffer, bf16 rows to Y.
// Measured (rocprofv3, tools/smallp_probe.py: the 14 small-P launches of one GlobalSPFN step, operands cold):
//   first version (128-row tiles, chunk-serial)   ~220 us     same with double-buffered panel   174 us
//   this kernel                                    112 us     (2048 x 1280 -> 256: 43 -> 9.8 us)
// What is left is launch + two memory round trips per workgroup: with stores, statistics and MFMAs removed the
// sum only drops to 98 us.

// Data gradient of a small layer with the reduction of the layer below riding on it (cpfn_mlp_dgrad_small):
//   BST:   pass 1 of the BatchNorm backward of the layer BELOW from the tile being stored (sum g_z, sum g_z y with the
//          ReLU mask from that layer's pre-BN output Yb): its stand-alone cpfn_bn_relu_bwd launch disappears.
// (Round 2 also formed g_y on the operand load here and in the 64 x 64 weight gradient: every 64-column block re-forms the
//  whole panel, the two kernels got 4-6 us slower each and cancelled the saved launch — removed in round 3.)
struct SmallpBwdArgs {
  const unsigned short *Yb;                 // BST: [P, ldy] like Y
  const float *b_scale, *b_shift;           // BST: [N]
};

template <int RT>
struct SmallpLds {
  static constexpr int TT = RT / 16, LDT = 64 + 8, RAW_TILE = 4 * 32 * LDT * 2, RAW_RED = 4 * 4 * TT * 64 * 16;
  __attribute__((aligned(16))) unsigned char raw[RAW_TILE > RAW_RED ? RAW_TILE : RAW_RED];
  __attribute__((aligned(16))) float ss[2][SP_SS_MAX];
};

// (body with the workgroup's position as arguments: mlp_gemm_smallp_kernel runs it on its own grid, mlp_bwd_small_kernel on
//  the tail of a grid whose head is a small layer's weight gradient)
template <int RT, bool STATS, bool WT, bool BST>
__device__ __forceinline__ void mlp_gemm_smallp_body(
    SmallpLds<RT> &lds, int bx, int by,
    const unsigned short *__restrict__ A, int lda, int a_bytes, const unsigned short *__restrict__ W,
    int P, int K, int N, unsigned short *__restrict__ Y, int ldy, float *__restrict__ stats_partial,
    const float *__restrict__ a_scale, const float *__restrict__ a_shift, const SmallpBwdArgs &bw,
    const SeamOut &so = SeamOut(), const SeamIn &si = SeamIn()) {
  constexpr int TT = RT / 16, D = SP_DEPTH, LDT = 64 + 8;
  static_assert(!(STATS && BST) && (!BST || WT), "the riding reduction belongs to the data gradient");
  unsigned char *s_raw = lds.raw;
  float (*s_ss)[SP_SS_MAX] = lds.ss;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int lr = lane & 15, lq = lane >> 4;
  const int n0 = by * 64, row0 = bx * RT;
  const bool transform = a_scale || si.acc;
  // scale / shift of the previous layer: read, or FOLDED from that layer's sums (seam.h: no finalize launch in between) — the
  // accumulator words are requested here, the arithmetic runs behind the first operand requests (K <= 512: two channels per lane)
  // (replicas <= 4 on this route — cpfn_mlp_gemm_seam checks — so the words held across the first requests cost 20 registers;
  //  a second channel per lane, K > 256, is folded in one piece behind them)
  SeamFoldRegsT<4> fq;
  if (si.acc) {
    if (t < K) seam_fold_issue(si, t, fq);
  } else if (a_scale) {
    for (int e = t; e < K; e += 256) { s_ss[0][e] = a_scale[e]; s_ss[1][e] = a_shift[e]; }
    __syncthreads();
  }
  if (STATS && so.acc && t == 0 && bx == 0 && by == 0) seam_counters(so);
  const int S = K / 32;
  // Buffer loads (SGPR base + 32-bit lane offset, hardware bounds check): a pipeline slot past the end of K gets an
  // out-of-range offset, which returns zeros WITHOUT touching memory — the loop body stays straight-line (counted
  // vmcnt waits) and the tail slots cost nothing.  (Plain loads with clamped addresses re-read real data there:
  // with K = 256 that was 4x the traffic, and the kernel was slower than the one it replaces.)
  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void *)A, 0, a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void *)W, 0, N * K * 2, 0x00020000);
  unsigned aoff[TT];   // byte offsets
#pragma unroll
  for (int tt = 0; tt < TT; ++tt) {
    int p = row0 + tt * 16 + lr;
    p = p < P ? p : P - 1;
    aoff[tt] = ((unsigned)p * lda + 8 * lq) * 2;
  }
  unsigned woff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int e = lane + 64 * i;   // WT: 16-byte piece e of the [32 k][64 n] slice: k row e>>3, columns 8(e&7)..
    woff[i] = WT ? ((unsigned)(e >> 3) * N + n0 + (e & 7) * 8) * 2 : ((unsigned)(n0 + i * 16 + lr) * K + 8 * lq) * 2;
  }
  const unsigned wstep = WT ? 32u * N * 2 : 64u;
  unsigned short *tile = (unsigned short *)s_raw + wave * 32 * LDT;   // this wave's [32 k][64 n] slice (WT only)
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4;   // (a plain vector type: HIP's uint4 struct blocks SROA here)
  u32x4 ra[D][TT];
  u32x4 rw[D][4];
  auto issue = [&](int d, int s) __attribute__((always_inline)) {
    const unsigned oob = s < S ? 0u : 0x80000000u;
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) ra[d][tt] = __builtin_amdgcn_raw_buffer_load_b128(rs_a, (aoff[tt] + s * 64) | oob, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) rw[d][i] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, (woff[i] + s * wstep) | oob, 0, 0);
  };
  f32x4 acc[4][TT];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) acc[i][tt] = (f32x4){0, 0, 0, 0};
#pragma unroll
  for (int d = 0; d < D; ++d) issue(d, wave + 4 * d);
  if (si.acc) {
    float sc, sh;
    if (t < K) { seam_fold_finish(si, t, bx == 0 && by == 0, fq, sc, sh); s_ss[0][t] = sc; s_ss[1][t] = sh; }
    if (t + 256 < K) {
      seam_fold_issue(si, t + 256, fq);
      seam_fold_finish(si, t + 256, bx == 0 && by == 0, fq, sc, sh);
      s_ss[0][t + 256] = sc; s_ss[1][t + 256] = sh;
    }
    __syncthreads();
  }
  const int cnt = (S + 3) / 4;
  // (one instantiation of the stage body per pipeline slot: the slot index must be a compile-time constant so that
  //  ra / rw stay in registers)
  auto stage = [&](auto slot, int i0) __attribute__((always_inline)) {
      constexpr int d = decltype(slot)::value;
      const int s = wave + 4 * (i0 + d);
      if (s < S) {
        bf16x8 af[TT], wf[4];
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) af[tt] = __builtin_bit_cast(bf16x8, ra[d][tt]);
        if (WT) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int e = lane + 64 * i;
            *(u32x4 *)&tile[(e >> 3) * LDT + (e & 7) * 8] = rw[d][i];
          }
          // (wave-private tile: LDS executes one wave's instructions in order, and the compiler keeps the
          //  may-alias write -> transposing read -> next write order: no barrier of any kind is needed)
#pragma unroll
          for (int i = 0; i < 4; ++i) wf[i] = tr_frag<LDT>(tile, i * 16, lane);
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i) wf[i] = __builtin_bit_cast(bf16x8, rw[d][i]);
        }
        if (transform) {   // BatchNorm + ReLU of the previous layer applied to the operand on the fly
          float sc[8], sh[8];
          const int k0 = s * 32 + 8 * lq;
          *(cpfn_f32x4 *)&sc[0] = *(const cpfn_f32x4 *)&s_ss[0][k0]; *(cpfn_f32x4 *)&sc[4] = *(const cpfn_f32x4 *)&s_ss[0][k0 + 4];
          *(cpfn_f32x4 *)&sh[0] = *(const cpfn_f32x4 *)&s_ss[1][k0]; *(cpfn_f32x4 *)&sh[4] = *(const cpfn_f32x4 *)&s_ss[1][k0 + 4];
#pragma unroll
          for (int tt = 0; tt < TT; ++tt) af[tt] = bn_relu_frag(af[tt], sc, sh);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int tt = 0; tt < TT; ++tt)
            acc[i][tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], af[tt], acc[i][tt], 0, 0, 0);
      }
      // the slot is refilled AFTER its operands were consumed (no register copies: occupancy, not per-wave depth,
      // is what hides the latency here — measured: depth 2 beats depth 4 on every shape)
      issue(d, wave + 4 * (i0 + d + D));
  };
  static_assert(D == 2, "the pipeline slots are spelled out below");
  for (int i0 = 0; i0 < cnt; i0 += D) {
    stage(std::integral_constant<int, 0>{}, i0);
    stage(std::integral_constant<int, 1>{}, i0);
  }
  // K-partials of the four waves -> LDS in register layout [src wave][nt][tt][lane] (conflict-free 16-byte stores);
  // wave w then owns channel block nt = w and adds the four partials in wave order (fixed: reproducible)
  __syncthreads();   // the tiles are dead (s_raw is reused)
  f32x4 *s_part = (f32x4 *)s_raw;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) s_part[((wave * 4 + i) * TT + tt) * 64 + lane] = acc[i][tt];
  __syncthreads();
  const int n = n0 + wave * 16 + 4 * lq;
  f32x4 sm = {0, 0, 0, 0}, sq = {0, 0, 0, 0};
#pragma unroll
  for (int tt = 0; tt < TT; ++tt) {
    f32x4 v = s_part[((0 * 4 + wave) * TT + tt) * 64 + lane];
#pragma unroll
    for (int w = 1; w < 4; ++w) v += s_part[((w * 4 + wave) * TT + tt) * 64 + lane];
    const int p = row0 + tt * 16 + lr;
    if (p < P) {
      if (STATS) { sm += v; sq += v * v; }
      bf16x4 o = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
      *(bf16x4 *)(Y + (size_t)p * ldy + n) = o;
      if (BST) {     // (on the ROUNDED gradient, as the stand-alone pass would read it back)
        const bf16x4 yb = *(const bf16x4 *)(bw.Yb + (size_t)p * ldy + n);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float yv = (float)yb[r];
          const float z = fmaf(bw.b_scale[n + r], yv, bw.b_shift[n + r]) > 0.f ? (float)o[r] : 0.f;
          sm[r] += z;
          sq[r] = fmaf(z, yv, sq[r]);
        }
      }
    }
  }
  if (STATS || BST) {
#pragma unroll
    for (int r = 0; r < 4; ++r) { sm[r] = row16_sum(sm[r]); sq[r] = row16_sum(sq[r]); }
    if (lr == 0) {
      if (STATS && so.acc) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { seam_add(so, N, 0, n + r, sm[r], (unsigned)bx); seam_add(so, N, 1, n + r, sq[r], (unsigned)bx); }
      } else {
        *(f32x4 *)&stats_partial[((size_t)bx * 2 + 0) * N + n] = sm;
        *(f32x4 *)&stats_partial[((size_t)bx * 2 + 1) * N + n] = sq;
      }
    }
  }
}

template <int RT, bool STATS, bool WT, bool BST = false>
__global__ __launch_bounds__(256) void mlp_gemm_smallp_kernel(
    const unsigned short *__restrict__ A, int lda, int a_bytes, const unsigned short *__restrict__ W,
    int P, int K, int N, unsigned short *__restrict__ Y, int ldy, float *__restrict__ stats_partial,
    const float *__restrict__ a_scale, const float *__restrict__ a_shift, unsigned long long *probe,
    const SmallpBwdArgs bw = SmallpBwdArgs(), const SeamOut so = SeamOut(), const SeamIn si = SeamIn()) {
  const unsigned long long probe_t0 = probe_begin(probe);
  __shared__ SmallpLds<RT> lds;
  mlp_gemm_smallp_body<RT, STATS, WT, BST>(lds, blockIdx.x, blockIdx.y, A, lda, a_bytes, W, P, K, N, Y, ldy, stats_partial, a_scale,
                                            a_shift, bw, so, si);
  probe_end(probe, probe_t0, 3);
}


// ---------------------------------------------------------------- weight gradient
// dW[n,k] = Σ_p Gy[p,n]·A[p,k]: the contraction runs over ROWS, so both MFMA operands are
// transposed tiles — staged row-major in LDS and read with ds_read_b64_tr_b16.
// grid (N/TN, ceil(K/TK), splits); partial[split][N][K] fp32.  The row loop is a 4-deep register pipeline:
// the 16-byte chunks of step i+4 are in flight while step i goes registers -> LDS -> transposed fragments ->
// MFMA, so a workgroup's time is its bytes, not (steps x memory latency) as in the first version (which
// had a 25 us floor on every layer).  128x128 tiles read each operand once for the 128-wide layers.

template <int TN, int TK>
struct WgradLds {
  __attribute__((aligned(16))) unsigned short g[WG_STEP * (TN + 8)];
  __attribute__((aligned(16))) unsigned short a[WG_STEP * (TK + 8)];
};

// (body with the workgroup's position as arguments, like mlp_gemm_smallp_body)
template <int TN, int TK>
__device__ __forceinline__ void mlp_wgrad_body(WgradLds<TN, TK> &lds, int bx, int by, int bz,
                                               const unsigned short *__restrict__ Gy, int ldg,
                                               const unsigned short *__restrict__ A, int lda,
                                               const int *__restrict__ gidx, long long P, int N, int K,
                                               long long rows_per_split, float *__restrict__ partial,
                                               const float *__restrict__ a_scale, const float *__restrict__ a_shift) {
  constexpr int LDN = TN + 8, LDK = TK + 8;       // LDS row strides (elements)
  constexpr int CG = TN / 64, CA = TK / 64;       // 16-byte chunks per thread and step
  constexpr int MI = TN / 32, MJ = TK / 32;       // MFMA tiles per wave (wave sub-tile = TN/2 x TK/2)
  unsigned short *s_g = lds.g, *s_a = lds.a;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int n0 = bx * TN, k0 = by * TK;
  const long long p0 = (long long)bz * rows_per_split, p1 = min(P, p0 + rows_per_split);
  if (p0 >= p1) {   // empty split: its partial slab must still be zero
    float *o = partial + (size_t)bz * N * K;
    for (int e = t; e < TN * TK; e += 256) {
      const int n = n0 + e / TK, k = k0 + e % TK;
      if (n < N && k < K) o[(size_t)n * K + k] = 0.f;
    }
    return;
  }
  const int wn = (wave >> 1) * (TN / 2), wk = (wave & 1) * (TK / 2);
  f32x4 acc[MI][MJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < MJ; ++j) acc[i][j] = (f32x4){0, 0, 0, 0};
  // chunk c = t + 256 i of a step: row c / (T/8), column 8 (c % (T/8))
  uint4 vg[WG_DEPTH][CG], va[WG_DEPTH][CA];
  // optional BatchNorm + ReLU of the PREVIOUS layer on the A operand (a lane's chunk columns never change)
  float asc[CA][8], ash[CA][8];
  if (a_scale) {
#pragma unroll
    for (int i = 0; i < CA; ++i) {
      const int col = min(k0 + ((t + 256 * i) % (TK / 8)) * 8, K - 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) { asc[i][j] = a_scale[col + j]; ash[i][j] = a_shift[col + j]; }
    }
  }
  auto issue = [&](int sidx, long long base) {
#pragma unroll
    for (int i = 0; i < CG; ++i) {
      const int c = t + 256 * i;
      const long long p = min(base + c / (TN / 8), p1 - 1);      // clamped: always a valid row, zeroed at store time
      vg[sidx][i] = *(const uint4 *)(Gy + p * ldg + n0 + (c % (TN / 8)) * 8);
    }
#pragma unroll
    for (int i = 0; i < CA; ++i) {
      const int c = t + 256 * i;
      const long long p = min(base + c / (TK / 8), p1 - 1);
      const long long ar = gidx ? (long long)gidx[p] : p;
      const int col = min(k0 + (c % (TK / 8)) * 8, K - 8);
      va[sidx][i] = *(const uint4 *)(A + ar * lda + col);
    }
  };
  auto stage = [&](int sidx, long long base) {
#pragma unroll
    for (int i = 0; i < CG; ++i) {
      const int c = t + 256 * i, r = c / (TN / 8);
      uint4 v = vg[sidx][i];
      if (base + r >= p1) v = (uint4){0, 0, 0, 0};
      *(uint4 *)&s_g[r * LDN + (c % (TN / 8)) * 8] = v;
    }
#pragma unroll
    for (int i = 0; i < CA; ++i) {
      const int c = t + 256 * i, r = c / (TK / 8), col = (c % (TK / 8)) * 8;
      uint4 v = va[sidx][i];
      if (a_scale) v = __builtin_bit_cast(uint4, bn_relu_frag(__builtin_bit_cast(bf16x8, v), asc[i], ash[i]));
      if (base + r >= p1 || k0 + col >= K) v = (uint4){0, 0, 0, 0};
      *(uint4 *)&s_a[r * LDK + col] = v;
    }
  };
#pragma unroll
  for (int d = 0; d < WG_DEPTH; ++d) issue(d, p0 + (long long)d * WG_STEP);
  for (long long base0 = p0; base0 < p1; base0 += WG_STEP * WG_DEPTH) {
#pragma unroll
    for (int d = 0; d < WG_DEPTH; ++d) {
      // no "if (base < p1)" here: a skipped stage would make the number of loads in flight path-dependent and
      // the compiler falls back to vmcnt(0) drains; steps past the end stage zeros instead (rows_per_split is
      // a multiple of WG_STEP*WG_DEPTH, so only the last split of a ragged P ever does that)
      const long long base = base0 + (long long)d * WG_STEP;
      __syncthreads();
      stage(d, base);
      __syncthreads();
      issue(d, base + WG_STEP * WG_DEPTH);
      bf16x8 fg[MI], fa[MJ];
#pragma unroll
      for (int i = 0; i < MI; ++i) fg[i] = tr_frag<LDN>(s_g, wn + 16 * i, lane);
#pragma unroll
      for (int j = 0; j < MJ; ++j) fa[j] = tr_frag<LDK>(s_a, wk + 16 * j, lane);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < MJ; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fg[i], fa[j], acc[i][j], 0, 0, 0);
    }
  }
  // D[row = n-local 4(lane>>4)+r][col = k-local lane&15]
  float *o = partial + (size_t)bz * N * K;
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < MJ; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + wn + i * 16 + 4 * (lane >> 4) + r, k = k0 + wk + j * 16 + (lane & 15);
        if (n < N && k < K) o[(size_t)n * K + k] = acc[i][j][r];
      }
}

// workgroup `L` of a (gx, gy, gz = splits) weight-gradient grid -> (tile x, tile y, split z) with all tiles of a split on one XCD
__device__ __forceinline__ void wgrad_xcd_map_linear(int L, int gx, int gy, int gz, int &bx, int &by, int &bz) {
  const int tiles = gx * gy;
  if (gz & 7) { bx = L % gx; by = (L / gx) % gy; bz = L / tiles; return; }
  const int j = L >> 3, t = j % tiles;
  bz = (L & 7) + 8 * (j / tiles);
  bx = t % gx; by = t / gx;
}
__device__ __forceinline__ void wgrad_xcd_map(int &bx, int &by, int &bz, int gx, int gy, int gz) {
  wgrad_xcd_map_linear(bx + gx * (by + gy * bz), gx, gy, gz, bx, by, bz);
}

template <int TN, int TK>
__global__ __launch_bounds__(256) void mlp_wgrad_kernel(const unsigned short *__restrict__ Gy, int ldg,
                                                        const unsigned short *__restrict__ A, int lda,
                                                        const int *__restrict__ gidx, long long P, int N, int K,
                                                        long long rows_per_split, float *__restrict__ partial,
                                                        const float *__restrict__ a_scale,
                                                        const float *__restrict__ a_shift,
                                                        unsigned long long *probe = nullptr) {
  const unsigned long long probe_t0 = probe_begin(probe);
  __shared__ WgradLds<TN, TK> lds;
  // XCD-aware (round 6): the (N / TN) x (K / TK) tiles of one row split read the SAME rows of g_y and of the input; consecutive
  // workgroup ids go round-robin over the 8 XCDs, so with the plain (x, y, z) order every split's rows were fetched into every L2
  // (1.64 x the launch's bytes from HBM in round 5's counters).  Here XCD k takes splits k, k + 8, ..., all tiles of a split in a row.
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  wgrad_xcd_map(bx, by, bz, (int)gridDim.x, (int)gridDim.y, (int)gridDim.z);
  mlp_wgrad_body<TN, TK>(lds, bx, by, bz, Gy, ldg, A, lda, gidx, P, N, K, rows_per_split, partial, a_scale, a_shift);
  probe_end(probe, probe_t0, 4);
}

// A SMALL layer's weight gradient and data gradient in ONE launch (cpfn_mlp_bwd_small): both read the same g_y and do not
// depend on each other, and as two launches of 3-10 us each on the step's chain they cost a kernel boundary (2.5-3 us) plus
// the shorter of the two durations more than they must — seven times per backward pass (sa3, sfp1, sfp2).  The grid's first
// nW workgroups are mlp_wgrad_kernel<64,64>'s, the rest mlp_gemm_smallp_kernel<32,false,true,BST>'s; the two bodies share
// the workgroup's LDS (a union) and — at 32-row tiles — have the same register footprint (116 / 128), so neither loses
// occupancy.  Same arithmetic, same partial layouts: bit-identical to the two launches.
struct BwdSmallGrid { int nW, wgx, wgy, dgx; };
template <bool BST>
__global__ __launch_bounds__(256) void mlp_bwd_small_kernel(
    BwdSmallGrid gr, const unsigned short *__restrict__ Gy, int ldg, const unsigned short *__restrict__ A, int lda, long long P,
    int N, int K, long long rows_per_split, float *__restrict__ partial, const float *__restrict__ a_scale,
    const float *__restrict__ a_shift, const unsigned short *__restrict__ W, int g_bytes, unsigned short *__restrict__ Gout, int ldo,
    float *__restrict__ stats_partial, const SmallpBwdArgs bw, unsigned long long *probe) {
  const unsigned long long probe_t0 = probe_begin(probe);
  __shared__ union U { WgradLds<64, 64> w; SmallpLds<32> d; __device__ U() {} } lds;
  const int b = blockIdx.x;
  if (b < gr.nW) {
    int bx, by, bz;            // (all tiles of a row split on one XCD: see mlp_wgrad_kernel)
    wgrad_xcd_map_linear(b, gr.wgx, gr.wgy, gr.nW / (gr.wgx * gr.wgy), bx, by, bz);
    mlp_wgrad_body<64, 64>(lds.w, bx, by, bz, Gy, ldg, A, lda, nullptr, P, N, K, rows_per_split, partial, a_scale, a_shift);
  } else {
    const int d = b - gr.nW;
    // in the small-P kernel's terms: operand = g_y [P, N] (contraction over the layer's N output channels), outputs = K channels
    mlp_gemm_smallp_body<32, false, true, BST>(lds.d, d % gr.dgx, d / gr.dgx, Gy, ldg, g_bytes, W, (int)P, N, K, Gout, ldo, stats_partial,
                                               nullptr, nullptr, bw);
  }
  probe_end(probe, probe_t0, 6);
}


}  // namespace

// ============================================================================ C ABI

int cpfn_smallp_gemm_launch(const unsigned short *a, int lda, const unsigned short *w, int w_trans, long long P, int K, int N,
                            unsigned short *y, int ldy, float *stats_partial, const float *a_scale, const float *a_shift, int gx,
                            hipStream_t st, const cpfn_seam_out *seam_out, const cpfn_seam_in *seam_in) {
  dim3 grid(gx, N / 64);
  const int a_bytes = (int)(((P - 1) * lda + K) * 2);
  if (seam_out || seam_in) {      // forward layers only (cpfn_mlp_gemm_seam): statistics on, plain weight layout
    if (w_trans || (!stats_partial && !seam_out)) return CPFN_EINVAL;
    const SeamOut so = seam_out_arg(seam_out);
    const SeamIn si = seam_in_arg(seam_in);
    if (sp_rows(P, N) == 32)
      mlp_gemm_smallp_kernel<32, true, false><<<grid, 256, 0, st>>>(a, lda, a_bytes, w, (int)P, K, N, y, ldy, stats_partial, a_scale, a_shift, probe_slot(grid), SmallpBwdArgs(), so, si);
    else
      mlp_gemm_smallp_kernel<64, true, false><<<grid, 256, 0, st>>>(a, lda, a_bytes, w, (int)P, K, N, y, ldy, stats_partial, a_scale, a_shift, probe_slot(grid), SmallpBwdArgs(), so, si);
    return cpfn_launch_status();
  }
#define CPFN_SMALLP(RT_)                                                                                              \
  do {                                                                                                                \
    if (stats_partial && w_trans) mlp_gemm_smallp_kernel<RT_, true, true><<<grid, 256, 0, st>>>(a, lda, a_bytes, w, (int)P, K, N, y, ldy, stats_partial, a_scale, a_shift, probe_slot(grid));  \
    else if (stats_partial) mlp_gemm_smallp_kernel<RT_, true, false><<<grid, 256, 0, st>>>(a, lda, a_bytes, w, (int)P, K, N, y, ldy, stats_partial, a_scale, a_shift, probe_slot(grid));       \
    else if (w_trans) mlp_gemm_smallp_kernel<RT_, false, true><<<grid, 256, 0, st>>>(a, lda, a_bytes, w, (int)P, K, N, y, ldy, nullptr, a_scale, a_shift, probe_slot(grid));                  \
    else mlp_gemm_smallp_kernel<RT_, false, false><<<grid, 256, 0, st>>>(a, lda, a_bytes, w, (int)P, K, N, y, ldy, nullptr, a_scale, a_shift, probe_slot(grid));                             \
  } while (0)
    if (sp_rows(P, N) == 32) CPFN_SMALLP(32); else CPFN_SMALLP(64);
#undef CPFN_SMALLP
  return cpfn_launch_status();
}

extern "C" int cpfn_mlp_dgrad_small_ok(long long P, int N, int K) {
  // (N, the contraction length here, is NOT bounded by SP_SS_MAX: that is the size of the operand transform's scale / shift vectors,
  //  which a data gradient does not use.  Until round 6 it was, and sa3's 1024 <- 512 layer ran its weight gradient, its data gradient
  //  and the reduction of the layer below as three launches instead of one.)
  return P > 0 && P <= SP_MAX_ROWS && N > 0 && (N & 31) == 0 && N <= 4096 && K > 0 && (K & 63) == 0 &&
         P * N * 2 < (1LL << 31) && (long long)N * K * 2 < (1LL << 31);
}

// (declared after cpfn_mlp_gemm_blocks: rows of stats_partial = cpfn_mlp_gemm_blocks(P, K))
extern "C" int cpfn_mlp_dgrad_small(const void *Gy, const void *W, long long P, int N, int K, void *Gout, int ldo,
                                    const void *bwd_y, const float *b_scale, const float *b_shift, float *stats_partial,
                                    void *stream) {
  if (!cpfn_mlp_dgrad_small_ok(P, N, K) || !Gy || !W || !Gout || (ldo & 3) || ldo < K || !bwd_y || !b_scale || !b_shift ||
      !stats_partial)
    return CPFN_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  // in the kernel's terms: operand A = Gz [P, N] (contraction over the layer's N output channels), outputs = K channels
  const int gx = cpfn_mlp_gemm_blocks(P, K);
  dim3 grid(gx, K / 64);
  const int a_bytes = (int)(((P - 1) * N + N) * 2);
  SmallpBwdArgs bw;
  bw.Yb = (const unsigned short *)bwd_y; bw.b_scale = b_scale; bw.b_shift = b_shift;
  const unsigned short *a = (const unsigned short *)Gy, *w = (const unsigned short *)W;
  unsigned short *y = (unsigned short *)Gout;
#define CPFN_DGRAD_SMALL(RT_)                                                                                             \
  mlp_gemm_smallp_kernel<RT_, false, true, true><<<grid, 256, 0, st>>>(a, N, a_bytes, w, (int)P, N, K, y, ldo, stats_partial, nullptr, nullptr, probe_slot(grid), bw)
  if (sp_rows(P, K) == 32) CPFN_DGRAD_SMALL(32); else CPFN_DGRAD_SMALL(64);
#undef CPFN_DGRAD_SMALL
  return cpfn_launch_status();
}


// output tile for one (P, N, K): 128 x 128 where the layer is wide and long enough to fill the chip with 128-tiles,
// 128 x 64 for the long 64 -> 128 layer (sa1: both operands read once instead of the input twice), 64 x 64 otherwise
static inline void wgrad_tile(long long P, int N, int K, int *TN, int *TK) {
  if (P >= 32768 && N % 128 == 0 && K >= 128) { *TN = 128; *TK = 128; }
  else if (P >= 32768 && N % 128 == 0 && K == 64) { *TN = 128; *TK = 64; }
  else { *TN = 64; *TK = 64; }
}

extern "C" int cpfn_mlp_wgrad_splits(long long P, int N, int K) {
  int TN, TK;
  wgrad_tile(P, N, K, &TN, &TK);
  const long long tiles = (long long)((N + TN - 1) / TN) * ((K + TK - 1) / TK);
  const long long target = TN == 128 ? 512 : 1024;     // workgroups (256 / 512 / 2048 for the 64-tiles: no measurable difference)
  long long s = (target + tiles - 1) / tiles;
  if (s > 256) s = 256;   // bound the partial buffer / reduce depth (128 and 512 measured: +40 us per step each)
  const long long max_s = (P + 127) / 128;            // at least 128 rows (one pipeline depth) per split
  if (s > max_s) s = max_s;
  if (s < 1) s = 1;
  return (int)s;
}

extern "C" int cpfn_mlp_wgrad(const void *Gy, int ldg, const void *A, int lda, const int *gidx, long long P, int N,
                              int K, const float *a_scale, const float *a_shift, float *workspace, float *dW,
                              void *stream) {
  if (P <= 0 || N <= 0 || K < 8 || (N & 63) || (K & 31) || !Gy || !A || !workspace || (ldg & 7) || (lda & 7) ||
      (!a_scale != !a_shift))
    return CPFN_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int splits = cpfn_mlp_wgrad_splits(P, N, K);
  long long rps = (P + splits - 1) / splits;
  rps = ((rps + WG_STEP * WG_DEPTH - 1) / (WG_STEP * WG_DEPTH)) * (WG_STEP * WG_DEPTH);
  const unsigned short *g = (const unsigned short *)Gy, *a = (const unsigned short *)A;
  int TN, TK;
  wgrad_tile(P, N, K, &TN, &TK);
  if (TN == 128 && TK == 128) {
    dim3 grid(N / 128, (K + 127) / 128, splits);
    mlp_wgrad_kernel<128, 128><<<grid, 256, 0, st>>>(g, ldg, a, lda, gidx, P, N, K, rps, workspace, a_scale, a_shift, probe_slot_all(grid));
  } else if (TN == 128) {
    dim3 grid(N / 128, (K + 63) / 64, splits);
    mlp_wgrad_kernel<128, 64><<<grid, 256, 0, st>>>(g, ldg, a, lda, gidx, P, N, K, rps, workspace, a_scale, a_shift, probe_slot_all(grid));
  } else {
    dim3 grid(N / 64, (K + 63) / 64, splits);
    mlp_wgrad_kernel<64, 64><<<grid, 256, 0, st>>>(g, ldg, a, lda, gidx, P, N, K, rps, workspace, a_scale, a_shift, probe_slot_all(grid));
  }
  const long long n = (long long)N * K;
  if (dW) cpfn_launch_split_reduce(workspace, splits, n, dW, st);    // NULL: the caller batches it (cpfn_multi_split_reduce)
  return cpfn_launch_status();
}

// Weight gradient + data gradient of a small layer as ONE launch (mlp_bwd_small_kernel): workspace as cpfn_mlp_wgrad leaves it
// (cpfn_mlp_wgrad_splits(P,N,K) slabs), Gout [P,K] = Gy . W (W: the FORWARD panel [N][K]); bwd_y (optional, + b_scale / b_shift +
// stats_partial [cpfn_mlp_bwd_small_blocks(P)][2][K]): pass 1 of the BatchNorm backward of the layer below on the stored tile.
extern "C" int cpfn_mlp_bwd_small_ok(long long P, int N, int K) {
  return cpfn_mlp_wgrad_apply_ok(P, N, K) && cpfn_mlp_dgrad_small_ok(P, N, K);
}
extern "C" int cpfn_mlp_bwd_small_blocks(long long P) { return (int)((P + 31) / 32); }
extern "C" int cpfn_mlp_bwd_small(const void *Gy, int ldg, const void *A, int lda, const void *W, long long P, int N, int K,
                                  const float *a_scale, const float *a_shift, float *workspace, void *Gout, int ldo,
                                  const void *bwd_y, const float *b_scale, const float *b_shift, float *stats_partial,
                                  void *stream) {
  if (!cpfn_mlp_bwd_small_ok(P, N, K) || !Gy || !A || !W || !workspace || !Gout || ldg != N || (lda & 7) || lda < K || (ldo & 3) ||
      ldo < K || (!a_scale != !a_shift) || (bwd_y && (!b_scale || !b_shift || !stats_partial)))
    return CPFN_EINVAL;
  const int splits = cpfn_mlp_wgrad_splits(P, N, K);
  long long rps = (P + splits - 1) / splits;
  rps = ((rps + WG_STEP * WG_DEPTH - 1) / (WG_STEP * WG_DEPTH)) * (WG_STEP * WG_DEPTH);
  BwdSmallGrid gr;
  gr.wgx = N / 64; gr.wgy = (K + 63) / 64; gr.nW = gr.wgx * gr.wgy * splits;
  gr.dgx = cpfn_mlp_bwd_small_blocks(P);
  const dim3 grid(gr.nW + gr.dgx * (K / 64));
  SmallpBwdArgs bw;
  bw.Yb = (const unsigned short *)bwd_y; bw.b_scale = b_scale; bw.b_shift = b_shift;
  const int g_bytes = (int)(((P - 1) * N + N) * 2);
  hipStream_t st = (hipStream_t)stream;
  if (bwd_y)
    mlp_bwd_small_kernel<true><<<grid, 256, 0, st>>>(gr, (const unsigned short *)Gy, ldg, (const unsigned short *)A, lda, P, N, K, rps,
                                                     workspace, a_scale, a_shift, (const unsigned short *)W, g_bytes,
                                                     (unsigned short *)Gout, ldo, stats_partial, bw, probe_slot_all(grid));
  else
    mlp_bwd_small_kernel<false><<<grid, 256, 0, st>>>(gr, (const unsigned short *)Gy, ldg, (const unsigned short *)A, lda, P, N, K, rps,
                                                      workspace, a_scale, a_shift, (const unsigned short *)W, g_bytes,
                                                      (unsigned short *)Gout, ldo, nullptr, bw, probe_slot_all(grid));
  return cpfn_launch_status();
}

// the 64 x 64-tile weight gradient with cpfn_bn_bwd_apply folded in (the layers whose data gradient is cpfn_mlp_dgrad_small)
extern "C" int cpfn_mlp_wgrad_apply_ok(long long P, int N, int K) {
  int TN, TK;
  if (P <= 0 || N <= 0 || K < 8 || (N & 63) || (K & 31)) return 0;
  wgrad_tile(P, N, K, &TN, &TK);
  return TN == 64 && TK == 64;
}
