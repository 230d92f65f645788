// Per-point MLP stacks: the backward of a dense layer as ONE pass over its gradient (weight gradient + data gradient +
// the BatchNorm-backward reduction of the layer below).  See mlp_fwd.hip for the data layout.
#include "mlp_common.h"

#ifndef CPFN_BWD_COEF32
#define CPFN_BWD_COEF32 1
#endif

namespace {

// ---------------------------------------------------------------- weight gradient + data gradient in one pass
// A dense layer's backward reads its BatchNorm-adjoint gradient g_y twice: mlp_wgrad (g_y^T . A) and the data-gradient
// GEMM (g_y . W).  Here ONE kernel does both from one read: the g_y tile that the weight-gradient loop stages in LDS
// anyway (row-major, so its rows are also MFMA point fragments) is multiplied with the transposed weight panel (filled
// once per workgroup) and the STEP x TK slab of the data gradient leaves through an LDS patch one step later as 16-byte
// row-contiguous stores.  BST: as in the streaming GEMM, pass 1 of the BatchNorm backward of the layer BELOW (sum g_z,
// sum g_z.y) is taken from the slab being stored and the matching rows of that layer's pre-BN output, requested one step
// ahead.  APPLY: Gy is the gradient with respect to the layer's ACTIVATED output and the BatchNorm-backward apply pass
// (g_y = c0 . [scale . y + shift > 0] . g + c1 . y + c2, rounded to bf16 exactly as cpfn_bn_bwd_apply stores it) runs on
// the staged chunks from the layer's own pre-BN output Yr: g_y is never written to or read from memory.
// Shapes <TN, TK, STEP>: <128,128,32>, <256,128,32>, <64,64,64>, <128,64,64> (layer N -> channels of g_y, K -> channels of its input;
// STEP rows per step, 128 rows in flight).  Grid (1, 1, splits), the split layout of mlp_wgrad_kernel: same partials,
// bit for bit.
// EIGHT waves: for <128,128> a wave's share of the dW tile is 32 x 64 (32 accumulator registers) and a thread stages one
// 16-byte chunk per tensor and step, which keeps the kernel under 224 registers.  That matters inside the replayed step:
// a wave of > 256 registers cannot be placed on a SIMD that hosts a wave of the geometry branch, and the first version
// (4 waves, 304 registers) then ran in two rounds — 52-61 us instead of 36 (in-kernel probe, tools/dbg/probe_timeline.py).
struct BwdApplyArgs {                       // APPLY != 0: what forms g_y on the staged chunks
  const unsigned short *Yr;                  // the layer's own pre-BN output [P, TN]
  const float *coef, *y_scale, *y_shift;     // finalize coefficients [3][TN], forward scale / shift (ReLU mask)
  const unsigned long long *drop_seed;       // APPLY == 3: dropout on the layer's output (cpfn_bn_bwd_apply's)
  unsigned thresh16;
  float inv_keep;
  const unsigned char *pool_arg;             // APPLY == 2 (max-pooled layer): arg-max row of every (group, channel) ...
  const unsigned short *pool_yarg;           // ... the pre-BN value there; Gy is then the POOLED gradient [P / pool_k, TN]
  int pool_k;
  long long groups;
  const float *xt_xyz;                       // XT: [P,3] fp32 coordinates that are three more input channels of the layer
  float *xt_partial;                         //     [splits][TN][3]: split partials of their weight-gradient columns
  const int *g_idx;                          // XT + gather (round 6): the input rows are A[(p / g_rpc) * g_nsrc + g_idx[p]] — the
  int g_rpc, g_nsrc;                         //     grouped rows of sa2's first layer read out of the per-cloud table (mlp_fwd.hip, GatherIn)
  const float *xw_xyz;                       // XW: [P,3] fp32 coordinates, the ONLY input of the layer below (sa1's first layer) ...
  float *xw_partial;                         //     [splits][7][TK]: S1 = sum g_z x_j, S2 = sum y x_j (j < 3), S3 = sum x_j — see below
};

// (Round 2 also had an instantiation that RECOMPUTED sa1's first-layer output from the coordinates inside the 64 -> 64
//  shape instead of reading it: 134 MB fewer reads bought 2 us of 65 — the shape is bound by VALU + LDS issue — and it was
//  removed in round 3; the recompute lives on in cpfn_smallk_wgrad_apply_xyz, where it pays.)
// XT (sa2's first layer, see stream_tile): the layer has three more input channels, the fp32 coordinates xyz [P,3].  They need
// no data gradient (coordinates are inputs) and their weight-gradient columns dWx [TN,3] = g_y^T . xyz ride along: the
// 32 x 3 coordinate tile of a step is staged as bf16 beside the input tile and costs the four waves that own wk = 0 MI MFMAs
// more per step.
// XW (round 6; with BST): the layer BELOW is an fp32-xyz first layer (sa1: y = W0 . x, three coordinates in, no data gradient).
// Its weight gradient dW0[c][j] = sum_p g_y[p,c] x[p,j], g_y = c0 g_z + c1 y + c2, is linear in the BatchNorm coefficients — which
// do not exist yet when this kernel forms g_z — so the riding reduction also accumulates S1 = sum g_z x_j, S2 = sum y x_j and
// S3 = sum x_j per split; cpfn_multi_split_reduce's coefficient form (bn.hip) finishes c0 S1 + c1 S2 + c2 S3.  With Gout = NULL
// the slab is not stored either: the 67 MB gradient w.r.t. that layer's output and the 30 us launch that read it are gone.
template <int TN, int TK, int STEP, bool BST, int APPLY, bool XT = false, bool BDROP = false, bool XW = false>
__global__ __launch_bounds__(512) void mlp_bwd_fused_kernel(
    const unsigned short *__restrict__ Gy, int ldg, const unsigned short *__restrict__ A, int lda,
    const unsigned short *__restrict__ W /* forward weight panel [TN][TK] bf16 */, long long P, long long rows_per_split,
    float *__restrict__ partial, unsigned short *__restrict__ Gout, int ldo, const float *__restrict__ a_scale,
    const float *__restrict__ a_shift, const unsigned short *__restrict__ Yb, const float *__restrict__ b_scale,
    const float *__restrict__ b_shift, float *__restrict__ stats_partial, const BwdApplyArgs ap,
    unsigned long long *probe = nullptr) {
  const unsigned long long probe_t0 = probe_begin(probe);
  const unsigned short *__restrict__ Yr = ap.Yr;
  // BDROP (with BST, no apply pass: the fc2 heads): the layer BELOW ends in the fused dropout, i.e. the slab leaving here is the
  // gradient w.r.t. the DROPPED activation — the riding reduction scales it by the mask recomputed from the 8-byte seed, as
  // bn_relu_bwd_kernel<true> does (that launch, 15 us on the fc1 features, is then not made).
  static_assert(!BDROP || (BST && APPLY == 0), "the layer below's dropout belongs to the riding reduction of a linear layer");
  static_assert(!XW || (BST && !BDROP && !XT), "the xyz weight-gradient sums ride on the reduction of the layer below");
  const unsigned long long bdrop_seed = BDROP ? *ap.drop_seed : 0ull;
  // DEPTH = 2 steps of rows in flight for every shape (round 4; the 32-row-step shapes of 128 channels had four).  The riding
  // reduction's rows (yb below) are consumed a step after they are requested, and loads return in order: whatever was requested
  // before them — all but the newest prefetch — has to be there by then, so slots three and four never were in flight when it
  // mattered; without them the 128 -> 128 instantiation needs 164 registers instead of 206 and the step is 7 us faster (same box).
  // (rows in flight: 64 for the 32-row steps, 128 for the 64-row steps.  The 64-row-step shapes
  //  are NOT waiting for memory: taking 134 MB of the 64 -> 64 kernel's reads away (recomputing them) saved 2 us of 65, and 256 rows in
  //  flight instead of 128 made both of them 5 % slower (registers); their apply / statistics / conversion VALU work and
  //  LDS traffic per row are what a 64-channel row costs.)
  constexpr int NT = 512, LDN = TN + 8, LDK = TK + 8, DEPTH = 2, KSTEPS = STEP / 32;
  constexpr int CPRG = TN / 8, CPRA = TK / 8;                 // 16-byte chunks per row of the TN- / TK-wide tensors
  constexpr int NG = STEP * CPRG / NT;                        // g_y chunks per thread and step
  // TK-wide tensors (input, data-gradient slab, y of the layer below): RPA rows per pass, TA of the 512 threads busy
  // (TK = 192: 24 chunks per row -> 16 rows x 24 = 384 threads, two passes per 32-row step)
  constexpr int RPA = (NT / CPRA >= 64 ? 64 : NT / CPRA >= 32 ? 32 : 16) < STEP ? (NT / CPRA >= 64 ? 64 : NT / CPRA >= 32 ? 32 : 16) : STEP;
  constexpr int TA = RPA * CPRA, NA = STEP / RPA;
  constexpr int MI = TN / 64, MJ = TK / 32;                   // dW tiles per wave (waves 4 x 2 over TN x TK)
  constexpr int CHB = TK / 16, TPW = (STEP / 16) * CHB / 8;   // 16-channel blocks of the data-gradient slab, its tiles per wave
  static_assert(NG >= 1 && NA >= 1 && TA <= NT && (WG_STEP * WG_DEPTH) % (STEP * DEPTH) == 0 && (STEP / 16) * CHB == 8 * TPW, "shape");
  static_assert(!BST || (64 % CPRA == 0), "the riding reduction needs a power-of-two chunk count per row");
  // DB (the 64-row-step shapes: sa1, run after the geometry work, so their LDS footprint is free): the three row tiles are
  // double-buffered by step parity, which leaves ONE barrier per step
  // (stage -> barrier -> store previous slab / MFMAs) instead of two — the waves may drift a step apart and the VALU-heavy
  // staging of one overlaps the LDS / MFMA phase of another.
  // (round 4: the 128 -> 128 shapes too.  Round 2 measured nothing for them — their steps were then paced by the drained prefetch
  //  pipeline, §9 — and their 87-90 KB would not have fitted beside the inverse-index build's 80 KB of that time (64 KB now); with
  //  the drain gone: 34.5 -> 32.5 us stand-alone, -9 us per step.  The 256-channel shape stays at two barriers: 141 KB.)
  constexpr bool DB = STEP == 64 || (STEP == 32 && TN == 128 && TK == 128);
  static_assert(!DB || DEPTH % 2 == 0, "the buffer of a step is its pipeline slot's parity");
  __shared__ __attribute__((aligned(16))) unsigned short s_g2[DB ? 2 : 1][STEP * LDN];
  __shared__ __attribute__((aligned(16))) unsigned short s_a2[DB ? 2 : 1][STEP * LDK];
  __shared__ __attribute__((aligned(16))) unsigned short s_o2[DB ? 2 : 1][STEP * LDK];
  __shared__ __attribute__((aligned(16))) unsigned short s_wt[TK * LDN];
  constexpr int LDX = 16 + 8;
  __shared__ __attribute__((aligned(16))) unsigned short s_x2[XT ? (DB ? 2 : 1) * STEP * LDX : 8];
  static_assert(!XT || (STEP * 3 <= NT && MI >= 1), "xyz tail: one float per thread and step");
  f32x4 accx[XT ? MI : 1];
#pragma unroll
  for (int i = 0; i < (XT ? MI : 1); ++i) accx[i] = (f32x4){0, 0, 0, 0};
  float vxt[XT ? DEPTH : 1];
  static_assert(!BST || sizeof(float) * 8 * 2 * TK <= sizeof(unsigned short) * STEP * LDN, "the statistics reduction reuses s_g");
  float(*s_red)[2][TK] = (float(*)[2][TK])s_g2[0];  // cross-wave reduction of the statistics: after the last step only
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, lr = lane & 15, lq = lane >> 4;
  const long long p0 = (long long)blockIdx.z * rows_per_split, p1 = min(P, p0 + rows_per_split);
  float *o = partial + (size_t)blockIdx.z * TN * TK;
  if (XT) {      // columns 3..15 of the coordinate tile stay zero
    for (int e = t; e < (DB ? 2 : 1) * STEP * LDX; e += NT) s_x2[e] = 0;
    __syncthreads();
  }
  if (p0 >= p1) {   // empty split: its partial slab (and its statistics row) must still be zero
    for (int e = t; e < TN * TK; e += NT) o[e] = 0.f;
    if (XT) for (int e = t; e < TN * 3; e += NT) ap.xt_partial[(size_t)blockIdx.z * TN * 3 + e] = 0.f;
    if (BST) for (int e = t; e < 2 * TK; e += NT) stats_partial[(size_t)blockIdx.z * 2 * TK + e] = 0.f;
    if (XW) for (int e = t; e < 7 * TK; e += NT) ap.xw_partial[(size_t)blockIdx.z * 7 * TK + e] = 0.f;
    probe_end(probe, probe_t0, 5);
    return;
  }
  const int wn = (wave >> 1) * (TN / 4), wk = (wave & 1) * (TK / 2);
  f32x4 acc[MI][MJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < MJ; ++j) acc[i][j] = (f32x4){0, 0, 0, 0};
  // chunk i of thread t: row t / CPR + i (NT / CPR), columns 8 (t % CPR) — a thread's columns never change
  const int grow = t / CPRG, gcol = (t % CPRG) * 8, arow = (TA == NT ? t : t % TA) / CPRA, acol = (t % CPRA) * 8;
  const bool a_live = TA == NT || t < TA;
  uint4 vg[APPLY == 2 ? 1 : DEPTH][NG], va[DEPTH][NA], vy[APPLY ? DEPTH : 1][NG];
  uint4 vgp[APPLY == 2 ? DEPTH : 1], vya[APPLY == 2 ? DEPTH : 1];       // pooled gradient / arg-max value of the step's group
  uint2 var_[APPLY == 2 ? DEPTH : 1];                                    // arg-max row (8 channels, one byte each)
  // (pooled: a step never straddles two groups — the host checks pool_k % STEP == 0)
  float asc[8], ash[8];
  if (a_scale) {
#pragma unroll
    for (int j = 0; j < 8; ++j) { asc[j] = a_scale[acol + j]; ash[j] = a_shift[acol + j]; }
  }
  // APPLY: the five per-channel vectors wait in LDS and are re-read at every stage (40 registers otherwise: with them
  // the <128,128> kernel needs 237, and two such waves plus a 56-register wave of the geometry branch do not fit one SIMD)
  // (64-row-step shapes — sa1, after the geometry work has ended, bound by VALU + LDS issue rather than memory — keep them
  //  in registers: 10 LDS reads per stage less)
  // (round 4: with two steps of rows in flight the dense 128 -> 128 apply instantiation has the registers too — 164 + 40 —; its
  //  vectors go through LDS once, COEF_VIA_LDS: forty 4-byte global loads per lane in the prologue cost more than they saved;
  //  fc1's dropout instantiation too: 214 registers, -1 us)
  constexpr bool COEF_VIA_LDS = CPFN_BWD_COEF32 && (APPLY == 1 || APPLY == 3) && STEP == 32 && TN <= 128;
  constexpr bool COEF_REGS = APPLY != 0 && (STEP == 64 || COEF_VIA_LDS);
  __shared__ __attribute__((aligned(16))) float s_cf[APPLY && (!COEF_REGS || COEF_VIA_LDS) ? 5 * TN : 4];
  float cfr[COEF_REGS ? 5 : 1][8];
  if (COEF_VIA_LDS) {
    for (int e = t; e < 3 * TN; e += NT) s_cf[e] = ap.coef[e];
    if (t < TN) { s_cf[3 * TN + t] = ap.y_scale[t]; s_cf[4 * TN + t] = ap.y_shift[t]; }
  } else if (COEF_REGS) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      cfr[0][j] = ap.coef[gcol + j]; cfr[1][j] = ap.coef[TN + gcol + j]; cfr[2][j] = ap.coef[2 * TN + gcol + j];
      cfr[COEF_REGS ? 3 : 0][j] = ap.y_scale[gcol + j]; cfr[COEF_REGS ? 4 : 0][j] = ap.y_shift[gcol + j];
    }
  } else if (APPLY) {
    for (int e = t; e < 3 * TN; e += NT) s_cf[e] = ap.coef[e];
    if (t < TN) { s_cf[3 * TN + t] = ap.y_scale[t]; s_cf[4 * TN + t] = ap.y_shift[t]; }
    // (visible after the first barrier of the step loop — two-barrier shapes; the one-barrier shapes STAGE before their first
    //  barrier and get one of their own below)
  }
  float bsc[8], bsh[8], st_s[8], st_q[8];
  if (BST) {
#pragma unroll
    for (int j = 0; j < 8; ++j) { bsc[j] = b_scale[acol + j]; bsh[j] = b_shift[acol + j]; st_s[j] = 0.f; st_q[j] = 0.f; }
  }
  float xw1[XW ? 8 : 1][3], xw2[XW ? 8 : 1][3], xw3[3] = {0.f, 0.f, 0.f};      // XW: the lane's shares of S1, S2 (its 8 channels), S3
  if (XW) {
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int q = 0; q < 3; ++q) { xw1[j][q] = 0.f; xw2[j][q] = 0.f; }
  }
  int gnx[XT ? NA : 1];        // XT + gather: table rows of the NEXT issue's input rows (consecutive issues take consecutive steps)
  if (XT && ap.g_idx) {
#pragma unroll
    for (int i = 0; i < NA; ++i) gnx[XT ? i : 0] = ap.g_idx[min(p0 + arow + i * RPA, p1 - 1)];
  }
  auto issue = [&](int sidx, long long base) {
    if (APPLY == 2) {
      const long long grp = min(base / ap.pool_k, ap.groups - 1);
      vgp[sidx] = *(const uint4 *)(Gy + grp * TN + gcol);
      vya[sidx] = *(const uint4 *)(ap.pool_yarg + grp * TN + gcol);
      var_[sidx] = *(const uint2 *)(ap.pool_arg + grp * TN + gcol);
    }
#pragma unroll
    for (int i = 0; i < NG; ++i) {
      const long long p = min(base + grow + i * (NT / CPRG), p1 - 1);      // clamped: always a valid row, zeroed at store time
      if (APPLY != 2) vg[sidx][i] = *(const uint4 *)(Gy + p * ldg + gcol);
      if (APPLY) vy[sidx][i] = *(const uint4 *)(Yr + p * ldg + gcol);
    }
    if (a_live) {
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        const long long p = min(base + arow + i * RPA, p1 - 1);
        if (XT && ap.g_idx) {
          const long long row = (long long)((int)p / ap.g_rpc) * ap.g_nsrc + gnx[XT ? i : 0];
          va[sidx][i] = *(const uint4 *)(A + row * lda + acol);
          gnx[XT ? i : 0] = ap.g_idx[min(base + STEP + arow + i * RPA, p1 - 1)];      // for the next issue (base + STEP)
        } else {
          va[sidx][i] = *(const uint4 *)(A + p * lda + acol);
        }
      }
    }
    if (XT && t < STEP * 3) {              // float t of the step's contiguous 32 x 3 block (rows past the split's end: clamped, zeroed at stage time)
      const long long p = min(base + t / 3, p1 - 1);
      vxt[XT ? sidx : 0] = ap.xt_xyz[p * 3 + t % 3];
    }
  };
  auto stage = [&](int sidx, long long base, int buf) {
    unsigned short *s_g = s_g2[buf], *s_a = s_a2[buf];
    float cf0[8], cf1[8], cf2[8], ysc[8], ysh[8];
    if (APPLY && COEF_REGS) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { cf0[j] = cfr[0][j]; cf1[j] = cfr[1][j]; cf2[j] = cfr[2][j]; ysc[j] = cfr[3][j]; ysh[j] = cfr[4][j]; }
    } else if (APPLY) {
      int zero;                                        // opaque 0: keeps these loop-invariant reads INSIDE the loop
      asm volatile("v_mov_b32 %0, 0" : "=v"(zero));
      const float *cp = &s_cf[gcol + zero];
      *(float4 *)&cf0[0] = *(const float4 *)&cp[0];          *(float4 *)&cf0[4] = *(const float4 *)&cp[4];
      *(float4 *)&cf1[0] = *(const float4 *)&cp[TN];         *(float4 *)&cf1[4] = *(const float4 *)&cp[TN + 4];
      *(float4 *)&cf2[0] = *(const float4 *)&cp[2 * TN];     *(float4 *)&cf2[4] = *(const float4 *)&cp[2 * TN + 4];
      *(float4 *)&ysc[0] = *(const float4 *)&cp[3 * TN];     *(float4 *)&ysc[4] = *(const float4 *)&cp[3 * TN + 4];
      *(float4 *)&ysh[0] = *(const float4 *)&cp[4 * TN];     *(float4 *)&ysh[4] = *(const float4 *)&cp[4 * TN + 4];
    }
    float pz[8];          // APPLY == 2: masked pooled gradient and arg-max row of this step's group, per channel
    int pk[8];
    if (APPLY == 2) {
      const unsigned gw[4] = {vgp[sidx].x, vgp[sidx].y, vgp[sidx].z, vgp[sidx].w};
      const unsigned yw[4] = {vya[sidx].x, vya[sidx].y, vya[sidx].z, vya[sidx].w};
      const unsigned aw[2] = {var_[sidx].x, var_[sidx].y};
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float ya = __uint_as_float((j & 1) ? (yw[j >> 1] & 0xffff0000u) : (yw[j >> 1] << 16));
        const float gp = __uint_as_float((j & 1) ? (gw[j >> 1] & 0xffff0000u) : (gw[j >> 1] << 16));
        pz[j] = fmaf(ysc[j], ya, ysh[j]) > 0.f ? gp : 0.f;
        pk[j] = (int)((aw[j >> 2] >> (8 * (j & 3))) & 0xffu);
      }
    }
    const int kbase = APPLY == 2 ? (int)(base % ap.pool_k) : 0;
#pragma unroll
    for (int i = 0; i < NG; ++i) {
      const int r = grow + i * (NT / CPRG);
      uint4 g4 = APPLY == 2 ? (uint4){0, 0, 0, 0} : vg[sidx][i];
      if (APPLY & 1) {     // (the arithmetic of bn_bwd_apply_kernel<true>, element for element; APPLY == 3: with dropout)
        const uint4 y4 = vy[sidx][i];
        const unsigned gw[4] = {g4.x, g4.y, g4.z, g4.w}, yw[4] = {y4.x, y4.y, y4.z, y4.w};
        unsigned ow[4];
        float f[8];
        if (APPLY == 3)
          dropout_factors(*ap.drop_seed, (unsigned long long)(min(base + r, p1 - 1) * CPRG + (gcol >> 3)), ap.thresh16, ap.inv_keep, f);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float y0 = __uint_as_float(yw[j] << 16), y1 = __uint_as_float(yw[j] & 0xffff0000u);
          float z0 = __uint_as_float(gw[j] << 16), z1 = __uint_as_float(gw[j] & 0xffff0000u);
          if (APPLY == 3) { z0 *= f[2 * j]; z1 *= f[2 * j + 1]; }
          z0 = fmaf(ysc[2 * j], y0, ysh[2 * j]) > 0.f ? z0 : 0.f;
          z1 = fmaf(ysc[2 * j + 1], y1, ysh[2 * j + 1]) > 0.f ? z1 : 0.f;
          const unsigned lo = f2bf(fmaf(cf0[2 * j], z0, fmaf(cf1[2 * j], y0, cf2[2 * j])));
          const unsigned hi = f2bf(fmaf(cf0[2 * j + 1], z1, fmaf(cf1[2 * j + 1], y1, cf2[2 * j + 1])));
          ow[j] = lo | (hi << 16);
        }
        g4 = (uint4){ow[0], ow[1], ow[2], ow[3]};
      }
      if (APPLY == 2) {     // (the arithmetic of bn_pool_bwd_apply_kernel, element for element)
        const uint4 y4 = vy[sidx][i];
        const unsigned yw[4] = {y4.x, y4.y, y4.z, y4.w};
        const int k = kbase + r;
        unsigned ow[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float y0 = __uint_as_float(yw[j] << 16), y1 = __uint_as_float(yw[j] & 0xffff0000u);
          const unsigned lo = f2bf(fmaf(cf0[2 * j], pk[2 * j] == k ? pz[2 * j] : 0.f, fmaf(cf1[2 * j], y0, cf2[2 * j])));
          const unsigned hi = f2bf(fmaf(cf0[2 * j + 1], pk[2 * j + 1] == k ? pz[2 * j + 1] : 0.f, fmaf(cf1[2 * j + 1], y1, cf2[2 * j + 1])));
          ow[j] = lo | (hi << 16);
        }
        g4 = (uint4){ow[0], ow[1], ow[2], ow[3]};
      }
      if (base + r >= p1) g4 = (uint4){0, 0, 0, 0};
      *(uint4 *)&s_g[r * LDN + gcol] = g4;
    }
    if (a_live) {
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        const int r = arow + i * RPA;
        uint4 a4 = va[sidx][i];
        if (a_scale) a4 = __builtin_bit_cast(uint4, bn_relu_frag(__builtin_bit_cast(bf16x8, a4), asc, ash));
        if (base + r >= p1) a4 = (uint4){0, 0, 0, 0};
        *(uint4 *)&s_a[r * LDK + acol] = a4;
      }
    }
    if (XT && t < STEP * 3) {
      const int r = t / 3;
      s_x2[(DB ? buf : 0) * STEP * LDX + r * LDX + t % 3] = base + r >= p1 ? (unsigned short)0 : f2bf(vxt[XT ? sidx : 0]);
    }
  };
  // the STEP x TK data-gradient slab of the PREVIOUS step leaves here (its LDS patch was completed before this step's
  // first barrier): NA 16-byte pieces per thread
  // yb: the rows of the layer below's pre-BN output that match the slab of step d, which leaves at step d + 1.  They are requested
  // IN FRONT of step d's prefetch.  (Until round 4 they were requested
  // BEHIND the prefetch of step d + DEPTH: memory returns a wave's loads in order, so waiting for them half a step later drained
  // the whole prefetch pipeline once per step — the four steps of rows in flight had half a step to arrive.  An ablation
  // without the riding reduction ran 37 % faster for 19 % fewer bytes.  A second register set, requested a whole step ahead, is
  // 6 % faster still when the kernel runs alone on HBM-resident operands and LOSES inside the step: 4-8 registers more per wave
  // are what a one-pass launch cannot afford beside the geometry work)
  uint4 yb[NA];
  float xr[XW ? NA : 1][3];          // XW: the coordinates of the same rows
  auto issue_yb = [&](long long base) {
    if (BST) {
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        const long long pr = min(base + arow + i * RPA, p1 - 1);
        yb[i] = *(const uint4 *)(Yb + pr * ldo + acol);
        if (XW) {
#pragma unroll
          for (int q = 0; q < 3; ++q) xr[XW ? i : 0][q] = ap.xw_xyz[pr * 3 + q];
        }
      }
    }
  };
  auto store_prev = [&](long long pbase, int buf) {
    const unsigned short *s_o = s_o2[buf];
    if (!a_live) return;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int r = arow + i * RPA;
      const long long p = pbase + r;
      const uint4 v = *(const uint4 *)&s_o[r * LDK + acol];
      if (p < p1) {
        if (!XW || Gout) {   // (`nt` store: see stream_tile in mlp_fwd.hip; XW with Gout = NULL: nobody reads this gradient)
          typedef __attribute__((ext_vector_type(4))) unsigned u32x4nt;
          __builtin_nontemporal_store((u32x4nt){v.x, v.y, v.z, v.w}, (u32x4nt *)(Gout + p * ldo + acol));
        }
        if (BST) {
          const uint4 ybv = yb[i];
          const unsigned g4[4] = {v.x, v.y, v.z, v.w}, y4[4] = {ybv.x, ybv.y, ybv.z, ybv.w};
          float df[8];
          if (BDROP) dropout_factors(bdrop_seed, (unsigned long long)((p * ldo + acol) >> 3), ap.thresh16, ap.inv_keep, df);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float g0 = __uint_as_float(g4[j] << 16), g1 = __uint_as_float(g4[j] & 0xffff0000u);
            if (BDROP) { g0 *= df[2 * j]; g1 *= df[2 * j + 1]; }
            const float y0 = __uint_as_float(y4[j] << 16), y1 = __uint_as_float(y4[j] & 0xffff0000u);
            const float z0 = fmaf(bsc[2 * j], y0, bsh[2 * j]) > 0.f ? g0 : 0.f;
            const float z1 = fmaf(bsc[2 * j + 1], y1, bsh[2 * j + 1]) > 0.f ? g1 : 0.f;
            st_s[2 * j] += z0; st_s[2 * j + 1] += z1;
            st_q[2 * j] = fmaf(z0, y0, st_q[2 * j]); st_q[2 * j + 1] = fmaf(z1, y1, st_q[2 * j + 1]);
            if (XW) {
#pragma unroll
              for (int q = 0; q < 3; ++q) {
                const float xq = xr[XW ? i : 0][q];
                xw1[XW ? 2 * j : 0][q] = fmaf(z0, xq, xw1[XW ? 2 * j : 0][q]);
                xw1[XW ? 2 * j + 1 : 0][q] = fmaf(z1, xq, xw1[XW ? 2 * j + 1 : 0][q]);
                xw2[XW ? 2 * j : 0][q] = fmaf(y0, xq, xw2[XW ? 2 * j : 0][q]);
                xw2[XW ? 2 * j + 1 : 0][q] = fmaf(y1, xq, xw2[XW ? 2 * j + 1 : 0][q]);
              }
            }
          }
          if (XW && acol == 0) {           // (every row once: the lane of its first chunk)
#pragma unroll
            for (int q = 0; q < 3; ++q) xw3[q] += xr[XW ? i : 0][q];
          }
        }
      }
    }
  };
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) issue(d, p0 + (long long)d * STEP);
  fill_w_panel<TK, LDN, NT>(s_wt, W, TN, TK, 0, 0, TN, 1, t);      // s_wt[k_out][n]: the forward weight [n][k] transposed
  // One-barrier (DB) shapes whose apply pass reads its per-channel vectors from LDS at every stage — fc1's dropout instantiation and
  // the pooled 128 -> 128 one since the 128-channel shapes are double-buffered — stage step 0 BEFORE the loop's first barrier: the
  // vectors written above by other lanes must be visible first.  (Found by tests/test_gpu_fused_mlp.py failing once in ~15 runs
  // after the shapes became one-barrier: a first stage that read the vectors early.)
  if (DB && APPLY && !COEF_REGS) __syncthreads();
  if (COEF_VIA_LDS) {
    __syncthreads();
#pragma unroll
    for (int v5 = 0; v5 < 5; ++v5) {
      *(float4 *)&cfr[COEF_REGS ? v5 : 0][0] = *(const float4 *)&s_cf[v5 * TN + gcol];
      *(float4 *)&cfr[COEF_REGS ? v5 : 0][4] = *(const float4 *)&s_cf[v5 * TN + gcol + 4];
    }
  }
  long long prev = -1;
  for (long long base0 = p0; base0 < p1; base0 += STEP * DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const long long base = base0 + (long long)d * STEP;
      constexpr int dummy_ = 0; (void)dummy_;
      const int buf = DB ? (d & 1) : 0;
      unsigned short *s_g = s_g2[buf], *s_a = s_a2[buf], *s_o = s_o2[buf];
      if (DB) {
        // ONE barrier per step.  Why that orders every LDS access (tests/test_gpu_concurrency.py replays these shapes
        // hundreds of times beside the geometry graph and compares bit for bit): step d writes the row tiles of parity
        // d & 1, the barrier, reads the slab patch of parity (d - 1) & 1 (completed by every wave BEFORE it arrived at this
        // barrier), then reads the row tiles of parity d & 1 and writes the patch of parity d & 1.  A wave can only be
        // one barrier ahead of the slowest one, i.e. staging step d + 1 into the OTHER parity while the slowest still
        // reads step d's tiles; the tiles / patch of parity d & 1 are written again at step d + 2, behind barrier d + 1,
        // which the slowest wave only reaches after all its reads of step d.
        stage(d, base, buf);
        __syncthreads();
        if (prev >= 0) store_prev(prev, buf ^ 1);
      } else {
        __syncthreads();
        if (prev >= 0) store_prev(prev, DB ? buf ^ 1 : 0);
        stage(d, base, buf);
        __syncthreads();
      }
      // (requested HERE, not right behind store_prev a staging phase earlier: the longer live range costs registers, and the
      //  step beside the geometry work 30 us — same-box three-way comparison)
      issue_yb(base);
      issue(d, base + STEP * DEPTH);
      // ---- weight gradient: transposed fragments of both tiles (as mlp_wgrad_kernel), 32 rows per MFMA
#pragma unroll
      for (int kk = 0; kk < KSTEPS; ++kk) {
        bf16x8 fg[MI], fa[MJ];
#pragma unroll
        for (int i = 0; i < MI; ++i) fg[i] = tr_frag<LDN>(s_g + kk * 32 * LDN, wn + 16 * i, lane);
#pragma unroll
        for (int j = 0; j < MJ; ++j) fa[j] = tr_frag<LDK>(s_a + kk * 32 * LDK, wk + 16 * j, lane);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < MJ; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fg[i], fa[j], acc[i][j], 0, 0, 0);
        if (XT && wk == 0) {         // (wave-uniform) the coordinate columns of the weight gradient
          const bf16x8 fx = tr_frag<LDX>(s_x2 + (DB ? buf : 0) * STEP * LDX + kk * 32 * LDX, 0, lane);
#pragma unroll
          for (int i = 0; i < MI; ++i) accx[XT ? i : 0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fg[i], fx, accx[XT ? i : 0], 0, 0, 0);
        }
      }
      if constexpr (CHB == 8 || CHB == 4) {
        // ---- data gradient of the same rows: this wave's output channels 16 cb .. +15 of rows 16 rb0 .. +31 (two tiles
        //      that share the weight fragment)
        //      (the 8 waves cover CHB column blocks x 8 / CHB row-tile pairs per trip: one trip for 32 x 128 and 64 x 64,
        //       two for the heads' 64-row step of 128 channels)
        const int cb = wave % CHB;
#pragma unroll
        for (int rb0 = (wave / CHB) * 2; rb0 < STEP / 16; rb0 += (8 / CHB) * 2) {
          f32x4 ad[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll
          for (int ks = 0; ks < TN / 32; ++ks) {
            const bf16x8 pf0 = *(const bf16x8 *)&s_g[(rb0 * 16 + lr) * LDN + ks * 32 + 8 * lq];
            const bf16x8 pf1 = *(const bf16x8 *)&s_g[(rb0 * 16 + 16 + lr) * LDN + ks * 32 + 8 * lq];
            const bf16x8 wf = *(const bf16x8 *)&s_wt[(cb * 16 + lr) * LDN + ks * 32 + 8 * lq];
            ad[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, pf0, ad[0], 0, 0, 0);
            ad[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, pf1, ad[1], 0, 0, 0);
          }
#pragma unroll
          for (int tt = 0; tt < 2; ++tt) {
            const f32x4 v = ad[tt];
            const bf16x4 ov = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
            *(bf16x4 *)&s_o[((rb0 + tt) * 16 + lr) * LDK + cb * 16 + 4 * lq] = ov;
          }
        }
      } else {
        // ---- (TK = 192: 2 x 12 tiles of 16 x 16) tile q = wave + 8 i of the slab
        f32x4 ad[TPW];
#pragma unroll
        for (int i = 0; i < TPW; ++i) ad[i] = (f32x4){0, 0, 0, 0};
#pragma unroll
        for (int ks = 0; ks < TN / 32; ++ks) {
#pragma unroll
          for (int i = 0; i < TPW; ++i) {
            const int q = wave + 8 * i, cb = q % CHB, rb = q / CHB;
            const bf16x8 pf = *(const bf16x8 *)&s_g[(rb * 16 + lr) * LDN + ks * 32 + 8 * lq];
            const bf16x8 wf = *(const bf16x8 *)&s_wt[(cb * 16 + lr) * LDN + ks * 32 + 8 * lq];
            ad[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, pf, ad[i], 0, 0, 0);
          }
        }
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
          const int q = wave + 8 * i, cb = q % CHB, rb = q / CHB;
          const f32x4 v = ad[i];
          const bf16x4 ov = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
          *(bf16x4 *)&s_o[(rb * 16 + lr) * LDK + cb * 16 + 4 * lq] = ov;
        }
      }
      prev = base;
    }
  }
  __syncthreads();
  store_prev(prev, DB ? (DEPTH - 1) & 1 : 0);
  // D[row = n-local 4(lane>>4)+r][col = k-local lane&15]
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < MJ; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        o[(size_t)(wn + i * 16 + 4 * (lane >> 4) + r) * TK + wk + j * 16 + (lane & 15)] = acc[i][j][r];
  if (XT && wk == 0 && (lane & 15) < 3) {
    float *ox = ap.xt_partial + (size_t)blockIdx.z * TN * 3;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) ox[(size_t)(wn + i * 16 + 4 * (lane >> 4) + r) * 3 + (lane & 15)] = accx[XT ? i : 0][r];
  }
  if (BST) {   // threads that share a column chunk (t % CPRA): shuffles inside the wave, then the eight waves through LDS
#pragma unroll
    for (int j = 0; j < 8; ++j) {
#pragma unroll
      for (int m = CPRA; m < 64; m <<= 1) { st_s[j] += __shfl_xor(st_s[j], m, 64); st_q[j] += __shfl_xor(st_q[j], m, 64); }
    }
    if (lane < CPRA) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { s_red[wave][0][lane * 8 + j] = st_s[j]; s_red[wave][1][lane * 8 + j] = st_q[j]; }
    }
    __syncthreads();
    for (int e = t; e < 2 * TK; e += NT) {
      const int which = e / TK, c = e - which * TK;
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) v += s_red[w][which][c];
      stats_partial[((size_t)blockIdx.z * 2 + which) * TK + c] = v;
    }
  }
  if (XW) {
    // the same two-level sum for S1, S2 (3 x TK values each) and S3 (3 values), one after the other through the same LDS patch
    static_assert(!XW || sizeof(float) * 8 * 3 * TK <= sizeof(unsigned short) * STEP * LDN, "the XW reduction reuses s_g");
    float(*s_x)[3][TK] = (float(*)[3][TK])s_g2[0];
    float *xo = ap.xw_partial + (size_t)blockIdx.z * 7 * TK;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      float (&xs)[XW ? 8 : 1][3] = pass == 0 ? xw1 : xw2;
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
          for (int m = CPRA; m < 64; m <<= 1) xs[XW ? j : 0][q] += __shfl_xor(xs[XW ? j : 0][q], m, 64);
      __syncthreads();
      if (lane < CPRA) {
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
          for (int q = 0; q < 3; ++q) s_x[wave][q][lane * 8 + j] = xs[XW ? j : 0][q];
      }
      __syncthreads();
      for (int e = t; e < 3 * TK; e += NT) {
        const int q = e / TK, c = e - q * TK;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) v += s_x[w][q][c];
        xo[(pass * 3 + q) * TK + c] = v;
      }
    }
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
      for (int m = CPRA; m < 64; m <<= 1) xw3[q] += __shfl_xor(xw3[q], m, 64);
    __syncthreads();
    if (lane == 0) {
#pragma unroll
      for (int q = 0; q < 3; ++q) s_x[wave][q][0] = xw3[q];      // (one word per row of the patch: three adjacent words would be a 96-bit DS write, banned)
    }
    __syncthreads();
    if (t < TK) {
      float v = 0.f;
      if (t < 3) {
#pragma unroll
        for (int w = 0; w < 8; ++w) v += s_x[w][t][0];
      }
      xo[6 * TK + t] = v;
    }
  }
  probe_end(probe, probe_t0, 5);
}


}  // namespace

// ============================================================================ C ABI

extern "C" int cpfn_mlp_bwd_fused_ok(long long P, int N, int K) {
  const bool shape = (N == 128 && K == 128) || (N == 64 && K == 64) || (N == 128 && K == 64) || (N == 256 && K == 128) ||
                     (N == 64 && K == 128);      // (64 <- 128: the fc2 heads, padded to 64 outputs; linear: no apply pass)
  return shape && P > SP_MAX_ROWS && P >= 32768;
}

static int bwd_fused_launch(const void *Gy, int ldg, const void *A, int lda, const void *W, long long P, int N, int K,
                            const float *a_scale, const float *a_shift, float *workspace, void *Gout, int ldo,
                            const void *bwd_y, const float *b_scale, const float *b_shift, float *stats_partial,
                            const void *apply_y, const float *apply_coef, const float *y_scale, const float *y_shift,
                            const unsigned long long *drop_seed, float drop_p, const unsigned char *pool_arg,
                            const void *pool_yarg, int pool_k, const float *xt_xyz,
                            float *xt_partial, const float *xw_xyz, float *xw_partial, void *stream,
                            const int *g_idx = nullptr, int g_rpc = 0, int g_nsrc = 0) {
  if (g_idx && (!xt_xyz || g_rpc <= 0 || g_nsrc <= 0 || P % g_rpc)) return CPFN_EINVAL;
  // xw (the xyz weight-gradient sums of the layer below riding on its reduction): the 64 <- 64 shape with the dense apply pass,
  // sa1's second layer; Gout may then be NULL (nobody needs the gradient w.r.t. the first layer's output)
  if ((!xw_xyz) != (!xw_partial)) return CPFN_EINVAL;
  if (xw_xyz && !(N == 64 && K == 64 && bwd_y && apply_y && !drop_seed && pool_k == 0 && !xt_xyz)) return CPFN_EINVAL;
  if (!cpfn_mlp_bwd_fused_ok(P, N, K) || !Gy || !A || !W || !workspace || (!Gout && !xw_xyz) || (ldg & 7) || (lda & 7) || (ldo & 7) ||
      ldg < N || lda < K || ldo < K || (!a_scale != !a_shift) || (bwd_y && (!b_scale || !b_shift || !stats_partial)) ||
      (apply_y && (!apply_coef || !y_scale || !y_shift)))
    return CPFN_EINVAL;
  if (apply_y && ldg != N) return CPFN_EINVAL;     // (the kernel walks apply_y with the gradient's row stride)
  // drop_seed: with apply_y the layer's OWN fused dropout (fc1); without it — the 64 <- 128 heads shape only, with bwd_y — the
  // dropout of the layer BELOW, whose output feeds this linear layer (its mask scales the riding reduction)
  const bool below_drop = drop_seed && !apply_y && bwd_y && N == 64 && K == 128;
  if ((drop_seed && !below_drop && (!apply_y || pool_k > 0)) || (drop_seed && !(drop_p >= 0.f && drop_p < 1.f)) || pool_k < 0)
    return CPFN_EINVAL;
  const int step = (K >= 128 && N != 64) ? 32 : 64;
  // xyz tail (three fp32 coordinate channels beside the K bf16 ones; sa2's first layer): the 128 -> 128 shape with the dense
  // apply pass and no layer below
  if ((!xt_xyz) != (!xt_partial)) return CPFN_EINVAL;
  if (xt_xyz && !(N == 128 && K == 128 && apply_y && !bwd_y && !drop_seed && pool_k == 0)) return CPFN_EINVAL;
  if (pool_k > 0 && (!apply_y || !pool_arg || !pool_yarg || pool_k > 255 || pool_k % step || P % pool_k || ldg != N))
    return CPFN_EINVAL;
  const int splits = cpfn_mlp_wgrad_splits(P, N, K);
  long long rps = (P + splits - 1) / splits;
  rps = ((rps + WG_STEP * WG_DEPTH - 1) / (WG_STEP * WG_DEPTH)) * (WG_STEP * WG_DEPTH);
  const dim3 grid(1, 1, splits);
  hipStream_t st = (hipStream_t)stream;
  const unsigned short *g = (const unsigned short *)Gy, *a = (const unsigned short *)A, *w = (const unsigned short *)W,
                       *yb = (const unsigned short *)bwd_y;
  unsigned short *go = (unsigned short *)Gout;
  BwdApplyArgs ap;
  ap.Yr = (const unsigned short *)apply_y; ap.coef = apply_coef; ap.y_scale = y_scale; ap.y_shift = y_shift;
  ap.drop_seed = drop_seed; ap.thresh16 = dropout_thresh16(drop_seed ? drop_p : 0.f); ap.inv_keep = drop_seed ? 1.f / (1.f - drop_p) : 1.f;
  ap.pool_arg = pool_arg; ap.pool_yarg = (const unsigned short *)pool_yarg; ap.pool_k = pool_k > 0 ? pool_k : 1;
  ap.groups = pool_k > 0 ? P / pool_k : 1;
  ap.xt_xyz = xt_xyz; ap.xt_partial = xt_partial;
  ap.xw_xyz = xw_xyz; ap.xw_partial = xw_partial;
  ap.g_idx = g_idx; ap.g_rpc = g_rpc; ap.g_nsrc = g_nsrc;
  const int mode = !apply_y ? 0 : (pool_k > 0 ? 2 : (drop_seed ? 3 : 1));
#define CPFN_BWD_FUSED(TN_, TK_, STEP_, BST_, APPLY_)                                                                      \
  mlp_bwd_fused_kernel<TN_, TK_, STEP_, BST_, APPLY_><<<grid, 512, 0, st>>>(g, ldg, a, lda, w, P, rps, workspace, go, ldo, \
                                                                            a_scale, a_shift, yb, b_scale, b_shift,       \
                                                                            stats_partial, ap, probe_slot_all(grid))
  // (mode 3 — the layer's own fused dropout — is instantiated for the 128 -> 128 shape only: fc1 is the one layer that ends in it)
  if (mode == 3 && !(N == 128 && K == 128)) return CPFN_EINVAL;
#define CPFN_BWD_FUSED_SHAPE(TN_, TK_, STEP_)                                  \
  do {                                                                         \
    if (bwd_y) {                                                               \
      if (mode == 2) CPFN_BWD_FUSED(TN_, TK_, STEP_, true, 2);                 \
      else if (mode == 1) CPFN_BWD_FUSED(TN_, TK_, STEP_, true, 1);            \
      else if (mode == 3) { if constexpr (TN_ == 128 && TK_ == 128) CPFN_BWD_FUSED(TN_, TK_, STEP_, true, 3); }  \
      else CPFN_BWD_FUSED(TN_, TK_, STEP_, true, 0);                           \
    } else {                                                                   \
      if (mode == 2) CPFN_BWD_FUSED(TN_, TK_, STEP_, false, 2);                \
      else if (mode == 1) CPFN_BWD_FUSED(TN_, TK_, STEP_, false, 1);           \
      else if (mode == 3) { if constexpr (TN_ == 128 && TK_ == 128) CPFN_BWD_FUSED(TN_, TK_, STEP_, false, 3); } \
      else CPFN_BWD_FUSED(TN_, TK_, STEP_, false, 0);                          \
    }                                                                          \
  } while (0)
  if (xw_xyz)
    mlp_bwd_fused_kernel<64, 64, 64, true, 1, false, false, true><<<grid, 512, 0, st>>>(g, ldg, a, lda, w, P, rps, workspace, go, ldo,
                                                                                        a_scale, a_shift, yb, b_scale, b_shift,
                                                                                        stats_partial, ap, probe_slot_all(grid));
  else if (xt_xyz)
    mlp_bwd_fused_kernel<128, 128, 32, false, 1, true><<<grid, 512, 0, st>>>(g, ldg, a, lda, w, P, rps, workspace, go, ldo, a_scale,
                                                                             a_shift, yb, b_scale, b_shift, stats_partial, ap,
                                                                             probe_slot_all(grid));
  else if (N == 128 && K == 128) CPFN_BWD_FUSED_SHAPE(128, 128, 32);
  else if (N == 256) CPFN_BWD_FUSED_SHAPE(256, 128, 32);
  else if (N == 64 && K == 128) {
    if (mode != 0 || (bwd_y && !below_drop)) return CPFN_EINVAL;     // (heads: linear; what may ride is fc1's dropped reduction)
    if (bwd_y)
      mlp_bwd_fused_kernel<64, 128, 64, true, 0, false, true><<<grid, 512, 0, st>>>(g, ldg, a, lda, w, P, rps, workspace, go, ldo,
                                                                                a_scale, a_shift, yb, b_scale, b_shift,
                                                                                stats_partial, ap, probe_slot_all(grid));
    else
      CPFN_BWD_FUSED(64, 128, 64, false, 0);
  } else if (N == 64 && K == 64) CPFN_BWD_FUSED_SHAPE(64, 64, 64);
  else CPFN_BWD_FUSED_SHAPE(128, 64, 64);
#undef CPFN_BWD_FUSED_SHAPE
#undef CPFN_BWD_FUSED
  return cpfn_launch_status();
}

extern "C" int cpfn_mlp_bwd_fused(const void *Gy, int ldg, const void *A, int lda, const void *W, long long P, int N, int K,
                                  const float *a_scale, const float *a_shift, float *workspace, void *Gout, int ldo,
                                  const void *bwd_y, const float *b_scale, const float *b_shift, float *stats_partial,
                                  const void *apply_y, const float *apply_coef, const float *y_scale, const float *y_shift,
                                  const unsigned long long *drop_seed, float drop_p, const unsigned char *pool_arg,
                                  const void *pool_yarg, int pool_k, const float *xt_xyz,
                                  float *xt_partial, void *stream) {
  return bwd_fused_launch(Gy, ldg, A, lda, W, P, N, K, a_scale, a_shift, workspace, Gout, ldo, bwd_y, b_scale, b_shift, stats_partial,
                          apply_y, apply_coef, y_scale, y_shift, drop_seed, drop_p, pool_arg, pool_yarg, pool_k, xt_xyz, xt_partial,
                          nullptr, nullptr, stream);
}

// The 64 <- 64 one-pass launch (sa1's second layer) with the weight-gradient sums of the fp32-xyz FIRST layer riding on its
// reduction of that layer (mlp_bwd_fused_kernel, XW): xw_xyz [P,3] the first layer's input, xw_partial [splits][7][64] (S1, S2, S3;
// finished by cpfn_multi_split_reduce with cpfn_reduce_desc.coef).  Gout may be NULL: the gradient w.r.t. the first layer's output
// is then never stored.
extern "C" int cpfn_mlp_bwd_fused_xw(const void *Gy, const void *A, const void *W, long long P, int N, int K, const float *a_scale,
                                     const float *a_shift, float *workspace, void *Gout, const void *bwd_y, const float *b_scale,
                                     const float *b_shift, float *stats_partial, const void *apply_y, const float *apply_coef,
                                     const float *y_scale, const float *y_shift, const float *xw_xyz, float *xw_partial,
                                     void *stream) {
  if (!xw_xyz || !xw_partial) return CPFN_EINVAL;
  return bwd_fused_launch(Gy, N, A, K, W, P, N, K, a_scale, a_shift, workspace, Gout, K, bwd_y, b_scale, b_shift, stats_partial, apply_y,
                          apply_coef, y_scale, y_shift, nullptr, 0.f, nullptr, nullptr, 0, nullptr, nullptr, xw_xyz, xw_partial, stream);
}

// cpfn_mlp_bwd_fused's xyz-tail form (sa2's first layer: 128 <- 128 + 3 coordinates, dense apply pass, no layer below) with the
// layer's input rows GATHERED from the per-cloud table while loading (cpfn_mlp_gemm_xyz_gather's operand): the [P, 128] grouped
// copy is not read — it does not exist.  ldg = N = 128, lda = ldo = K = 128.
extern "C" int cpfn_mlp_bwd_fused_xt_gather(const void *Gy, const void *table, const int *gidx, int rows_per_cloud, int n_src,
                                            const void *W, long long P, int N, int K, float *workspace, void *Gout,
                                            const void *apply_y, const float *apply_coef, const float *y_scale, const float *y_shift,
                                            const float *xt_xyz, float *xt_partial, void *stream) {
  if (!gidx) return CPFN_EINVAL;
  return bwd_fused_launch(Gy, N, table, K, W, P, N, K, nullptr, nullptr, workspace, Gout, K, nullptr, nullptr, nullptr, nullptr, apply_y,
                          apply_coef, y_scale, y_shift, nullptr, 0.f, nullptr, nullptr, 0, xt_xyz, xt_partial, nullptr, nullptr, stream,
                          gidx, rows_per_cloud, n_src);
}
