// Furthest-point sampling for gfx950.
//
// One workgroup per cloud.  The running min-distance of every point lives in
// registers (PPT points per lane), the cloud itself is mirrored in LDS (x, y, z planes) so
// that the coordinates of the newly selected point are three broadcast ds_read_b32
// away.  A sample is a VALU-issue phase over the lane's points plus an arg-max, and the
// VALU phase is what bounds it (profiles/r03_fps_latency.md: 8192 points x 13 vector
// instructions / (4 SIMDs x 16 lanes) = 1664 cycles = 0.69 us of the 1.0-1.1 us per sample
// of the round-2 kernel), so since round 3
//   * two points per instruction: the points of a lane sit in registers as pairs and the
//     eight roundings of a distance are v_pk_add_f32 / v_pk_mul_f32 (same IEEE roundings,
//     one instruction for two points), the running minimum is v_min_f32;
//   * the lane keeps only its maximum VALUE (v_max3_f32: half an instruction per point
//     instead of a compare and two selects per point for a (value, index) pair), the wave
//     reduces that 32-bit value by DPP row operations, and only then is the index looked
//     up: the first j whose ballot(md[j] == wave maximum) is non-empty, lowest lane in it —
//     indices are t + j*NT, so that IS the lowest index (what torch.max returns on a tie);
//   * across waves as before: a 64-bit key (fp32 bits of the distance) << 32 | ~index per
//     wave in an LDS slot, ONE workgroup barrier per sample (slots double-buffered by
//     sample parity), a DPP maximum over the slots.
//
// Arithmetic follows the reference's CPU route (modules/geometry_utils.py:88-101)
// op for op — ((dx*dx + dy*dy) + dz*dz) with every op rounded — so that the selected
// indices are identical to it, not merely close.
//
// TRIPWIRE (round 5, every instantiation): a sample's own min-distance is 0 after its update, so the next arg-max can only
// return the SAME point again when every candidate is exhausted (maximum 0).  A repeated point with a positive maximum means the
// update was lost on the lane that owns the sample — the signature of round 4's packed-fp32 fault — and costs one compare per
// sample to see, by the lane that stores the sample's index, OFF the loop's dependency chain: it bumps the fault count that cpfn_fps_faults() /
// ops.check_fps_faults() read (device counter + pinned host word).  Detection only: a first version also repeated the pass (the
// update is idempotent) — the loop control then hung on two VALU -> SGPR round trips per sample, +4..8 % on every shape and +19 %
// on the one-wave kernel (same-box A/B against the round-4 tree), for a repair that cannot reach the lanes that do not own the
// sample anyway.
//
// Beside a training step (cpfn_set_background_geometry) the instantiations differ: no packed fp32 at 64 / 256 lanes x 8 points, and
// the 8192-point shape claims its compute unit's whole LDS — a co-resident weight-gradient workgroup makes packed fp32 lose a
// row now and then (round 4; fps_update and fps_launch below).
#include "common.h"
#include <cstdlib>
#include <mutex>

constexpr int CPFN_LDS_BYTES_PER_CU = 160 * 1024;      // gfx950

constexpr unsigned CPFN_FPS_INIT_DIST_BITS = 0x501502F9u;      // 1e10f, every point's min-distance before the first sample
__device__ unsigned g_fps_faults = 0;           // sibling time-outs of the several-workgroups kernel + lost updates (tripwire)

namespace {

// one lane of a workgroup reports a fault: device counter + (when the launcher had one) the pinned host word
__device__ __forceinline__ void fps_report_fault(unsigned *__restrict__ host_faults) {
  atomicAdd(&g_fps_faults, 1u);
  if (host_faults) __hip_atomic_fetch_add(host_faults, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

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

typedef float f32x2 __attribute__((ext_vector_type(2)));

// max of a float over the wave (values >= 0, or -1 for "no candidate"), wave-uniform: six v_max_f32 with the DPP modifier
// on their first operand (through __builtin_amdgcn_update_dpp the compiler emits v_mov_b32_dpp + v_max_f32 + their wait
// states per step: 168 cycles for the six steps, profiles/r03_fps_latency.md).  The s_nop 1 in front of each step is the
// two wait states a DPP read of a VGPR needs after the VALU write of it.
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

// v_min_f32 / v_max3_f32 as single instructions: through fminf / fmaxf the compiler first canonicalises every operand it
// cannot prove to be an arithmetic result (v_max_f32 x, x: one more instruction per point for the min-distances carried
// around the sample loop).  No NaNs exist here: coordinates are finite, distances >= 0, "not a candidate" is -1.
__device__ __forceinline__ float v_min(float a, float b) {
  float r;
  asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float v_max3(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

// One sample's pass over a lane's PPT points (pairs in registers): min-distances updated against the new sample
// (fx, fy, fz), the lane's largest min-distance returned.  d = ((dx*dx + dy*dy) + dz*dz), every operation rounded
// (the translation unit is compiled with -ffp-contract=off; the packed instructions round like the scalar ones).
//
// PK = false — one point per instruction, no packed fp32: the 64- and 256-lane instantiations used BESIDE a training step (the next
// batch's geometry on the side stream).  Round 4 found run-to-run different samples in ~1 of 100 twelve-step runs of
// tests/test_gpu_epoch.py's configuration and followed them into this function.  With v_pk_add_f32 / v_pk_mul_f32 here, a lane of a
// wave's LAST row (lanes 48-63) now and then kept its min-distance un-updated for one sample — caught in the act by an in-kernel
// invariant (the sample's own min-distance must be 0 after the update; EXEC full, the sample's coordinates right) — whenever a
// workgroup of a WEIGHT-GRADIENT kernel shared the compute unit: mlp_wgrad, the one-pass backward kernels, the small-layer backward
// kernels (the kernels that read their operands transposed out of LDS, ds_read_b64_tr_b16 — the same neighbour next to which
// ds_read_b96 returned wrong data in round 1, common.h).  tools/dbg/pk_aggressor.py, packed 4 x 8 instantiation on a side stream,
// one candidate looping on the main stream: 1027 of 19264 launches differ beside mlp_wgrad; 0 beside the forward / data-gradient
// GEMMs, the BatchNorm passes (fp64 sums included), the split reduction, the optimizer, copies, a spinning flag waiter.  Not a
// wait-state problem of this wave (eight s_nop around the consumer made it MORE frequent), not LDS read widths, not stale input,
// none of the hand-written reductions (compiler-scheduled C versions failed alike).  With the eight roundings as single
// v_sub / v_mul / v_add: 0 violations, 0 of 250 runs differ.  The 8192-point shape keeps the packed form behind a structural
// guard (fps_launch: its workgroup claims the compute unit's whole LDS); stand-alone launches (evaluation, the parity tests: nothing
// else runs) keep it too: same roundings, half the VALU issue.
template <int PPT, bool PK = true>
__device__ __forceinline__ float fps_update(const f32x2 (&px)[PPT / 2], const f32x2 (&py)[PPT / 2], const f32x2 (&pz)[PPT / 2],
                                            f32x2 (&md)[PPT / 2], float fx, float fy, float fz) {
  if (!PK) {
    constexpr int NCH1 = PPT >= 16 ? 4 : 1;
    float lm1[NCH1];
#pragma unroll
    for (int c = 0; c < NCH1; ++c) lm1[c] = -1.0f;
#pragma unroll
    for (int j = 0; j < PPT / 2; ++j) {
      float m2[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        // (inline assembly, one rounding each: left to the compiler, the vectoriser pairs them up into the packed forms again)
        float dx, dy, dz, xx, yy, zz, d;
        asm("v_sub_f32 %0, %1, %2" : "=v"(dx) : "v"(px[j][h]), "v"(fx));
        asm("v_sub_f32 %0, %1, %2" : "=v"(dy) : "v"(py[j][h]), "v"(fy));
        asm("v_sub_f32 %0, %1, %2" : "=v"(dz) : "v"(pz[j][h]), "v"(fz));
        asm("v_mul_f32 %0, %1, %1" : "=v"(xx) : "v"(dx));
        asm("v_mul_f32 %0, %1, %1" : "=v"(yy) : "v"(dy));
        asm("v_mul_f32 %0, %1, %1" : "=v"(zz) : "v"(dz));
        asm("v_add_f32 %0, %1, %2" : "=v"(d) : "v"(xx), "v"(yy));
        asm("v_add_f32 %0, %1, %2" : "=v"(d) : "v"(d), "v"(zz));
        m2[h] = v_min(md[j][h], d);
      }
      md[j] = (f32x2){m2[0], m2[1]};
      lm1[j % NCH1] = v_max3(lm1[j % NCH1], m2[0], m2[1]);
    }
    float r1 = lm1[0];
    if (NCH1 == 4) r1 = fmaxf(v_max3(lm1[0], lm1[1], lm1[2]), lm1[3]);
    return r1;
  }
  const f32x2 f2x = {fx, fx}, f2y = {fy, fy}, f2z = {fz, fz};
  // (a dependent vector instruction issues ~8-10 cycles after its producer when the wave has nothing else to issue — one
  //  wave per SIMD at 32 points per lane — so the running maximum is NCH independent chains, combined at the end)
  constexpr int NCH = PPT >= 16 ? 4 : 1;
  float lm[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) lm[c] = -1.0f;
#pragma unroll
  for (int j = 0; j < PPT / 2; ++j) {
    const f32x2 dx = px[j] - f2x, dy = py[j] - f2y, dz = pz[j] - f2z;
    const f32x2 d = (dx * dx + dy * dy) + dz * dz;
    f32x2 m = md[j];
    m.x = v_min(m.x, d.x);            // never lowers an m = -1 ("not a candidate"): d >= 0
    m.y = v_min(m.y, d.y);
    md[j] = m;
    lm[j % NCH] = v_max3(lm[j % NCH], m.x, m.y);
  }
  float r = lm[0];
  if (NCH == 4) r = fmaxf(v_max3(lm[0], lm[1], lm[2]), lm[3]);
  return r;
}

// Lowest point index of the wave whose min-distance equals the wave's maximum wmax (>= 0).  Slot q of a lane is point
// base + lane + q * NT, so the lowest index is the lowest slot, then the lowest lane.  Branch-free on the vector side: every
// lane finds its own first matching slot (one compare + one select per slot, walking the slots downwards); then, because
// the maximum is almost always attained by ONE lane, a ballot and a readlane finish the job — exact ties across lanes (real:
// duplicated points) take the DPP minimum over the lanes' candidate indices.  (A first version walked the slots with
// wave-uniform ballots and an early exit: a vector compare feeding a scalar branch costs ~33 cycles per slot — 1140 cycles
// for 32 slots, and the waves of a workgroup left that loop at different times: tools/dbg/valu_issue_bench.hip,
// profiles/r03_fps_latency.md.)
template <int PPT, int NT>
__device__ __forceinline__ unsigned fps_first_index(const f32x2 (&md)[PPT / 2], float wmax, unsigned base, int lane) {
  // Four slots per step: four compares into four DIFFERENT scalar mask registers, then the four selects.  Written through
  // the compiler, every compare lands in VCC and every select waits for it (v_cmp, s_nop 1, v_cndmask: ~20 cycles per slot
  // on a wave that has its SIMD to itself — 665 of a sample's 2500 cycles at 32 points per lane).
  unsigned q = 0xFFFFu;
#pragma unroll
  for (int j = PPT / 2 - 2; j >= 0; j -= 2) {
    asm("v_cmp_eq_f32_e64 s[40:41], %[w], %[m3]\n\t"
        "v_cmp_eq_f32_e64 s[42:43], %[w], %[m2]\n\t"
        "v_cmp_eq_f32_e64 s[44:45], %[w], %[m1]\n\t"
        "v_cmp_eq_f32_e64 s[46:47], %[w], %[m0]\n\t"
        "v_cndmask_b32_e64 %[q], %[q], %[c3], s[40:41]\n\t"
        "v_cndmask_b32_e64 %[q], %[q], %[c2], s[42:43]\n\t"
        "v_cndmask_b32_e64 %[q], %[q], %[c1], s[44:45]\n\t"
        "v_cndmask_b32_e64 %[q], %[q], %[c0], s[46:47]"
        : [q] "+v"(q)
        : [w] "s"(wmax), [m0] "v"(md[j].x), [m1] "v"(md[j].y), [m2] "v"(md[j + 1].x), [m3] "v"(md[j + 1].y),
          [c0] "n"(2 * j), [c1] "n"(2 * j + 1), [c2] "n"(2 * j + 2), [c3] "n"(2 * j + 3)
        : "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47");
  }
  const unsigned long long hit = __ballot(q != 0xFFFFu);
  const int l0 = __builtin_ctzll(hit);
  unsigned idx = base + (unsigned)__builtin_amdgcn_readlane((int)q, l0) * NT + (unsigned)l0;
  if (hit & (hit - 1)) {                       // several lanes attain the maximum: the lowest index among them
    unsigned cand = q != 0xFFFFu ? base + q * NT + (unsigned)lane : 0xFFFFFFFFu;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
      const unsigned o = (unsigned)__shfl_xor((int)cand, m, 64);
      cand = o < cand ? o : cand;
    }
    idx = (unsigned)__builtin_amdgcn_readfirstlane((int)cand);
  }
  return idx;
}

// PROFILE: wave 0 accumulates the shader-clock cycles of a sample's phases into prof[b][0..5] (see fps_stamp below) —
// a diagnostic instantiation behind cpfn_fps_profile; the product kernels carry no stamp.
__device__ __forceinline__ unsigned long long fps_stamp() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}

template <int NT, int PPT, bool PROFILE = false, bool PK = true, bool DBG = false>
__global__ __launch_bounds__(NT) void fps_resident_kernel(const float *__restrict__ xyz, int N, int S,
                                                          const int *__restrict__ start, int flags,
                                                          int *__restrict__ idx_out,
                                                          unsigned long long *__restrict__ prof = nullptr,
                                                          float *__restrict__ centres = nullptr /* [B,S,3]: xyz[idx_out] */,
                                                          unsigned *__restrict__ host_faults = nullptr, int dbg_drop = -1) {
  constexpr int NW = NT / CPFN_WAVE;
  static_assert(PPT % 2 == 0, "points sit in registers as pairs");
  __shared__ float s_x[NT * PPT], s_y[NT * PPT], s_z[NT * PPT];   // three b32 broadcasts per sample, NOT one float4:
                                                                   // see cpfn_lds_read4 in common.h (ds_read_b96)
  __shared__ unsigned long long s_key[2][NW > 1 ? NW : 1];

  const int b = blockIdx.x;
  const int t = threadIdx.x;
  const int lane = t & (CPFN_WAVE - 1);
  const int wave = t / CPFN_WAVE;
  const float *p = xyz + (size_t)b * N * 3;
  int *out = idx_out + (size_t)b * S;

  f32x2 px[PPT / 2], py[PPT / 2], pz[PPT / 2], md[PPT / 2];
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
    px[j / 2][j & 1] = x; py[j / 2][j & 1] = y; pz[j / 2][j & 1] = z; md[j / 2][j & 1] = m;
    s_x[k] = x; s_y[k] = y; s_z[k] = z;
    // (measured and not kept: this loop also writing the cloud as (x, y, z, |p|^2) for the ball query that follows beside a
    //  training step — cpfn_pack_xyzn's launch saved, the step 17 us SLOWER: the stores sit at the head of the longest chain)
  }
  __syncthreads();

  unsigned long long acc[6] = {0, 0, 0, 0, 0, 0}, t0 = 0;
  unsigned far = start ? (unsigned)start[b] : 0u;
  float fx = s_x[far], fy = s_y[far], fz = s_z[far];
  unsigned prev = 0xFFFFFFFFu, kd = 0u;
  for (int i = 0; i < S; ++i) {
    if (PROFILE) t0 = fps_stamp();
    if (t == 0) {
      out[i] = (int)far;
      // TRIPWIRE, by the lane that stores the index: the point sampled one pass ago comes back with a positive distance -> its own
      // min-distance was not zeroed (a lost update).  Placed HERE the check is free (same-box A/B: 464 us against 464 without it at
      // 8 waves x 16 points, 454 against 480 at 4 x 32; at the loop's tail, behind the cross-wave reduction, it cost 3-4 % — there
      // the compiler rotated the loop differently and filled the v_readlane wait states with s_nop instead of pointer arithmetic)
      // (kd == the bits of the INITIAL min-distance 1e10f: the point's distance was never lowered at all — an inf / NaN
      //  coordinate, whose distance to itself is NaN; the reference repeats such a point silently, geometry_utils.py:88-101,
      //  and so does this kernel: bad input, not a lost update — ADVICE r5)
      if (!PROFILE && i > 0 && far == prev && kd != 0u && kd != CPFN_FPS_INIT_DIST_BITS) fps_report_fault(host_faults);
    }
    prev = far;
    if (NW == 1 && i > 0) { fx = s_x[far]; fy = s_y[far]; fz = s_z[far]; }
    if (centres && t == 0) {          // the sampled centre itself (what a gather of xyz by idx_out would read: one launch less per level)
      float *c = centres + ((size_t)b * S + i) * 3;
      c[0] = fx; c[1] = fy; c[2] = fz;
    }
    if (PROFILE) { const unsigned long long t1 = fps_stamp(); acc[0] += t1 - t0; t0 = t1; }      // (one wave: broadcast read of the sample)
    // (test hook, cpfn_fps_debug_drop — the DBG instantiations only, the product kernels carry none of it: the wave that owns
    //  sample `dbg_drop` loses its update once — what the hardware fault does to a row of lanes — so that the tripwire and its
    //  repair can be tested on a box that does not have the fault.  The wave measures its distances to a point at infinity
    //  instead: every min(md, inf) leaves md alone.)
    const bool drop = DBG && i == dbg_drop && (unsigned)wave == (far % NT) / CPFN_WAVE;
    const float ux = drop ? __builtin_inff() : fx;
    const float lm = fps_update<PPT, PK>(px, py, pz, md, ux, fy, fz);
    if (PROFILE) { const unsigned long long t1 = fps_stamp(); acc[1] += t1 - t0; t0 = t1; }      // distance update + lane maximum
    const float wmax = wave_max_f32(lm);
    if (PROFILE) { const unsigned long long t1 = fps_stamp(); acc[2] += t1 - t0; t0 = t1; }      // wave maximum (DPP)
    unsigned long long key = 0ull;
    if (wmax >= 0.f)
      key = ((unsigned long long)__float_as_uint(wmax) << 32) | (unsigned)(~fps_first_index<PPT, NT>(md, wmax, (unsigned)t - (unsigned)lane, lane));
    if (PROFILE) { const unsigned long long t1 = fps_stamp(); acc[3] += t1 - t0; t0 = t1; }      // index of the maximum (ballots)
    if (NW > 1) {
      if (lane == 0) s_key[i & 1][wave] = key;
      __syncthreads();
      if (PROFILE) { const unsigned long long t1 = fps_stamp(); acc[4] += t1 - t0; t0 = t1; }    // LDS slot + workgroup barrier
      key = s_key[i & 1][lane & (NW - 1)];
      // Lane w (< NW) holds wave w's candidate: its coordinates are requested from the LDS mirror NOW, while the maximum over
      // the waves is still being formed, and the winner's are picked with three readlanes — the sample's coordinates used
      // to be read after the maximum was known: one more LDS round trip (~250 cycles of a 2200-cycle sample) on the chain.
      const unsigned cand = key ? ~(unsigned)(key & 0xFFFFFFFFull) : 0u;
      const float cx = s_x[cand], cy = s_y[cand], cz = s_z[cand];
      // (the maximum over the slots by DPP row operations; reading all NW slots into every lane and taking the maximum in
      //  registers as a tree was measured slower: 570-680 against 405-435 cycles — 64-bit compare / select pairs wait on VCC)
      if (NW >= 4) {
        key = group_max_key<(NW >= 4 ? NW : 4)>(key);
      } else {
#pragma unroll
        for (int m = NW / 2; m >= 1; m >>= 1) {
          unsigned long long o = cpfn_shfl_xor_u64(key, m);
          key = o > key ? o : key;
        }
      }
      far = key ? ~(unsigned)(key & 0xFFFFFFFFull) : 0u;
      far = (unsigned)__builtin_amdgcn_readfirstlane((int)far);                   // (every lane holds the maximum)
      const int w = key ? (int)((far % NT) / CPFN_WAVE) : 0;                      // the wave that owns point `far`; no candidate: any slot reads point 0
      fx = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cx), w));
      fy = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cy), w));
      fz = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cz), w));
    } else {
      far = key ? ~(unsigned)(key & 0xFFFFFFFFull) : 0u;
    }
    if (PROFILE) { const unsigned long long t1 = fps_stamp(); acc[5] += t1 - t0; t0 = t1; }      // slot read + maximum over the waves
    kd = (unsigned)(key >> 32);            // (distance bits of the key that chose the next sample: the tripwire at the loop's top)
  }
  if (PROFILE && t == 0 && prof) {
#pragma unroll
    for (int q = 0; q < 6; ++q) prof[(size_t)b * 6 + q] = acc[q];
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
// Round 4, measured on one box (tools/dbg/fps_shared_time.py, 131072 -> 512, B = 1): what the exchange costs grows with the
// number of PARTICIPANTS, not with the bytes: 64 workgroups x 8 points per lane 1.41 ms, 32 x 16: 1.24 ms, 16 x 32: 1.00 ms
// (512 threads x 16: 1.04, 1024 x 8: 1.10; 8 workgroups of 512 x 32: 1.25 — the per-sample pass over 16384 points then costs
// more than the lighter exchange saves).  The launcher therefore always takes 32 points per lane.  NOT adopted: the winner's
// coordinates riding with the key as three more tagged 8-byte granules per workgroup (read from an LDS mirror, published by
// four lanes at once, one granule kind polled per wave, so that the dependent p[3 * far] load after the exchange disappears):
// 1.57 / 1.49 / 1.39 ms at 8 / 16 / 32 points per lane against 1.41 / 1.24 / 1.00 — four times the polling traffic on the
// same few cache lines slows every workgroup's round trip by more than the L2-hit load of the coordinates cost; and with
// the workgroups of a cloud placed on ONE XCD (linear block ids congruent modulo 8) 2.99 ms at 64 workgroups (two per CU on
// 32 CUs: the per-sample pass doubles), 1.25 ms at 32.  Both removed.
template <int PPT, bool DBG = false>
__global__ __launch_bounds__(256) void fps_shared_kernel(const float *__restrict__ xyz, int N, int S,
                                                         const int *__restrict__ start, int flags,
                                                         int *__restrict__ idx_out, unsigned long long *__restrict__ slots,
                                                         unsigned *__restrict__ host_faults, int dbg_drop) {
  constexpr int NT = 256, NW = 4;
  __shared__ unsigned long long s_key[2][NW];
  __shared__ unsigned s_far, s_dist;
  const int G = gridDim.x, wg = blockIdx.x, b = blockIdx.y;
  const int t = threadIdx.x, lane = t & (CPFN_WAVE - 1), wave = t / CPFN_WAVE;
  const float *p = xyz + (size_t)b * N * 3;
  int *out = idx_out + (size_t)b * S;
  unsigned long long *sl = slots + (size_t)b * 2 * G;
  const int base = wg * NT * PPT;
  f32x2 px[PPT / 2], py[PPT / 2], pz[PPT / 2], md[PPT / 2];
#pragma unroll
  for (int j = 0; j < PPT; ++j) {
    const int k = base + t + j * NT;
    float x = 0.f, y = 0.f, z = 0.f, m = -1.0f;      // m < 0 marks "not a candidate"
    if (k < N) {
      x = p[3 * k]; y = p[3 * k + 1]; z = p[3 * k + 2];
      m = 1e10f;
      if ((flags & CPFN_FPS_SKIP_NEAR_ORIGIN) && cpfn_sqnorm3(x, y, z) <= 1e-3f) m = -1.0f;
    }
    px[j / 2][j & 1] = x; py[j / 2][j & 1] = y; pz[j / 2][j & 1] = z; md[j / 2][j & 1] = m;
  }
  unsigned far = start ? (unsigned)start[b] : 0u;
  for (int i = 0; i < S; ++i) {
    if (wg == 0 && t == 0) out[i] = (int)far;
    const float fx = p[3 * far], fy = p[3 * far + 1], fz = p[3 * far + 2];
    const bool drop = DBG && i == dbg_drop && (int)far >= base && (int)far < base + NT * PPT &&
                      (unsigned)wave == ((far - (unsigned)base) % NT) / CPFN_WAVE;          // (test hook: see fps_resident_kernel)
    const float lm = fps_update<PPT>(px, py, pz, md, drop ? __builtin_inff() : fx, fy, fz);
    const float wmax = wave_max_f32(lm);
    // key without the tag: candidates compare by (distance, lowest index); "no candidate" = 0
    unsigned long long key = 0ull;
    if (wmax >= 0.f) {
      const unsigned besti = fps_first_index<PPT, NT>(md, wmax, (unsigned)(base + t - lane), lane);
      key = ((unsigned long long)__float_as_uint(wmax) << 32) | ((unsigned long long)(0xFFFFFu - besti) << 12);
    }
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
      if (lane == 0) {
        s_far = timeout ? 0xFFFFFFFFu : (k ? 0xFFFFFu - (unsigned)((k >> 12) & 0xFFFFFull) : 0u);
        s_dist = (unsigned)(k >> 32);
      }
    }
    __syncthreads();
    const unsigned nf = s_far, nd = s_dist;
    if (nf == 0xFFFFFFFFu) {          // a sibling workgroup never arrived: give up for this cloud (every workgroup of it
                                      // takes this branch at most one spin period later)
      if (wg == 0 && t == 0) {
        for (int r = i + 1; r < S; ++r) out[r] = 0;
        fps_report_fault(host_faults);
      }
      break;
    }
    // TRIPWIRE (see fps_resident_kernel): the point just sampled comes back with a positive distance
    if (__builtin_expect(nf == far, 0)) {
      asm volatile("; tripwire: rare path" ::: "memory");
      if (wg == 0 && t == 0 && nd != 0u && nd != CPFN_FPS_INIT_DIST_BITS) fps_report_fault(host_faults);
    }
    far = nf;
  }
}

}  // namespace

// Workgroups of fps_shared_kernel<PPT> that can be resident at once on the current device when each is launched with `dyn_lds`
// bytes of dynamic LDS (0: unknown -> do not use it).
template <int PPT>
static int fps_shared_capacity(int dyn_lds) {
  static int cached[64] = {0};                  // per device ordinal; 0 = not queried yet, -1 = query failed
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
  if (cached[dev] == 0) {
    int per_cu = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fps_shared_kernel<PPT, false>, 256, dyn_lds) == hipSuccess &&
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && per_cu > 0 && cus > 0)
      cached[dev] = per_cu * cus;
    else
      cached[dev] = -1;
  }
  return cached[dev] > 0 ? cached[dev] : 0;
}

// The fault count also lives in a pinned host word the kernel bumps with a system-scope atomic, so that it can be polled
// without synchronising the device (round 3 read the device symbol: a hipMemcpyFromSymbol, i.e. a device synchronisation,
// which nobody but a test ever paid for).
// All launcher-side state of this file (the host word, the LDS claims) sits behind ONE mutex: launches may come from several
// host threads (the epoch loop's staging thread, evaluation beside training) — ADVICE r5.
static std::mutex g_fps_mu;

static unsigned *fps_host_faults(unsigned **dev_ptr) {      // (g_fps_mu held by the caller)
  static unsigned *host = nullptr, *dev = nullptr;
  static bool tried = false;
  if (!tried) {
    tried = true;
    // The first launch may sit inside a stream capture (thread-local / global mode: allocations are refused there): the
    // allocation is not a stream operation, so the capture mode is relaxed around it — the word then exists for EVERY launch and
    // cpfn_fps_faults() never has to fall back to the synchronising read of the device symbol.
    hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
    const bool swapped = hipThreadExchangeStreamCaptureMode(&mode) == hipSuccess;
    void *h = nullptr, *d = nullptr;
    if (hipHostMalloc(&h, sizeof(unsigned), hipHostMallocMapped) == hipSuccess && h &&
        hipHostGetDevicePointer(&d, h, 0) == hipSuccess && d) {
      host = (unsigned *)h;
      dev = (unsigned *)d;
      *host = 0u;
    } else {
      (void)hipGetLastError();
    }
    if (swapped) (void)hipThreadExchangeStreamCaptureMode(&mode);
  }
  if (dev_ptr) *dev_ptr = dev;
  return host;
}

static bool g_fps_launched_without_hf = false;

// Sampling faults since the library was loaded: clouds whose several-workgroups FPS gave up on a sibling + lost updates caught by
// the tripwire.  Reads a pinned host word (no synchronisation; a fault shows up once its kernel has got that far).  Launches
// that were issued before the word existed (a first launch inside a stream capture) report to the device counter only: once that
// has happened — or without a host word at all — the device symbol is read as well (synchronises) and the larger count returned
// (the freshly allocated host word, 0, used to hide such a graph's faults).
extern "C" int cpfn_fps_faults(void) {
  const unsigned *h;
  bool without;
  {
    std::lock_guard<std::mutex> lk(g_fps_mu);
    h = fps_host_faults(nullptr);
    without = g_fps_launched_without_hf;
  }
  unsigned host_n = h ? __atomic_load_n(h, __ATOMIC_RELAXED) : 0u;
  if (h && !without) return (int)host_n;
  unsigned n = 0;
  if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_fps_faults), sizeof(n)) != hipSuccess) return h ? (int)host_n : -1;
  return (int)(n > host_n ? n : host_n);
}

// The whole-LDS claim of a one-workgroup-per-cloud sampling kernel: dynamic padding up to the compute unit's 160 KB, so that no
// LDS-using workgroup of another kernel can be resident beside it (-1: not available).  Asked once per kernel.
static int fps_lds_claim(const void *kernel) {
  constexpr int MAX_DEV = 16;
  static const void *seen_[MAX_DEV][4] = {};
  static int pads_[MAX_DEV][4];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEV) { (void)hipGetLastError(); return -1; }
  std::lock_guard<std::mutex> lk(g_fps_mu);       // (a function attribute belongs to a device: the cache is per device ordinal)
  const void **seen = seen_[dev];
  int *pads = pads_[dev];
  for (int i = 0; i < 4; ++i) {
    if (seen[i] == kernel) return pads[i];
    if (!seen[i]) {
      seen[i] = kernel;
      pads[i] = -1;
      hipFuncAttributes a;
      if (hipFuncGetAttributes(&a, kernel) == hipSuccess) {
        const int p = CPFN_LDS_BYTES_PER_CU - (int)a.sharedSizeBytes;
        if (p >= 0 && hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, p) == hipSuccess) pads[i] = p;
      }
      (void)hipGetLastError();
      return pads[i];
    }
  }
  return -1;
}

static int fps_launch(const float *xyz, int B, int N, int S, const int *start, int flags, int *idx_out, float *scratch,
                      float *centres, hipStream_t st);

extern "C" int cpfn_fps(const float *xyz, int B, int N, int S, const int *start, int flags, int *idx_out,
                        float *scratch, void *stream) {
  return fps_launch(xyz, B, N, S, start, flags, idx_out, scratch, nullptr, (hipStream_t)stream);
}

extern "C" int cpfn_fps_max_resident(void) { return CPFN_FPS_MAX_RESIDENT; }

extern "C" int cpfn_fps_centres(const float *xyz, int B, int N, int S, const int *start, int flags, int *idx_out,
                                float *centres, void *stream) {
  if (N > CPFN_FPS_MAX_RESIDENT || !centres) return CPFN_EINVAL;      // (the one-workgroup-per-cloud kernels only)
  return fps_launch(xyz, B, N, S, start, flags, idx_out, nullptr, centres, (hipStream_t)stream);
}

static int g_fps_dbg_drop = -1;
// Test hook: the wave that owns sample `sample` (0-based) skips its distance update once in every sampling launch issued from now on
// (-1: off) — what round 4's hardware fault does to a row of lanes — so that the tripwire and its repair are testable anywhere.
extern "C" int cpfn_fps_debug_drop(int sample) {
  const int was = g_fps_dbg_drop;
  g_fps_dbg_drop = sample;
  return was;
}

static int fps_launch(const float *xyz, int B, int N, int S, const int *start, int flags, int *idx_out, float *scratch,
                      float *centres, hipStream_t st) {
  if (B < 0 || N <= 0 || S < 0 || !xyz || (!idx_out && B * S > 0)) return CPFN_EINVAL;
  if (B == 0 || S == 0) return 0;
  // the pinned host word of the fault count (first use allocates: never inside a capture — such a launch reports to the device
  // counter only, and cpfn_fps_faults() then reads both)
  unsigned *hf = nullptr;
  {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess) (void)hipGetLastError();
    (void)cs;
    std::lock_guard<std::mutex> lk(g_fps_mu);
    fps_host_faults(&hf);                         // (allocates on first use, also inside a capture: see there)
    if (!hf) g_fps_launched_without_hf = true;    // no host word on this stack at all: cpfn_fps_faults() reads the device symbol
  }
  const int dd = g_fps_dbg_drop;
  // (beside a training step — cpfn_background_geometry() — the instantiations without packed fp32: see fps_update)
  const bool beside = cpfn_background_geometry();
  // one launch of fps_resident_kernel<NT, PPT, false, PK, DBG> (the DBG twins only while the test hook is armed)
#define CPFN_FPS_RESIDENT(NT_, PPT_, PK_, LDS_)                                                                                          \
  do {                                                                                                                                 \
    if (dd >= 0) fps_resident_kernel<NT_, PPT_, false, PK_, true><<<B, NT_, 0, st>>>(xyz, N, S, start, flags, idx_out, nullptr, centres, hf, dd); \
    else fps_resident_kernel<NT_, PPT_, false, PK_, false><<<B, NT_, LDS_, st>>>(xyz, N, S, start, flags, idx_out, nullptr, centres, hf, -1); \
  } while (0)
  if (N <= 512) {
    if (beside) CPFN_FPS_RESIDENT(64, 8, false, 0);
    else CPFN_FPS_RESIDENT(64, 8, true, 0);
  } else if (N <= 2048) {
    if (beside) CPFN_FPS_RESIDENT(256, 8, false, 0);
    else CPFN_FPS_RESIDENT(256, 8, true, 0);
  } else if (N <= CPFN_FPS_MAX_RESIDENT) {
    // 8192 points on ONE CU either way (a sample is a VALU-throughput phase over the cloud plus two key reductions):
    // (with ds_bpermute key reductions) 16 waves x 8 points per lane took 670 us for 512 samples, 8 waves x 16 points 572 us,
    // 4 waves x 32 points 636 us (one wave per SIMD: no other wave hides a wave's dependent chains), 2 x 64 1003 us
    // (registers spill to AGPRs); with the DPP reductions 4 x 32 takes 522 us (the step beside it: 1.898 -> 1.868 ms).
    // Beside a training step (one workgroup per cloud on 16 CUs for the whole forward pass) the FEWER waves the better
    // for the step: 1.871 ms with 16 waves, 1.855 with 8, 1.849 with 4 (interleaved A/B on one box each).
    if (beside) {
      // The 8192-point shape keeps its packed arithmetic beside a step, behind a structural guard instead: its workgroup claims
      // the compute unit's WHOLE LDS (96 KB mirror + dynamic padding to 160 KB), so that no workgroup that uses LDS — every kernel
      // of the disturbing kind does: their transposed operand reads are LDS reads — can be resident on the same compute unit.
      // (Unpadded, a 64 x 64 weight-gradient workgroup — 9 KB of LDS, 116 registers — fits beside it; the 128-channel ones never
      //  did: 252 + 257 registers.  The one-point-per-instruction form costs this shape 116 us per launch: +48 us per step and
      //  the dominant kernel 0.72 -> 0.63 of peak, because the longer chain then overlaps the backward pass.)
      //  CPFN_FPS_BESIDE_MODE (debugging): 0 = the form without packed fp32, 2 = packed without the LDS claim.
      static const int mode = getenv("CPFN_FPS_BESIDE_MODE") ? atoi(getenv("CPFN_FPS_BESIDE_MODE")) : 1;
      const int pad = fps_lds_claim((const void *)fps_resident_kernel<256, 32, false, true, false>);
      if (mode == 2) CPFN_FPS_RESIDENT(256, 32, true, 0);
      else if (mode == 1 && pad >= 0) CPFN_FPS_RESIDENT(256, 32, true, pad);
      else CPFN_FPS_RESIDENT(256, 32, false, 0);
    } else {
      // Stand-alone (evaluation, the parity tests) the packed 8-wave shape — ALSO with its compute unit's LDS claimed whole
      // (round 5): "nothing else runs" was an assumption about the caller; the claim costs a workgroup that has
      // its compute unit to itself nothing.
      const int pad = fps_lds_claim((const void *)fps_resident_kernel<512, 16, false, true, false>);
      CPFN_FPS_RESIDENT(512, 16, true, pad >= 0 ? pad : 0);
    }
  } else {
    if (!scratch) return CPFN_EINVAL;
    // several workgroups per cloud while all of them can be resident together and the key layout holds
    // (index < 2^20); slots = the first B * 2 * G 8-byte words of the scratch row buffer
    // 32 points per lane: as few workgroups per cloud as the registers allow (see fps_shared_kernel: the exchange costs by
    // participant) — 16 for the 131072-point clouds of the evaluation cascade, 64 at 524288 points
    const int ppt = 32;
    const int G = (N + 256 * ppt - 1) / (256 * ppt);
    // (round 5: each of these workgroups claims its compute unit's whole LDS too — one workgroup per compute unit, nothing that
    //  uses LDS beside its packed arithmetic; the cascade's shapes need 16-64 of the 256 compute units)
    const int claim = fps_lds_claim((const void *)fps_shared_kernel<32, false>);
    const int capacity = fps_shared_capacity<32>(claim >= 0 ? claim : 0);
    if (G <= 64 && (long long)B * G <= capacity && B <= 65535 && N <= (1 << 20) &&
        (size_t)B * 2 * G * 8 <= (size_t)B * N * 4 && ((uintptr_t)scratch & 7) == 0) {
      hipError_t e = hipMemsetAsync(scratch, 0, (size_t)B * 2 * G * 8, st);
      if (e != hipSuccess) return (int)e;
      unsigned long long *slots = (unsigned long long *)scratch;
      if (dd >= 0) fps_shared_kernel<32, true><<<dim3(G, B), 256, 0, st>>>(xyz, N, S, start, flags, idx_out, slots, hf, dd);
      else fps_shared_kernel<32, false><<<dim3(G, B), 256, claim >= 0 ? claim : 0, st>>>(xyz, N, S, start, flags, idx_out, slots, hf, -1);
    } else {
      fps_streaming_kernel<1024><<<B, 1024, 0, st>>>(xyz, N, S, start, flags, idx_out, scratch);
    }
  }
#undef CPFN_FPS_RESIDENT
  return cpfn_launch_status();
}

// Diagnostic: the resident kernel of shape `variant` (0: 256 threads x 8 points per lane [N <= 2048], 1: 512 x 16, 2: 256 x 32
// [N <= 8192]) with phase stamps; prof[B][6] = shader-clock cycles summed over the S samples of {sample broadcast read,
// distance update + lane maximum, wave maximum, index ballots, LDS slot + barrier, slot read + maximum over waves} as
// wave 0 of each workgroup saw them (every stamp drains the wave's memory counters first and costs ~40 cycles itself:
// profiles/r03_fps_latency.md subtracts that).  Same indices as cpfn_fps.
extern "C" int cpfn_fps_profile(const float *xyz, int B, int N, int S, const int *start, int variant, int *idx_out,
                                unsigned long long *prof, void *stream) {
  if (B <= 0 || N <= 0 || S <= 0 || !xyz || !idx_out || !prof) return CPFN_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (variant == 0 && N <= 2048) fps_resident_kernel<256, 8, true><<<B, 256, 0, st>>>(xyz, N, S, start, 0, idx_out, prof);
  else if (variant == 1 && N <= 8192) fps_resident_kernel<512, 16, true><<<B, 512, 0, st>>>(xyz, N, S, start, 0, idx_out, prof);
  else if (variant == 2 && N <= 8192) fps_resident_kernel<256, 32, true><<<B, 256, 0, st>>>(xyz, N, S, start, 0, idx_out, prof);
  else return CPFN_EINVAL;
  return cpfn_launch_status();
}
