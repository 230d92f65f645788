"""Scratch build of the sampling kernel with RUN-TIME ablation switches, for the packed-fp32 hunt (the method that
found the in-order-return drain in round 4: one build, a device word selects what is skipped, so every variant has the same code
around it).  Not the product:
    python tools/dbg/fps_ablate.py .tabl            # git archive HEAD -> .tabl, csrc/sampling.hip patched, library built there
    cd .tabl && python tools/dbg/pk_aggressor.py 6 ablate      (on the GPU box)
Switches (cpfn_dbg_fps_abl(mask); they act on the STAND-ALONE, packed instantiations, which is what pk_aggressor.py launches):
    1    no index / centre stores (the two EXEC-masked blocks at the head of a sample)
    2    the arg-max index by compiler-scheduled C instead of the four-compare assembly blocks
    4    the wave maximum through __builtin_amdgcn_update_dpp instead of the assembly
    8    the sample's coordinates re-read from the LDS mirror by index AFTER the maximum is known (no v_readlane of candidates)
    16   the maximum over the waves' keys by reading all slots into every lane (no DPP on the 64-bit keys)
    32   the next sample by a fixed schedule instead of the arg-max (everything still computed and kept alive)
    64   s_waitcnt + scheduling fence in front of the distance update
The in-kernel invariant of round 4 (after the update the sample's own min-distance is 0) counts into g_fps_dbg[8] as before.

Round 5: build-time VARIANTS on top of that (second argument; VERDICT r4 #1b), each run with `pk_aggressor.py <s> ablate0`:
    pairs    the sample's coordinates as true {f, f} VGPR pairs (no op_sel_hi:[1,0] on a pair whose high half is an unrelated live register)
    staged   the packed update stage by stage over all pairs (every consumer of a packed result >= 3 independent instructions behind it)
    lds      every v_readlane with an SGPR lane select replaced by an LDS broadcast (arg-max slot and the candidates' coordinates)
    diag     the RICH invariant (`pk_aggressor.py <s> diag`): every lane re-computes every distance one float at a time and compares the
             packed distance AND the new min-distance; per event: sample, lane, slot, packed d, scalar d, scalar d to the PREVIOUS
             sample (a stale operand would give exactly that), old and new min-distance; g_fps_dbg[0..3] = events by 16-lane row,
             [4] = packed d equals the distance to the previous sample, [5] = packed d wrong, [6] = minimum not applied to a right d."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))


def sub(s, old, new, count=1):
    assert s.count(old) == count, (s.count(old), old[:80])
    return s.replace(old, new)


def main():
    dst = os.path.abspath(sys.argv[1] if len(sys.argv) > 1 else ".tabl")
    variant = sys.argv[2] if len(sys.argv) > 2 else "abl"
    assert variant in ("abl", "pairs", "staged", "lds", "diag"), variant
    os.makedirs(dst, exist_ok=False)
    subprocess.check_call("git archive HEAD | tar -x -C '%s'" % dst, shell=True, cwd=ROOT)
    p = os.path.join(dst, "cpfn_amd", "csrc", "sampling.hip")
    s = open(p).read()
    s = sub(s, "namespace {", '''__device__ int g_fps_dbg[64];
__device__ int g_fps_abl;
extern "C" __attribute__((visibility("default"))) int cpfn_dbg_fps_read(int *out, int reset) {
  int e = (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fps_dbg), sizeof(int) * 64);
  if (reset) { int z[64] = {0}; e |= (int)hipMemcpyToSymbol(HIP_SYMBOL(g_fps_dbg), z, sizeof(z)); }
  return e;
}
extern "C" __attribute__((visibility("default"))) int cpfn_dbg_fps_abl(int mask) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_fps_abl), &mask, sizeof(int));
}
namespace {''')
    # C forms of the two assembly helpers
    s = sub(s, "// One sample's pass over a lane's PPT points (pairs in registers)", '''__device__ __forceinline__ float wave_max_f32_c(float v) {
#define CPFN_DPP_MAX(CTRL, RM)                                                                                              \\
  {                                                                                                                         \\
    const int o = __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), CTRL, RM, 0xF, false); \\
    v = fmaxf(v, __builtin_bit_cast(float, o));                                                                             \\
  }
  CPFN_DPP_MAX(0xB1, 0xF) CPFN_DPP_MAX(0x4E, 0xF) CPFN_DPP_MAX(0x141, 0xF) CPFN_DPP_MAX(0x140, 0xF) CPFN_DPP_MAX(0x142, 0xA)
  CPFN_DPP_MAX(0x143, 0xC)
#undef CPFN_DPP_MAX
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
// One sample's pass over a lane's PPT points (pairs in registers)''')
    s = sub(s, "PROFILE: wave 0 accumulates the shader-clock cycles", '''(ablation build) the arg-max index in C:
template <int PPT, int NT>
__device__ __forceinline__ unsigned fps_first_index_c(const f32x2 (&md)[PPT / 2], float wmax, unsigned base, int lane) {
  unsigned q = 0xFFFFu;
#pragma unroll
  for (int j = PPT - 1; j >= 0; --j) q = md[j / 2][j & 1] == wmax ? (unsigned)j : q;
  const unsigned long long hit = __ballot(q != 0xFFFFu);
  const int l0 = __builtin_ctzll(hit);
  unsigned idx = base + (unsigned)__builtin_amdgcn_readlane((int)q, l0) * NT + (unsigned)l0;
  if (hit & (hit - 1)) {
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
// PROFILE: wave 0 accumulates the shader-clock cycles''')
    MARK = "// Any N: min-distances in a global scratch row"          # (what follows the one-workgroup-per-cloud kernel)
    head, tail = s.split(MARK, 1)
    s = head
    s = sub(s, "  unsigned long long acc[6] = {0, 0, 0, 0, 0, 0}, t0 = 0;",
            "  unsigned long long acc[6] = {0, 0, 0, 0, 0, 0}, t0 = 0;\n  const int abl = PK ? g_fps_abl : 0;      // (wave-uniform: a scalar load)\n  float keep = 0.f;")
    s = sub(s, "    if (t == 0) out[i] = (int)far;", "    if (t == 0 && !(abl & 1)) out[i] = (int)far;")
    s = sub(s, "    if (centres && t == 0) {          // the sampled centre itself", "    if (centres && t == 0 && !(abl & 1)) {          // the sampled centre itself")
    s = sub(s, "    const float lm = fps_update<PPT, PK>(px, py, pz, md, fx, fy, fz);", '''    if (abl & 64) { __builtin_amdgcn_s_waitcnt(0); __builtin_amdgcn_sched_barrier(0); }
    const float lm = fps_update<PPT, PK>(px, py, pz, md, fx, fy, fz);
    if ((unsigned)t == far % NT) {                     // the lane that owns the sample: its min-distance is 0 now
      float m = -2.f;
#pragma unroll
      for (int j = 0; j < PPT; ++j) if ((unsigned)j == far / NT) m = md[j / 2][j & 1];
      if (m != 0.f) {
        const int k = atomicAdd(&g_fps_dbg[8], 1);
        if (k < 6) { g_fps_dbg[16 + 8 * k] = i; g_fps_dbg[17 + 8 * k] = (int)far; }
      }
    }''')
    s = sub(s, "    const float wmax = wave_max_f32(lm);", "    const float wmax = (abl & 4) ? wave_max_f32_c(lm) : wave_max_f32(lm);")
    s = sub(s, "      key = ((unsigned long long)__float_as_uint(wmax) << 32) | (unsigned)(~fps_first_index<PPT, NT>(md, wmax, (unsigned)t - (unsigned)lane, lane));",
            '''      key = ((unsigned long long)__float_as_uint(wmax) << 32) |
            (unsigned)(~((abl & 2) ? fps_first_index_c<PPT, NT>(md, wmax, (unsigned)t - (unsigned)lane, lane)
                                   : fps_first_index<PPT, NT>(md, wmax, (unsigned)t - (unsigned)lane, lane)));''')
    s = sub(s, '''      if (NW >= 4) {
        key = group_max_key<(NW >= 4 ? NW : 4)>(key);
      } else {''', '''      if (abl & 16) {
        unsigned long long best = 0ull;
#pragma unroll
        for (int w_ = 0; w_ < NW; ++w_) { const unsigned long long o = s_key[i & 1][w_]; best = o > best ? o : best; }
        key = best;
      } else if (NW >= 4) {
        key = group_max_key<(NW >= 4 ? NW : 4)>(key);
      } else {''')
    s = sub(s, '''      fz = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cz), w));''',
            '''      fz = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cz), w));
      if (abl & 32) {                 // the next sample by schedule: the arg-max above stays alive through `keep`
        keep += fx + (float)far;
        far = (unsigned)((i + 1) * 1103 + 7 * b) % (unsigned)N;
      }
      if (abl & (8 | 32)) { fx = s_x[far]; fy = s_y[far]; fz = s_z[far]; }''')
    s = sub(s, '''  if (PROFILE && t == 0 && prof) {''', '''  if (keep == 12345.678f && idx_out) idx_out[0] = -1;          // (never true: keeps the ablated values alive)
  if (PROFILE && t == 0 && prof) {''')
    if variant == "pairs":
        s = sub(s, "  const f32x2 f2x = {fx, fx}, f2y = {fy, fy}, f2z = {fz, fz};",
                "  f32x2 f2x = {fx, fx}, f2y = {fy, fy}, f2z = {fz, fz};\n"
                "  asm volatile(\"\" : \"+v\"(f2x), \"+v\"(f2y), \"+v\"(f2z));       // real pairs: the compiler cannot fold them into op_sel_hi")
    if variant == "staged":
        s = sub(s, """#pragma unroll
  for (int j = 0; j < PPT / 2; ++j) {
    const f32x2 dx = px[j] - f2x, dy = py[j] - f2y, dz = pz[j] - f2z;
    const f32x2 d = (dx * dx + dy * dy) + dz * dz;
    f32x2 m = md[j];""", """  f32x2 sdx[PPT / 2], sdy[4], sdz[4];
#pragma unroll
  for (int c = 0; c < PPT / 2; c += 4) {                 // four pairs (eight points) at a time, stage by stage
#pragma unroll
    for (int j = 0; j < 4; ++j) { sdx[c + j] = px[c + j] - f2x; sdy[j] = py[c + j] - f2y; sdz[j] = pz[c + j] - f2z; }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 4; ++j) { sdx[c + j] = sdx[c + j] * sdx[c + j]; sdy[j] = sdy[j] * sdy[j]; sdz[j] = sdz[j] * sdz[j]; }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 4; ++j) sdx[c + j] = sdx[c + j] + sdy[j];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 4; ++j) sdx[c + j] = sdx[c + j] + sdz[j];
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int j = 0; j < PPT / 2; ++j) {
    const f32x2 d = sdx[j];
    f32x2 m = md[j];""")
    if variant == "lds":
        # the arg-max slot of the winning lane and the candidates' coordinates through LDS words instead of v_readlane with an SGPR select
        s = sub(s, "  __shared__ unsigned long long s_key[2][NW > 1 ? NW : 1];",
                "  __shared__ unsigned long long s_key[2][NW > 1 ? NW : 1];\n  __shared__ __attribute__((aligned(16))) float s_bc[NW > 1 ? NW : 1][4];")
        s = sub(s, "      fx = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cx), w));\n"
                   "      fy = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cy), w));\n"
                   "      fz = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cz), w));",
                "      if (lane == w) *(cpfn_f32x4 *)&s_bc[wave][0] = (cpfn_f32x4){cx, cy, cz, 0.f};\n"
                "      { const cpfn_f32x4 bc = cpfn_lds_read4(&s_bc[wave][0]); fx = bc.x; fy = bc.y; fz = bc.z; }   // (same wave wrote them: program order in LDS)")
        s = sub(s, "  unsigned idx = base + (unsigned)__builtin_amdgcn_readlane((int)q, l0) * NT + (unsigned)l0;",
                "  unsigned idx = base + (unsigned)__shfl((int)q, l0, 64) * NT + (unsigned)l0;        // ds_bpermute instead of v_readlane", 2)
    if variant == "diag":
        a = s.index("template <int PPT, bool PK = true>\n__device__ __forceinline__ float fps_update(")
        b = s.index("\n}\n", a) + 3
        twin = s[a:b].replace("float fps_update(", "float fps_update_dd(f32x2 (&dd)[PPT / 2], ")
        twin = sub(twin, "    md[j] = m;\n    lm[j % NCH] = v_max3(lm[j % NCH], m.x, m.y);",
                   "    md[j] = m;\n    dd[j] = d;\n    lm[j % NCH] = v_max3(lm[j % NCH], m.x, m.y);")
        s = s[:b] + twin + s[b:]
        old_inv = s[s.index("    if ((unsigned)t == far % NT) {                     // the lane that owns the sample"):s.index("    const float wmax = (abl & 4)")]
        s = s.replace(old_inv, """    if (PK) {
#pragma unroll
      for (int j = 0; j < PPT; ++j) {
        float ds, dp_, e;
        {
          float dx, dy, dz, xx, yy, zz;
          asm volatile("v_sub_f32 %0, %1, %2" : "=v"(dx) : "v"(px[j / 2][j & 1]), "v"(fx));
          asm volatile("v_sub_f32 %0, %1, %2" : "=v"(dy) : "v"(py[j / 2][j & 1]), "v"(fy));
          asm volatile("v_sub_f32 %0, %1, %2" : "=v"(dz) : "v"(pz[j / 2][j & 1]), "v"(fz));
          asm volatile("v_mul_f32 %0, %1, %1" : "=v"(xx) : "v"(dx));
          asm volatile("v_mul_f32 %0, %1, %1" : "=v"(yy) : "v"(dy));
          asm volatile("v_mul_f32 %0, %1, %1" : "=v"(zz) : "v"(dz));
          asm volatile("v_add_f32 %0, %1, %2" : "=v"(ds) : "v"(xx), "v"(yy));
          asm volatile("v_add_f32 %0, %1, %2" : "=v"(ds) : "v"(ds), "v"(zz));
          asm volatile("v_sub_f32 %0, %1, %2" : "=v"(dx) : "v"(px[j / 2][j & 1]), "v"(pfx));
          asm volatile("v_sub_f32 %0, %1, %2" : "=v"(dy) : "v"(py[j / 2][j & 1]), "v"(pfy));
          asm volatile("v_sub_f32 %0, %1, %2" : "=v"(dz) : "v"(pz[j / 2][j & 1]), "v"(pfz));
          asm volatile("v_mul_f32 %0, %1, %1" : "=v"(xx) : "v"(dx));
          asm volatile("v_mul_f32 %0, %1, %1" : "=v"(yy) : "v"(dy));
          asm volatile("v_mul_f32 %0, %1, %1" : "=v"(zz) : "v"(dz));
          asm volatile("v_add_f32 %0, %1, %2" : "=v"(dp_) : "v"(xx), "v"(yy));
          asm volatile("v_add_f32 %0, %1, %2" : "=v"(dp_) : "v"(dp_), "v"(zz));
        }
        const float dpk = dd[j / 2][j & 1], mo = mold[j / 2][j & 1], mn = md[j / 2][j & 1];
        e = v_min(mo, ds);
        if ((dpk != ds || mn != e) && mo >= 0.f) {
          const int k = atomicAdd(&g_fps_dbg[8], 1);
          atomicAdd(&g_fps_dbg[lane >> 4], 1);
          if (dpk == dp_ && dpk != ds) atomicAdd(&g_fps_dbg[4], 1);
          if (dpk != ds) atomicAdd(&g_fps_dbg[5], 1);
          if (dpk == ds && mn != e) atomicAdd(&g_fps_dbg[6], 1);
          if (k < 6) {
            int *r = &g_fps_dbg[16 + 8 * k];
            r[0] = i; r[1] = t; r[2] = j; r[3] = __float_as_int(dpk); r[4] = __float_as_int(ds); r[5] = __float_as_int(dp_);
            r[6] = __float_as_int(mo); r[7] = __float_as_int(mn);
          }
        }
      }
    }
    pfx = fx; pfy = fy; pfz = fz;
""")
        s = sub(s, "    const float lm = fps_update<PPT, PK>(px, py, pz, md, fx, fy, fz);",
                "    f32x2 dd[PPT / 2], mold[PPT / 2];\n#pragma unroll\n    for (int j = 0; j < PPT / 2; ++j) { mold[j] = md[j]; dd[j] = md[j]; }\n"
                "    const float lm = fps_update_dd<PPT, PK>(dd, px, py, pz, md, fx, fy, fz);")
        s = sub(s, "  float keep = 0.f;", "  float keep = 0.f, pfx = 0.f, pfy = 0.f, pfz = 0.f;")
    open(p, "w").write(s + MARK + tail)
    subprocess.check_call([sys.executable, "-m", "cpfn_amd.build"], cwd=dst)
    for f in ("pk_aggressor.py", "step_repro.py"):
        subprocess.check_call(["cp", os.path.join(ROOT, "tools", "dbg", f), os.path.join(dst, "tools", "dbg", f)])
    print("built", dst)


if __name__ == "__main__":
    main()
