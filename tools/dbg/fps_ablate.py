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
The in-kernel invariant of round 4 (after the update the sample's own min-distance is 0) counts into g_fps_dbg[8] as before."""
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
    open(p, "w").write(s + MARK + tail)
    subprocess.check_call([sys.executable, "-m", "cpfn_amd.build"], cwd=dst)
    for f in ("pk_aggressor.py", "step_repro.py"):
        subprocess.check_call(["cp", os.path.join(ROOT, "tools", "dbg", f), os.path.join(dst, "tools", "dbg", f)])
    print("built", dst)


if __name__ == "__main__":
    main()
