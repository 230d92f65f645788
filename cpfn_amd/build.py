"""Builds libcpfn_hip.so in-tree with hipcc for gfx950 (no GPU needed to compile).

    python -m cpfn_amd.build [--force]

One translation unit per kernel family, linked into one shared object whose only
exported symbols are the `extern "C"` entry points declared in include/cpfn_hip.h.
Geometry files are compiled with -ffp-contract=off: their index outputs must be
bit-identical to the reference's CPU arithmetic, so the compiler may not fuse a
multiply into an add unless the source says fma.
"""
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
SO = os.path.join(HERE, "libcpfn_hip.so")
ARCH = "gfx950"

# file -> extra flags
SOURCES = {
    "abi.hip": [],
    # (-fno-slp-vectorize: the kernels of the geometry pass run BESIDE the training step's weight-gradient workgroups, next to which
    #  packed fp32 loses a row now and then — DESIGN.md §4; what pairs up scalar float code into v_pk_*_f32 is the SLP vectoriser.
    #  The sampling kernels' stand-alone instantiations keep their explicit two-point arithmetic: vector types, not SLP.)
    "sampling.hip": ["-ffp-contract=off", "-fno-slp-vectorize"],
    "neighbors.hip": ["-ffp-contract=off", "-fno-slp-vectorize"],
    "gather.hip": ["-ffp-contract=off"],
    "fitters.hip": [],
    "fit_algebra.hip": ["-ffp-contract=off"],
    "mlp_fwd.hip": [],
    "mlp_small.hip": [],
    # (-fno-slp-vectorize, round 6: with the SLP vectoriser's v_pk_fma_f32 the xyz weight-gradient sums that ride on the 64 <- 64
    #  one-pass kernel came out DIFFERENT from run to run in a few accumulators — 2e-4 relative, same inputs, one kernel alone on the
    #  chip, no neighbour: packed fp32 in a kernel whose other waves read their operands transposed out of LDS, the round-4 fault
    #  inside ONE kernel (tools/dbg/xw_sens3.py).  Without it: bit-identical runs, 2e-8 from an fp64 reference,
    #  same registers, +2 us per step.  CPFN_BWD_SLP=1 builds the faulty form for the reproducer.)
    "mlp_bwd_fused.hip": [] if os.environ.get("CPFN_BWD_SLP") == "1" else ["-fno-slp-vectorize"],
    "bn.hip": [],
    "losses.hip": [],
    "merging.hip": [],
    "metrics.hip": [],
    "optim.hip": [],
}
COMMON = ["--offload-arch=" + ARCH, "-O3", "-fPIC", "-std=c++17", "-fvisibility=hidden",
          "-Wall", "-Wno-unused-function"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _deps(src):
    d = [os.path.join(CSRC, src), os.path.join(HERE, "..", "include", "cpfn_hip.h")]
    d += [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    return d


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def _compile(src, flags, force):
    obj = os.path.join(OBJ, src.replace(".hip", ".o"))
    if force or _stale(obj, _deps(src)):
        cmd = [_hipcc()] + COMMON + flags + ["-save-temps=obj", "-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, " ".join(cmd), r.stderr[-4000:]))
        _check_isa(src)
        return obj, True
    return obj, False


# 96-bit DS instructions are banned: ds_read_b96 was measured to return wrong data on gfx950 while another
# kernel's workgroup on the same CU writes LDS heavily (csrc/common.h: cpfn_lds_read4).  The compiler emits
# them when three of four floats of a 16-byte LDS element are used; the device assembly kept by
# -save-temps is scanned after every compile so that a new one cannot slip in.
_BANNED_ISA = re.compile(r"\bds_(read|write|load|store)_b96\b")
# Scratch (private segment) is banned too: a dynamically indexed local array silently moves to scratch memory — it cost
# the Adam kernel 25 us per launch, and the fitters' algebra 44 MB of HBM writes per backward launch until round 3 made its
# last three run-time indices (Jacobi's (p, q), the eigenvalue sort's permutation, the plane frame's `pick`) static.
_SCRATCH_OK = ()
# Packed fp32 is banned from the kernels that run beside a training step's backward pass (the next batch's geometry): all of
# neighbors.hip, and the sampling instantiations without it (fps_resident_kernel<.., PROFILE = false, PK = false>).
_PACKED_F32 = re.compile(r"\bv_pk_(add|mul|fma)_f32\b")
_NO_PACKED = {"neighbors.hip": lambda name: True,
              "mlp_bwd_fused.hip": lambda name: os.environ.get("CPFN_BWD_SLP") != "1",
              "sampling.hip": lambda name: "fps_resident_kernel" in name and "ELb0ELb0EEEv" in name}     # <.., PROFILE = false, PK = false>
_KERNEL_BODY = re.compile(r"^(_Z\w+):[^\n]*\n(.*?)^\.Lfunc_end", re.M | re.S)
_KERNEL_META = re.compile(r"\.name:\s+(\S+)\s*\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)")


def _check_isa(src):
    stem = src.replace(".hip", "")
    temps = [f for f in os.listdir(OBJ) if (f.startswith(stem + "-hip-") or f.startswith(stem + "-host-") or
                                            f.startswith(src + "-hip-")) and f != stem + ".o"]
    asm = [f for f in temps if f.startswith(stem + "-hip-amdgcn") and f.endswith(".s")]
    try:
        if not asm:
            raise RuntimeError("device assembly of %s not found under %s (needed for the ISA check)" % (src, OBJ))
        text = open(os.path.join(OBJ, asm[0])).read()
        hits = _BANNED_ISA.findall(text)
        if hits:
            os.remove(os.path.join(OBJ, stem + ".o"))
            raise RuntimeError("%s: %d banned 96-bit DS instruction(s) in the gfx950 code (see csrc/common.h, "
                               "cpfn_lds_read4)" % (src, len(hits)))
        if src in _NO_PACKED:
            packed = [(n, len(_PACKED_F32.findall(body))) for n, body in _KERNEL_BODY.findall(text)
                      if _NO_PACKED[src](n) and _PACKED_F32.search(body)]
            if packed:
                os.remove(os.path.join(OBJ, stem + ".o"))
                raise RuntimeError("%s: packed fp32 in kernels that run beside a training step: %s (DESIGN.md section 4)" % (src, packed))
            if src == "sampling.hip" and not any(_NO_PACKED[src](n) for n, _ in _KERNEL_BODY.findall(text)):
                os.remove(os.path.join(OBJ, stem + ".o"))
                raise RuntimeError("sampling.hip: the instantiations without packed fp32 were not found (mangled name changed?)")
        spilled = [(n, int(b)) for n, b in _KERNEL_META.findall(text) if int(b) > 0 and not any(k in n for k in _SCRATCH_OK)]
        if spilled:
            os.remove(os.path.join(OBJ, stem + ".o"))
            raise RuntimeError("%s: kernels using scratch memory: %s (dynamically indexed local array or register "
                               "spill; restructure, or add to _SCRATCH_OK with a reason)" % (src, spilled))
    finally:
        for f in temps:                      # -save-temps leaves ~8 MB per source; only the object is kept
            os.remove(os.path.join(OBJ, f))


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    srcs = {s: f for s, f in SOURCES.items() if os.path.exists(os.path.join(CSRC, s))}
    with ThreadPoolExecutor(max_workers=min(6, len(srcs))) as ex:
        res = list(ex.map(lambda kv: _compile(kv[0], kv[1], force), srcs.items()))
    objs = [o for o, _ in res]
    if force or any(c for _, c in res) or _stale(SO, objs):
        cmd = [_hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", SO] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s" % r.stderr[-4000:])
        if verbose:
            print("linked", SO)
    return SO


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
