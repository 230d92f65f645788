#!/usr/bin/env python
"""Register / LDS / scratch footprint of every kernel of one csrc/*.hip file (compiles it to /tmp with -save-temps
and reads the code-object metadata): `python tools/kernel_regs.py mlp_fwd.hip [filter]`.  Occupancy on CDNA4: a wave's
VGPRs + AGPRs come out of one 512-entry file per SIMD lane, so > 256 means ONE wave per SIMD."""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cpfn_amd import build as B
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
d = tempfile.mkdtemp(prefix="kregs")
subprocess.check_call([B._hipcc()] + B.COMMON + B.SOURCES.get(src, []) + ["-save-temps=obj", "-c", os.path.join(B.CSRC, src), "-o", os.path.join(d, "o.o")],
                      stderr=subprocess.DEVNULL)
asm = [f for f in os.listdir(d) if f.endswith(".s") and "amdgcn" in f][0]
t = open(os.path.join(d, asm)).read()
def demangle(n):
    try:
        return subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()
    except OSError:
        return n
for b in t.split("  - .agpr_count:")[1:]:
    name = re.search(r"\.name:\s+(\S+)", b).group(1)
    if flt and flt not in name:
        continue
    g = lambda k: re.search(r"\.%s:\s+(\d+)" % k, b).group(1)
    print("%-100s vgpr %3s agpr %3s sgpr %3s lds %6s scratch %s" % (demangle(name)[:100], g("vgpr_count"), re.match(r"\s+(\d+)", b).group(1),
          g("sgpr_count"), g("group_segment_fixed_size"), g("private_segment_fixed_size")))
