"""CPU: the C-ABI library builds, loads and exports every symbol include/cpfn_hip.h
declares (no compute calls — there is no GPU here)."""
import ctypes
import os
import re

from conftest import ROOT


def _declared():
    text = open(os.path.join(ROOT, "include", "cpfn_hip.h")).read()
    return sorted(set(re.findall(r"CPFN_API\s+[\w\s\*]+?\b(cpfn_\w+)\s*\(", text)))


def test_header_symbols_exported_and_bound():
    from cpfn_amd import build, lib
    so = build.build()
    assert os.path.exists(so)
    h = ctypes.CDLL(so)
    names = _declared()
    assert len(names) >= 15
    for n in names:
        assert hasattr(h, n), "symbol %s declared in cpfn_hip.h but not exported" % n
        assert n in lib.SIGNATURES, "symbol %s has no ctypes prototype in cpfn_amd/lib.py" % n
    assert set(lib.SIGNATURES) == set(names)
    header = open(os.path.join(ROOT, "include", "cpfn_hip.h")).read()
    version = int(re.search(r"#define\s+CPFN_ABI_VERSION\s+(\d+)", header).group(1))
    assert lib.lib().cpfn_abi_version() == version == lib.ABI_VERSION
    assert b"gfx950" in lib.lib().cpfn_build_info()


def test_ctypes_prototypes_have_the_header_s_argument_counts():
    """A signature that changes in the header but not in lib.py (or the other way round) is a silent argument mismatch
    for ctypes: every prototype must at least have the declared number of parameters."""
    from cpfn_amd import lib
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "cpfn_hip.h")).read(), flags=re.S)
    decls = re.findall(r"CPFN_API\s+[\w\s\*]+?\b(cpfn_\w+)\s*\(([^;]*?)\)\s*;", text)
    assert len(decls) == len(_declared())
    for name, params in decls:
        params = params.strip()
        n = 0 if params in ("", "void") else params.count(",") + 1
        assert n == len(lib.SIGNATURES[name]), "%s: header declares %d parameters, lib.py binds %d" % (
            name, n, len(lib.SIGNATURES[name]))


def test_product_never_imports_oracle():
    """The product path must not route through the oracle (or any CPU fallback)."""
    pkg = os.path.join(ROOT, "cpfn_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f
                assert "libcpfn_oracle" not in src, f
                assert "cpu_standins" not in src, f          # (the CPU stand-ins of the gloo tests are test infrastructure too)


def test_bad_arguments_raise_runtime_error():
    import pytest
    import torch
    from cpfn_amd import cuda_ops
    with pytest.raises(RuntimeError, match="CPU not supported"):
        cuda_ops.farthest_point_sampling(torch.zeros(1, 8, 3), 2)
    with pytest.raises(RuntimeError, match="CPU not supported"):
        cuda_ops.ball_query(torch.zeros(1, 2, 3), torch.zeros(1, 8, 3), 0.2, 4)


def test_build_rejects_packed_fp32_beside_a_step_and_96_bit_lds_reads():
    """cpfn_amd/build.py scans the device assembly of every compile (two hardware interactions found next to the
    weight-gradient kernels).  The patterns, on synthetic assembly: a 96-bit DS read anywhere; packed fp32 in ANY kernel of
    neighbors.hip and in the sampling instantiations meant to be free of it — and not in the stand-alone ones."""
    from cpfn_amd import build as b
    assert b._BANNED_ISA.search("\tds_read_b96 v[0:2], v3\n") and not b._BANNED_ISA.search("\tds_read_b128 v[0:3], v4\n")
    beside = "_ZN12_GLOBAL__N_119fps_resident_kernelILi256ELi8ELb0ELb0EEEvPKfiiPKiiPiPyPf"
    alone = "_ZN12_GLOBAL__N_119fps_resident_kernelILi256ELi8ELb0ELb1EEEvPKfiiPKiiPiPyPf"
    text = ("%s: ; @x\n\tv_sub_f32_e32 v1, v2, v3\n\tv_pk_mul_f32 v[0:1], v[2:3], v[2:3]\n.Lfunc_end0:\n"
            "%s: ; @y\n\tv_pk_add_f32 v[0:1], v[2:3], v[4:5]\n\tv_pk_mov_b32 v[0:1], v[2:3], v[4:5]\n.Lfunc_end1:\n" % (beside, alone))
    bodies = dict(b._KERNEL_BODY.findall(text))
    assert set(bodies) == {beside, alone}
    sel = b._NO_PACKED["sampling.hip"]
    assert sel(beside) and not sel(alone)
    assert len(b._PACKED_F32.findall(bodies[beside])) == 1 and len(b._PACKED_F32.findall(bodies[alone])) == 1   # (v_pk_mov is not arithmetic)
    assert b._NO_PACKED["neighbors.hip"]("anything")
    assert "-fno-slp-vectorize" in b.SOURCES["neighbors.hip"] and "-fno-slp-vectorize" in b.SOURCES["sampling.hip"]


def test_background_geometry_switch_is_per_thread():
    """cpfn_set_background_geometry is read at launch time by the launching thread: as a process-global (until round 5) two threads
    capturing at once got each other's setting.  Host-side state only: no GPU needed."""
    import ctypes
    import threading
    from cpfn_amd import build
    h = ctypes.CDLL(build.SO)
    h.cpfn_set_background_geometry.argtypes = [ctypes.c_int]
    h.cpfn_set_background_geometry.restype = ctypes.c_int
    assert h.cpfn_set_background_geometry(1) == 0
    seen = []

    def other():
        seen.append(h.cpfn_set_background_geometry(0))          # this thread's own setting: still the default
        seen.append(h.cpfn_set_background_geometry(1))
        seen.append(h.cpfn_set_background_geometry(1))
    t = threading.Thread(target=other)
    t.start()
    t.join()
    assert seen == [0, 0, 1]
    assert h.cpfn_set_background_geometry(0) == 1               # ... and the first thread's is untouched by the other's
