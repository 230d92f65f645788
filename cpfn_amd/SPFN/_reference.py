"""Pass-through to the reference's OWN host-side SPFN helpers when this package stands in for `SPFN`
(cpfn_amd.dropin).

The reference's data path and evaluation scripts call numpy-only helpers that live in the same modules as the
device math this package replaces (Utils/dataset_utils.py:79 `fitter_factory.create_primitive_from_dict`,
:112-120 `{plane,sphere,cylinder,cone}_fitter.extract_parameter_data_as_dict`,
SPFN/metric_implementation.py:593-599 `*_fitter.extract_predicted_parameters_as_json`) and in
SPFN/primitives.py.  They are host-side GT parsing / JSON export: out of scope to rebuild —
and they must not be shadowed either.  So any name this package does not define is looked up in the
reference's own file, loaded from wherever the user's checkout lies on `sys.path`.  Nothing is copied or
restated; without a checkout on `sys.path` the lookup raises an AttributeError that says so.

How a reference file is loaded.  The reference's files import each other by absolute name
(`from SPFN.primitives import Plane`, `from SPFN.geometry_utils import weighted_plane_fitting_tensorflow`),
and `SPFN` may at that moment be aliased to this package.  The file is therefore executed with `SPFN` and
`SPFN.*` in `sys.modules` TEMPORARILY bound to a private package whose `__path__` is the reference's `SPFN/`
directory, so that the reference's helper sees the reference's siblings; afterwards the previous bindings are
put back.  The privately loaded modules are cached and reused by later loads (one `Plane` class for everybody).
Submodules this package does not override at all (`primitives`) are in addition published as
`SPFN.<name>` / `cpfn_amd.SPFN.<name>`, so that `from SPFN.primitives import Plane` in user code gives the class
the helpers test with `isinstance`.
"""
import importlib
import importlib.util
import os
import sys
import threading
import types

_HERE = os.path.dirname(os.path.abspath(__file__))
_PKG = __name__.rsplit(".", 1)[0]                      # "cpfn_amd.SPFN"
_lock = threading.RLock()
_private = {}                                          # "SPFN.<name>" -> privately loaded reference module
_ref_dir = None


def reference_dir():
    """The reference's `SPFN/` directory: the first `<root>/SPFN/primitives.py` on sys.path that is not ours."""
    global _ref_dir
    if _ref_dir is not None and os.path.isfile(os.path.join(_ref_dir, "primitives.py")):
        return _ref_dir
    for root in list(sys.path):
        cand = os.path.abspath(os.path.join(root or ".", "SPFN"))
        if cand != _HERE and os.path.isfile(os.path.join(cand, "primitives.py")):
            _ref_dir = cand
            return cand
    raise ImportError("no reference checkout on sys.path (looked for <root>/SPFN/primitives.py): put the root of "
                      "erictuanle/CPFN on sys.path — it is there when the reference's scripts run from their directory")


def overridden(name):
    """True if this package has its own `<name>.py` (the device path); False for the reference-only files."""
    return os.path.isfile(os.path.join(_HERE, name + ".py"))


def reset():
    """Forget the located checkout and every privately loaded module (tests)."""
    global _ref_dir
    with _lock:
        _private.clear()
        _ref_dir = None
        pkg = sys.modules.get(_PKG)
        for key in [k for k in sys.modules if k.startswith(("SPFN.", _PKG + "."))]:
            mod = sys.modules[key]
            if getattr(mod, "__cpfn_reference_file__", False):
                del sys.modules[key]
                if pkg is not None and pkg.__dict__.get(key.rsplit(".", 1)[1]) is mod:
                    delattr(pkg, key.rsplit(".", 1)[1])
        if pkg is not None:
            pkg.__path__[:] = [p for p in pkg.__path__ if os.path.abspath(p) == _HERE]


def reference_module(name):
    """The reference's own `SPFN/<name>.py` as a module object (loaded once)."""
    key = "SPFN." + name
    with _lock:
        if key in _private:
            return _private[key]
        rdir = reference_dir()
        if not os.path.isfile(os.path.join(rdir, name + ".py")):
            raise ImportError("the reference has no SPFN/%s.py (looked in %s)" % (name, rdir))
        saved = {k: m for k, m in sys.modules.items() if k == "SPFN" or k.startswith("SPFN.")}
        for k in saved:
            del sys.modules[k]
        shadow = types.ModuleType("SPFN")
        shadow.__path__ = [rdir]
        shadow.__cpfn_reference_file__ = True
        sys.modules["SPFN"] = shadow
        # modules of the reference that are already loaded — privately, or publicly because we do not override them
        for k, m in list(_private.items()) + [(k, m) for k, m in saved.items()
                                              if getattr(m, "__cpfn_reference_file__", False) and k != "SPFN"]:
            sys.modules[k] = m
            setattr(shadow, k.rsplit(".", 1)[1], m)
        try:
            importlib.import_module(key)
            for k, m in list(sys.modules.items()):
                if k.startswith("SPFN.") and k not in _private:
                    m.__cpfn_reference_file__ = True
                    _private[k] = m
        finally:
            for k in [k for k in sys.modules if k == "SPFN" or k.startswith("SPFN.")]:
                del sys.modules[k]
            sys.modules.update(saved)
        _publish_unoverridden()
        return _private[key]


def _publish_unoverridden():
    """Reference-only submodules (`primitives`) become `SPFN.<name>` / `cpfn_amd.SPFN.<name>` for everybody."""
    pkg = sys.modules.get(_PKG)
    for key, mod in _private.items():
        name = key.rsplit(".", 1)[1]
        if overridden(name) or pkg is None:
            continue
        sys.modules.setdefault(_PKG + "." + name, mod)
        if sys.modules.get("SPFN") is pkg:
            sys.modules.setdefault(key, mod)
        if name not in pkg.__dict__:
            setattr(pkg, name, mod)


# What may fall through to the reference's own file: host-side GT parsing / JSON export and the TensorFlow twins of its
# self-checks.  Everything else a module of this package does not define RAISES: a compute function of the reference that
# silently ran op by op from the reference's file (`guarded_matrix_solve_ls`, whose reference body calls the removed
# `torch.solve`) is worse than an error that names it.
PASS_THROUGH = ("create_primitive_from_dict", "creates_json", "batched_gather")
PASS_THROUGH_PREFIXES = ("extract_",)
PASS_THROUGH_SUFFIXES = ("_tensorflow",)


def passes_through(name):
    return name in PASS_THROUGH or name.startswith(PASS_THROUGH_PREFIXES) or name.endswith(PASS_THROUGH_SUFFIXES)


def module_fallback(module_name):
    """A module-level `__getattr__` for `cpfn_amd/SPFN/<module_name>.py`: host-side helper names (`passes_through`) the
    device path does not define come from the reference's same-named file; any other name raises."""
    def __getattr__(name):
        if name.startswith("__"):
            raise AttributeError(name)
        if not passes_through(name):
            raise AttributeError("%s.%s has no %r: only host-side helpers (create_primitive_from_dict, extract_*, creates_json, "
                                 "batched_gather, *_tensorflow) are passed through to the reference's SPFN/%s.py; a compute "
                                 "function must be defined on the device path" % (_PKG, module_name, name, module_name))
        try:
            ref = reference_module(module_name)
        except ImportError as e:
            raise AttributeError("%s.%s has no %r; it is a host-side helper of the reference, and the reference's own "
                                 "SPFN/%s.py could not be loaded: %s" % (_PKG, module_name, name, module_name, e))
        try:
            return getattr(ref, name)
        except AttributeError:
            raise AttributeError("neither %s.%s nor the reference's SPFN/%s.py defines %r"
                                 % (_PKG, module_name, module_name, name)) from None
    return __getattr__


def package_getattr(name):
    """`__getattr__` of the package itself: `from SPFN import primitives` / `SPFN.primitives`."""
    if name.startswith("__"):
        raise AttributeError(name)
    if overridden(name):
        return importlib.import_module(_PKG + "." + name)
    try:
        return reference_module(name)
    except ImportError as e:
        raise AttributeError("%s has no submodule %r and the reference's SPFN/%s.py could not be loaded: %s"
                             % (_PKG, name, name, e))


def attach():
    """Called by dropin.install(): if a checkout is on sys.path already, make `import SPFN.primitives` work through
    the import system as well (the reference's directory behind ours on the package's `__path__`; our files win)."""
    pkg = importlib.import_module(_PKG)
    try:
        rdir = reference_dir()
    except ImportError:
        return None
    for name in sorted(f[:-3] for f in os.listdir(rdir) if f.endswith(".py")):
        if not overridden(name) and name != "__init__":
            try:
                reference_module(name)
            except Exception:          # a reference-only file that needs a package this machine lacks: stays lazy
                pass
    if rdir not in pkg.__path__:
        pkg.__path__.append(rdir)
    _publish_unoverridden()
    return rdir
