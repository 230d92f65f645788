"""Drop-in for the reference's `SPFN` package: the device math is this package's; the reference's host-side
helpers (`primitives`, `create_primitive_from_dict`, `extract_*`) resolve to the reference's own files
(`_reference.py`)."""
from . import _reference


def __getattr__(name):
    return _reference.package_getattr(name)
