"""CPU oracle for the CPFN hot path — TEST INFRASTRUCTURE, never shipped.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package.  ``cpfn_amd`` (the product) never does: it fails
loudly when its HIP library is missing instead of falling back to this code.

Contents
--------
* ``cpfn_oracle.c`` / ``geometry.py`` — scalar-fp32 C restatement of the
  reference's ``fast=False`` geometry route (FPS, ball-query, 3-NN,
  interpolation, grouping), loaded through ctypes.
* ``spfn.py`` — torch-CPU restatement of the SPFN weighted-TLS fitters and the
  losses that consume them.
* ``pn2.py`` — torch-CPU restatement of the PointNet++ encoder/decoder and of
  one training step (the "reference CPU path" timed by ``bench.py``).

Parity status: **pinned** by ``tests/golden/*.npz`` — outputs of the reference's
own Python, imported in the build container by ``tests/golden/make_golden.py``.
The reference ships no executable tests or golden vectors of its own
(SURVEY.md §4), and its native extension is CUDA-only and cannot be built
here, so those fixtures are the only pin.
"""
