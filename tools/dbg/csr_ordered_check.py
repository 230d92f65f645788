"""The "ordered" inverse index (CPFN_CSR_THREADS < 0: the default) against numpy's stable argsort on adversarial inputs, and its fall-back count."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.getcwd())
from cpfn_amd import ops          # noqa: E402

assert ops.CSR_THREADS < 0, "run with CPFN_CSR_THREADS < 0"
dev = torch.device("cuda:0")
rng = np.random.default_rng(5)
bad = 0
for name, B, E, M, gen in [
        ("one target", 4, 8192, 512, lambda: np.full((4, 8192), 7)),
        ("two targets alternating", 4, 8192, 512, lambda: np.tile(np.arange(8192) % 2 * 300, (4, 1))),
        ("uniform", 16, 24576, 512, lambda: rng.integers(0, 512, (16, 24576))),
        ("uniform small", 16, 1536, 128, lambda: rng.integers(0, 128, (16, 1536))),
        ("skewed", 16, 24576, 512, lambda: np.minimum(rng.geometric(0.02, (16, 24576)) - 1, 511)),
        ("runs of 64", 8, 8192, 512, lambda: np.repeat(rng.integers(0, 512, (8, 128)), 64, axis=1)),
        ("odd sizes", 3, 1000, 77, lambda: rng.integers(0, 77, (3, 1000))),
        ("out of range", 2, 4096, 64, lambda: rng.integers(-5, 70, (2, 4096)))]:
    idx = gen().astype(np.int32)
    off, ent = ops.csr_build(torch.from_numpy(idx).to(dev), M)
    off, ent = off.cpu().numpy(), ent.cpu().numpy()
    cl = np.clip(idx, 0, M - 1)
    ok = True
    for b in range(B):
        ok &= np.array_equal(ent[b], np.argsort(cl[b], kind="stable"))
        ok &= np.array_equal(off[b], np.concatenate([[0], np.cumsum(np.bincount(cl[b], minlength=M))]))
    print("%-26s %s" % (name, "ok" if ok else "DIFFERENT"))
    bad += not ok
print("fall-backs:", ops.csr_fallbacks(), " different:", bad)
sys.exit(1 if bad else 0)
