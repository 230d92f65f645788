"""Repeats the two epoch tests that compare a replayed run with a second run bit for bit, in ONE process, and counts mismatches:
    python tools/dbg/flaky_epoch.py [iterations] [NAME=0 ...]
NAME=0/1 sets a module switch of cpfn_amd.fused_mlp / cpfn_amd.autograd_ops first (REDUCE_RIDE, SKIP_JOIN, ...): which change a
run-to-run difference goes away with.  Run from the root of the tree under test (tests/ is taken from there)."""
import os
import sys
import time
import traceback

sys.path.insert(0, os.getcwd())


def main():
    it = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 20
    from cpfn_amd import autograd_ops, fused_mlp
    for kv in sys.argv[1:]:
        if "=" in kv:
            k, v = kv.split("=")
            for mod in (fused_mlp, autograd_ops):
                if hasattr(mod, k):
                    setattr(mod, k, bool(int(v)))
                    print("set", mod.__name__, k, bool(int(v)), flush=True)
    from tests import test_gpu_epoch as te
    names = [n for n in ("test_local_spfn_with_feature_inputs_through_the_epoch_function",
                         "test_patch_selection_epoch_equals_trainer_steps_by_hand",
                         "test_epoch_function_equals_trainer_steps_by_hand") if hasattr(te, n)]
    bad = {n: 0 for n in names}
    t0 = time.time()
    for i in range(it):
        for n in names:
            try:
                getattr(te, n)()
            except AssertionError:
                bad[n] += 1
                msg = traceback.format_exc().splitlines()
                print("iteration", i, n, "MISMATCH:", msg[-1][:200], flush=True)
    print("RESULT", os.getcwd(), sys.argv[1:], {k[:28]: v for k, v in bad.items()}, "of", it, "in %.0f s" % (time.time() - t0), flush=True)


if __name__ == "__main__":
    main()
