"""Repeats the short bf16-vs-fp32 comparison of tests/test_gpu_trainer.py to see how often its band is exceeded."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tools"))
import torch
import bf16_vs_fp32_training as cmp
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    res = cmp.run(steps=400, B=8, N=4096, n_train=24, n_held=8, dev=torch.device("cuda:0"), floor_scale=float(os.environ.get("CPFN_FLOOR_SCALE", "2.5")), seeds=(11, 22, 33))
    bad = {k: (round(v["abs_diff_of_means"], 4), round(v["allowed"], 4)) for k, v in res["comparison"].items() if not v["within_band"]}
    print(rep, "ok" if res["ok"] else "FAIL", bad,
          "mIoU", [round(res[r]["metrics"]["mIoU"], 3) for r in sorted(res) if "_seed" in r], flush=True)
