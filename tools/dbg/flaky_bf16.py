import sys, os, json, torch
sys.path.insert(0, "tools")
import bf16_vs_fp32_training as cmp
for i in range(8):
    res = cmp.run(steps=400, B=8, N=4096, n_train=24, n_held=8, dev=torch.device("cuda:0"), floor_scale=2.0)
    bad = {k: {a: round(b, 4) for a, b in v.items()} for k, v in res["comparison"].items() if v["abs_diff_of_means"] > v["allowed"]}
    worst = max(res["comparison"].items(), key=lambda kv: kv[1]["abs_diff_of_means"] / kv[1]["allowed"])
    print(i, "ok" if res["ok"] else "FAIL", bad, "worst ratio", worst[0], round(worst[1]["abs_diff_of_means"] / worst[1]["allowed"], 2),
          "mIoU", [round(res[r]["metrics"]["mIoU"], 3) for r in ("bf16_seedA", "bf16_seedB", "fp32_seedA", "fp32_seedB")], flush=True)
