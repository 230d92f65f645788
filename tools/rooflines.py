#!/usr/bin/env python
"""Per-kernel-family roofline table of one training step (profiles/r02_rooflines.json) from

  --trace  DIR   rocprofv3 --kernel-trace of `bench.py` with graph replay  -> launches/step and us/step of the REPLAYED steps
  --fetch  DIR   rocprofv3 --pmc FETCH_SIZE  of `bench.py --no-graphs`     -> HBM read bytes per launch (x2 on gfx950)
  --write  DIR   rocprofv3 --pmc WRITE_SIZE  of `bench.py --no-graphs`     -> HBM write bytes per launch
  --mfma   DIR   rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_VALU
                 SQ_WAVE_CYCLES (optional)                                   -> matrix-core / vector-ALU occupancy
  --census FILE  `bench.py --census-out FILE`: algorithmic bytes per entry point and step

Counter corrections as MI355X_MICROARCH.md prescribes: FETCH_SIZE / WRITE_SIZE are KiB; FETCH_SIZE is doubled on gfx950
(it tallies 128-B requests at 64 B for wide coalesced reads).  The counter passes run with eager launches (a replayed graph's
dispatches carry no per-kernel counters); the same kernels run in both.
"""
import argparse, collections, csv, glob, json, re, sys

# kernel function name -> C-ABI entry point that launches it (one entry point may launch several kernels)
ENTRY = {
    "mlp_gemm_stream_kernel": "cpfn_mlp_gemm", "mlp_gemm_smallp_kernel": "cpfn_mlp_gemm", "mlp_gemm_kernel": "cpfn_mlp_gemm",
    "mlp_wgrad_kernel": "cpfn_mlp_wgrad", "mlp_bwd_fused_kernel": "cpfn_mlp_bwd_fused", "mlp_bwd_small_kernel": "cpfn_mlp_bwd_small",
    "multi_split_reduce_kernel": "cpfn_multi_split_reduce",
    "bn_finalize_kernel": "cpfn_bn_finalize", "bn_bwd_finalize_kernel": "cpfn_bn_bwd_finalize",
    "bn_bwd_finalize_ride_kernel": "cpfn_bn_bwd_finalize_ride",
    "bn_relu_apply_kernel": "cpfn_bn_relu_apply", "bn_relu_maxpool_kernel": "cpfn_bn_relu_maxpool",
    "bn_relu_bwd_kernel": "cpfn_bn_relu_bwd", "bn_bwd_apply_kernel": "cpfn_bn_bwd_apply",
    "bn_pool_bwd_apply_kernel": "cpfn_bn_pool_bwd_apply", "smallk_fwd_kernel": "cpfn_smallk_fwd", "smallk_fwd_cast_kernel": "cpfn_smallk_fwd",
    "smallk_wgrad_kernel": "cpfn_smallk_wgrad", "colsum_f32_kernel": "cpfn_colsum_f32",
    "csr_gather_sum_kernel": "cpfn_csr_gather_sum_bf16", "group_concat_bf16_kernel": "cpfn_group_concat_bf16",
    "interp_rows_bf16_kernel": "cpfn_interp_rows_bf16", "concat_pos_feats_kernel": "cpfn_concat_pos_feats_bf16",
    "moments_fwd_kernel": "cpfn_fit_moments_fwd", "fit_params_bwd_algebra_kernel": "cpfn_fit_algebra_bwd", "moments_bwd_kernel": "cpfn_fit_moments_bwd",
    "cone_fwd_kernel": "cpfn_cone_pass_fwd", "cone_bwd_kernel": "cpfn_cone_pass_bwd",
    "fit_algebra_fwd_kernel": "cpfn_fit_algebra_fwd", "fit_algebra_bwd_kernel": "cpfn_fit_algebra_bwd",
    "fit_pack_fwd_kernel": "cpfn_fit_pack_fwd", "fit_pack_bwd_kernel": "cpfn_fit_pack_bwd",
    "head_post_fwd_kernel": "cpfn_head_post_fwd", "head_post_bwd_kernel": "cpfn_head_post_bwd",
    "seg_stats_bwd_kernel": "cpfn_seg_stats_bwd", "residue_fwd_kernel": "cpfn_residue_fwd",
    "adam_flat_kernel": "cpfn_adam_flat", "multi_copy_kernel": "cpfn_multi_copy", "multi_cast_kernel": "cpfn_multi_cast",
    "fps_resident_kernel": "cpfn_fps", "fps_streaming_kernel": "cpfn_fps", "ball_query_kernel": "cpfn_ball_query",
    "three_nn_kernel": "cpfn_three_nn", "three_weights_kernel": "cpfn_three_weights", "csr_build_kernel": "cpfn_csr_build",
    "gather_rows_kernel": "cpfn_gather_rows", "group_xyz_centered_kernel": "cpfn_group_xyz_centered",
}
# entry points that launch a kernel family of another entry point: their algorithmic bytes are added to that family
CENSUS_MERGE = {"cpfn_mlp_dgrad_small": "cpfn_mlp_gemm", "cpfn_mlp_wgrad_apply": "cpfn_mlp_wgrad",
                "cpfn_smallk_wgrad_apply": "cpfn_smallk_wgrad", "cpfn_smallk_wgrad_apply_xyz": "cpfn_smallk_wgrad",
                "cpfn_mlp_bwd_fused_xyz": "cpfn_mlp_bwd_fused"}
HBM_PEAK = 8.0e12
CLOCK_HZ, N_SIMD = 2.4e9, 1024


def base(name):
    n = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return re.split(r"[<(]", n)[0].strip()


def one(d, pat):
    f = glob.glob(d + "/**/" + pat, recursive=True)
    if not f:
        sys.exit("no %s under %s" % (pat, d))
    return f[0]


def replayed(trace_dir):
    rows = list(csv.DictReader(open(one(trace_dir, "*kernel_trace.csv"))))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if "adam_flat_kernel" in r["Kernel_Name"]]
    steps = range(7, min(14, len(idx) - 1))        # steps 8..14 of the run: replayed (5 warm-up incl. capture + 1 census)
    agg = collections.defaultdict(lambda: [0, 0.0])
    for k in steps:
        for r in rows[idx[k] + 1:idx[k + 1] + 1]:
            agg[base(r["Kernel_Name"])][0] += 1
            agg[base(r["Kernel_Name"])][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    n = len(steps)
    wall = (int(rows[idx[steps[-1] + 1]]["End_Timestamp"]) - int(rows[idx[steps[0]]]["End_Timestamp"])) / 1e3 / n
    return {k: (v[0] / n, v[1] / n) for k, v in agg.items()}, wall


def pmc(d, counters):
    out = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for r in csv.DictReader(open(one(d, "*counter_collection.csv"))):
        if r["Counter_Name"] in counters:
            e = out[base(r["Kernel_Name"])][r["Counter_Name"]]
            e[0] += float(r["Counter_Value"])
            e[1] += 1
    return out


def main():
    ap = argparse.ArgumentParser()
    for a in ("trace", "fetch", "write", "census"):
        ap.add_argument("--" + a, required=True)
    ap.add_argument("--mfma", default=None)
    ap.add_argument("--traffic-out", default=None, help="also write the per-family traffic file bench.py reads")
    args = ap.parse_args()
    times, wall_us = replayed(args.trace)
    fetch, write = pmc(args.fetch, ["FETCH_SIZE"]), pmc(args.write, ["WRITE_SIZE"])
    mf = pmc(args.mfma, ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU_MFMA_MOPS_BF16", "SQ_ACTIVE_INST_VALU",
                         "SQ_WAVE_CYCLES"]) if args.mfma else {}
    census = json.load(open(args.census))
    for k, into in CENSUS_MERGE.items():
        if k in census:
            a, b = census.pop(k), census.get(into, [0, 0])
            census[into] = [a[0] + b[0], a[1] + b[1]]
    fam = collections.OrderedDict()
    for k, (n, us) in sorted(times.items(), key=lambda kv: -kv[1][1]):
        key = ENTRY.get(k, k)
        e = fam.setdefault(key, {"kernels": [], "launches_per_step": 0.0, "us_per_step": 0.0, "hbm_read": 0.0, "hbm_write": 0.0,
                                 "pmc_launches": 0, "mfma_busy": 0.0, "sq_busy": 0.0, "mops": 0.0, "valu_active": 0.0, "wave_cycles": 0.0})
        e["kernels"].append(k)
        e["launches_per_step"] += n
        e["us_per_step"] += us
        if k in fetch:
            f, w = fetch[k]["FETCH_SIZE"], write.get(k, {}).get("WRITE_SIZE", [0.0, 0])
            # per-STEP traffic of this kernel = per-launch average x launches per step
            e["hbm_read"] += 2.0 * 1024.0 * f[0] / max(f[1], 1) * n
            e["hbm_write"] += 1024.0 * w[0] / max(w[1], 1) * n
            e["pmc_launches"] += f[1]
        if k in mf:
            g = lambda c: mf[k].get(c, [0.0, 0])[0] / max(mf[k].get(c, [0.0, 0])[1], 1) * n      # per step
            e["mfma_busy"] += g("SQ_VALU_MFMA_BUSY_CYCLES")
            e["sq_busy"] += g("SQ_BUSY_CYCLES")
            e["mops"] += g("SQ_INSTS_VALU_MFMA_MOPS_BF16")
            e["valu_active"] += g("SQ_ACTIVE_INST_VALU")
            e["wave_cycles"] += g("SQ_WAVE_CYCLES")
    table, tot_alg, tot_us, tot_traffic = [], 0.0, 0.0, 0.0
    for key, e in fam.items():
        alg = census.get(key, [0, 0])
        row = {"family": key, "kernels": e["kernels"], "launches_per_step": round(e["launches_per_step"], 2),
               "us_per_step": round(e["us_per_step"], 2)}
        if alg[1]:
            row["algorithmic_bytes_per_step"] = alg[1]
            row["achieved_TBps"] = round(alg[1] / (e["us_per_step"] * 1e-6) / 1e12, 3)
            row["frac_of_hbm_peak"] = round(alg[1] / (e["us_per_step"] * 1e-6) / HBM_PEAK, 3)
            tot_alg += alg[1]
        if e["pmc_launches"]:
            row["hbm_traffic_bytes_per_step"] = round(e["hbm_read"] + e["hbm_write"])
            row["hbm_read_bytes_per_step"] = round(e["hbm_read"])
            row["hbm_write_bytes_per_step"] = round(e["hbm_write"])
            tot_traffic += e["hbm_read"] + e["hbm_write"]
            if alg[1]:
                row["traffic_over_algorithmic"] = round((e["hbm_read"] + e["hbm_write"]) / alg[1], 3)
        if e["sq_busy"]:
            # matrix-core utilisation: cycles a SIMD's MFMA pipe was busy, summed over the chip, against the kernel time
            # x 1024 SIMDs x the shader clock (nominal 2.4 GHz; the clock under load is not read here)
            simd_cycles = e["us_per_step"] * 1e-6 * CLOCK_HZ * N_SIMD
            row["mfma_busy_cycles_per_step"] = round(e["mfma_busy"])
            row["mfma_utilisation"] = round(e["mfma_busy"] / simd_cycles, 4)
            # SQ_ACTIVE_INST_VALU and SQ_WAVE_CYCLES count quad-cycles per wave: the share of a resident wave's time in
            # which it issues vector-ALU instructions
            row["valu_active_over_wave_cycles"] = round(e["valu_active"] / e["wave_cycles"], 4) if e["wave_cycles"] else None
            row["mfma_mops_bf16_per_step"] = round(e["mops"])
            # achieved matrix throughput from the op counter: MOPS counts 512 FLOP units
            row["mfma_TFLOPs"] = round(e["mops"] * 512 / (e["us_per_step"] * 1e-6) / 1e12, 1)
        tot_us += e["us_per_step"]
        table.append(row)
    out = {"what": "one replayed GlobalSPFN training step, 16 x 8192 points, bf16 (bench.py defaults), MI355X",
           "step_wall_us": round(wall_us, 1), "kernel_us_per_step_all_streams": round(tot_us, 1),
           "algorithmic_bytes_per_step_instrumented": tot_alg, "hbm_traffic_bytes_per_step_measured": round(tot_traffic),
           "step_frac_of_hbm_peak_algorithmic": round(tot_alg / (wall_us * 1e-6) / HBM_PEAK, 3),
           "step_frac_of_hbm_peak_traffic": round(tot_traffic / (wall_us * 1e-6) / HBM_PEAK, 3),
           "notes": "us from the rocprofv3 kernel trace of replayed steps; traffic = 2*FETCH_SIZE + WRITE_SIZE (KiB counters, "
                    "FETCH doubled on gfx950) from eager counter passes; algorithmic bytes = every operand read once, every result "
                    "written once (bench.py --census-out); mfma_utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (kernel time x 2.4 GHz x 1024 SIMDs)",
           "families": table}
    print(json.dumps(out, indent=1))
    if args.traffic_out:
        out_t = {"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over `bench.py --no-graphs`; FETCH_SIZE "
                         "doubled per MI355X_MICROARCH.md (gfx950), counters in KiB; tools/rooflines.py"}
        for f in ("cpfn_mlp_gemm", "cpfn_mlp_wgrad", "cpfn_mlp_bwd_fused"):
            if f in fam and fam[f]["pmc_launches"]:
                e = fam[f]
                n = e["launches_per_step"]
                out_t[f] = {"launches_per_step": n, "hbm_bytes_per_launch": (e["hbm_read"] + e["hbm_write"]) / n,
                            "fetch_bytes_per_launch_corrected_x2": e["hbm_read"] / n, "write_bytes_per_launch": e["hbm_write"] / n}
        json.dump(out_t, open(args.traffic_out, "w"), indent=1)


if __name__ == "__main__":
    main()
