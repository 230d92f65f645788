"""HBM traffic per launch of the roofline kernel family from two rocprofv3 --pmc passes
(FETCH_SIZE and WRITE_SIZE, collected separately as MI355X_MICROARCH.md prescribes).

    python tools/pmc_traffic.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> [substring] > profiles/rNN_mlp_gemm_traffic.json

Corrections applied (MI355X_MICROARCH.md, HBM / rocprofv3 section): both counters are in KiB; on gfx950
FETCH_SIZE reports half the bytes of wide coalesced reads and is doubled.
"""
import csv
import glob
import json
import sys


def per_launch(d, counter, sub):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    assert f, "no counter_collection.csv under " + d
    tot, n = 0.0, 0
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] == counter and sub in r["Kernel_Name"]:
            tot += float(r["Counter_Value"]) * 1024.0
            n += 1
    return tot / max(n, 1), n


def main():
    sub = sys.argv[3] if len(sys.argv) > 3 else "mlp_gemm"
    fetch, n1 = per_launch(sys.argv[1], "FETCH_SIZE", sub)
    write, n2 = per_launch(sys.argv[2], "WRITE_SIZE", sub)
    print(json.dumps({
        "kernel": "kernels whose name contains '%s' (entry point cpfn_mlp_gemm)" % sub,
        "launches_sampled": n1,
        "fetch_size_bytes_per_launch_raw": fetch,
        "fetch_size_bytes_per_launch_corrected_x2": 2 * fetch,
        "write_size_bytes_per_launch": write,
        "hbm_bytes_per_launch": 2 * fetch + write,
        "note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over `bench.py --steps 3 --warmup 3 "
                "--no-graphs --no-cpu-baseline`; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half the bytes "
                "of wide coalesced reads); counters are in KiB"}, indent=1))


if __name__ == "__main__":
    main()
