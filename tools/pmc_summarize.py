#!/usr/bin/env python3
"""Condenses the rocprofv3 output of tools/prof_round.sh (gpurun_out/prof_<round>/*, KP_ROUND, default r03) into
profiles/<round>_pmc_summary.json (per-kernel mean of every counter, per launch) and copies the
kernel-trace statistics to profiles/<round>_bench_kernel_stats.csv.  For each pass the newest run
(most recently written files) is used."""
import csv, glob, json, os, re, shutil, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROUND = os.environ.get("KP_ROUND", "r03")
SRC = os.path.join(ROOT, "gpurun_out", "prof_" + ROUND)


def newest(pattern):
    fs = glob.glob(pattern)
    return max(fs, key=os.path.getmtime) if fs else None          # newest run of this pass


def short(name):
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name)


def main():
    out = {"command": "rocprofv3 --pmc <counter set> --kernel-trace -- python3 bench.py --no-cpu-baseline --no-mpc --no-extras --steps 50 --warmup 5 "
                      "(one run per counter set; tools/prof_round.sh, condensed by tools/pmc_summarize.py)",
           "kernels": defaultdict(dict)}
    for d in ("pmc_fetch", "pmc_write", "pmc_mfma", "pmc_lds"):
        f = newest(os.path.join(SRC, d, "*", "*_counter_collection.csv"))
        if not f:
            continue
        acc = defaultdict(lambda: defaultdict(list))
        for r in csv.DictReader(open(f)):
            acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in acc.items():
            if not k.startswith("kp_"):
                continue
            for c, v in cs.items():
                out["kernels"][k][c] = sum(v) / len(v)
    st = newest(os.path.join(SRC, "trace", "*", "*_kernel_stats.csv"))
    if st:
        for r in csv.DictReader(open(st)):
            k = short(r["Name"])
            if k.startswith("kp_"):
                out["kernels"][k]["avg_us"] = float(r["AverageNs"]) / 1e3
                out["kernels"][k]["calls"] = int(r["Calls"])
        shutil.copy(st, os.path.join(ROOT, "profiles", ROUND + "_bench_kernel_stats.csv"))
    out["notes"] = {
        "FETCH_SIZE/WRITE_SIZE": "rocprofv3 reports kilobytes per launch; per MI355X_MICROARCH.md gfx950 FETCH_SIZE reports half the bytes of a wide "
                                 "coalesced read: it is doubled for roofline.traffic (our reads are 8 B per lane, a width the guide "
                                 "calls uncalibrated, so the figure is an upper-side estimate); WRITE_SIZE is taken as is",
        "SQ_*": "summed over the shader engines, per launch"}
    out["kernels"] = dict(out["kernels"])
    g = [k for k in out["kernels"] if k.startswith("kp_gram3_kernel")]
    if g:
        k = out["kernels"][g[0]]
        out["dominant_kernel"] = g[0]
        out["traffic_note"] = ("FETCH_SIZE/WRITE_SIZE are in KB per dispatch; per MI355X_MICROARCH.md the gfx950 FETCH_SIZE under-reports wide "
                               "coalesced reads by 2x (8-byte-per-lane reads are uncalibrated): corrected fetch = 2*FETCH_SIZE")
        out["gram_fetch_bytes_corrected"] = 2.0 * k["FETCH_SIZE"] * 1024
        out["gram_write_bytes"] = k["WRITE_SIZE"] * 1024
        out["gram_traffic_bytes_per_launch"] = out["gram_fetch_bytes_corrected"] + out["gram_write_bytes"]
        out["gram_algorithmic_bytes"] = 12000000
        out["gram_mfma_instructions"] = k["SQ_INSTS_VALU_MFMA_MOPS_F64"]
        out["gram_executed_flop"] = k["SQ_INSTS_VALU_MFMA_MOPS_F64"] * 512
        out["gram_algorithmic_flop"] = 33902400000.0
        out["gram_mfma_busy_cycles_per_instr"] = k["SQ_VALU_MFMA_BUSY_CYCLES"] / k["SQ_INSTS_VALU_MFMA_MOPS_F64"]
        out["gram_lds_conflict_fraction"] = k["SQ_LDS_BANK_CONFLICT"] / k["SQ_LDS_IDX_ACTIVE"]
    json.dump(out, open(os.path.join(ROOT, "profiles", ROUND + "_pmc_summary.json"), "w"), indent=1)
    g = [k for k in out["kernels"] if k.startswith("kp_gram3_kernel")]
    for k in g:
        print(k, json.dumps(out["kernels"][k], indent=1))


if __name__ == "__main__":
    main()
