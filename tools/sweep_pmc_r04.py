#!/usr/bin/env python3
"""Condenses gpurun_out/prof_r04_sweep (tools/prof_r04.sh) into profiles/r04_sweep_pmc_summary.json and copies the kernel
statistics to profiles/r04_sweep_kernel_stats.csv: per kernel the mean of every counter per launch, and for the Gram pass the
VALU instructions per f64 MFMA (SQ_INSTS_VALU counts the MFMAs too: the ratio minus one is what rides beside each MFMA)."""
import csv, glob, json, os, re, shutil
from collections import defaultdict
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof_r04_sweep")
short = lambda n: re.sub(r"\(.*$", "", re.sub(r"^void ", "", n))
OLD = json.load(open(os.path.join(ROOT, "profiles", "r04_sweep_pmc_summary.json"))) if os.path.exists(os.path.join(ROOT, "profiles", "r04_sweep_pmc_summary.json")) else {"kernels": {}}
out = {"command": "rocprofv3 --pmc <set> --kernel-trace -- python3 tools/sweep_profile.py 1024 (one run per counter set, tools/prof_r04.sh)", "kernels": defaultdict(dict)}
for d in ("pmc_valu", "pmc_lds", "pmc_fetch", "pmc_write"):
    fs = glob.glob(os.path.join(SRC, d, "*", "*_counter_collection.csv"))
    if not fs:
        continue
    acc = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(max(fs, key=os.path.getmtime))):
        acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        if k.startswith("kp_"):
            for c, v in cs.items():
                out["kernels"][k][c] = sum(v) / len(v)
st = glob.glob(os.path.join(SRC, "trace", "*", "*_kernel_stats.csv"))
if st:
    st = max(st, key=os.path.getmtime)
    for r in csv.DictReader(open(st)):
        k = short(r["Name"])
        if k.startswith("kp_"):
            out["kernels"][k]["avg_us"] = float(r["AverageNs"]) / 1e3
            out["kernels"][k]["max_us"] = float(r["MaxNs"]) / 1e3
            out["kernels"][k]["calls"] = int(r["Calls"])
    shutil.copy(st, os.path.join(ROOT, "profiles", "r04_sweep_kernel_stats.csv"))
out["kernels"] = dict(out["kernels"])
# a counter pass that did not finish (the LDS set sat past its 300 s limit on the last run): its counters are carried over from the
# previous summary and marked
for k, v in out["kernels"].items():
    if "SQ_INSTS_LDS" not in v and "SQ_INSTS_LDS" in OLD["kernels"].get(k, {}):
        for c in ("SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_WAIT_INST_ANY"):
            if c in OLD["kernels"][k]:
                v[c] = OLD["kernels"][k][c]
        v["lds_counters_from"] = "the previous run of this script (kernel before the pair permutation of the operand reads)"
for k, v in out["kernels"].items():
    if "SQ_INSTS_VALU" in v and v.get("SQ_INSTS_VALU_MFMA_MOPS_F64"):
        v["valu_per_mfma_incl_the_mfma"] = v["SQ_INSTS_VALU"] / v["SQ_INSTS_VALU_MFMA_MOPS_F64"]
    if v.get("SQ_LDS_IDX_ACTIVE"):
        v["lds_conflict_fraction"] = v.get("SQ_LDS_BANK_CONFLICT", 0.0) / v["SQ_LDS_IDX_ACTIVE"]
json.dump(out, open(os.path.join(ROOT, "profiles", "r04_sweep_pmc_summary.json"), "w"), indent=1)
for k, v in out["kernels"].items():
    if "gram" in k:
        print(k, json.dumps(v, indent=1))
