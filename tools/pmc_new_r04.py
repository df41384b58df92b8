#!/usr/bin/env python3
"""Condenses gpurun_out/prof_r04_new (tools/prof_r04_new_kernels.sh) into profiles/r04_new_kernels_pmc_summary.json: per kernel the mean of
every counter per launch (FETCH_SIZE / WRITE_SIZE in the units the counter reports: 64-byte and KB-class units are left as they
come, see the guide's gfx950 notes; the byte figures below apply FETCH_SIZE x 2 KB... no conversion is made here)."""
import csv, glob, json, os, re
from collections import defaultdict
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof_r04_new")
short = lambda n: re.sub(r"\(.*$", "", re.sub(r"^void ", "", n.replace("(anonymous namespace)::", "")))
out = {"command": "rocprofv3 --pmc <set> --kernel-trace -- python3 tools/{lasso_illcond_probe.py 2 1 bilinear | prelift_time.py | prelift_ext_time.py} (tools/prof_r04_new_kernels.sh)", "kernels": {}}
want = ("kp_lasso_path_kernel", "kp_gram3_prelift_kernel", "kp_gram3_prelift_ext_kernel", "kp_gram3_kernel<3, 3, false, false, true>")
for d in sorted(glob.glob(os.path.join(SRC, "*_*"))):
    if not os.path.isdir(d):
        continue
    fs = glob.glob(os.path.join(d, "*", "*_counter_collection.csv"))
    if not fs:
        continue
    acc = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        k = short(r["Kernel_Name"])
        if any(k.startswith(w) for w in want):
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    tr = glob.glob(os.path.join(d, "*", "*_kernel_trace.csv"))
    dur = defaultdict(list)
    if tr:
        for r in csv.DictReader(open(tr[0])):
            k = short(r["Kernel_Name"])
            if any(k.startswith(w) for w in want):
                dur[k].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
    for k, cs in acc.items():
        e = out["kernels"].setdefault(k, {})
        for c, v in cs.items():
            e[c] = sum(v) / len(v)
        if dur[k]:
            e.setdefault("us_under_counters", {})[os.path.basename(d)] = sum(dur[k]) / len(dur[k])
            e["launches"] = len(dur[k])
for k, v in out["kernels"].items():
    if v.get("SQ_LDS_IDX_ACTIVE"):
        v["lds_conflict_fraction"] = v.get("SQ_LDS_BANK_CONFLICT", 0.0) / v["SQ_LDS_IDX_ACTIVE"]
    if v.get("SQ_WAVE_CYCLES") and v.get("SQ_INSTS_VALU") is not None:
        v["valu_insts_per_wave_cycle"] = v["SQ_INSTS_VALU"] / v["SQ_WAVE_CYCLES"]
json.dump(out, open(os.path.join(ROOT, "profiles", "r04_new_kernels_pmc_summary.json"), "w"), indent=1)
print(json.dumps(out, indent=1)[:4000])
