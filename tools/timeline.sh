#!/bin/bash
# kernel timeline of the pipelined fit loop: gaps on the Gram stream and the overlap with the solve stream
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/timeline
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O -- python3 $R/bench.py --no-cpu-baseline --no-mpc --steps 30 --warmup 5 > $O.log 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob("$O/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n): return n.split("(")[0].replace("void ", "")[:28]
g = [r for r in rows if "gram3_kernel" in r["Kernel_Name"]]
g = g[10:-2]
t0 = int(g[0]["Start_Timestamp"])
per = (int(g[-1]["Start_Timestamp"]) - t0) / (len(g) - 1)
dur = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in g) / len(g)
gaps = [int(g[i + 1]["Start_Timestamp"]) - int(g[i]["End_Timestamp"]) for i in range(len(g) - 1)]
print("gram period %.1f us, duration %.1f us, gap between consecutive gram kernels mean %.1f us (min %.1f max %.1f)" % (per / 1e3, dur / 1e3, sum(gaps) / len(gaps) / 1e3, min(gaps) / 1e3, max(gaps) / 1e3))
a, b = int(g[4]["Start_Timestamp"]), int(g[7]["Start_Timestamp"])
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if a <= s < b: print("%9.1f %9.1f  %7.1f us  q%s  %s" % ((s - a) / 1e3, (e - a) / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), short(r["Kernel_Name"])))
PY
