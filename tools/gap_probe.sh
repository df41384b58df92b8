#!/bin/bash
# kernel timeline of the pipelined bench: what sits between two consecutive Gram kernels on the device
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/gap_probe
rm -rf $O; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O -- python3 $R/bench.py --no-cpu-baseline --no-mpc --no-extras --no-one-caller --steps 50 --warmup 5 > $O/run.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$O/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60], r.get("Stream_Id", r.get("Queue_Id", "?"))) for r in rows]
g = [i for i, e in enumerate(ev) if "kp_gram3_kernel" in e[2]]
# the timed region: the last 50 Gram launches that are followed by solves... print a window of 3 consecutive grams near the end
import statistics
gaps = [(ev[g[i + 1]][0] - ev[g[i]][1]) / 1e3 for i in range(len(g) - 60, len(g) - 10)]
durs = [(ev[i][1] - ev[i][0]) / 1e3 for i in g[-60:-10]]
print("gram dur us mean %.1f  gap between consecutive gram kernels us mean %.1f min %.1f max %.1f" % (statistics.mean(durs), statistics.mean(gaps), min(gaps), max(gaps)))
i0 = g[-30]
t0 = ev[i0][0]
for e in ev[i0: i0 + 40]:
    print("%9.1f %9.1f  q=%s  %s" % ((e[0] - t0) / 1e3, (e[1] - t0) / 1e3, e[3], e[2]))
PY
