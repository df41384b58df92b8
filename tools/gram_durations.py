#!/usr/bin/env python3
"""Launch-by-launch durations of the dominant Gram kernel from the rocprofv3 --kernel-trace pass of tools/prof_round.sh
-> profiles/r02_gram_launch_durations.txt.  Shows why the stats file's average over ALL launches of the process is higher
than the kernel time bench.py reports for its timed region (device clock ramp after idle or lightly loaded stretches)."""
import csv, glob, os
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
fs = sorted(glob.glob(os.path.join(ROOT, "gpurun_out/prof_r02/trace/*/*_kernel_trace.csv")), key=os.path.getmtime)
rows = [r for r in csv.DictReader(open(fs[-1])) if "kp_gram3_kernel" in r["Kernel_Name"]]
d = np.array([(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows])
d = d[np.argsort([int(r["Start_Timestamp"]) for r in rows])]
PRE, WARM, FILL, STEPS = 256, 5, 32, 50        # bench.py: untimed queue of 256 fits, --warmup 5, 32 Gram-only launches, --steps 50
t0, t1 = PRE + WARM + FILL, PRE + WARM + FILL + STEPS
out = ["kp_gram3_kernel<6,3,false> launch by launch (us, launch order), rocprofv3 --kernel-trace of",
       "python3 bench.py --no-cpu-baseline --no-mpc --no-extras --steps 50 --warmup 5 (the run behind r02_bench_kernel_stats.csv).",
       f"{len(d)} launches: mean {d.mean():.1f} us = the stats file's AverageNs; min {d.min():.1f}, max {d.max():.1f}.",
       "After an idle or lightly loaded stretch (process start; the synchronous latency / fetch section with its one-workgroup",
       "Cholesky) the kernel takes ~470 us and comes down to ~405 us over the next ~60 launches (~25 ms of sustained load) as the",
       "device clock ramps up.  bench.py therefore queues 256 untimed fits before its warm-up, and 32 Gram-only launches between",
       "the warm-up's synchronisation (which ends in a light batch of factorisations) and the timed region:",
       f"launches {t0}..{t1 - 1} (the timed region): mean {d[t0:t1].mean():.1f} us, min {d[t0:t1].min():.1f}, max {d[t0:t1].max():.1f} - what bench.py's HIP events",
       "report as kernel_ms.gram and what roofline.achieved is computed from; the launches after it belong to the latency",
       "section (7 synchronous fits, slow again) and to the 3 x 64 fits streamed from host memory.", ""]
for a in range(0, len(d), 16):
    out.append(f"{a:4d}: " + " ".join(f"{x:5.0f}" for x in d[a:a + 16]))
open(os.path.join(ROOT, "profiles/r02_gram_launch_durations.txt"), "w").write("\n".join(out) + "\n")
print("\n".join(out[:9]))
