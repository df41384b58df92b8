"""Kernel timeline of ONE synchronous kp_fit (KP_NO_ASYNC): run under `rocprofv3 --kernel-trace --output-format csv -d DIR -- python3
tools/fit_timeline.py`, then `python tools/fit_timeline.py DIR` prints start offset, duration and the gap before every kernel of
the last fit."""
import csv, glob, os, sys
if len(sys.argv) > 1:
    f = sorted(glob.glob(os.path.join(sys.argv[1], "*", "*_kernel_trace.csv")), key=os.path.getmtime)[-1]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if "kp_gram3_kernel" in r["Kernel_Name"]]
    last = rows[idx[-1]:]
    t0, prev_end = int(last[0]["Start_Timestamp"]), None
    for r in last:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f} us  gap {0.0 if prev_end is None else (s - prev_end) / 1e3:7.1f} us  {r['Kernel_Name'][:60]}")
        prev_end = e
    print(f"first start -> last end: {(prev_end - t0) / 1e3:.1f} us")
    sys.exit(0)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["KP_NO_ASYNC"] = "1"
import time, numpy as np, koopman_realizations_amd as kra, bench
ctx = kra.Context(0); a, b, u = bench.synth_pairs(100000)
basis = kra.Basis(ctx, "bilinear", 6, 3, [("poly", kra.poly_exponent_table(6, 3)[6:])]); snaps = kra.Snapshots(ctx, a, b, u)
for _ in range(30):
    kra.fit(ctx, basis, snaps, fetch=False)
lat = []
for _ in range(8):
    t1 = time.perf_counter(); kra.fit(ctx, basis, snaps, fetch=False); lat.append(time.perf_counter() - t1)
print("sync fit latency ms", np.median(lat) * 1e3)
