"""Wide-dictionary points of bench.py alone (fourier degree 1 on six states: linear W = 738, bilinear W = 2 940): Gram pass and solve
times of synchronous fits.  KP_LIB_PATH=<experimental build> for A/B runs."""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import koopman_realizations_amd as kra
import bench
ctx = kra.Context(0)
r = bench.bench_wide_points(ctx, kra)
for k, v in r.items():
    print(os.environ.get("KP_LIB_PATH", "default").split("/")[-1], k, "gram_ms %.3f solve_ms %.3f fit_ms %.3f frac %.3f" % (v["gram_ms"], v["solve_ms"], v["ms_per_fit"], v["roofline"]["frac"]))
