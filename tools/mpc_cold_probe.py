"""The three MPC figures of bench.py alone (independent states = cold solves, closed loop = warm steps, one batch):
python tools/mpc_cold_probe.py"""
import os, sys, json, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import koopman_realizations_amd as kra, bench
ctx = kra.Context(0)
a, b, u = bench.synth_pairs(100000)
basis = kra.Basis(ctx, "bilinear", 6, 3, [("poly", kra.poly_exponent_table(6, 3)[6:])])
snaps = kra.Snapshots(ctx, a, b, u)
args = argparse.Namespace(mpc_steps=300, mpc_batch=4096)
r = bench.bench_mpc(ctx, kra, basis, snaps, args)
print({k: (round(float(v), 2) if isinstance(v, (int, float)) else None) for k, v in r.items() if isinstance(v, (int, float))})
