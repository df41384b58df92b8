#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_fit.py -x -q -m gpu 2>&1 | tail -3
for NQ in 8 4; do for WG in 1 2 3; do KP_GRAM3_NQ=$NQ KP_GRAM3_WGPCU=$WG python - <<PY
import sys, numpy as np
sys.path.insert(0,'.')
import koopman_realizations_amd as kra, bench
ctx=kra.Context(0); a,b,u=bench.synth_pairs(100000)
basis=kra.Basis(ctx,"bilinear",6,3,[("poly",kra.poly_exponent_table(6,3)[6:])]); snaps=kra.Snapshots(ctx,a,b,u)
t=[]
for i in range(8):
    kra.fit_gram(ctx,basis,snaps,fetch=False); t.append((ctx.timer(0), ctx.timer(6)))
t=np.array(t[2:]).mean(axis=0)
print("NQ=$NQ WGPCU=$WG gram ms %.4f reduce ms %.4f" % (t[0], t[1]))
PY
done; done
