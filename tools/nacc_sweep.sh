#!/bin/bash
cd $GRAFT_REPO_ROOT
for N in 8 11 12 16; do KP_GRAM2_NACC=$N python - <<PY
import sys, numpy as np
sys.path.insert(0,'.')
import koopman_realizations_amd as kra, bench
ctx=kra.Context(0); a,b,u=bench.synth_pairs(100000)
basis=kra.Basis(ctx,"bilinear",6,3,[("poly",kra.poly_exponent_table(6,3)[6:])]); snaps=kra.Snapshots(ctx,a,b,u)
t=[]
for i in range(8):
    kra.fit_gram(ctx,basis,snaps,fetch=False); t.append(ctx.timer(0))
print("nacc=$N gram ms", np.round(np.mean(t[2:]),4))
PY
done
