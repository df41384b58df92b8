#!/bin/bash
# gram kernel timing versus quads per job (KP_GRAM3_NQ override); usage: g3_nq.sh "<NQ list>" [old]
cd $GRAFT_REPO_ROOT
[ "$2" = old ] && export KP_GRAM3_OLD=1
for NQ in ${1:-8 7 6 5 4}; do KP_GRAM3_NQ=$NQ python - <<PY
import sys, numpy as np
sys.path.insert(0,'.')
import koopman_realizations_amd as kra, bench
ctx=kra.Context(0); a,b,u=bench.synth_pairs(100000)
basis=kra.Basis(ctx,"bilinear",6,3,[("poly",kra.poly_exponent_table(6,3)[6:])]); snaps=kra.Snapshots(ctx,a,b,u)
t=[]
for i in range(10):
    kra.fit_gram(ctx,basis,snaps,fetch=False); t.append((ctx.timer(0), ctx.timer(6)))
t=np.array(t[3:]).mean(axis=0)
print("NQ=$NQ gram ms %.4f reduce ms %.4f" % (t[0], t[1]))
PY
done
