#!/bin/bash
# timing-only ablations of the Gram kernel (results are wrong by design; library is restored afterwards)
# usage: tools/abl.sh "<ablation ids>" "<NQ values>"
cd $GRAFT_REPO_ROOT
L=koopman-realizations_amd/libkoopman_hip.so
cp $L /tmp/orig.so
for A in ${1:-0 1 5 6 7}; do if [ $A != 0 ]; then cp tools/libkp_abl$A.so $L; else cp /tmp/orig.so $L; fi
for NQ in ${2:-7}; do KP_GRAM3_NQ=$NQ python - <<PY
import sys, numpy as np
sys.path.insert(0,'.')
import koopman_realizations_amd as kra, bench
ctx=kra.Context(0); a,b,u=bench.synth_pairs(100000)
basis=kra.Basis(ctx,"bilinear",6,3,[("poly",kra.poly_exponent_table(6,3)[6:])]); snaps=kra.Snapshots(ctx,a,b,u)
t=[]
for i in range(8):
    kra.fit_gram(ctx,basis,snaps,fetch=False); t.append(ctx.timer(0))
print("ablate=$A NQ=$NQ gram ms", np.round(np.mean(t[2:]),4))
PY
done; done
cp /tmp/orig.so $L
