#!/bin/bash
# timing-only ablations of kp_chol_kernel (KP_CHOL_ABL: 1 no trailing tiles, 2 no diagonal block, 3 no L21)
cd $GRAFT_REPO_ROOT
L=koopman-realizations_amd/libkoopman_hip.so
cp $L /tmp/new.so
for V in 0 1 2 3; do
  if [ $V != 0 ]; then cp tools/libkp_chol$V.so $L; else cp /tmp/new.so $L; fi
  python - <<PY
import sys, numpy as np, time
sys.path.insert(0,'.')
import koopman_realizations_amd as kra
ctx=kra.Context(0)
rng=np.random.default_rng(0); P=rng.standard_normal((2000,336)); G=P.T@P; C=rng.standard_normal((336,336))
t=[]
for i in range(10):
    try: ctx.fit_solve(G,C)
    except Exception as e: pass
    t.append(ctx.timer(1))
print("chol ablation $V solve ms", np.round(np.min(t[2:]),4))
PY
done
cp /tmp/new.so $L
