#!/bin/bash
# A/B of two library builds on the same box: tools/ab.sh <alt .so>
cd $GRAFT_REPO_ROOT
L=koopman-realizations_amd/libkoopman_hip.so
cp $L /tmp/new.so
for rep in 1 2; do
for V in new alt; do
  if [ $V = alt ]; then cp $1 $L; else cp /tmp/new.so $L; fi
  echo -n "$V "; python bench.py --no-cpu-baseline --no-mpc --steps 100 --warmup 5 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['fit_latency_ms'], d['kernel_ms'])"
done; done
cp /tmp/new.so $L
