#!/bin/bash
cd $GRAFT_REPO_ROOT
L=koopman-realizations_amd/libkoopman_hip.so
cp $L /tmp/new.so
for V in new alt new alt; do
  if [ $V = alt ]; then cp $1 $L; else cp /tmp/new.so $L; fi
  echo -n "$V "; python bench.py --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())['mpc']; print(d['single_steps_per_s'], d['single_kernel_us'], d['batch_problems_per_s'], d['batch_kernel_ms'], d['batch_solved'])"
done
cp /tmp/new.so $L
