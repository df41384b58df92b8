#!/bin/bash
# MPC A/B over several library builds on one box: tools/ab_mpc2.sh <alt1.so> <alt2.so> ...
cd $GRAFT_REPO_ROOT
L=koopman-realizations_amd/libkoopman_hip.so
cp $L /tmp/new.so
for rep in 1 2; do
for V in new "$@"; do
  if [ $V = new ]; then cp /tmp/new.so $L; else cp $V $L; fi
  echo -n "$V "; python bench.py --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())['mpc']; print(d['single_steps_per_s'], d['single_kernel_us'], d.get('closed_loop_steps_per_s'), d['batch_problems_per_s'], d['batch_kernel_ms'], d['batch_solved'])"
done; done
cp /tmp/new.so $L
