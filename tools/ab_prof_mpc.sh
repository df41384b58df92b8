#!/bin/bash
# MPC phase profile (tools/prof_mpc.py) for the current library and alternatives: tools/ab_prof_mpc.sh <alt.so> ...
cd $GRAFT_REPO_ROOT
L=koopman-realizations_amd/libkoopman_hip.so
cp $L /tmp/new.so
for V in new "$@"; do
  if [ $V = new ]; then cp /tmp/new.so $L; else cp $V $L; fi
  echo "== $V"; python tools/prof_mpc.py 2>/dev/null | tail -5
  echo "-- cold"; KP_MPC_NO_WARM=1 python tools/prof_mpc.py 2>/dev/null | tail -3
done
cp /tmp/new.so $L
