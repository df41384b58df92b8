#!/bin/bash
# library variant with the QP solver's per-phase cycle counters printed by problem 0 (KP_QP_PROF): tools/libkp_qpprof.so
cd "$(dirname "$0")/../koopman-realizations_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I. -DKP_QP_PROF -c kp_mpc.hip -o /tmp/kp_mpc_prof.o &&
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/libkp_qpprof.so $(ls *.o | grep -v kp_mpc.o) /tmp/kp_mpc_prof.o
