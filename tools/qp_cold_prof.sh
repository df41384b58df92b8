cd $GRAFT_REPO_ROOT
L=koopman-realizations_amd/libkoopman_hip.so
cp $L /tmp/new.so; cp tools/libkp_qpprof.so $L
KP_MPC_NO_WARM=1 python tools/prof_mpc.py 2>&1 | tail -30
cp /tmp/new.so $L
