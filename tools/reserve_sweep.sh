#!/bin/bash
# pipelined fit rate versus the number of CUs kept free for the solve stream (KP_RESERVE_CUS)
cd $GRAFT_REPO_ROOT
for R in ${1:-24 16 12 8}; do echo -n "reserve=$R "; KP_RESERVE_CUS=$R python bench.py --steps 200 --warmup 10 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['fit_latency_ms'], d['kernel_ms'])"; done
