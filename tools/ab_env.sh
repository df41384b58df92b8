#!/bin/bash
# headline A/B over environment settings on one box: tools/ab_env.sh "VAR=1" "VAR2=x" ...   ("-" = no setting)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for V in - "$@"; do
  echo -n "$V "; ( if [ "$V" != "-" ]; then export $V; fi; python bench.py --no-cpu-baseline --no-mpc --steps 200 --warmup 10 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['fit_latency_ms'], d['kernel_ms'])" )
done; done
