#!/bin/bash
# kernel time of kp_gram3_prelift_kernel under the compile-time ablations PRE_ABL (1: no stores, 2: no multiply-adds, 4: one matrix row;
# tools/abl/libkp_pre<N>.so built by hand, selected through KP_LIB_PATH)
cd /tmp && export TMPDIR=/tmp
for a in 0 1 2 3 4 5; do
  export KP_LIB_PATH=$GRAFT_REPO_ROOT/tools/abl/libkp_pre$a.so
  rm -rf /tmp/pp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -- python3 $GRAFT_REPO_ROOT/tools/prelift_time.py > /tmp/pp.log 2>&1
  f=$(find /tmp/pp -name "*kernel_stats.csv" | head -1)
  python3 - "$f" "$a" <<'PY'
import csv, sys
try:
    for r in csv.DictReader(open(sys.argv[1])):
        if "prelift" in r["Name"]:
            print("abl", sys.argv[2], r["Name"][:40], round(float(r["AverageNs"]) / 1e3, 1), "us")
except Exception as e:
    print("abl", sys.argv[2], "failed", e); print(open("/tmp/pp.log").read()[-500:])
PY
done
