#!/bin/bash
# timing-only ablations of kp_trsm2_kernel: tools/abl_libs/lib_{A,B,C}.so = no MFMAs / no tile loads / no barrier
cd $GRAFT_REPO_ROOT
L=koopman-realizations_amd/libkoopman_hip.so
cp $L /tmp/base.so
export TMPDIR=/tmp
for V in base A B C; do
  if [ $V = base ]; then cp /tmp/base.so $L; else cp tools/abl_libs/lib_$V.so $L; fi
  rm -rf /tmp/pp_$V
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp_$V -- python3 $GRAFT_REPO_ROOT/tools/chol_prof.py 336 > /dev/null 2>&1)
  f=$(ls /tmp/pp_$V/*/*kernel_stats.csv | head -1)
  echo -n "$V: "; python3 -c "
import csv
for r in csv.DictReader(open('$f')):
    if 'trsm2' in r['Name']: print('trsm2 avg us', float(r['AverageNs'])/1e3)
"
done
cp /tmp/base.so $L
