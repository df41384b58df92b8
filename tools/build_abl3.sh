#!/bin/bash
# builds timing-only ablation variants of the library: tools/libkp_abl<A>.so  (KP_ABL3=A in kp_gram3.hip; headline instantiation only)
# 1: no lift  7: no weight multiplies  8: no end-of-tile table build  9: no barrier  10: no MFMAs  11: MFMAs + operand reads only
cd "$(dirname "$0")/../koopman-realizations_amd/csrc"
for A in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I. -DKP_G3_DEV -DKP_ABL3=$A -c kp_gram3.hip -o /tmp/kp_gram3_abl$A.o &&
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/libkp_abl$A.so $(ls *.o | grep -v kp_gram3.o) /tmp/kp_gram3_abl$A.o -ldl &
done
wait
