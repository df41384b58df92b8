#!/bin/bash
# builds timing-only ablation variants of the library: tools/libkp_abl<A>.so  (KP_ABL3=A in kp_gram3.hip)
cd "$(dirname "$0")/../koopman-realizations_amd/csrc"
for A in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I. -DKP_ABL3=$A -c kp_gram3.hip -o /tmp/kp_gram3_abl$A.o &&
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/libkp_abl$A.so $(ls *.o | grep -v kp_gram3.o) /tmp/kp_gram3_abl$A.o &
done
wait
