#!/bin/bash
# build an experimental variant of the library: scripts/exp_build.sh "-DPQ_K11=32 -DPQ_KXX=16"
set -e
cd /root/repo/polars_quant_amd/csrc
for f in runtime overlap momentum misc pattern backtest fused suite factor; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function $1 -c $f.hip -o /tmp/exp_$f.o &
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../libpolars_quant_hip.so /tmp/exp_*.o
