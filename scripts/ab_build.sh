#!/bin/bash
# A/B variant of the library: scripts/ab_build.sh <name> "<extra -D flags>" [all | <tu>]  -> ab/libpq_<name>.so
# The experiment switches of csrc/experiments.h need -DPQ_EXPERIMENTS as well (a product build never sets it).
# Default: only suite.hip (the job-grid kernels bench.py times) is rebuilt with the flags; `all` rebuilds every TU.
# `ab/` is git-ignored; delete it after use (it travels to the GPU box with every gpurun push).
set -e
cd /root/repo/polars_quant_amd/csrc
mkdir -p /root/repo/ab /tmp/ab_$1
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function -Wno-bitwise-instead-of-logical -Wno-parentheses"
if [ "$3" = all ]; then
  for f in runtime wt overlap momentum misc pattern backtest fused suite factor plugin comm strategy; do /opt/rocm/bin/hipcc $F $2 -c $f.hip -o /tmp/ab_$1/$f.o & done; wait
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o /root/repo/ab/libpq_$1.so /tmp/ab_$1/*.o -ldl
else
  TU=${3:-suite}   # the one TU to rebuild with the flags (default: suite.hip)
  /opt/rocm/bin/hipcc $F $2 -c $TU.hip -o /tmp/ab_$1/$TU.o
  OBJS=""; for f in runtime wt overlap momentum misc pattern backtest fused factor plugin suite comm strategy; do if [ $f = $TU ]; then OBJS="$OBJS /tmp/ab_$1/$TU.o"; else OBJS="$OBJS $f.o"; fi; done
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o /root/repo/ab/libpq_$1.so $OBJS -ldl
fi
