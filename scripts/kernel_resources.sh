#!/bin/bash
# Registers / scratch / spills of the light job kernel with ONE op at a time (what each op costs INSIDE seq_jobs_kernel<0>, under its
# 192-VGPR cap), CPU only:   scripts/kernel_resources.sh [extra flags] -- Op1 'Op2<1>' ...      (no ops: every op of a suite step)
# The build is an analysis build (-DPQ_EXPERIMENTS -DPQ_ANALYZE_LIGHT=X(Op)) and is never linked.
cd "$(dirname "$0")/../polars_quant_amd/csrc"
FL=""; while [ $# -gt 0 ] && [ "$1" != "--" ]; do FL="$FL $1"; shift; done; [ "$1" = "--" ] && shift
[ $# -eq 0 ] && set -- CciOp MavpSma32Op VolumeAllOp SarPairOp DmiAtrOp StochRsiOp 'StochOp<0>' 'StochOp<1>' UltoscOp MidpriceOp ApoPpoOp CmoRsiOp \
  MacdextOp EmaAllOp BbandsOp MacdOp MidpointOp DmPairOp KamaOp T3Op WmaOp TrimaOp SmaOp MacdPairOp AtrAllOp AdAllOp
i=0
for op in "$@"; do
  tag=$(echo "$op" | tr -c 'A-Za-z0-9' '_')
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -DPQ_EXPERIMENTS $FL "-DPQ_ANALYZE_LIGHT=X($op)" \
      -Rpass-analysis=kernel-resource-usage -c suite.hip -o /tmp/kr_$tag.o > /tmp/kr_$tag.log 2>&1
    echo "$op $(grep -A14 'Function Name: _Z15seq_jobs_kernelILi0E' /tmp/kr_$tag.log | grep -E ' VGPRs:|VGPRs Spill|SGPRs Spill|ScratchSize' | sed 's/.*remark: *//; s/\[-R.*//' | tr '\n' ' ')" ) &
  i=$((i+1)); [ $((i % 8)) = 0 ] && wait
done; wait
