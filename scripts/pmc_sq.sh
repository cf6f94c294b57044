#!/bin/bash
# SQ counter passes over bench.py (run on the GPU box): bash scripts/pmc_sq.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for p in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_INSTS_SMEM SQ_INSTS_BRANCH"; do
  i=$((i+1))
  timeout -k 10 250 rocprofv3 --kernel-trace --pmc $p --output-format csv -d "$R/gpurun_out/sq_$i" -- python3 "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > "$R/gpurun_out/sq_$i.log" 2>&1 || exit 1
done
cd "$R" && python scripts/pmc_sq.py gpurun_out/sq_1 gpurun_out/sq_2 gpurun_out/sq_3 > gpurun_out/sq.txt
rm -rf gpurun_out/sq_1 gpurun_out/sq_2 gpurun_out/sq_3
cat gpurun_out/sq.txt
