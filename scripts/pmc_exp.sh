#!/bin/bash
# SQ counters for one exp_time.py task list (run on the GPU box): bash scripts/pmc_exp.sh "<task,list>"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for p in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INSTS_BRANCH" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  ( cd "$R" && timeout -k 10 250 rocprofv3 --kernel-trace --pmc $p --output-format csv -d "$R/gpurun_out/pe_$i" -- python3 scripts/exp_time.py "$1" > "$R/gpurun_out/pe_$i.log" 2>&1 ) || exit 1
done
cd "$R" && python scripts/pmc_sq.py gpurun_out/pe_1 gpurun_out/pe_2 gpurun_out/pe_3 | tr ' ' '\n' | grep -v "^$"
rm -rf gpurun_out/pe_1 gpurun_out/pe_2 gpurun_out/pe_3
