mkdir -p gpurun_out/d6
timeout -k 10 300 python scripts/exp_split.py > gpurun_out/d6/split.txt 2>&1
PQ_LIB_PATH=/root/repo/ab/libpq_nostore.so timeout -k 10 300 python scripts/exp_split.py > gpurun_out/d6/split_nostore.txt 2>&1
grep -v "^ROW-only\|amdgpu" gpurun_out/d6/split.txt; echo; grep -v "^ROW-only\|amdgpu" gpurun_out/d6/split_nostore.txt
