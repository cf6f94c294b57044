mkdir -p gpurun_out/d8
timeout -k 10 300 python scripts/exp_solo.py > gpurun_out/d8/solo_base.txt 2>&1
PQ_LIB_PATH=/root/repo/ab/libpq_pf2all.so timeout -k 10 300 python scripts/exp_solo.py > gpurun_out/d8/solo_pf2.txt 2>&1
paste <(sort -k1,1 gpurun_out/d8/solo_base.txt | grep " ms") <(sort -k1,1 gpurun_out/d8/solo_pf2.txt | grep " ms") | awk '{printf "%-22s %7.3f -> %7.3f\n",$1,$2,$7}'
