mkdir -p gpurun_out/d9; rm -f gpurun_out/d9/ab.txt
for v in base r1 r4; do
  if [ $v = base ]; then L=; else L="PQ_LIB_PATH=/root/repo/ab/libpq_$v.so"; fi
  echo "== $v" >> gpurun_out/d9/ab.txt
  env $L timeout -k 10 300 python scripts/exp_time.py cdl_all 2>&1 | grep " ms" >> gpurun_out/d9/ab.txt
  bash scripts/bench_short.sh $L >> gpurun_out/d9/ab.txt 2>&1
  bash scripts/bench_short.sh $L >> gpurun_out/d9/ab.txt 2>&1
done
cat gpurun_out/d9/ab.txt
