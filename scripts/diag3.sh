mkdir -p gpurun_out/d3; rm -f gpurun_out/d3/ab.txt
for v in base "$@"; do
  if [ $v = base ]; then L=; else L="PQ_LIB_PATH=/root/repo/ab/libpq_$v.so"; fi
  echo "== $v" >> gpurun_out/d3/ab.txt
  bash scripts/bench_short.sh $L >> gpurun_out/d3/ab.txt 2>&1
  bash scripts/bench_short.sh $L >> gpurun_out/d3/ab.txt 2>&1
done
cat gpurun_out/d3/ab.txt
