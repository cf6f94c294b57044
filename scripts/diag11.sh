mkdir -p gpurun_out/d11; rm -f gpurun_out/d11/ab.txt
run() { echo "== $*" >> gpurun_out/d11/ab.txt; bash scripts/bench_short.sh "$@" >> gpurun_out/d11/ab.txt 2>&1; bash scripts/bench_short.sh "$@" >> gpurun_out/d11/ab.txt 2>&1; }
run X=0
run PQ_HEAVY_LDS_PAD=49152
run PQ_HEAVY_LDS_PAD=65536
cat gpurun_out/d11/ab.txt
