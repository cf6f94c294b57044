mkdir -p gpurun_out/d5; rm -f gpurun_out/d5/ab.txt
run() { echo "== $*" >> gpurun_out/d5/ab.txt; bash scripts/bench_short.sh "$@" >> gpurun_out/d5/ab.txt 2>&1; bash scripts/bench_short.sh "$@" >> gpurun_out/d5/ab.txt 2>&1; }
run X=0
run PQ_ROW_LATE_CHAIN=2
run PQ_ROW_LATE_CHAIN=1
run PQ_NO_ROW_SPLIT=1
run PQ_ROW_LATE_CHAIN=2 PQ_ROW_EARLY=0.3
run PQ_ROW_LATE_CHAIN=2 PQ_ROW_EARLY=0.7
run PQ_ROW_EARLY=0.0 PQ_ROW_LATE_CHAIN=2
cat gpurun_out/d5/ab.txt
