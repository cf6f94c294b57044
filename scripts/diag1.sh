set -x
mkdir -p gpurun_out/d1
bash scripts/bench_short.sh > gpurun_out/d1/bench_short.txt 2>&1
timeout -k 10 300 python scripts/exp_split.py > gpurun_out/d1/split.txt 2>&1
timeout -k 10 400 python scripts/exp_solo.py > gpurun_out/d1/solo.txt 2>&1
PQ_SUITE_DEBUG=1 timeout -k 10 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/d1/dbg.txt 2>&1
echo done
