set -x
mkdir -p gpurun_out/d2
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/d2/pytest.txt 2>&1 || { tail -30 gpurun_out/d2/pytest.txt; exit 1; }
tail -3 gpurun_out/d2/pytest.txt
bash scripts/bench_short.sh > gpurun_out/d2/bench_short.txt 2>&1
cat gpurun_out/d2/bench_short.txt
timeout -k 10 400 python scripts/exp_solo.py > gpurun_out/d2/solo.txt 2>&1
echo done
