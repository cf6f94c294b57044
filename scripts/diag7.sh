mkdir -p gpurun_out/d7; rm -f gpurun_out/d7/ab.txt
for st in 0 2528 2560; do
echo "== stride $st" >> gpurun_out/d7/ab.txt
timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --stride $st 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), [(g['kernel'][-3:], round(g['avg_ms'],2)) for g in d['roofline']['grids']])" >> gpurun_out/d7/ab.txt
done
cat gpurun_out/d7/ab.txt
