#!/bin/bash
b() { lib=$1; shift; if [ -n "$lib" ]; then export PQ_LIB_PATH=$PWD/ab/libpq_$lib.so; else unset PQ_LIB_PATH; fi; timeout -k 10 200 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib $*', round(d['ms_per_step'],3), [(g['kernel'][-3:], g['n_jobs'], round(g['avg_ms'],2)) for g in d['roofline']['grids']])"; }
for i in 1 2 3; do
b ""
b head
done
