#!/bin/bash
b() { lib=$1; shift; if [ -n "$lib" ]; then export PQ_LIB_PATH=$PWD/ab/libpq_$lib.so; else unset PQ_LIB_PATH; fi; timeout -k 10 200 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib $*', round(d['ms_per_step'],3))"; }
for i in 1 2; do
b ""
b "" --stride 2520
b cap152
b cap120
done
