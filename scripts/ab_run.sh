#!/bin/bash
# A/B in ONE session on the GPU box: scripts/ab_run.sh [-n rounds] <name> ...   ("head" = the product library, others = ab/libpq_<name>.so)
N=2; if [ "$1" = -n ]; then N=$2; shift 2; fi
for i in $(seq $N); do
for n in "$@"; do
  if [ $n = head ]; then unset PQ_LIB_PATH; else export PQ_LIB_PATH=ab/libpq_$n.so; fi
  python bench.py --steps 30 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$n', round(d['ms_per_step'],3))"
done; done
