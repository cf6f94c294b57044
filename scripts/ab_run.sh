#!/bin/bash
# A/B in ONE session on the GPU box: scripts/ab_run.sh <name> ...   ("head" = the product library, others = ab/libpq_<name>.so)
for i in 1 2; do
for n in "$@"; do
  if [ $n = head ]; then unset PQ_LIB_PATH; else export PQ_LIB_PATH=ab/libpq_$n.so; fi
  python bench.py --steps 30 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$n', round(d['ms_per_step'],3), 'dominant', d['roofline'].get('achieved'))"
done; done
