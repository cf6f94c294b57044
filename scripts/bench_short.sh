#!/bin/bash
# quick A/B: bash scripts/bench_short.sh [ENV=VAL ...]   -> ms/step and per-grid ms
env "$@" timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), [(g['kernel'][-3:], round(g['avg_ms'],2)) for g in d['roofline']['grids']])"
