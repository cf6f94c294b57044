#!/bin/bash
# residency of one step: residency.sh <tag> [ENV=VAL...]
tag=$1; shift
env "$@" PQ_SUITE_DEBUG=2 timeout -k 10 200 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/wg_$tag.log 2>&1
python3 scripts/wg_residency.py gpurun_out/wg_$tag.log | head -12
rm -f gpurun_out/wg_$tag.log
