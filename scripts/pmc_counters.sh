#!/bin/bash
# usage: pmc_counters.sh <tag> "<counters>"   (GPU box)
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/pmc_$1
rm -rf "$OUT" && mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --pmc $2 --output-format csv -d "$OUT/raw" -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > /dev/null 2> "$OUT/err.txt"
( cd "$R" && python scripts/pmc_generic.py "$OUT/raw" > "$OUT/summary.txt" )
rm -rf "$OUT/raw"
cat "$OUT/summary.txt"
