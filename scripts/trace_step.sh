#!/bin/bash
# usage: trace_step.sh <tag> [lib]   (run on the GPU box)
R=${GRAFT_REPO_ROOT:-$PWD}
if [ -n "$2" ]; then export PQ_LIB_PATH=$R/ab/libpq_$2.so; fi
OUT=$R/gpurun_out/trace_$1
rm -rf "$OUT" && mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > "$OUT/bench.json" 2> "$OUT/trace.err"
( cd "$R" && python scripts/trace_summary.py "$OUT/trace" > "$OUT/step_timeline.txt" )
rm -rf "$OUT/trace"
head -12 "$OUT/step_timeline.txt"
