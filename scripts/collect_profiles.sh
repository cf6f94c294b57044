#!/bin/bash
# Run on the GPU box (gpurun -- 'bash scripts/collect_profiles.sh'): regenerates every artifact under profiles/ from the
# current build.  rocprofv3 gets the program itself after `--` (no env / shell hop), counters in their own passes.
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/prof
rm -rf "$OUT" && mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline"
# 1. kernel trace + stats
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $CMD > "$OUT/bench_under_rocprof.json" 2> "$OUT/trace.err"
cp "$(find "$OUT/trace" -name '*kernel_stats.csv' | head -1)" "$OUT/kernel_stats.csv"
( cd "$R" && python scripts/trace_summary.py "$OUT/trace" > "$OUT/step_timeline.txt" )
# 2. HBM traffic counters, one pass each
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- $CMD > /dev/null 2> "$OUT/pmc_fetch.err"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- $CMD > /dev/null 2> "$OUT/pmc_write.err"
( cd "$R" && python scripts/pmc_summary.py "$OUT/pmc_fetch" "$OUT/pmc_write" "$OUT/r04" 23 > "$OUT/pmc_summary.txt" )
# 3. workgroup residency of one step (device timestamps)
( cd "$R" && PQ_SUITE_DEBUG=2 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/wg.log" 2>&1; python scripts/wg_residency.py "$OUT/wg.log" > "$OUT/wg_residency.txt" )
# 4. the plain bench line (with the CPU baseline) for reference, the end-to-end figures, the backtest alone on shards (strong scaling
#    projection), the phase profile of the wave backtest is a separate build (scripts/prof_backtest.py)
( cd "$R" && python3 bench.py > "$OUT/bench.json" 2> "$OUT/bench.err" )
( cd "$R" && python3 bench.py --e2e --no-cpu-baseline > "$OUT/bench_e2e.json" 2>> "$OUT/bench.err" )
( cd "$R" && python3 scripts/strong_scaling_1gpu.py > "$OUT/strong_scaling_1gpu.json" 2>> "$OUT/bench.err" )
( cd "$R" && python3 scripts/bench_backtest.py > "$OUT/bench_backtest.json" 2>> "$OUT/bench.err" )
( cd "$R" && python3 scripts/bench_strategy.py > "$OUT/bench_strategy.json" 2>> "$OUT/bench.err" )
( cd "$R" && python3 scripts/measure_tolerance.py > "$OUT/tolerance.json" 2>> "$OUT/bench.err" )
# 5. rocprofv3 kernel stats of the config-3 backtest alone
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_bt" -- python3 $R/scripts/bench_backtest.py > /dev/null 2> "$OUT/trace_bt.err"
f=$(find "$OUT/trace_bt" -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" "$OUT/backtest_kernel_stats.csv"; rm -rf "$OUT/trace_bt"
ls -la "$OUT"
rm -rf "$OUT/trace" "$OUT/pmc_fetch" "$OUT/pmc_write" "$OUT/wg.log"
ls -la "$OUT"; tail -1 "$OUT/bench.json" | cut -c1-400
