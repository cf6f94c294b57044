#!/bin/bash
# Run on the GPU box (gpurun -- 'bash scripts/collect_profiles.sh'): regenerates the step's own evidence under profiles/ from the
# current build (kernel stats, PMC traffic, residency, bench lines, SQ counters); scripts/collect_profiles2.sh the side benchmarks
# (shards, exchange, plugin, ragged, wave forms, microbenchmarks) -- two gpurun calls at ONE commit, each inside the 20-minute limit.  rocprofv3 gets the program itself after `--` (no env / shell hop), counters in their own passes.
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/prof
rm -rf "$OUT" && mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary"   # (the step's own kernels only)
# 1. kernel trace + stats
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $CMD > "$OUT/bench_under_rocprof.json" 2> "$OUT/trace.err"
cp "$(find "$OUT/trace" -name '*kernel_stats.csv' | head -1)" "$OUT/kernel_stats.csv"
( cd "$R" && python scripts/trace_summary.py "$OUT/trace" > "$OUT/step_timeline.txt" )
# 2. HBM traffic counters, one pass each
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- $CMD > /dev/null 2> "$OUT/pmc_fetch.err"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- $CMD > /dev/null 2> "$OUT/pmc_write.err"
( cd "$R" && python scripts/pmc_summary.py "$OUT/pmc_fetch" "$OUT/pmc_write" "$OUT/r06" 23 > "$OUT/pmc_summary.txt" )
cp "$OUT/r06_pmc_traffic.json" "$R/profiles/r06_pmc_traffic.json"   # (on the box: the bench runs below quote it; it carries this build's source hash)
# 3. workgroup residency of one step (device timestamps)
( cd "$R" && PQ_SUITE_DEBUG=2 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > "$OUT/wg.log" 2>&1; python scripts/wg_residency.py "$OUT/wg.log" > "$OUT/wg_residency.txt" )
# 4. the plain bench line (with the CPU baseline) for reference, the end-to-end figures, the backtest alone on shards (strong scaling
#    projection), the phase profile of the wave backtest is a separate build (scripts/prof_backtest.py)
( cd "$R" && python3 bench.py > "$OUT/bench.json" 2> "$OUT/bench.err" )
( cd "$R" && python3 bench.py --e2e --no-cpu-baseline > "$OUT/bench_e2e.json" 2>> "$OUT/bench.err" )
( cd "$R" && python3 scripts/strong_scaling_1gpu.py > "$OUT/strong_scaling_1gpu.json" 2>> "$OUT/bench.err" )
( cd "$R" && python3 scripts/bench_backtest.py > "$OUT/bench_backtest.json" 2>> "$OUT/bench.err" )
( cd "$R" && python3 scripts/bench_strategy.py > "$OUT/bench_strategy.json" 2>> "$OUT/bench.err" )
( cd "$R" && python3 scripts/measure_tolerance.py > "$OUT/tolerance.json" 2>> "$OUT/bench.err" )
# 4c. per-job accounting of the compute wave inside a step (a PQ_PROFILE_WAVES build made on the CPU box: scripts/ab_build.sh prof
#     "-DPQ_EXPERIMENTS -DPQ_PROFILE_WAVES" suite)
if [ -f "$R/ab/libpq_prof.so" ]; then ( cd "$R" && PQ_LIB_PATH=ab/libpq_prof.so PQ_SUITE_DEBUG=1 python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-secondary 2>&1 | grep "pq prof" | tail -40 > "$OUT/wave_profile.txt" ); fi
cp "$R/polars_quant_amd/csrc/suite.resources.txt" "$OUT/kernel_resources.txt" 2>/dev/null || true
# 5. rocprofv3 kernel stats of the config-3 backtest alone
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_bt" -- python3 $R/scripts/bench_backtest.py > /dev/null 2> "$OUT/trace_bt.err"
f=$(find "$OUT/trace_bt" -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" "$OUT/backtest_kernel_stats.csv"; rm -rf "$OUT/trace_bt"
# 6. instruction-side counters of the step (SQ passes; kernels serialised by the profiler), VALU-busy per SIMD derived in pmc_sq.py
cd /tmp
i=0
for p in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_INSTS_SMEM SQ_INSTS_BRANCH"; do
  i=$((i+1))
  timeout -k 10 250 rocprofv3 --kernel-trace --pmc $p --output-format csv -d "$OUT/sq_$i" -- python3 "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > "$OUT/sq_$i.log" 2>&1 || true
done
( cd "$R" && python scripts/pmc_sq.py "$OUT/sq_1" "$OUT/sq_2" "$OUT/sq_3" > "$OUT/sq_counters.txt" 2>> "$OUT/bench.err" )
rm -rf "$OUT/sq_1" "$OUT/sq_2" "$OUT/sq_3" "$OUT"/sq_*.log
ls -la "$OUT"
rm -rf "$OUT/trace" "$OUT/pmc_fetch" "$OUT/pmc_write" "$OUT/wg.log"
ls -la "$OUT"; tail -1 "$OUT/bench.json" | cut -c1-400
