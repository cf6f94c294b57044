#!/bin/bash
# Part 2 of the evidence collection (see scripts/collect_profiles.sh): the side benchmarks.  gpurun -- 'bash scripts/collect_profiles2.sh'
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/prof2
rm -rf "$OUT" && mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R"
# a. small shards: the step on 625 / 1 250 / 2 500 symbols (scripts/strong_scaling_1gpu.py), kernel timeline and per-job end times of a 625-symbol step
python3 scripts/strong_scaling_1gpu.py > "$OUT/strong_scaling_1gpu.json" 2>> "$OUT/err.txt"
( cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace625" -- python3 "$R/scripts/step_time.py" 625 > "$OUT/trace625.log" 2>&1 )
python3 scripts/trace_summary.py "$OUT/trace625" > "$OUT/shard625_timeline.txt" 2>> "$OUT/err.txt" || true
rm -rf "$OUT/trace625"
( timeout -k 10 120 python3 scripts/shard_jobs.py 625 > /dev/null 2> "$OUT/shard625_jobs.log" || true ); awk '/last step/{p=1} p' "$OUT/shard625_jobs.log" | grep "pq suite" > "$OUT/shard625_jobs.txt" || true
# b. the exchange: through Python / torch streams, from C through the C ABI alone (default and highest priority of the exchange stream), the runtime alone
timeout -k 10 200 python3 scripts/bench_gather.py 2>> "$OUT/err.txt" | sed -n '/^{/,$p' > "$OUT/bench_gather.json"
( cd scripts/ubench && /opt/rocm/bin/hipcc -O2 -std=c++17 gather_cabi.cpp -o gather_cabi -I../../include -L../../polars_quant_amd -lpolars_quant_hip -Wl,-rpath,"$R/polars_quant_amd" \
  && /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 xstream.hip -o xstream && /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 halfwave.hip -o halfwave \
  && /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 stepshape.hip -o stepshape && /opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 pipes.hip -o pipes )
{ echo '{"default_priority":'; timeout -k 10 100 scripts/ubench/gather_cabi 625 400 | sed -n '/^{/,$p'; echo ', "highest_priority":'; PQ_COMM_PRIO=high timeout -k 10 100 scripts/ubench/gather_cabi 625 400 | sed -n '/^{/,$p'; echo '}'; } > "$OUT/gather_cabi.json" 2>> "$OUT/err.txt"
timeout -k 10 200 scripts/ubench/xstream > "$OUT/ubench_xstream.json" 2>> "$OUT/err.txt"
timeout -k 10 100 scripts/ubench/halfwave > "$OUT/ubench_halfwave.txt" 2>> "$OUT/err.txt"
timeout -k 10 100 scripts/ubench/stepshape 50 > "$OUT/ubench_stepshape.json" 2>> "$OUT/err.txt"   # the step's launch shape by stream / compute pipe
timeout -k 10 60 scripts/ubench/pipes > "$OUT/ubench_pipes.txt" 2>> "$OUT/err.txt"
# c. the plugin end to end from host Arrow buffers; ragged batches (incl. a recorded ragged suite)
timeout -k 10 400 python3 scripts/bench_plugin.py > "$OUT/bench_plugin.json" 2>> "$OUT/err.txt"
timeout -k 10 400 python3 scripts/bench_ragged.py > "$OUT/bench_ragged.json" 2>> "$OUT/err.txt"
# d. backtest: sizes, phase profile of the wave kernel (a PQ_BTW_PROF build made on the CPU box: scripts/ab_build.sh btwprof -DPQ_BTW_PROF backtest)
timeout -k 10 200 python3 scripts/bench_backtest_sizes.py > "$OUT/bench_backtest_sizes.txt" 2>> "$OUT/err.txt"
if [ -f "$R/ab/libpq_btwprof.so" ]; then ( PQ_LIB_PATH=ab/libpq_btwprof.so timeout -k 10 200 python3 scripts/prof_backtest.py > "$OUT/btw_phase_profile.txt" 2>> "$OUT/err.txt" || true ); fi
# 4b. the wave-per-symbol indicator kernels alone against the lane-per-symbol bodies (direct C-ABI calls), their per-phase device time
#     (a PQ_WT_PROF build made on the CPU box: scripts/ab_build.sh wtprof -DPQ_WT_PROF wt) and the in-suite A/B
( timeout -k 10 300 python3 scripts/bench_wt.py 2>> "$OUT/err.txt" | tail -1 > "$OUT/bench_wt.json" )
if [ -f "$R/ab/libpq_wtprof.so" ]; then ( PQ_LIB_PATH=ab/libpq_wtprof.so PQ_WT_ALL=1 PQ_MIDPRICE_SEQ=1 timeout -k 10 200 python3 scripts/prof_wt.py > "$OUT/wt_phase_profile.txt" 2>> "$OUT/err.txt" || true ); fi
( for v in "PQ_NO_WT=1" "PQ_WT_SUITE=1 PQ_WT_ALL=1" "PQ_WT_SUITE=1 PQ_WT_OPS=atr,midpoint" "PQ_MIDPRICE_ROW=1"; do
    echo "$v: $(env $v python3 bench.py --steps 30 --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "import sys,json; print(round(json.loads(sys.stdin.readline())['ms_per_step'],3))") ms per step"; done > "$OUT/wt_in_suite_ab.txt" )
ls -la "$OUT"
