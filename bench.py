#!/usr/bin/env python3
"""bench.py -- indicator + backtest rows/sec on MI355X (contract: see the task statement / DESIGN.md §Measurement).

A step = one pass of the hot path over one synthetic symbol-major OHLCV block already resident in HBM:
the full talib suite (every function of SURVEY 8(a), Python-wrapper default parameters, all 61 candlestick
recognisers fused) + the fused MACD-cross per-symbol backtest with summary.  N GPUs: each rank owns its own
shard of symbols (static split, no data-path collective); the only exchange is one all_gather of the
[n_local, 8] summary table per step.  rows = symbols x days.

  --scaling weak   (default) every GPU gets --symbols symbols (5000): per-GPU work fixed
  --scaling strong the --symbols symbols are split over the GPUs (BASELINE config 3: 5000 symbols over 8 GPUs = 625 each)
  --e2e            additionally times the step END TO END from host Arrow-style buffers: pq_host_register'ed OHLCV columns ->
                   H2D -> step -> D2H of the summary table (and, second figure, of every output); reported in config.e2e,
                   never as `value`
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

import numpy as np  # noqa: E402
import torch  # noqa: E402

N_SYM, T_DAYS, SEED = 5000, 2520, 0x5EED0002
COPY_GBS = 4900.0   # a grid-stride 16-byte copy kernel on MI355X, read + written (scripts/ubench/copybw.hip, profiles/r03_ubench_copybw.txt; torch copy_: 4400)
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); measured here: streaming read 6.1, fill 6.6 TB/s


def make_inputs(n_sym: int, T: int, seed: int, device):
    """SURVEY 8(d) synthetic OHLCV (polars_quant_amd/synthetic.py: bit-identical to the generator the tests use)."""
    from polars_quant_amd.synthetic import gen_ohlcv
    d = gen_ohlcv(seed, n_sym, T, 0)
    return {k: torch.from_numpy(v).to(device) for k, v in d.items()}


def cpu_baseline(sample_syms: int, T: int):
    from oracle import pq_oracle as oracle   # the ONLY use of oracle/ here: the timed CPU baseline leg
    from polars_quant_amd.synthetic import gen_ohlcv
    d = gen_ohlcv(SEED, sample_syms, T, 0)
    # the GPU box allots 16 host cores per GPU (and caps worker pools there); use at most that many
    cores = min(len(os.sched_getaffinity(0)), 16)
    oracle.suite_bench({k: v[:8] for k, v in d.items()}, cores)  # spin up the OpenMP team
    t0 = time.perf_counter(); oracle.suite_bench(d, cores); t_all = time.perf_counter() - t0
    small = {k: v[: max(8, sample_syms // 8)] for k, v in d.items()}
    t0 = time.perf_counter(); oracle.suite_bench(small, 1); t_one = time.perf_counter() - t0
    rows = sample_syms * T
    cpu = "?"
    try:
        cpu = next(l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name"))
    except Exception:  # noqa: BLE001
        pass
    import subprocess
    cc = subprocess.run(["gcc", "--version"], capture_output=True, text=True).stdout.splitlines()[0] if True else "gcc"
    return {"value": rows / t_all, "unit": "rows/s", "cores": cores, "kind": "port",
            "sample": f"{sample_syms} symbols x {T} days, same suite+backtest, oracle (scalar C restatement of the reference; the Rust "
                      f"reference cannot be built here), {cc}, -O3 -ffp-contract=off (no -march=native: the .so travels between hosts), "
                      f"OpenMP over symbols on {cores} threads of '{cpu}'",
            "value_1thread": small["close"].shape[0] * T / t_one}


def end_to_end(suite, ohlcv, n_local, T, dev):
    """The step from HOST buffers, through the C ABI's own copy entry points: the five OHLCV columns live in page-locked
    (pq_host_register) DENSE host arrays as an Arrow buffer handed over by the caller would; H2D into the pitched device
    columns (pq_memcpy_h2d_pitched) -> step -> D2H of the [n, 8] summary table.
      serial      copies, then the step (round 2's form)
      overlapped  loader.DeviceFrame.upload on a copy stream, one event per column, and the step split into stages by the columns
                  a task needs (Suite.record_staged): the consumers of `close` run while `high` is still on the bus
    Third figure: D2H of every output column as well (10.9 GB at full size)."""
    import ctypes as C
    from polars_quant_amd._lib import check, lib
    from polars_quant_amd.api import ctx
    from polars_quant_amd.loader import DeviceFrame, HostFrame
    L, h = lib(), ctx(dev.index)
    host = HostFrame([str(i) for i in range(n_local)], np.arange(T), {k: np.ascontiguousarray(v.cpu().numpy()) for k, v in ohlcv.items()})
    frame = DeviceFrame(columns=dict(ohlcv), stride=suite.stride)
    frame.register(host)
    summ = np.empty((n_local, 8))
    check(L.pq_host_register(summ.ctypes.data_as(C.c_void_p), summ.nbytes))
    outs = [t for ts in suite.out.values() for t in ts] + list(suite.pat.values()) + suite.bt
    big = np.empty(max(t.numel() * t.element_size() for t in outs), dtype=np.uint8)
    check(L.pq_host_register(big.ctypes.data_as(C.c_void_p), big.nbytes))
    order = list(suite.STAGE_ORDER)

    def serial(all_outputs):
        frame.upload(host, order=order, copy_stream=torch.cuda.current_stream(dev))
        suite.run(ohlcv)
        check(L.pq_memcpy_d2h(h, summ.ctypes.data_as(C.c_void_p), C.c_void_p(suite.summary.data_ptr()), summ.nbytes))
        if all_outputs:
            for t in outs:
                es = t.element_size()
                check(L.pq_memcpy_d2h_pitched(h, big.ctypes.data_as(C.c_void_p), T * es, C.c_void_p(t.data_ptr()), suite.stride * es, T * es, n_local))
        torch.cuda.synchronize()

    def overlapped():
        frame.upload(host, order=order)
        suite.run_staged(frame.events)
        check(L.pq_memcpy_d2h(h, summ.ctypes.data_as(C.c_void_p), C.c_void_p(suite.summary.data_ptr()), summ.nbytes))
        torch.cuda.synchronize()

    res = {}
    for label, fn, reps in (("summary_only", lambda: serial(False), 5), ("all_outputs", lambda: serial(True), 2)):
        fn()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        dt = (time.perf_counter() - t0) / reps
        res[label] = {"ms": dt * 1e3, "rows_per_s": n_local * T / dt}
    ref_summary = suite.summary.clone()
    res["stages"] = suite.record_staged(ohlcv)        # (replaces the single recorded suite of this object)
    overlapped()
    t0 = time.perf_counter()
    for _ in range(5):
        overlapped()
    dt = (time.perf_counter() - t0) / 5
    res["summary_only_overlapped"] = {"ms": dt * 1e3, "rows_per_s": n_local * T / dt,
                                      "same_summary_bits": bool(torch.equal(ref_summary.view(torch.int64), suite.summary.view(torch.int64)))}
    res["h2d_bytes"] = sum(a.nbytes for a in host.columns.values())
    res["d2h_bytes_all_outputs"] = sum(t.numel() * t.element_size() for t in outs)
    frame.unregister()
    for a in (summ, big):
        check(L.pq_host_unregister(a.ctypes.data_as(C.c_void_p)))
    return res


def launch_ranks(args) -> int:
    """`python bench.py --gpus N` without a launcher: N child processes, one per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*
    as torch.distributed.run would set them).  The parent has made no HIP call (torch.cuda.device_count() does not initialise
    the runtime), so the children are ordinary fresh processes; rank 0 prints the JSON line on the inherited stdout."""
    import socket
    import subprocess
    n = args.gpus
    have = torch.cuda.device_count()
    if have < n and not args.dry_run:
        print(f"bench.py: --gpus {n} needs {n} visible GPUs, this host shows {have}; not running a {have}-GPU job under the name "
              f"of an {n}-GPU one", file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, str(Path(__file__).resolve())] + sys.argv[1:], env=env))
    rcs = [p.wait() for p in procs]
    return max(abs(rc) for rc in rcs)


def dry_run(args):
    """What each rank would own, through the same rendezvous (gloo, CPU only): rank 0 prints one JSON line."""
    import torch.distributed as dist
    from polars_quant_amd.distributed import shard_range
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
    if args.scaling == "strong":
        lo, hi = shard_range(args.symbols, rank, world)
        mine = {"rank": rank, "local_rank": int(os.environ.get("LOCAL_RANK", "0")), "symbols": [lo, hi], "seed": SEED}
    else:
        mine = {"rank": rank, "local_rank": int(os.environ.get("LOCAL_RANK", "0")), "symbols": [rank * args.symbols, (rank + 1) * args.symbols],
                "seed": SEED + rank}
    shards = [mine]
    if world > 1:
        shards = [None] * world
        dist.all_gather_object(shards, mine)
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": world, "scaling": args.scaling, "days": args.days,
                          "symbols_total": args.symbols if args.scaling == "strong" else args.symbols * world,
                          "world_size_seen": dist.get_world_size() if world > 1 else 1, "backend": "gloo" if world > 1 else None,
                          "shards": shards}))
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--symbols", type=int, default=N_SYM, help="symbols per GPU (weak scaling) / in total (strong scaling)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak")
    ap.add_argument("--e2e", action="store_true", help="also time the step end to end from registered host buffers")
    ap.add_argument("--days", type=int, default=T_DAYS)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--stride", type=int, default=0,
                    help="row pitch of the device columns in elements (0 = days rounded up to a multiple of 16 = 128 B; = days: dense)")
    ap.add_argument("--dry-run", action="store_true",
                    help="launch / rendezvous / shard ranges only (gloo on the CPU, no GPU work): what `--gpus N` would run where")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args))   # this process never touches the GPU: it starts one fresh process per rank
    if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={os.environ['WORLD_SIZE']} (launch with --nproc-per-node {args.gpus})", file=sys.stderr)
        sys.exit(2)
    if args.dry_run:
        return dry_run(args)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist_
        dist = dist_
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    assert torch.cuda.is_available(), "bench.py needs an MI355X; there is no CPU path"
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    from polars_quant_amd.suite import Suite
    from polars_quant_amd.distributed import gather_summaries, shard_range
    T = args.days
    if args.scaling == "strong":     # one data set of --symbols symbols, split statically over the ranks
        lo, hi = shard_range(args.symbols, rank, world)
        n_local, n_total = hi - lo, args.symbols
        full = make_inputs(args.symbols, T, SEED, torch.device("cpu"))
        ohlcv = {k: v[lo:hi].contiguous().to(dev) for k, v in full.items()}
        del full
    else:
        n_local, n_total = args.symbols, args.symbols * world
        ohlcv = make_inputs(n_local, T, SEED + rank, dev)       # every rank: its own symbols
    # Device columns are pitched like a hipMallocPitch allocation: a row pitch that is a multiple of 128 B makes every 64 / 128-byte
    # tile piece one aligned cache line (dense 2520-element rows start at odd multiples of 64 B): -8 % per step.  `--stride <days>`
    # measures the dense layout.
    stride = args.stride or (T + 15) // 16 * 16
    if stride != T:  # re-house the inputs with the padded row pitch
        for k in list(ohlcv):
            buf = torch.zeros((n_local, stride), dtype=torch.float64, device=dev)
            buf[:, :T] = ohlcv[k]
            ohlcv[k] = buf[:, :T]
    suite = Suite(n_local, T, dev, stride=stride)

    def step():
        suite.run(ohlcv)
        if world > 1:   # the one exchange of the path: per-symbol summary rows to every rank (RCCL over xGMI)
            gather_summaries(suite.summary, n_total)

    suite.record(ohlcv)   # one-time: turn the step's calls into job grids (not part of the timed region)
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    suite.set_timing(True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # ---- roofline of the dominant kernel ----------------------------------------------------------------
    # The step's device time is dominated by seq_jobs_kernel<0>: the tiled bodies of all sequential jobs, two launches per
    # step (one per LDS class) that run CONCURRENTLY with each other and with
    # the row-parallel kernels.  Each launch was bracketed by HIP events on its own launch stream during the timed steps
    # above (pq_suite_set_timing).  Contract figure: achieved = mean algorithmic bytes per launch / mean launch duration
    # (= what `rocprofv3 --kernel-trace --stats` reports as that kernel's average duration, profiles/).  Because the
    # launches overlap, each one sees only its share of the chip; `suite_algorithmic_GBps` in `config` is the rate of the
    # whole step.
    grids = [g for g in suite.grid_stats() if g["runs"] > 0]
    rows_local = n_local * T
    dom = [g for g in grids if g["kernel"] == "seq_jobs_kernel<0>"]
    n_launch = sum(g["runs"] for g in dom)
    if n_launch:
        mean_ms = sum(g["avg_ms"] * g["runs"] for g in dom) / n_launch
        mean_bytes = sum(g["alg_bytes"] * g["runs"] for g in dom) / n_launch
        achieved = mean_bytes / (mean_ms * 1e-3) / 1e9
    else:   # e.g. an odd --days / --stride: no tiled grid exists, every job ran in the gather kernel
        mean_ms = mean_bytes = achieved = None
    span_ms, span_bytes = suite.span_stats(0)   # the two launches as one concurrent set
    suite_bytes = suite.suite_bytes_per_row() * rows_local
    suite_gbs = suite_bytes / (elapsed / args.steps) / 1e9
    # HBM traffic of that kernel from the PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate runs of this
    # command; summary committed by scripts/pmc_summary.py).  Only valid for the configuration it was collected on.
    traffic = None
    pmc = ROOT / "profiles" / "r03_pmc_traffic.json"   # (collected by scripts/collect_profiles.sh; says so in the line: roofline.traffic_source)
    if pmc.exists() and n_local == N_SYM and T == T_DAYS:
        kernels = json.loads(pmc.read_text())["kernels"]
        k = kernels.get("seq_jobs_kernel<0>")
        traffic = k["hbm_bytes_per_launch"] if k else None
        # every kernel of the step (the PMC file holds means per launch; seq_jobs_kernel<0> is launched twice per step)
        step_traffic = sum(v["hbm_bytes_per_launch"] * (2 if name == "seq_jobs_kernel<0>" else 1)
                           for name, v in kernels.items() if not name.startswith(("at::", "__amd")))   # torch's own kernels excluded
    else:
        step_traffic = None

    if rank == 0:
        rows_total = n_total * T * args.steps
        line = {
            "metric": "indicator+backtest rows/sec, 5000 sym x 2520 day f64 OHLCV",
            "value": rows_total / elapsed, "unit": "rows/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "collective": ({"backend": dist.get_backend(), "world_size_seen": dist.get_world_size(),
                            "per_step": "one all_gather of the [n_local, 8] summary rows",
                            "c_abi_gather": "pq_comm_init + pq_gather_summaries run once AFTER this line and are compared with this result; "
                                            "outcome on stderr"}
                           if world > 1 else None),
            "config": {"workload": f"full talib suite ({len(suite.tasks()) - 2} indicator calls + 61 fused candlestick "
                                   f"recognisers) + fused MACD-cross backtest with summary, {n_local} symbols x {T} days "
                                   "f64 OHLCV per GPU, inputs resident in HBM",
                       "symbols_per_gpu": n_local, "symbols_total": n_total, "days": T, "row_pitch_elements": stride,
                       "parallelism": f"symbol-sharded x{world}",
                       "algorithmic_bytes_per_row": suite.suite_bytes_per_row(),
                       "suite_algorithmic_GBps": suite_gbs, "suite_frac_of_hbm_peak": suite_gbs / HBM_PEAK_GBS},
            "roofline": {"bound": "hbm", "kernel": "seq_jobs_kernel<0>", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": (achieved / HBM_PEAK_GBS) if achieved else None, "traffic": traffic,
                         "traffic_source": "profiles/r03_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, not measured in this run)" if traffic else None,
                         # the whole step against what this chip does for a read/write MIX: its L2-miss bytes (PMC) per second, and
                         # the rate of a streaming copy kernel (read + written, scripts/ubench/copybw.hip)
                         "step_traffic": step_traffic,
                         "step_traffic_GBps": step_traffic / (elapsed / args.steps) / 1e9 if step_traffic and world == 1 else None,
                         "copy_GBps_measured": COPY_GBS,
                         "algorithmic_bytes_per_launch": mean_bytes, "avg_launch_ms": mean_ms,
                         "launches_per_step": len(dom),
                         "concurrent_set": {"note": "the launches of this kernel overlap inside a step: their summed algorithmic bytes "
                                                    "over the time from the earliest start to the latest end (HIP events)",
                                            "span_ms": span_ms, "algorithmic_bytes": span_bytes,
                                            "achieved": span_bytes / (span_ms * 1e-3) / 1e9 if span_ms else None,
                                            "frac": span_bytes / (span_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if span_ms else None},
                         "grids": [{k: g[k] for k in ("kernel", "avg_ms", "alg_bytes", "n_jobs", "lds_bytes")} for g in grids]},
        }
        if args.e2e and world == 1:
            line["config"]["e2e"] = end_to_end(suite, ohlcv, n_local, T, dev)
        if world == 1 and stride != T and args.scaling == "weak" and not args.no_cpu_baseline:
            # for the record: the same step on the DENSE layout (row pitch = days), outside the timed region above
            suite.close()
            dense_in = {k: v.contiguous() for k, v in ohlcv.items()}
            dense = Suite(n_local, T, dev, stride=T)
            dense.record(dense_in)
            for _ in range(args.warmup):
                dense.run(dense_in)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                dense.run(dense_in)
            torch.cuda.synchronize()
            line["config"]["dense_layout"] = {"row_pitch_elements": T, "ms_per_step": (time.perf_counter() - t0) / args.steps * 1e3}
            dense.close()
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(4096, T)
        print(json.dumps(line))
    if world > 1:
        # The same exchange through the C ABI's own entry points (pq_comm_init + pq_gather_summaries: what a non-Python host calls),
        # once, after the measurement has been printed, compared bit for bit with the torch.distributed result.  It runs on a
        # helper thread with a deadline: a second communicator that cannot be built must not be able to hold the job.
        import threading
        box = {}

        def cross_check():
            try:
                torch.cuda.set_device(dev)   # (the current device is per thread)
                from polars_quant_amd.distributed import CabiComm
                ref = gather_summaries(suite.summary, n_total)
                comm = CabiComm(dev, rank, world)
                got = comm.gather_summaries(suite.summary, n_total)
                torch.cuda.synchronize()
                box["result"] = "ok: identical to the torch.distributed gather" if torch.equal(got.view(torch.int64), ref.view(torch.int64)) else "MISMATCH"
                comm.close()
            except Exception as e:  # noqa: BLE001
                box["result"] = f"failed: {e}"

        th = threading.Thread(target=cross_check, daemon=True)
        th.start()
        th.join(90.0)
        if rank == 0:
            print(f"[bench] C-ABI gather (pq_comm_init + pq_gather_summaries over {world} ranks): {box.get('result', 'no answer within 90 s')}",
                  file=sys.stderr, flush=True)
        if th.is_alive():
            sys.stdout.flush()
            os._exit(0)   # the measurement is out; do not wait for a collective that will not complete
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
