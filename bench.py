#!/usr/bin/env python3
"""bench.py -- indicator + backtest rows/sec on MI355X (contract: see the task statement / DESIGN.md §Measurement).

A step = one pass of the hot path over one synthetic symbol-major OHLCV block already resident in HBM:
the full talib suite (every function of SURVEY 8(a), Python-wrapper default parameters, all 61 candlestick
recognisers fused) + the fused MACD-cross per-symbol backtest with summary.  N GPUs: each rank owns its own
shard of symbols (static split, no data-path collective); the only exchange is one all_gather of the
[n_local, 8] summary table per step.  rows = symbols x days.

  --scaling strong (default for --gpus N > 1: BASELINE.json's metric is 5000 symbols AT 1/2/4/8 GPUs) the --symbols symbols are
                   split over the GPUs (5000 symbols over 8 GPUs = 625 each); the weak figure is carried as `weak_scaling`
  --scaling weak   every GPU gets --symbols symbols (5000): per-GPU work fixed (the only meaning at --gpus 1)
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
FUSED_FLOOR_BYTES_PER_ROW = 928   # SURVEY 8(d) second denominator: OHLCV read once (40 B) + every output of the step written once
                                  # (64 + 17 f64 indicator columns, 62 int32 columns, 3 backtest columns = 888 B)
ROUND = "r06"          # evidence under profiles/ is named per round (scripts/collect_profiles.sh writes profiles/<ROUND>_*)
PMC_FILE = ROOT / "profiles" / f"{ROUND}_pmc_traffic.json"


def source_hash() -> str:
    """sha256 over the kernel sources (csrc/*.hip, csrc/*.h): the PMC traffic file is only quoted for the build it was collected on
    (scripts/pmc_summary.py writes the same hash into it; .git does not travel to the GPU box)"""
    import hashlib
    h = hashlib.sha256()
    host_only = {"plugin.hip", "comm.hip"}     # no device code: the Polars plugin exporter and the RCCL binding
    for f in sorted(list((ROOT / "polars_quant_amd" / "csrc").glob("*.hip")) + list((ROOT / "polars_quant_amd" / "csrc").glob("*.h"))):
        if f.name in host_only:
            continue
        h.update(f.name.encode()); h.update(f.read_bytes())
    return h.hexdigest()[:16]


def make_inputs(n_sym: int, T: int, seed: int, device):
    """SURVEY 8(d) synthetic OHLCV (polars_quant_amd/synthetic.py: bit-identical to the generator the tests use)."""
    from polars_quant_amd.synthetic import gen_ohlcv
    d = gen_ohlcv(seed, n_sym, T, 0)
    return {k: torch.from_numpy(v).to(device) for k, v in d.items()}


def cpu_baseline(sample_syms: int, T: int):
    """the scalar C restatement, -O3 -march=native (built on this host), OpenMP over symbols on every core
    this process may use, on the bench's own data set (all 5000 symbols)."""
    from oracle import pq_oracle as oracle   # the ONLY use of oracle/ here: the timed CPU baseline leg
    from polars_quant_amd.synthetic import gen_ohlcv
    d = gen_ohlcv(SEED, sample_syms, T, 0)
    avail = len(os.sched_getaffinity(0))
    native = oracle.native_lib() is not None
    # "all cores" = every core this process may use; a GPU box gives a job a CPU SHARE smaller than the affinity mask (the first
    # run on 256 threads was slower than 16): the thread count is swept over {16, 32, 64, all} and the best one reported as `cores`
    reps, best = 3, None
    for cores in sorted({min(c, avail) for c in (16, 32, 64, avail)}):
        oracle.suite_bench({k: v[:2 * cores] for k, v in d.items()}, cores, native=native)  # spin up the OpenMP team
        t0 = time.perf_counter()
        for _ in range(reps):
            oracle.suite_bench(d, cores, native=native)
        t = (time.perf_counter() - t0) / reps
        if best is None or t < best[0]:
            best = (t, cores)
    t_all, cores = best
    small = {k: v[: max(8, sample_syms // 16)] for k, v in d.items()}
    t0 = time.perf_counter(); oracle.suite_bench(small, 1, native=native); t_one = time.perf_counter() - t0
    rows = sample_syms * T
    cpu = "?"
    try:
        cpu = next(l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name"))
    except Exception:  # noqa: BLE001
        pass
    import subprocess
    cc = subprocess.run(["gcc", "--version"], capture_output=True, text=True).stdout.splitlines()[0]
    flags = "-O3 -march=native -ffp-contract=off (built on this host)" if native else "-O3 -ffp-contract=off (the -march=native build failed here: portable build)"
    return {"value": rows / t_all, "unit": "rows/s", "cores": cores, "kind": "port",
            "sample": f"{sample_syms} symbols x {T} days (the bench's own data set), same suite+backtest, mean of {reps} passes; oracle = scalar C "
                      f"restatement of the reference (the Rust reference cannot be built here), {cc}, {flags}, "
                      f"OpenMP over symbols on {cores} threads (best of 16 / 32 / 64 / all {avail}) of '{cpu}'",
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
    # the frame uploads into the columns the recorded suite READS: the suite re-houses inputs handed over at a slow row pitch (a dense
    # [n, 2520] tensor) into its own pitched buffers, and an upload into the caller's tensors would neither reach the step nor fit them
    cols = dict(getattr(suite, "_ohlcv", None) or suite.house(ohlcv))
    frame = DeviceFrame(columns=cols, stride=suite.stride)
    frame.register(host)
    summ = np.empty((n_local, 8))
    check(L.pq_host_register(summ.ctypes.data_as(C.c_void_p), summ.nbytes))
    outs = [t for ts in suite.out.values() for t in ts] + list(suite.pat.values()) + suite.bt
    big = np.empty(max(t.numel() * t.element_size() for t in outs), dtype=np.uint8)
    check(L.pq_host_register(big.ctypes.data_as(C.c_void_p), big.nbytes))
    order = list(suite.STAGE_ORDER)

    def serial(all_outputs):
        frame.upload(host, order=order, copy_stream=torch.cuda.current_stream(dev))
        suite.run()                                    # (the upload went into the columns the recording reads)
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
    res["stages"] = suite.record_staged(cols)         # (replaces the single recorded suite of this object; reads the buffers the frame fills)
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


def visible_gpus():
    """number of GPUs this process would see, WITHOUT touching the HIP / HSA runtime (torch.cuda.device_count() may fall back to
    hipGetDeviceCount, which initialises it -- and a GPU-initialised parent must not start the ranks): kfd topology nodes with SIMDs,
    narrowed by HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES.  None: unknown (then --gpus is trusted and the children fail loudly)."""
    nodes = Path("/sys/class/kfd/kfd/topology/nodes")
    if not nodes.is_dir():
        return 0              # no kfd driver: no AMD GPU on this host
    try:
        n = 0
        for node in sorted(nodes.iterdir()):
            props = dict(l.split(None, 1) for l in (node / "properties").read_text().splitlines() if " " in l)
            n += int(props.get("simd_count", "0")) > 0
    except Exception:  # noqa: BLE001
        return None
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def launch_ranks(args) -> int:
    """`python bench.py --gpus N` without a launcher: N child processes, one per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*
    as torch.distributed.run would set them).  The parent makes no HIP / HSA call at all -- GPUs are counted from the kfd topology
    in sysfs (visible_gpus) -- so the children are ordinary fresh processes; rank 0 prints the JSON line on the inherited stdout."""
    import socket
    import subprocess
    n = args.gpus
    have = visible_gpus()
    if have is not None and have < n and not args.dry_run:
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
    ap.add_argument("--symbols", type=int, default=N_SYM, help="symbols in total (strong scaling) / per GPU (weak scaling)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default=None,
                    help="default: strong when --gpus > 1 (BASELINE.json: 5000 symbols at 1/2/4/8 GPUs), weak at --gpus 1 (same thing)")
    ap.add_argument("--e2e", action="store_true", help="also time the step end to end from registered host buffers")
    ap.add_argument("--days", type=int, default=T_DAYS)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary figures (dense layout at 1 GPU, weak scaling at N GPUs)")
    ap.add_argument("--stride", type=int, default=0,
                    help="row pitch, in elements, of the INPUT columns the caller hands over (0 = days rounded up to a multiple of 16 = "
                         "128 B; = days: dense).  The suite owns its outputs and re-houses inputs handed over at a pitch that is not a "
                         "multiple of 128 B once, at record time (Suite.house); --exact-layout keeps the caller's pitch for everything")
    ap.add_argument("--rehearse-exchange", action="store_true",
                    help="run the N-GPU code path (communicator, exchange modes, backtest_only modes, weak-scaling secondary) with a world of one")
    ap.add_argument("--exact-layout", action="store_true", help="run on the caller's row pitch exactly (measures a slow layout as it is)")
    ap.add_argument("--dry-run", action="store_true",
                    help="launch / rendezvous / shard ranges only (gloo on the CPU, no GPU work): what `--gpus N` would run where")
    args = ap.parse_args()
    default_scaling = args.scaling is None
    if default_scaling:
        args.scaling = "strong" if args.gpus > 1 else "weak"

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
    # --rehearse-exchange: the code path of an N-GPU run -- process group, the C-ABI communicator built on a helper thread, the exchange
    # modes, the backtest_only figures, the weak-scaling secondary -- on ONE GPU with a world of one (a one-GPU box cannot run N > 1)
    multi = world > 1 or args.rehearse_exchange
    if args.rehearse_exchange:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    dist = None
    if multi:
        import torch.distributed as dist_
        dist = dist_
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if torch.cuda.device_count() <= local_rank:
            print(f"bench.py: rank {rank} needs GPU {local_rank}, this host shows {torch.cuda.device_count()} visible GPUs", file=sys.stderr)
            sys.exit(2)
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    assert torch.cuda.is_available(), "bench.py needs an MI355X; there is no CPU path"
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    from polars_quant_amd.suite import Suite
    from polars_quant_amd.distributed import gather_summaries, shard_range
    T = args.days
    stride = args.stride or (T + 15) // 16 * 16

    def build(scaling):
        """-> (suite, ohlcv, n_local, n_total) for this rank; inputs resident in HBM on the pitched layout"""
        if scaling == "strong":     # one data set of --symbols symbols, split statically over the ranks
            lo, hi = shard_range(args.symbols, rank, world)
            n_local, n_total = hi - lo, args.symbols
            full = make_inputs(args.symbols, T, SEED, torch.device("cpu"))
            ohlcv = {k: v[lo:hi].contiguous().to(dev) for k, v in full.items()}
            del full
        else:
            n_local, n_total = args.symbols, args.symbols * world
            ohlcv = make_inputs(n_local, T, SEED + rank, dev)       # every rank: its own symbols
        # Device columns are pitched like a hipMallocPitch allocation: a row pitch that is a multiple of 128 B makes every 64 / 128-byte
        # tile piece one aligned cache line (dense 2520-element rows start at odd multiples of 64 B).  `--stride <days>` measures the
        # dense layout.
        if stride != T:  # re-house the inputs with the padded row pitch
            for k in list(ohlcv):
                buf = torch.zeros((n_local, stride), dtype=torch.float64, device=dev)
                buf[:, :T] = ohlcv[k]
                ohlcv[k] = buf[:, :T]
        st = Suite(n_local, T, dev, stride=stride, exact_layout=args.exact_layout)
        st.record(ohlcv)   # one-time: turn the step's calls into job grids, re-house slow-pitch inputs (not part of the timed region)
        return st, ohlcv, n_local, n_total

    # ---- the one exchange of the path (world > 1): per-symbol summary rows to every rank, through the PRODUCT's own collective --
    # pq_comm_init + pq_gather_summaries of the C ABI (csrc/comm.hip: RCCL all-gather over xGMI on the context's stream).  The
    # communicator is built on a helper thread with a deadline; if it cannot be built, the step falls back to torch.distributed's
    # all_gather and the line says so (collective.timed).
    comm, comm_note, th = None, None, None
    if multi:
        import threading
        box = {}
        # (the suite's three side streams come into being with its first replay: that replay happens HERE, before the communicator and
        #  its stream(s) exist -- the runtime hands out its four hardware queues in creation order, and a step whose chains share a queue
        #  takes 5.0 instead of 3.9 ms)
        suite0, ohlcv0, _nl0, _nt0 = build(args.scaling)
        suite0.run(ohlcv0)
        torch.cuda.synchronize()

        def mk_comm():
            # The communicator lives on the context of the stream the steps run on (its in-series gather is then one more kernel on that
            # stream: +1.3 us at a world of one; bound to another stream every gather would pay two cross-stream dependencies, ~30 us).
            # The TRIAL exchange runs on the communicator's own stream and is waited for on the host from this helper thread: a
            # communicator that cannot be built -- or whose first collective never returns -- hangs this thread, not the steps' stream.
            try:
                torch.cuda.set_device(dev)   # (the current device is per thread)
                from polars_quant_amd.distributed import CabiComm
                c = CabiComm(dev, rank, world)
                trial = c.trial()
                if not torch.equal(trial[:, 0].cpu(), torch.arange(world, dtype=torch.float64)):
                    raise RuntimeError("trial gather returned the wrong rows")
                box["comm"] = c
            except Exception as e:  # noqa: BLE001
                box["err"] = str(e)

        th = threading.Thread(target=mk_comm, daemon=True)
        th.start()
        th.join(120.0)
        ok = torch.tensor([1 if "comm" in box else 0], device=dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)        # all ranks or none
        if int(ok.item()) == 1:
            comm = box["comm"]
        else:
            comm_note = box.get("err", "no answer within 120 s")

    from polars_quant_amd.distributed import OverlappedGather

    def time_loop(run_slot, local, n_total, steps, warmup, mode, on_timed=None):
        """`steps` timed steps of run_slot(slot) (which writes the summary rows local[slot]) + the per-step exchange:
          overlapped  the exchange of step k on the communicator's own stream beside the kernels of step k + 1 (two slots;
                      pq_gather_summaries_begin / _end, or torch.distributed's async collective if the C-ABI communicator is missing)
          serial      the exchange on the step's stream behind every step (round 4's form)
          kernel_only no exchange
        Every exchange completes inside the timed region (drain before the closing synchronize)."""
        og = OverlappedGather(n_total, local[0].shape[0], dev, comm=comm, local=local) if (multi and mode == "overlapped") else None

        def step(k):
            if not multi or mode == "kernel_only":
                run_slot(k & 1)
            elif og is not None:
                slot = og.acquire()
                run_slot(slot)
                og.begin(slot)
            else:
                run_slot(0)
                if comm is not None:
                    comm.gather_summaries(local[0], n_total)
                else:
                    gather_summaries(local[0], n_total)
        for k in range(warmup):
            step(k)
        if og is not None:
            og.drain()
        torch.cuda.synchronize()
        if on_timed:
            on_timed()
        if multi:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(steps):
            step(k)
        if og is not None:
            og.drain()
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if multi:
            tmax = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            el = float(tmax.item())
        return el

    def time_steps(st, ohlcv, n_total, steps, warmup, timing=False, mode="overlapped"):
        if multi and len(getattr(st, "_summaries", [])) < 2:   # two recordings of the step, one per summary buffer
            st.record(ohlcv, summaries=[st.summary, torch.empty_like(st.summary)])
        two = len(st._summaries) == 2
        # (run() without tensors: the inputs are resident -- where record() re-housed a slow-pitch column, resident means the suite's copy)
        return time_loop(lambda slot: st.run(slot=slot if two else 0), st._summaries if two else [st.summary, st.summary], n_total,
                         steps, warmup, mode, on_timed=(lambda: st.set_timing(True)) if timing else None)

    def backtest_only(ohlcv, n_local, n_total, steps, warmup, stride):
        """BASELINE config 3 by itself -- the MACD-cross backtest with summary on this rank's shard, DIRECT C-ABI calls (what
        north_star's `>= 6x at 8 GPUs on the symbol-sharded backtest` is about): ms per step in the three exchange modes."""
        import ctypes as C
        from polars_quant_amd._lib import Batch, BtParams, check, lib
        from polars_quant_amd._spec import BT_DEFAULTS
        from polars_quant_amd.api import ctx
        L, h, b, prm = lib(), ctx(dev.index), Batch(n_local, T, stride), BtParams(**BT_DEFAULTS)
        close = ohlcv["close"]
        curves = [torch.empty((n_local, stride), dtype=torch.float64, device=dev) for _ in range(3)]
        local = [torch.empty((n_local, 8), dtype=torch.float64, device=dev) for _ in range(2)]

        def run_slot(slot):
            check(L.pq_backtest_macd_cross(h, C.byref(b), C.c_void_p(close.data_ptr()), 12, 26, 9, C.byref(prm),
                                           *[C.c_void_p(t.data_ptr()) for t in curves], C.c_void_p(local[slot].data_ptr())))
        res = {"symbols_per_gpu": n_local, "symbols_total": n_total, "steps": steps,
               "what": "pq_backtest_macd_cross (signals + scan + summary, position / cash / equity columns written) on the rank's shard"}
        for m in (("overlapped", "serial", "kernel_only") if multi else ("kernel_only",)):
            el = time_loop(run_slot, local, n_total, steps, warmup, m)
            res[m + "_ms_per_step"] = el / steps * 1e3
        best = min(res["overlapped_ms_per_step"], res["serial_ms_per_step"]) if multi else res["kernel_only_ms_per_step"]
        if multi:
            res["chosen"] = "serial" if res["serial_ms_per_step"] <= res["overlapped_ms_per_step"] else "overlapped"
        # For the record, NOT part of `value`: two INDEPENDENT steps in flight (what a host that sweeps strategy parameters over resident
        # columns does): step k on stream k & 1, each stream with its own context and its own output columns, no exchange.  A 625-symbol
        # shard is 625 workgroups on 256 CUs, every wave at its own dependency latency -- a second step fills the idle SIMDs.
        try:
            streams = [torch.cuda.Stream(dev) for _ in range(2)]
            sets = []
            for st_ in streams:
                with torch.cuda.stream(st_):
                    sets.append((ctx(dev.index), [torch.empty((n_local, stride), dtype=torch.float64, device=dev) for _ in range(3)],
                                 torch.empty((n_local, 8), dtype=torch.float64, device=dev)))

            def run2(k):
                hh, cv, sm = sets[k & 1]
                with torch.cuda.stream(streams[k & 1]):
                    check(L.pq_backtest_macd_cross(hh, C.byref(b), C.c_void_p(close.data_ptr()), 12, 26, 9, C.byref(prm),
                                                   *[C.c_void_p(t.data_ptr()) for t in cv], C.c_void_p(sm.data_ptr())))
            for k in range(2 * warmup):
                run2(k)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k in range(2 * steps):
                run2(k)
            torch.cuda.synchronize()
            res["two_steps_in_flight_ms_per_step"] = (time.perf_counter() - t0) / (2 * steps) * 1e3
            res["two_steps_in_flight_same_summary"] = bool(torch.equal(sets[0][2].view(torch.int64), sets[1][2].view(torch.int64)))
        except Exception as e:  # noqa: BLE001
            res["two_steps_in_flight_ms_per_step"] = f"failed: {e}"
        res["value"] = n_total * T / (best * 1e-3)
        res["unit"] = "rows/s"
        return res

    suite, ohlcv, n_local, n_total = (suite0, ohlcv0, _nl0, _nt0) if multi else build(args.scaling)
    exchange_modes, mode = None, "kernel_only"
    if multi:
        # The exchange can run in series behind every step (one more small kernel on the step's stream) or double-buffered on the
        # communicator's own stream beside the next step.  Which one is cheaper is a property of the runtime and of the step's length
        # -- a cross-stream dependency costs tens of microseconds on this runtime (scripts/bench_gather.py: a 51 us backtest step
        # becomes 53 us with the gather in series and 136 us with it overlapped, world of one) -- so both are timed in short runs
        # outside the headline's timed region (max over ranks, so every rank picks the same) and the faster one is used.
        short = max(5, args.steps // 2)
        exchange_modes = {m + "_ms_per_step": time_steps(suite, ohlcv, n_total, short, 2, mode=m) / short * 1e3
                          for m in ("serial", "overlapped", "kernel_only")}
        mode = "serial" if exchange_modes["serial_ms_per_step"] <= exchange_modes["overlapped_ms_per_step"] else "overlapped"
        exchange_modes["chosen"] = mode
    elapsed = time_steps(suite, ohlcv, n_total, args.steps, args.warmup, timing=True, mode=mode)
    gather_check = None
    if multi and comm is not None:   # cross-check, outside the timed region: the C-ABI gather against torch.distributed's
        ref = gather_summaries(suite.summary, n_total)
        got = comm.gather_summaries(suite.summary, n_total)
        torch.cuda.synchronize()
        gather_check = bool(torch.equal(got.view(torch.int64), ref.view(torch.int64)))
    bt_only = None
    if not args.no_secondary:
        bt_only = backtest_only(suite._ohlcv, n_local, n_total, max(20, args.steps), 3, suite.stride)

    # ---- roofline of the dominant kernel ----------------------------------------------------------------
    # The step's device time is dominated by seq_jobs_kernel<0>: the tiled bodies of all sequential jobs, launched per LDS class, the
    # launches running CONCURRENTLY with each other and with the row-parallel kernels.  Each launch was bracketed by HIP events on
    # its own launch stream during the timed steps above (pq_suite_set_timing).  Contract figure: achieved = mean algorithmic bytes
    # per launch / mean launch duration (= what `rocprofv3 --kernel-trace --stats` reports as that kernel's average duration,
    # profiles/).  Because the launches overlap, each one sees only its share of the chip; the whole step is reported three ways:
    # per-call algorithmic bytes (config.suite_algorithmic_GBps), counter bytes (roofline.step_frac_counter_bytes) and the fused
    # floor (roofline.step_frac_fused_floor) -- SURVEY 8(d): "always state which denominator".
    grids = [g for g in suite.grid_stats() if g["runs"] > 0]
    rows_local = n_local * T
    by_kernel = {}
    for g in grids:
        by_kernel.setdefault(g["kernel"], []).append(g)
    dom_name = max(by_kernel, key=lambda k: sum(g["avg_ms"] * g["runs"] for g in by_kernel[k])) if by_kernel else None
    dom = by_kernel.get(dom_name, [])
    n_launch = sum(g["runs"] for g in dom)
    if n_launch:
        mean_ms = sum(g["avg_ms"] * g["runs"] for g in dom) / n_launch
        mean_bytes = sum(g["alg_bytes"] * g["runs"] for g in dom) / n_launch
        achieved = mean_bytes / (mean_ms * 1e-3) / 1e9
    else:
        mean_ms = mean_bytes = achieved = None
    span_ms, span_bytes = suite.span_stats(0)   # the launches of the tiled job kernel as one concurrent set
    ms_step = elapsed / args.steps * 1e3
    suite_bytes = suite.suite_bytes_per_row() * rows_local
    suite_gbs = suite_bytes / (ms_step * 1e-3) / 1e9
    fused_gbs = FUSED_FLOOR_BYTES_PER_ROW * rows_local / (ms_step * 1e-3) / 1e9
    # HBM traffic from the PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate runs of this command; summary committed by
    # scripts/pmc_summary.py).  Only quoted for the build (source hash) and the configuration it was collected on.
    traffic = step_traffic = traffic_note = None
    if PMC_FILE.exists() and n_local == N_SYM and T == T_DAYS and suite.stride == (T_DAYS + 15) // 16 * 16:
        pm = json.loads(PMC_FILE.read_text())
        if pm.get("source_hash") == source_hash():
            kernels = pm["kernels"]
            per_step = pm.get("launches_per_step", {})
            k = kernels.get(dom_name)
            traffic = k["hbm_bytes_per_launch"] if k else None
            step_traffic = sum(v["hbm_bytes_per_launch"] * per_step.get(name, 1) for name, v in kernels.items()
                               if not name.startswith(("at::", "__amd")))   # torch's own kernels excluded
            traffic_note = f"profiles/{PMC_FILE.name} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command on this build, source hash {pm['source_hash']}; not measured in this run)"
        else:
            traffic_note = f"profiles/{PMC_FILE.name} was collected on another build (source hash {pm.get('source_hash')} != {source_hash()}): not quoted"

    line = None
    if rank == 0:
        rows_total = n_total * T * args.steps
        line = {
            "metric": "indicator+backtest rows/sec, 5000 sym x 2520 day f64 OHLCV",
            "value": rows_total / elapsed, "unit": "rows/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_step, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "collective": ({"per_step": "one all-gather of the [n_local, 8] summary rows per step, completed inside the timed region; `step_ms.chosen` = "
                                        "the faster of: in series on the step's stream / double-buffered on the communicator's own stream beside "
                                        "the next step (both timed in short runs before the headline run)",
                            "timed": (("pq_gather_summaries" if mode == "serial" else "pq_gather_summaries_begin / _end") + " (C ABI, csrc/comm.hip: RCCL ncclAllGather)"
                                      if comm is not None
                                      else f"torch.distributed all_gather_into_tensor (the C-ABI communicator could not be built: {comm_note})"),
                            "step_ms": exchange_modes,
                            "torch_backend": dist.get_backend(), "world_size_seen": dist.get_world_size(),
                            "c_abi_equals_torch_gather": gather_check}
                           if multi else None),
            "backtest_only": bt_only,
            "config": {"workload": f"full talib suite ({len(suite.tasks()) - 2} indicator calls + 61 fused candlestick "
                                   f"recognisers) + fused MACD-cross backtest with summary, {n_total} symbols x {T} days "
                                   f"f64 OHLCV ({n_local} per GPU), inputs resident in HBM",
                       "symbols_per_gpu": n_local, "symbols_total": n_total, "days": T, "row_pitch_elements": suite.stride,
                       "input_pitch_elements": stride, "inputs_rehoused_at_record_time": sorted(suite._housed),
                       "parallelism": f"symbol-sharded x{world}",
                       "algorithmic_bytes_per_row": suite.suite_bytes_per_row(),
                       "suite_algorithmic_GBps": suite_gbs,
                       "suite_frac_of_hbm_peak_on_per_call_bytes": suite_gbs / HBM_PEAK_GBS},
            "roofline": {"bound": "hbm", "kernel": dom_name, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": (achieved / HBM_PEAK_GBS) if achieved else None, "traffic": traffic,
                         "traffic_source": traffic_note,
                         "denominators": "frac: per-call algorithmic bytes (SURVEY 8d B/row of every reference call a job replaces) of ONE launch of "
                                         "the dominant kernel / its duration; step_frac_counter_bytes: HBM bytes of the whole step from the PMC "
                                         "counters / step time / peak; step_frac_fused_floor: 928 B/row (inputs once + every output once) / step "
                                         "time / peak; config.suite_frac_of_hbm_peak_on_per_call_bytes: 1 856 B/row (one call per function, inputs "
                                         "re-read per call) / step time / peak -- may exceed what moved",
                         "fused_lower_bound_bytes_per_row": FUSED_FLOOR_BYTES_PER_ROW,
                         "step_frac_fused_floor": fused_gbs / HBM_PEAK_GBS if world == 1 else None,
                         "step_traffic": step_traffic,
                         "step_traffic_GBps": step_traffic / (ms_step * 1e-3) / 1e9 if step_traffic and world == 1 else None,
                         "step_frac_counter_bytes": step_traffic / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS if step_traffic and world == 1 else None,
                         "copy_GBps_measured": COPY_GBS,
                         "algorithmic_bytes_per_launch": mean_bytes, "avg_launch_ms": mean_ms,
                         "launches_per_step": len(dom),
                         "concurrent_set": {"note": "the launches of seq_jobs_kernel<0> overlap inside a step: their summed algorithmic bytes "
                                                    "over the time from the earliest start to the latest end (HIP events)",
                                            "span_ms": span_ms, "algorithmic_bytes": span_bytes,
                                            "achieved": span_bytes / (span_ms * 1e-3) / 1e9 if span_ms else None,
                                            "frac": span_bytes / (span_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if span_ms else None},
                         "grids": [{k: g[k] for k in ("kernel", "avg_ms", "alg_bytes", "n_jobs", "lds_bytes")} for g in grids]},
        }
        if args.e2e and world == 1:
            line["config"]["e2e"] = end_to_end(suite, ohlcv, n_local, T, dev)
    # ---- secondary figures, outside the timed region above
    if world == 1 and not multi and not args.no_secondary and n_local == N_SYM and args.scaling == "weak":
        # Strong-scaling proxy.  BASELINE's metric is 5 000 symbols IN TOTAL on 1 / 2 / 4 / 8 GPUs; the path has no data-path collective, so
        # a rank's step at N / G symbols is what ONE GPU needs for a shard of that size (+ the one summary all-gather).  No 8-GPU node is
        # given to this pool's driver, so the shard times are measured here, on this GPU, and the factors they project are spelled out
        # with the collective latency they ASSUME -- twice: the measured cost of pq_gather_summaries at a world of one, and a
        # conservative figure for eight ranks over xGMI.
        proxy = {"what": "the same recorded step / the MACD-cross backtest alone on the first N / G symbols of the data set, one GPU; projected factor at G GPUs = "
                         "t(N) / (t(N / G) + assumed collective latency)",
                 "assumed_collective_latency_ms": {"measured_world_of_one": 0.005, "conservative_8_ranks": 0.025,
                                                   "note": "0.005: pq_gather_summaries in series behind a step at a world of one (profiles/r05_bench_gather.json: "
                                                           "4.1 us + launch; RCCL's one-rank all-gather is a copy kernel); 0.025: assumed for an 8-rank ring "
                                                           "all-gather of 40 KB per rank over xGMI -- NOT measured (no multi-GPU node on this pool)"},
                 "suite_step_ms": {str(n_local): ms_step}, "backtest_only_ms": {str(n_local): bt_only["kernel_only_ms_per_step"] if bt_only else None}}
        for g_ in (2, 4, 8):
            n_sh = shard_range(n_local, 0, g_)[1]
            sub = {k: v[:n_sh] for k, v in suite._ohlcv.items()}          # (rows keep the pitched layout: used in place)
            sst = Suite(n_sh, T, dev, stride=suite.stride, exact_layout=args.exact_layout)
            sst.record(sub)
            el = time_steps(sst, sub, n_sh, max(10, args.steps), args.warmup)
            proxy["suite_step_ms"][str(n_sh)] = el / max(10, args.steps) * 1e3
            proxy.setdefault("suite_plan", {})[str(n_sh)] = sst.info()
            sst.close()
            proxy["backtest_only_ms"][str(n_sh)] = backtest_only(sub, n_sh, n_sh, max(20, args.steps), 3, suite.stride)["kernel_only_ms_per_step"]
        for key in ("suite_step_ms", "backtest_only_ms"):
            t = proxy[key]
            if t[str(n_local)] is None:
                continue
            proxy[key.replace("_ms", "_projected_factor")] = {
                lat: {f"{g_}gpu": t[str(n_local)] / (t[str(shard_range(n_local, 0, g_)[1])] + L) for g_ in (2, 4, 8)}
                for lat, L in (("collective_0.005ms", 0.005), ("collective_0.025ms", 0.025))}
        line["strong_scaling_proxy"] = proxy
    if world == 1 and suite.stride != T and not args.no_secondary and not args.no_cpu_baseline:
        # for the record: the same step on the DENSE layout (row pitch = days)
        suite.close()
        dense_in = {k: v.contiguous() for k, v in ohlcv.items()}
        dense = Suite(n_local, T, dev, stride=T, exact_layout=True)
        dense.record(dense_in)
        el = time_steps(dense, dense_in, n_total, args.steps, args.warmup)
        line["config"]["dense_layout"] = {"row_pitch_elements": T, "ms_per_step": el / args.steps * 1e3}
        dense.close()
    if multi and default_scaling and not args.no_secondary:
        # the weak-scaling figure (5000 symbols PER GPU) beside the BASELINE configuration (5000 symbols in total)
        suite.close()
        del ohlcv
        wsuite, wohlcv, wn_local, wn_total = build("weak")
        wsteps = max(5, args.steps // 2)
        el = time_steps(wsuite, wohlcv, wn_total, wsteps, 2, mode=mode)
        if rank == 0:
            line["weak_scaling"] = {"symbols_per_gpu": wn_local, "symbols_total": wn_total, "steps": wsteps, "ms_per_step": el / wsteps * 1e3,
                                    "value": wn_total * T * wsteps / el, "unit": "rows/s"}
        wsuite.close()
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(N_SYM, T)
        print(json.dumps(line), flush=True)
    if multi:
        if comm is not None:
            try:
                comm.close()
            except Exception:  # noqa: BLE001
                pass
        elif th is not None and th.is_alive():
            sys.stdout.flush()
            os._exit(0)   # the measurement is out; do not wait for a communicator that will not come up
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
