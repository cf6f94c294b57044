"""not-gpu, world_size 2 over gloo: the symbol sharding + the one summary gather of the multi-GPU path.

On CPU the product cannot compute (no HIP device), so each rank produces its shard's summary rows with the
oracle; what is under test is polars_quant_amd.distributed (static split, ragged gather, symbol order)."""
import os
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent


def _worker(rank, world, n_sym, port, q):
    sys.path.insert(0, str(ROOT))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import pq_oracle as oracle
    from polars_quant_amd.distributed import gather_summaries, shard_range
    T = 300
    d = oracle.gen_ohlcv(0x5EED0003, n_sym, T, 0)          # every rank can regenerate any symbol deterministically
    lo, hi = shard_range(n_sym, rank, world)
    close = d["close"][lo:hi]
    buy, sell = oracle.macd_cross_signals(close)
    _, _, _, summ = oracle.backtest(close, buy, sell)
    full = gather_summaries(torch.from_numpy(summ.reshape(hi - lo, 8)), n_sym)
    if rank == 0:
        q.put(full.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_sym", [10, 7])      # even split and ragged split
def test_two_rank_symbol_sharding(n_sym, oracle):
    from polars_quant_amd.distributed import shard_range
    assert shard_range(10, 0, 2) == (0, 5) and shard_range(10, 1, 2) == (5, 10)
    assert shard_range(7, 0, 2) == (0, 3) and shard_range(7, 1, 2) == (3, 7)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + n_sym
    procs = [ctx.Process(target=_worker, args=(r, 2, n_sym, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    d = oracle.gen_ohlcv(0x5EED0003, n_sym, 300, 0)
    buy, sell = oracle.macd_cross_signals(d["close"])
    _, _, _, exp = oracle.backtest(d["close"], buy, sell)
    assert got.shape == (n_sym, 8)
    assert (got.view(np.uint64) == exp.view(np.uint64)).all()


def _worker_days(rank, world, n_sym, T, port, q):
    sys.path.insert(0, str(ROOT))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import pq_oracle as oracle
    from polars_quant_amd.distributed import factor_ic_day_sharded, shard_range
    rng = np.random.default_rng(12)                       # every rank regenerates the same data and keeps its symbol shard
    f = rng.normal(size=(n_sym, T)); r = 0.3 * f + rng.normal(size=(n_sym, T))
    f[rng.random((n_sym, T)) < 0.05] = oracle.NULL
    lo, hi = shard_range(n_sym, rank, world)

    def compute(ff, rr, m):                                # CPU stand-in for api.factor_ic (no HIP device here)
        ic, nv = oracle.factor_ic(ff.numpy(), rr.numpy(), method=m)
        return torch.from_numpy(ic), torch.from_numpy(nv)
    res = {}
    for m in (0, 1):
        ic, nv = factor_ic_day_sharded(torch.from_numpy(f[lo:hi].copy()), torch.from_numpy(r[lo:hi].copy()), n_sym, method=m, compute=compute)
        res[m] = (ic.numpy(), nv.numpy())
    if rank == 1:                                          # any rank holds the full series
        q.put(res)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_sym,T", [(12, 40), (7, 33)])   # even and ragged splits of both axes
def test_two_rank_day_sharded_factor_ic(n_sym, T, oracle):
    """BASELINE config 4 across GPUs: symbol-sharded columns -> one all-to-all -> day-sharded IC / Rank-IC -> gather; the
    assembled series equals the single-process oracle bit for bit (cross-sections are independent per day)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000) + n_sym
    procs = [ctx.Process(target=_worker_days, args=(r, 2, n_sym, T, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    rng = np.random.default_rng(12)
    f = rng.normal(size=(n_sym, T)); r = 0.3 * f + rng.normal(size=(n_sym, T))
    f[rng.random((n_sym, T)) < 0.05] = oracle.NULL
    for m in (0, 1):
        eic, env = oracle.factor_ic(f, r, method=m)
        gic, gnv = got[m]
        assert (gnv == env).all()
        assert ((gic.view(np.uint64) == eic.view(np.uint64)) | (np.isnan(gic) & np.isnan(eic))).all()


def _run_bench(*argv, env=None):
    import json
    import os
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    e.update(env or {})
    p = subprocess.run([sys.executable, str(root / "bench.py"), *argv], capture_output=True, text=True, timeout=300, env=e)
    lines = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")]
    return p, lines


def test_bench_gpus_flag_launches_one_process_per_rank():
    """`python bench.py --gpus N` (no launcher) starts N ranks with the right shards; BASELINE config 3's strong split."""
    p, lines = _run_bench("--gpus", "2", "--dry-run", "--scaling", "strong")
    assert p.returncode == 0, p.stderr
    assert len(lines) == 1, "rank 0 prints exactly one line"
    d = lines[0]
    assert d["n_gpus"] == 2 and d["world_size_seen"] == 2 and d["symbols_total"] == 5000
    assert [s["symbols"] for s in d["shards"]] == [[0, 2500], [2500, 5000]]
    assert [s["rank"] for s in d["shards"]] == [0, 1] and [s["local_rank"] for s in d["shards"]] == [0, 1]
    p, lines = _run_bench("--gpus", "3", "--dry-run", "--scaling", "strong", "--symbols", "1000")
    assert [s["symbols"] for s in lines[0]["shards"]] == [[0, 333], [333, 666], [666, 1000]]       # ragged: floor(N r / G)
    p, lines = _run_bench("--gpus", "2", "--dry-run")                    # default at N > 1 = BASELINE's configuration: 5000 in total
    assert lines[0]["scaling"] == "strong" and [s["symbols"] for s in lines[0]["shards"]] == [[0, 2500], [2500, 5000]]
    p, lines = _run_bench("--gpus", "2", "--dry-run", "--scaling", "weak")                          # weak: 5000 per rank, own seeds
    assert [s["symbols"] for s in lines[0]["shards"]] == [[0, 5000], [5000, 10000]] and lines[0]["symbols_total"] == 10000
    assert lines[0]["shards"][0]["seed"] != lines[0]["shards"][1]["seed"]


def test_bench_refuses_to_run_fewer_gpus_than_asked():
    import torch
    have = torch.cuda.device_count()
    p, lines = _run_bench("--gpus", str(have + 2), "--steps", "1")
    assert p.returncode != 0 and not lines, "an N-GPU run on fewer devices must fail loudly, not print n_gpus: 1"
    assert "visible GPUs" in p.stderr
    # under an external launcher the flag must agree with WORLD_SIZE
    p, lines = _run_bench("--gpus", "2", "--dry-run", env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert p.returncode != 0 and "WORLD_SIZE" in p.stderr


def test_shard_range_of_the_c_abi_matches_python():
    import ctypes as C
    from polars_quant_amd._lib import lib
    from polars_quant_amd.distributed import shard_range
    L = lib()
    for n, g in ((5000, 8), (1000, 3), (7, 8), (0, 4), (10**12, 7)):
        for r in range(g):
            lo, hi = C.c_int64(), C.c_int64()
            assert L.pq_shard_range(n, r, g, C.byref(lo), C.byref(hi)) == 0
            assert (lo.value, hi.value) == shard_range(n, r, g)


def _worker_overlapped(rank, world, n_sym, port, q):
    sys.path.insert(0, str(ROOT))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from polars_quant_amd.distributed import OverlappedGather, shard_range
    lo, hi = shard_range(n_sym, rank, world)
    og = OverlappedGather(n_sym, hi - lo, "cpu")
    tables, held = {}, {}
    for k in range(6):
        slot = og.acquire()                                   # the exchange that used this slot two steps ago has completed
        if slot in held:
            tables[held[slot]] = og.all[slot].clone()
        # "the step": this rank's rows of step k (symbol s, column c -> 1000 k + 8 s + c), written into the slot's local buffer
        s_idx = torch.arange(lo, hi, dtype=torch.float64)[:, None]
        og.local[slot].copy_(1000.0 * k + 8.0 * s_idx + torch.arange(8, dtype=torch.float64)[None, :])
        og.begin(slot)                                        # returns with the collective in flight
        held[slot] = k
    og.drain()
    for slot, k in held.items():
        tables[k] = og.all[slot].clone()
    if rank == 1:
        q.put({k: v.numpy() for k, v in tables.items()})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_sym", [10, 7])      # even split (one all-gather) and ragged split (grouped broadcasts)
def test_two_rank_overlapped_double_buffered_gather(n_sym):
    """distributed.OverlappedGather over gloo, world 2: six steps, two slots, the exchange of step k in flight while step k + 1 fills
    the other slot; every step's table arrives complete, in symbol order, and belongs to ITS step."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000) + n_sym
    procs = [ctx.Process(target=_worker_overlapped, args=(r, 2, n_sym, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert sorted(got) == [0, 1, 2, 3, 4, 5]
    for k, table in got.items():
        exp = 1000.0 * k + 8.0 * np.arange(n_sym)[:, None] + np.arange(8)[None, :]
        assert table.shape == (n_sym, 8) and (table == exp).all(), k
