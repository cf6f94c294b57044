"""The reference's own call path, end to end from HOST Arrow buffers (SURVEY 8(d) "(b) end-to-end"): a frame of 5 000 symbols x 2 520
days sorted by symbol, as pyarrow arrays; `polars` itself is not installed in this image, so the caller's side of the plugin ABI is
played by pyarrow as in tests/test_polars_plugin.py.
  over        ONE call of `_polars_plugin_<f>_over` on the whole columns + the key column      (what this library adds)
  per_group   5 000 calls of `_polars_plugin_<f>`, one per group                              (what `.over("symbol")` does, momentum.py:13-16)
  oracle      the CPU restatement of the reference on the same host, ONE thread (a baseline, not a target; the suite-level figure on all cores is bench.py's cpu_baseline)
for ema / macd / cdlengulfing; `over` twice -- the first call uploads the columns, the second finds them in the input-column cache
(csrc/plugin.hip) --, with PQ_PLUGIN_CACHE_MB=0 and with PQ_PLUGIN_HOSTPOOL_MB=0 (every call's result in fresh host memory).   Prints one JSON object (profiles/<round>_bench_plugin.json)."""
import ctypes as C
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
import pyarrow as pa

import test_polars_plugin as tp
from oracle import pq_oracle as oracle

N, T = int(os.environ.get("PQ_BENCH_SYMBOLS", 5000)), 2520
L = tp._lib()
d = oracle.gen_ohlcv(0x5EED0002, N, T, 0)
cols = {k: pa.array(d[k].reshape(-1)) for k in ("open", "high", "low", "close")}
key = pa.array(np.repeat(np.arange(N, dtype=np.int32), T))
rows = N * T


def t_of(fn, reps=3):
    fn()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); best = min(best, time.perf_counter() - t0)
    return best


def over(name, names, **kw):
    return tp._plugin_call(L, name + "_over", [([cols[k]], k) for k in names] + [([key], "symbol")], kwargs=kw or None)


def per_group(name, names, n_groups, **kw):
    out = []
    for g in range(n_groups):
        out.append(tp._plugin_call(L, name, [([cols[k].slice(g * T, T)], k) for k in names], kwargs=kw or None))
    return out


out = {"symbols": N, "days": T, "rows": rows, "host": "pyarrow arrays (the caller's side of the Polars plugin ABI); results returned as Arrow arrays on the host",
       "functions": {}}
L.pq_plugin_cache_clear.restype = None
SAMPLE = 250     # per-group calls are timed on a sample of the groups and scaled (5 000 Python-driven calls of ~0.3 ms each)
for name, names, kw, ocall in (("ema", ("close",), {"timeperiod": 30}, lambda x: oracle.call("ema", x["close"], timeperiod=30)),
                               ("macd", ("close",), {}, lambda x: oracle.call("macd", x["close"])),
                               ("cdlengulfing", ("open", "high", "low", "close"), {}, lambda x: oracle.pattern("cdlengulfing", x["open"], x["high"], x["low"], x["close"]))):
    r = {}
    L.pq_plugin_cache_clear()
    t0 = time.perf_counter(); over(name, names, **kw); r["over_first_call_ms"] = (time.perf_counter() - t0) * 1e3
    r["over_cached_ms"] = t_of(lambda: over(name, names, **kw)) * 1e3
    os.environ["PQ_PLUGIN_CACHE_MB"] = "0"
    r["over_no_cache_ms"] = t_of(lambda: over(name, names, **kw)) * 1e3
    del os.environ["PQ_PLUGIN_CACHE_MB"]
    # the host side of the result columns comes from a pool of released blocks (csrc/plugin.hip HostBuf): a caller that KEEPS every result --
    # sixty expressions of one with_columns -- finds the pool empty and pays the page faults of fresh memory; that case:
    os.environ["PQ_PLUGIN_HOSTPOOL_MB"] = "0"
    r["over_cached_no_host_pool_ms"] = t_of(lambda: over(name, names, **kw)) * 1e3
    del os.environ["PQ_PLUGIN_HOSTPOOL_MB"]
    L.pq_plugin_cache_clear()
    t = t_of(lambda: per_group(name, names, SAMPLE, **kw), reps=2)
    r["per_group_ms"] = t / SAMPLE * N * 1e3
    r["per_group_note"] = f"{SAMPLE} of the {N} per-group calls timed, scaled"
    L.pq_plugin_cache_clear()
    sub = {k: d[k] for k in names}
    r["oracle_one_thread_ms"] = t_of(lambda: ocall(sub), reps=2) * 1e3 if hasattr(oracle, "call") else None
    for k in ("over_first_call_ms", "over_cached_ms", "over_no_cache_ms", "over_cached_no_host_pool_ms", "per_group_ms", "oracle_one_thread_ms"):
        if r.get(k):
            r[k.replace("_ms", "_rows_per_s")] = rows / (r[k] * 1e-3)
    out["functions"][name] = r
print(json.dumps(out, indent=1))
