"""A Strategy in front of the backtest, two ways (README.md:862-994 -> VectorizedBacktester): the stochastic strategy -- STOCH, its %K / %D
cross (pq_cross_signals), the oversold / overbought zones (pq_gate_signals) -- followed by pq_backtest_vectorized over 5000 x 2520:
  eager   four C-ABI calls, one launch (or more) each
  suite   the same four calls recorded once (pq_suite_begin / end) and replayed: the signal columns stay on the device between the
          indicator job, the rule kernels and the wave-per-symbol backtest, and the launches are a recorded plan
Prints one JSON object (profiles/r03_bench_strategy.json)."""
import ctypes as C
import json
import sys

sys.path.insert(0, ".")
import torch

from polars_quant_amd import api
from polars_quant_amd._lib import Batch, BtParams, check, lib
from polars_quant_amd._spec import BT_DEFAULTS
from polars_quant_amd.synthetic import gen_ohlcv

N, T = 5000, 2520
PITCH = (T + 15) // 16 * 16
d = gen_ohlcv(0x5EED0002, N, T, 0)
dev = torch.device("cuda")
col = {}
for k in ("high", "low", "close"):
    buf = torch.zeros((N, PITCH), dtype=torch.float64, device=dev)
    buf[:, :T] = torch.from_numpy(d[k]).to(dev)
    col[k] = buf
f64 = lambda: torch.empty((N, PITCH), dtype=torch.float64, device=dev)
u8 = lambda: torch.zeros((N, PITCH), dtype=torch.uint8, device=dev)
k_, d_, b0, s0, b1, s1 = f64(), f64(), u8(), u8(), u8(), u8()
pos, cash, eq, summ = f64(), f64(), f64(), torch.empty((N, 8), dtype=torch.float64, device=dev)
L, h, b, prm = lib(), api.ctx(0), Batch(N, T, PITCH), BtParams(**BT_DEFAULTS)
P = lambda t: C.c_void_p(t.data_ptr())


def calls():
    check(L.pq_stoch(h, C.byref(b), P(col["high"]), P(col["low"]), P(col["close"]), 5, 3, 0, 3, 0, P(k_), P(d_)))
    check(L.pq_cross_signals(h, C.byref(b), P(k_), P(d_), P(b0), P(s0)))
    check(L.pq_gate_signals(h, C.byref(b), P(k_), None, 0, C.c_double(20.0), C.c_double(80.0), P(b0), P(s0), P(b1), P(s1)))
    check(L.pq_backtest_vectorized(h, C.byref(b), P(col["close"]), P(b1), P(s1), None, C.byref(prm), P(pos), P(cash), P(eq), P(summ)))


def t_event(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps


eager_ms = t_event(calls)
torch.cuda.synchronize()
ref = [t.clone() for t in (pos, cash, eq, summ, b1, s1)]
check(L.pq_suite_begin(h, C.byref(b)))
calls()
suite = C.c_void_p()
check(L.pq_suite_end(h, C.byref(suite)))
for t in (pos, cash, eq, summ):
    t.fill_(-1.0)
suite_ms = t_event(lambda: check(L.pq_suite_run(h, suite)))
torch.cuda.synchronize()
same = all(torch.equal(a.view(torch.int64) if a.dtype == torch.float64 else a, r.view(torch.int64) if r.dtype == torch.float64 else r)
           for a, r in zip((pos[:, :T], cash[:, :T], eq[:, :T], summ, b1[:, :T], s1[:, :T]), (ref[0][:, :T], ref[1][:, :T], ref[2][:, :T], ref[3], ref[4][:, :T], ref[5][:, :T])))
check(L.pq_suite_destroy(h, suite))
print(json.dumps({"workload": f"Strategy.stoch -> VectorizedBacktester, {N} x {T}", "eager_ms": eager_ms, "recorded_suite_ms": suite_ms,
                  "identical_results": bool(same), "trades_total": float(summ[:, 7].sum())}, indent=1))
