"""Times pq_backtest_macd_cross / pq_backtest_vectorized alone on shards of 5000 / 2500 / 1250 / 625 symbols x 2520 days (pitched
columns as in bench.py) and prints the wave form's chunk statistics.  PQ_BT_LANE_FORM=1 times the lane-per-symbol form."""
import json
import sys

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch

from polars_quant_amd import api
from polars_quant_amd.synthetic import gen_ohlcv

T = int(sys.argv[1]) if len(sys.argv) > 1 else 2520
full = gen_ohlcv(0x5EED0002, 5000, T, 0)
PITCH = (T + 15) // 16 * 16


def t_event(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps


out = {"days": T, "macd_cross_ms": {}, "macd_cross_summary_only_ms": {}, "vectorized_ms": {}}
for n in (5000, 2500, 1250, 625):
    buf = torch.zeros((n, PITCH), dtype=torch.float64, device="cuda")
    buf[:, :T] = torch.from_numpy(full["close"][:n].copy()).cuda()
    close = buf[:, :T]
    b = api.Batch(n, T, PITCH)
    import ctypes as C
    from polars_quant_amd._lib import check, lib
    from polars_quant_amd.api import BtParams, BT_DEFAULTS, ctx
    prm = BtParams(**BT_DEFAULTS)
    pos, cash, eq = (torch.empty((n, PITCH), dtype=torch.float64, device="cuda") for _ in range(3))
    summ = torch.empty((n, 8), dtype=torch.float64, device="cuda")
    vp = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    h = ctx(0)
    def run(curves=True):
        check(lib().pq_backtest_macd_cross(h, C.byref(b), vp(buf), 12, 26, 9, C.byref(prm), vp(pos) if curves else None,
                                           vp(cash) if curves else None, vp(eq) if curves else None, vp(summ)))
    api.backtest_wave_stats(reset=True)
    out["macd_cross_ms"][n] = t_event(run)
    st = api.backtest_wave_stats(reset=True)
    out.setdefault("wave_stats_per_call", {})[n] = [v / 23 for v in st]
    out["macd_cross_summary_only_ms"][n] = t_event(lambda: run(False))
    bu, se = api.macd_cross_signals(close)
    bu = bu.contiguous(); se = se.contiguous()
    dense = close.contiguous()
    out["vectorized_ms"][n] = t_event(lambda: api.backtest_vectorized(dense, bu, se))
# HIP-graph replay of the shard step: the same call captured once on a side stream and replayed -- what a host that
# repeats a fixed step (a parameter sweep over the same columns) can do; launch-to-launch time of `reps` replays
out["macd_cross_graph_replay_ms"] = {}
for n in (5000, 625):
    try:
        buf = torch.zeros((n, PITCH), dtype=torch.float64, device="cuda")
        buf[:, :T] = torch.from_numpy(full["close"][:n].copy()).cuda()
        b = api.Batch(n, T, PITCH)
        pos, cash, eq = (torch.empty((n, PITCH), dtype=torch.float64, device="cuda") for _ in range(3))
        summ = torch.empty((n, 8), dtype=torch.float64, device="cuda")
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            hs = ctx(0)          # the context of the capture stream exists before the capture (its creation allocates)
            call = lambda: check(lib().pq_backtest_macd_cross(hs, C.byref(b), vp(buf), 12, 26, 9, C.byref(prm), vp(pos), vp(cash), vp(eq), vp(summ)))
            for _ in range(3): call()
            side.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):
                call()
            out["macd_cross_graph_replay_ms"][n] = t_event(g.replay, reps=50)
    except Exception as e:  # noqa: BLE001
        out["macd_cross_graph_replay_ms"][n] = f"capture failed: {e}"
# two INDEPENDENT steps in flight (stream k & 1, own context, own outputs): what a host sweeping strategy parameters over resident columns does
out["macd_cross_two_in_flight_ms"] = {}
for n in (5000, 625):
    buf = torch.zeros((n, PITCH), dtype=torch.float64, device="cuda")
    buf[:, :T] = torch.from_numpy(full["close"][:n].copy()).cuda()
    b = api.Batch(n, T, PITCH)
    streams = [torch.cuda.Stream() for _ in range(2)]
    sets = []
    for st_ in streams:
        with torch.cuda.stream(st_):
            sets.append((ctx(0), [torch.empty((n, PITCH), dtype=torch.float64, device="cuda") for _ in range(3)], torch.empty((n, 8), dtype=torch.float64, device="cuda")))
    cnt = [0]
    def run2():
        k = cnt[0]; cnt[0] += 1
        hh, cv, sm = sets[k & 1]
        with torch.cuda.stream(streams[k & 1]):
            check(lib().pq_backtest_macd_cross(hh, C.byref(b), vp(buf), 12, 26, 9, C.byref(prm), *[vp(t) for t in cv], vp(sm)))
    import time
    for _ in range(10): run2()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(400): run2()
    torch.cuda.synchronize()
    out["macd_cross_two_in_flight_ms"][n] = (time.perf_counter() - t0) / 400 * 1e3
t = out["macd_cross_ms"]
out["projected_speedup"] = {f"{g}gpu": t[5000] / t[5000 // g] for g in (2, 4, 8)}
tw = out["macd_cross_two_in_flight_ms"]
out["projected_speedup_two_in_flight_8gpu"] = tw[5000] / tw[625]
gr = out["macd_cross_graph_replay_ms"]
if all(isinstance(gr.get(n), float) for n in (5000, 625)):
    out["projected_speedup_graph_replay_8gpu"] = gr[5000] / gr[625]
out["rows_per_s_5000"] = 5000 * T / (t[5000] * 1e-3)
out["alg_GBps_5000"] = 5000 * T * 32 / (t[5000] * 1e-3) / 1e9
print(json.dumps(out, indent=1))
