"""Solo times of the functions that have a one-symbol-per-wavefront form (csrc/ops_wt.h) at 5000 x 2520 on the pitched layout:
wave form against the lane-per-symbol form (PQ_NO_WT=1), direct C-ABI calls on preallocated outputs.  GPU box only."""
import ctypes as C
import json
import os
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from polars_quant_amd import api  # noqa: E402
from polars_quant_amd._lib import Batch, check, lib  # noqa: E402
from polars_quant_amd.synthetic import gen_ohlcv  # noqa: E402

N, T, S = int(os.environ.get("N", 5000)), 2520, 2528
d = gen_ohlcv(0x5EED0002, N, T, 0)
dev = {}
for k, v in d.items():
    buf = torch.zeros((N, S), dtype=torch.float64, device="cuda")
    buf[:, :T] = torch.from_numpy(v).cuda()
    dev[k] = buf
b = Batch(N, T, S)
P = lambda k: C.c_void_p(dev[k].data_ptr())
outs = [torch.empty((N, S), dtype=torch.float64, device="cuda") for _ in range(7)]
V = lambda m: [C.c_void_p(t.data_ptr()) for t in outs[:m]]
h, L = api.ctx(0), lib()
CALLS = {
    "ema (1 in, 1 out)": (2, lambda: L.pq_ema(h, C.byref(b), P("close"), 30, *V(1))),
    "tema (1 in, 1 out)": (2, lambda: L.pq_tema(h, C.byref(b), P("close"), 30, *V(1))),
    "trix (1 in, 1 out)": (2, lambda: L.pq_trix(h, C.byref(b), P("close"), 30, *V(1))),
    "macd (1 in, 3 out)": (4, lambda: L.pq_macd(h, C.byref(b), P("close"), 12, 26, 9, *V(3))),
    "adx (3 in, 1 out)": (4, lambda: L.pq_adx(h, C.byref(b), P("high"), P("low"), P("close"), 14, *V(1))),
    "atr (3 in, 1 out)": (4, lambda: L.pq_atr(h, C.byref(b), P("high"), P("low"), P("close"), 14, *V(1))),
    "plus_dm (2 in, 1 out)": (3, lambda: L.pq_plus_dm(h, C.byref(b), P("high"), P("low"), 14, *V(1))),
    "ema_all (1 in, 4 out)": (5, lambda: L.pq_ema_all(h, C.byref(b), P("close"), 30, *V(4))),
    "macd_pair (1 in, 6 out)": (7, lambda: L.pq_macd_pair(h, C.byref(b), P("close"), 12, 26, 9, 9, *V(6))),
    "rsi (1 in, 1 out)": (2, lambda: L.pq_rsi(h, C.byref(b), P("close"), 14, *V(1))),
    "dm_pair (2 in, 2 out)": (4, lambda: L.pq_dm_pair(h, C.byref(b), P("high"), P("low"), 14, *V(2))),
    "dmi_all (3 in, 5 out)": (8, lambda: L.pq_dmi_all(h, C.byref(b), P("high"), P("low"), P("close"), 14, *V(5))),
    "atr_all (3 in, 2 out)": (5, lambda: L.pq_atr_all(h, C.byref(b), P("high"), P("low"), P("close"), 14, *V(2))),
    "midpoint (1 in, 1 out)": (2, lambda: L.pq_midpoint(h, C.byref(b), P("close"), 14, *V(1))),
    "midprice (2 in, 1 out)": (3, lambda: L.pq_midprice(h, C.byref(b), P("high"), P("low"), 14, *V(1))),
}


def timed(fn, reps=20):
    for _ in range(3):
        check(fn())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        check(fn())
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


res = {}
for name, (cols, fn) in CALLS.items():
    os.environ.pop("PQ_NO_WT", None)
    os.environ["PQ_WT_ALL"] = "1"   # every wave form, also those the library does not use by default (csrc/wt.hip)
    api.wt_stats(reset=True)
    t_wave = timed(fn)
    st = api.wt_stats()
    os.environ["PQ_NO_WT"] = "1"
    t_lane = timed(fn)
    gb = cols * N * T * 8 / 1e9
    res[name] = {"wave_ms": round(t_wave, 4), "lane_ms": round(t_lane, 4), "columns_GB": round(gb, 3), "wave_TBps": round(gb / t_wave, 3),
                 "lane_TBps": round(gb / t_lane, 3), "failed_chunks_per_call": st[1] / 23, "reruns_per_call": st[2] / 23}
    print(name, res[name], flush=True)
# the same functions on a RAGGED batch (pq_batch.offsets: the long columns of `.over("symbol")`, here N groups of T rows each, dense):
# the wave form against the per-lane gather body these batches ran before
import numpy as np  # noqa: E402
long_cols = {k: torch.from_numpy(np.ascontiguousarray(v.reshape(-1))).cuda() for k, v in d.items()}
offs = torch.arange(0, (N + 1) * T, T, dtype=torch.int64, device="cuda")
rb = Batch(N, T, N * T, C.c_void_p(offs.data_ptr()))
routs = [torch.empty(N * T, dtype=torch.float64, device="cuda") for _ in range(3)]
RP = lambda k: C.c_void_p(long_cols[k].data_ptr())
RV = lambda m: [C.c_void_p(t.data_ptr()) for t in routs[:m]]
RCALLS = {
    "ragged ema": (2, lambda: L.pq_ema(h, C.byref(rb), RP("close"), 30, *RV(1))),
    "ragged trix": (2, lambda: L.pq_trix(h, C.byref(rb), RP("close"), 30, *RV(1))),
    "ragged rsi": (2, lambda: L.pq_rsi(h, C.byref(rb), RP("close"), 14, *RV(1))),
    "ragged macd": (4, lambda: L.pq_macd(h, C.byref(rb), RP("close"), 12, 26, 9, *RV(3))),
    "ragged atr": (4, lambda: L.pq_atr(h, C.byref(rb), RP("high"), RP("low"), RP("close"), 14, *RV(1))),
    "ragged midpoint": (2, lambda: L.pq_midpoint(h, C.byref(rb), RP("close"), 14, *RV(1))),
    "ragged tema": (2, lambda: L.pq_tema(h, C.byref(rb), RP("close"), 30, *RV(1))),
    "ragged dema": (2, lambda: L.pq_dema(h, C.byref(rb), RP("close"), 30, *RV(1))),
    "ragged adx": (4, lambda: L.pq_adx(h, C.byref(rb), RP("high"), RP("low"), RP("close"), 14, *RV(1))),
    "ragged dx": (4, lambda: L.pq_dx(h, C.byref(rb), RP("high"), RP("low"), RP("close"), 14, *RV(1))),
    "ragged plus_dm": (3, lambda: L.pq_plus_dm(h, C.byref(rb), RP("high"), RP("low"), 14, *RV(1))),
    "ragged natr": (4, lambda: L.pq_natr(h, C.byref(rb), RP("high"), RP("low"), RP("close"), 14, *RV(1))),
}
rres = {}
for name, (cols, fn) in RCALLS.items():
    os.environ.pop("PQ_NO_WT", None)
    os.environ["PQ_WT_ALL"] = "1"
    t_wave = timed(fn, 10)
    os.environ["PQ_NO_WT"] = "1"
    t_lane = timed(fn, 5)
    rres[name] = {"wave_ms": round(t_wave, 4), "gather_ms": round(t_lane, 4)}
    print(name, rres[name], flush=True)
os.environ.pop("PQ_NO_WT", None)
print(json.dumps({"symbols": N, "days": T, "row_pitch_elements": S, "results": res, "ragged": rres}))
