"""Per-phase device time of the wave-per-symbol indicator kernels (a PQ_WT_PROF build: scripts/ab_build.sh wtprof -DPQ_WT_PROF wt;
run with PQ_LIB_PATH=ab/libpq_wtprof.so).  Prints kilo-cycles per wave (= per symbol) and phase."""
import ctypes as C
import os
import sys

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch

from polars_quant_amd import api
from polars_quant_amd._lib import Batch, check, lib
from polars_quant_amd.synthetic import gen_ohlcv

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
L.pq_wt_prof.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.c_int32]
CALLS = {
    "ema_all": lambda: L.pq_ema_all(h, C.byref(b), P("close"), 30, *V(4)),
    "macd_pair": lambda: L.pq_macd_pair(h, C.byref(b), P("close"), 12, 26, 9, 9, *V(6)),
    "rsi": lambda: L.pq_rsi(h, C.byref(b), P("close"), 14, *V(1)),
    "dmi_all": lambda: L.pq_dmi_all(h, C.byref(b), P("high"), P("low"), P("close"), 14, *V(5)),
    "atr_all": lambda: L.pq_atr_all(h, C.byref(b), P("high"), P("low"), P("close"), 14, *V(2)),
    "midpoint": lambda: L.pq_midpoint(h, C.byref(b), P("close"), 14, *V(1)),
}
NAMES = ["stage", "anchor seed", "horner+scan", "warm-up", "own walk", "verify+rerun", "emit", "stage_fn", "store", "map"]
out = (C.c_int64 * 16)()
for name, fn in CALLS.items():
    for _ in range(2):
        check(fn())
    check(L.pq_wt_prof(h, out, 1))
    reps = 5
    for _ in range(reps):
        check(fn())
    check(L.pq_wt_prof(h, out, 1))
    v = [x / (reps * N) / 1e3 for x in out]
    print(f"{name:10s} " + "  ".join(f"{nm} {x:.1f}" for nm, x in zip(NAMES, v)) + f"   sum {sum(v):.1f} kcycles = {sum(v) / 2.4:.1f} us")
