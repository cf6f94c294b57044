"""Per-phase device time of the wave-per-symbol backtest (a PQ_BTW_PROF build: scripts/ab_build.sh btwprof -DPQ_BTW_PROF backtest;
run with PQ_LIB_PATH=ab/libpq_btwprof.so).  Prints mean microseconds per wave: load, signals, walk + fill, summary."""
import ctypes as C
import sys

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch

from polars_quant_amd import api
from polars_quant_amd._lib import check, lib
from polars_quant_amd.synthetic import gen_ohlcv

T = 2520
for n in (5000, 625):
    close = torch.from_numpy(gen_ohlcv(0x5EED0002, n, T, 0)["close"]).cuda()
    L = lib()
    L.pq_backtest_wave_prof.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.c_int32]
    out = (C.c_int64 * 16)()
    for _ in range(3): api.backtest_macd_cross(close)
    check(L.pq_backtest_wave_prof(api.ctx(0), out, 1))
    reps = 10
    for _ in range(reps): api.backtest_macd_cross(close)
    check(L.pq_backtest_wave_prof(api.ctx(0), out, 1))
    v = [x / (reps * n) for x in out]   # s_memtime ticks = shader clocks (2.4 GHz, scripts/ubench/f64lat.hip)
    print(n, "head detail: init %.1f  seeding rows %.1f  lane-0 steady rows %.1f  hand-over %.1f  pass A + scan %.1f  pass B + scan %.1f  seeds %.1f" % tuple(x / 1e3 for x in (v[14], v[15], v[10], v[11], v[12], v[13], v[5])))
    print(n, "kilo-cycles per wave: load %.1f  signals [first chunk %.1f, warm-up chunks %.1f, own chunk %.1f, hand-over + bit test %.1f]  event marking %.1f  "
          "event list + reciprocals %.1f  chain %.1f  fill %.1f  summary %.1f  (sum %.1f = %.1f us)" %
          (v[0] / 1e3, (v[5] + v[10] + v[11] + v[12] + v[13]) / 1e3, v[6] / 1e3, v[7] / 1e3, v[4] / 1e3, v[1] / 1e3, v[8] / 1e3, v[9] / 1e3, v[2] / 1e3, v[3] / 1e3, sum(v) / 1e3, sum(v) / 2400.0))
