"""One-wave against four-wave form of pq_backtest_macd_cross over batch sizes (PQ_BT_WAVES forced per run): where the library's
choice of the form comes from (csrc/backtest.hip, bt_wave)."""
import os, sys, ctypes as C
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch
from polars_quant_amd import api
from polars_quant_amd._lib import Batch, BtParams, check, lib
from polars_quant_amd._spec import BT_DEFAULTS
from polars_quant_amd.synthetic import gen_ohlcv
T, PITCH = 2520, 2528
full = gen_ohlcv(0x5EED0002, 5000, T, 0)["close"]
def t_event(fn, reps):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps
for n in (256, 512, 625, 768, 900, 1024, 1250, 1536, 1792, 2048, 2500, 3072, 3750, 5000):
    buf = torch.zeros((n, PITCH), dtype=torch.float64, device="cuda"); buf[:, :T] = torch.from_numpy(full[:n].copy()).cuda()
    b, prm, h = Batch(n, T, PITCH), BtParams(**BT_DEFAULTS), api.ctx(0)
    o = [torch.empty((n, PITCH), dtype=torch.float64, device="cuda") for _ in range(3)]
    sm = torch.empty((n, 8), dtype=torch.float64, device="cuda")
    vp = lambda t: C.c_void_p(t.data_ptr())
    r = {}
    for wv in ("1", "4"):
        os.environ["PQ_BT_WAVES"] = wv
        r[wv] = round(1000 * t_event(lambda: check(lib().pq_backtest_macd_cross(h, C.byref(b), vp(buf), 12, 26, 9, C.byref(prm), *[vp(t) for t in o], vp(sm))), 60), 1)
    print(n, r, "<- 4 waves" if r["4"] < r["1"] else "")
