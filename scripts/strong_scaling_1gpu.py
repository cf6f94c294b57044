"""Strong-scaling projection from ONE GPU (run on the GPU box): BASELINE config 3 splits 5000 symbols over 8 GPUs (625 each).
A rank's step time at N/G symbols is what one GPU needs for a shard of that size (the path has no data-path collective; the
320 KB summary gather is latency only), so   projected speed-up at G GPUs = t(N) / t(N / G)   for
  (a) the MACD-cross backtest alone (pq_backtest_macd_cross, the kernel config 3 names) and
  (b) the whole suite step (what bench.py times).
Prints one JSON object; committed under profiles/."""
import json
import sys
import time

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import ctypes as C

import torch

from polars_quant_amd import api
from polars_quant_amd.suite import Suite
from polars_quant_amd.synthetic import gen_ohlcv

T = 2520
full = gen_ohlcv(0x5EED0002, 5000, T, 0)


def t_event(fn, reps):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps


PITCH = (T + 15) // 16 * 16     # device columns pitched to 128 B, as in bench.py
out = {"days": T, "row_pitch_elements": PITCH, "symbols_total": 5000, "backtest_macd_cross_ms": {}, "suite_step_ms": {}}
for n in (5000, 2500, 1250, 625):
    g = {}
    for k, v in full.items():
        buf = torch.zeros((n, PITCH), dtype=torch.float64, device="cuda")
        buf[:, :T] = torch.from_numpy(v[:n].copy()).cuda()
        g[k] = buf[:, :T]
    # the C ABI called directly on preallocated, pitched output columns (what bench.py's suite and a non-Python host do; the Python
    # convenience wrapper adds ~25 us of allocations per call, a third of the 625-symbol kernel)
    from polars_quant_amd._lib import Batch, BtParams, check, lib
    from polars_quant_amd._spec import BT_DEFAULTS
    _b, _prm, _h = Batch(n, T, PITCH), BtParams(**BT_DEFAULTS), api.ctx(0)
    _o = [torch.empty((n, PITCH), dtype=torch.float64, device="cuda") for _ in range(3)]
    _sm = torch.empty((n, 8), dtype=torch.float64, device="cuda")
    _vp = lambda t: C.c_void_p(t.data_ptr())
    _close = g["close"]
    out["backtest_macd_cross_ms"][n] = t_event(lambda: check(lib().pq_backtest_macd_cross(_h, C.byref(_b), _vp(_close), 12, 26, 9, C.byref(_prm),
                                                                                          *[_vp(t) for t in _o], _vp(_sm))), 20)
    st = Suite(n, T, "cuda", stride=PITCH)
    st.record(g)
    out.setdefault("suite_plan", {})[n] = st.info()      # phases / jobs / ROW launches of the plan this shard size records
    out["suite_step_ms"][n] = t_event(lambda: st.run(), 20)
    st.close()
for key in ("backtest_macd_cross_ms", "suite_step_ms"):
    t = out[key]
    out[key.replace("_ms", "_projected_speedup")] = {f"{g}gpu": t[5000] / t[5000 // g] for g in (2, 4, 8)}
print(json.dumps(out, indent=1))
