"""BASELINE config 5: 5000 symbols x 2520 days, leveraged/margin backtest with commission + slippage and the daily
portfolio metrics.  Prints rows/s of the HIP path (inputs resident in HBM) and of the oracle on the host cores."""
import sys, time; sys.path.insert(0, ".")
import numpy as np, torch
from polars_quant_amd import api
from oracle import pq_oracle as oracle
N, T = 5000, 2520
d = oracle.gen_ohlcv(0x5EED0002, N, T, 0)
close = torch.from_numpy(d["close"]).cuda()
buy, sell = api.macd_cross_signals(close)
bench = close[0].clone()
kw = dict(leverage=2.0, slippage=0.001, max_trades=64)
for _ in range(2):
    r = api.backtest_leveraged(close, buy, sell, bench, **kw); m = api.portfolio_metrics(r["total_value"], 1e5 * N, bench)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    r = api.backtest_leveraged(close, buy, sell, bench, **kw)
e1.record(); e1.synchronize()
ms_bt = e0.elapsed_time(e1) / 5
e0.record()
for _ in range(5):
    m = api.portfolio_metrics(r["total_value"], 1e5 * N, bench)
e1.record(); e1.synchronize()
ms_pm = e0.elapsed_time(e1) / 5
print(f"backtest kernel {ms_bt:.3f} ms (incl. output allocation), portfolio metrics {ms_pm:.3f} ms")
# the kernel alone: preallocated outputs, the C ABI called directly (what a host that keeps its buffers does)
import ctypes as C0
from polars_quant_amd._lib import Batch as B0, LevParams as LP0, check as ck0, lib as lib0
from polars_quant_amd._spec import LEV_DEFAULTS as LD0
_o = [torch.empty((N, T), dtype=torch.float64, device="cuda") for _ in range(3)]
_cnt, _sm = torch.zeros(N, dtype=torch.int32, device="cuda"), torch.empty((N, 8), dtype=torch.float64, device="cuda")
_prm, _b, _h = LP0(**{**LD0, "leverage": 2.0, "slippage": 0.001}), B0(N, T, T), api.ctx(0)
_vp = lambda t: C0.c_void_p(t.data_ptr())
def _k():
    ck0(lib0().pq_backtest_leveraged(_h, C0.byref(_b), _vp(close), _vp(buy), _vp(sell), _vp(bench), C0.byref(_prm), *[_vp(t) for t in _o], 0, _vp(_cnt),
                                     *([None] * 8), _vp(_sm)))
for _ in range(3): _k()
torch.cuda.synchronize(); e0.record()
for _ in range(10): _k()
e1.record(); e1.synchronize()
print(f"pq_backtest_leveraged alone, preallocated outputs: {e0.elapsed_time(e1) / 10:.3f} ms")
ms = ms_bt + ms_pm
alg = (8 + 2 + 24) * N * T + 8 * N * T  # price + 2 signals in, 3 columns out; the portfolio pass re-reads total_value
print(f"HIP: {ms:.3f} ms/step  {N*T/ms/1e3:.1f} M rows/s  {alg/ms/1e6:.0f} GB/s algorithmic  trades={int(r['trade_count'].sum())}")
ns = 512
b, s = buy[:ns].cpu().numpy(), sell[:ns].cpu().numpy()
t0 = time.perf_counter(); e = oracle.backtest_leveraged(d["close"][:ns], b, s, d["close"][0], leverage=2.0, slippage=0.001); oracle.portfolio_metrics(e["total_value"], 1e5 * ns, d["close"][0]); dt = time.perf_counter() - t0
print(f"oracle (1 thread, {ns} symbols): {ns*T/dt/1e6:.2f} M rows/s")

# --- the same engine as ONE job of a recorded suite: beside the indicator suite it is 79 workgroups among ~2400
import ctypes as C
from polars_quant_amd._lib import Batch, LevParams, check, lib
from polars_quant_amd._spec import LEV_DEFAULTS
from polars_quant_amd.suite import Suite
g = {k: torch.from_numpy(v).cuda() for k, v in d.items()}
st = Suite(N, T, "cuda", exact_layout=True)   # dense columns shared with the direct leveraged call
L, h = lib(), api.ctx(0)
cash, sv, tv = (torch.empty((N, T), dtype=torch.float64, device="cuda") for _ in range(3))
cnt = torch.zeros(N, dtype=torch.int32, device="cuda")
summ = torch.empty((N, 8), dtype=torch.float64, device="cuda")
prm = LevParams(**{**LEV_DEFAULTS, "leverage": 2.0, "slippage": 0.001})
vp = lambda t: C.c_void_p(t.data_ptr())
def timed(with_lev):
    st.close()
    check(L.pq_suite_begin(h, C.byref(st.batch)))
    for name in st.tasks(fused=True):
        st.run_one(name, g)
    if with_lev:
        check(L.pq_backtest_leveraged(h, C.byref(st.batch), vp(close), vp(buy), vp(sell), vp(bench), C.byref(prm), vp(cash), vp(sv), vp(tv), 0, vp(cnt),
                                      *([None] * 8), vp(summ)))
    out = C.c_void_p()
    check(L.pq_suite_end(h, C.byref(out)))
    st._suite = out; st._suites = [out]; st._summaries = [st.summary]
    for _ in range(3): st.run()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(10): st.run()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / 10
a, b_ = timed(False), timed(True)
ok = (tv.cpu().numpy().view(np.uint64) == r["total_value"].cpu().numpy().view(np.uint64)).all()
print(f"suite step without / with the leveraged backtest as one more job: {a:.3f} / {b_:.3f} ms  (+{b_ - a:.3f} ms in-suite vs {ms_bt:.3f} ms solo); "
      f"recorded result == direct call: {ok}")
