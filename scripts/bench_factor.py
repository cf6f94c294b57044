"""BASELINE config 4: 10000 symbols x 5040 days, factor IC / Rank-IC (+ rolling IC).  HIP path (inputs resident in HBM)
next to the oracle on one host thread (on a sample of the days)."""
import sys, time; sys.path.insert(0, ".")
import numpy as np, torch
from polars_quant_amd import api
from oracle import pq_oracle as oracle
N, T = 10000, 5040
g = torch.Generator(device="cuda"); g.manual_seed(1)
f = torch.randn((N, T), dtype=torch.float64, device="cuda", generator=g)
r = 0.1 * f + torch.randn((N, T), dtype=torch.float64, device="cuda", generator=g)
def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): out = fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps, out
ms_p, (ic, nv) = timed(lambda: api.factor_ic(f, r, 0))
ms_r, (ric, _) = timed(lambda: api.factor_ic(f, r, 1))
ms_w, _ = timed(lambda: api.rolling_ic(ic, 60))
rows = N * T
print(f"IC       {ms_p:8.2f} ms  {rows/ms_p/1e6:7.2f} G cells/s  {16*rows/ms_p/1e6:6.0f} GB/s algorithmic (2 f64 columns read once)")
print(f"Rank-IC  {ms_r:8.2f} ms  {rows/ms_r/1e6:7.2f} G cells/s")
print(f"rolling  {ms_w:8.3f} ms")
Ts = 64
fs, rs = f[:, :Ts].cpu().numpy().copy(), r[:, :Ts].cpu().numpy().copy()
t0 = time.perf_counter(); eic, _ = oracle.factor_ic(fs, rs, 0); t1 = time.perf_counter(); eric, _ = oracle.factor_ic(fs, rs, 1); t2 = time.perf_counter()
print(f"oracle, 1 thread, {Ts} days: IC {N*Ts/(t1-t0)/1e6:.1f} M cells/s, Rank-IC {N*Ts/(t2-t1)/1e6:.1f} M cells/s")
print("parity on the sample:", bool((ic[:Ts].cpu().numpy() == eic).all()), bool((ric[:Ts].cpu().numpy() == eric).all()))
