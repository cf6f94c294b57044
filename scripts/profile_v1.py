"""Per-kernel timing survey: every indicator once (default params) + fused patterns + MACD backtest."""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import polars_quant_amd as pq
from polars_quant_amd import api
from oracle import pq_oracle as oracle

N, T = int(sys.argv[1]) if len(sys.argv) > 1 else 5000, 2520
d = oracle.gen_ohlcv(0x5EED0002, N, T, 0)
g = {k: torch.from_numpy(v).cuda() for k, v in d.items()}
g["real"] = g["close"]
g["periods"] = torch.from_numpy(np.tile((2 + np.arange(T) % 29).astype(np.float64), (N, 1))).cuda()
torch.cuda.synchronize()
rows = N * T
res = []
for rep in range(2):
    for name in sorted(pq.SPEC):
        cols = pq.SPEC[name][0]
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = api.call(name, *[g[c] for c in cols])
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        nb = 8 * len(cols) + sum(8 if dt_ == "f8" else 4 for _, dt_ in pq.SPEC[name][2])
        if rep: res.append((name, dt * 1e3, nb * rows / dt / 1e9))
        del out
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = api.cdl_all(g["open"], g["high"], g["low"], g["close"])
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    if rep: res.append(("cdl_all", dt * 1e3, (32 + 244) * rows / dt / 1e9))
    del out
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = api.backtest_macd_cross(g["close"])
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    if rep: res.append(("backtest_macd", dt * 1e3, 32 * rows / dt / 1e9))
    del out
tot = sum(r[1] for r in res)
for name, ms, gbs in sorted(res, key=lambda r: -r[1]):
    print(f"{name:14s} {ms:9.3f} ms  {gbs:8.1f} GB/s")
print(f"TOTAL {tot:.2f} ms -> {rows / tot * 1e3 / 1e9:.3f} G rows/s")
