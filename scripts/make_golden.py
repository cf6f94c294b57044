"""Writes the committed golden fixtures tests/golden/*.npz: the CPU oracle's outputs for every function of the path on the
three SURVEY 8(d) data sets (clean random walk, pattern-rich, null-bearing), 4 symbols x 200 days, default parameters plus one
non-default parameter set per parametrised function.

The reference itself holds no vectors (tests/__init__.py:1-5) and cannot be built or imported here, so these are NOT
reference outputs: they freeze the oracle, so that an edit which shifts its semantics shows up in review as a changed
fixture (tests/test_golden.py recomputes and compares bit for bit; transcendental outputs to 1e-13).
    python scripts/make_golden.py          # rewrites tests/golden/
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
from oracle import pq_oracle as oracle  # noqa: E402
from pattern_kats import kat_series  # noqa: E402

N, T = 4, 200
ALT = {"timeperiod": 7, "timeperiod1": 3, "timeperiod2": 6, "timeperiod3": 11, "fastperiod": 5, "slowperiod": 13, "signalperiod": 4,
       "fastk_period": 8, "slowk_period": 4, "slowd_period": 5, "fastd_period": 4, "minperiod": 3, "maxperiod": 17,
       "nbdevup": 1.5, "nbdevdn": 2.5, "vfactor": 0.4, "acceleration": 0.02, "maximum": 0.2, "fastlimit": 0.5, "slowlimit": 0.05,
       "startvalue": 0.0, "offsetonreverse": 0.01, "accelerationinitlong": 0.02, "accelerationlong": 0.02, "accelerationmaxlong": 0.2,
       "accelerationinitshort": 0.03, "accelerationshort": 0.03, "accelerationmaxshort": 0.3, "period": 5, "method": 1, "window": 7}
NULL_TOLERANT = {"bbands", "dema", "ema", "kama", "ma", "mama", "mavp", "midpoint", "midprice", "sar", "sarext", "sma", "t3", "tema",
                 "trima", "wma", "apo", "ppo", "macdext", "stoch", "stochf", "atr", "natr", "trange", "ad", "adosc", "obv",
                 "avgprice", "medprice", "typprice", "wclprice", "returns", "rolling_max", "rolling_min"}


def datasets():
    clean = oracle.gen_ohlcv(0x601D0001, N, T, 0)
    rich = oracle.gen_ohlcv(0x601D0002, N, T, 1)
    # symbols 2 and 3 of the pattern-rich set: the hand-built firing sequences of tests/pattern_kats.py (all 60 satisfiable
    # recognisers fire in the fixture; cdl2crows cannot, pattern.rs:30-33)
    for sym, start in ((2, 0), (3, 44)):
        rich["open"][sym], rich["high"][sym], rich["low"][sym], rich["close"][sym] = kat_series(T, start)[:4]
    rng = np.random.default_rng(0x601D)
    holes = {}
    for k, v in clean.items():
        a = v.copy()
        m = rng.random(a.shape) < 0.02
        m[:, :2] = True
        m[1] = False
        m[2, 120:] = True
        a[m] = oracle.NULL
        holes[k] = a
    per = rng.integers(0, 40, size=(N, T)).astype(np.float64)
    out = {}
    for name, d in (("clean", clean), ("rich", rich), ("nulls", holes)):
        d = dict(d)
        d["real"] = d["close"]
        d["periods"] = per
        out[name] = d
    return out


def compute(d, null_bearing):
    res = {}
    spec = {**oracle.SPEC, **oracle.EXTRA}
    for name, (cols, pspec, outs) in sorted(spec.items()):
        if null_bearing and name not in NULL_TOLERANT:
            continue
        for tag, prm in (("", {}), ("@alt", {p: ALT[p] for p, _, _ in pspec if p in ALT})):
            if tag and not prm:
                continue
            vals = oracle.call(name, *[d[c] for c in cols], **prm)
            for (oname, _), v in zip(outs, vals):
                res[f"{name}{tag}.{oname}"] = v
    if not null_bearing:
        for nm in oracle.PATTERN_NAMES:
            res[f"{nm}.pattern"] = oracle.pattern(nm, d["open"], d["high"], d["low"], d["close"])
        buy, sell = oracle.macd_cross_signals(d["close"])
        res["macd_cross.buy"], res["macd_cross.sell"] = buy, sell
        pos, cash, eq, summ = oracle.backtest(d["close"], buy, sell, benchmark=d["open"], buy_slippage=0.01, position_size=0.5)
        res.update({"backtest.position": pos, "backtest.cash": cash, "backtest.equity": eq, "backtest.summary": summ})
        lev = oracle.backtest_leveraged(d["close"], buy, sell, benchmark=d["open"][0], max_trades=8, leverage=2.0, slippage=0.001)
        for k in ("cash", "stock_value", "total_value", "trade_count", "summary"):
            res[f"leveraged.{k}"] = lev[k]
        for k, v in lev["trades"].items():
            res[f"leveraged.trades.{k}"] = v
        res["portfolio.metrics"] = oracle.portfolio_metrics(lev["total_value"], 100000.0 * N, d["open"][0])
        fwd = np.roll(d["close"], -1, axis=1) / d["close"] - 1.0
        fwd[:, -1] = oracle.NULL
        (fac,) = oracle.call("rsi", d["close"], timeperiod=5)
        for m in (0, 1):
            ic, nv = oracle.factor_ic(fac, fwd, method=m)
            res[f"factor.ic{m}"], res[f"factor.nvalid{m}"] = ic, nv
        ric, rir = oracle.rolling_ic(res["factor.ic0"], 10)
        res["factor.rolling_ic"], res["factor.rolling_ir"] = ric, rir
        for rule, sig in (("cross", oracle.cross_signals(d["close"], d["open"])), ("band", oracle.band_signals(fac, 30.0, 70.0)),
                          ("channel0", oracle.channel_signals(d["close"], d["low"], d["high"], 0)),
                          ("channel1", oracle.channel_signals(d["close"], d["low"], d["high"], 1))):
            res[f"signals.{rule}.buy"], res[f"signals.{rule}.sell"] = sig
    return res


def main():
    out_dir = ROOT / "tests" / "golden"
    out_dir.mkdir(parents=True, exist_ok=True)
    for name, d in datasets().items():
        res = compute(d, null_bearing=(name == "nulls"))
        blob = {f"in.{k}": v for k, v in d.items()}
        blob.update({f"out.{k}": v for k, v in res.items()})
        np.savez_compressed(out_dir / f"oracle_{name}_{N}x{T}.npz", **blob)
        print(name, len(res), "outputs ->", out_dir / f"oracle_{name}_{N}x{T}.npz")


if __name__ == "__main__":
    main()
