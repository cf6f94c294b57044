"""ctypes loader for the CPU ORACLE (test infrastructure, not product code).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
Every C entry point has the shape  f(inputs..., n, params..., outputs...)  over ONE series;
`call()` loops over the rows of [N, T] inputs.  Nulls are the NaN bit pattern NULL_BITS.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
NULL_BITS = 0x7FF80000504E554C
NULL = np.array([NULL_BITS], dtype=np.uint64).view(np.float64)[0]
NULL_I32 = np.int32(-2147483648)

I, F = "i", "f"  # param kinds (int64 / double)

# name -> (input columns, [(param, kind, python-wrapper default)], [(output, dtype)])
# defaults follow python/polars_quant/talib/*.py
SPEC = {
    # overlap.py
    "bbands": (["real"], [("timeperiod", I, 20), ("nbdevup", F, 2.0), ("nbdevdn", F, 2.0)],
               [("bb_upper", "f8"), ("bb_middle", "f8"), ("bb_lower", "f8")]),
    "dema": (["real"], [("timeperiod", I, 30)], [("dema", "f8")]),
    "ema": (["real"], [("timeperiod", I, 30)], [("ema", "f8")]),
    "kama": (["real"], [("timeperiod", I, 30)], [("kama", "f8")]),
    "ma": (["real"], [("timeperiod", I, 30), ("matype", I, 0)], [("ma", "f8")]),
    "mama": (["real"], [("fastlimit", F, 0.0), ("slowlimit", F, 0.0)], [("mama", "f8"), ("fama", "f8")]),
    "mavp": (["real", "periods"], [("minperiod", I, 2), ("maxperiod", I, 30), ("matype", I, 0)], [("mavp", "f8")]),
    "midpoint": (["real"], [("timeperiod", I, 14)], [("midpoint", "f8")]),
    "midprice": (["high", "low"], [("timeperiod", I, 14)], [("midprice", "f8")]),
    "sar": (["high", "low"], [("acceleration", F, 0.0), ("maximum", F, 0.0)], [("sar", "f8")]),
    "sarext": (["high", "low"], [("startvalue", F, 0.0), ("offsetonreverse", F, 0.0),
                                ("accelerationinitlong", F, 0.0), ("accelerationlong", F, 0.0),
                                ("accelerationmaxlong", F, 0.0), ("accelerationinitshort", F, 0.0),
                                ("accelerationshort", F, 0.0), ("accelerationmaxshort", F, 0.0)], [("sarext", "f8")]),
    "sma": (["real"], [("timeperiod", I, 30)], [("sma", "f8")]),
    "t3": (["real"], [("timeperiod", I, 5), ("vfactor", F, 0.7)], [("t3", "f8")]),
    "tema": (["real"], [("timeperiod", I, 30)], [("tema", "f8")]),
    "trima": (["real"], [("timeperiod", I, 30)], [("trima", "f8")]),
    "wma": (["real"], [("timeperiod", I, 30)], [("wma", "f8")]),
    # momentum.py
    "adx": (["high", "low", "close"], [("timeperiod", I, 14)], [("adx", "f8")]),
    "adxr": (["high", "low", "close"], [("timeperiod", I, 14)], [("adxr", "f8")]),
    "apo": (["real"], [("fastperiod", I, 12), ("slowperiod", I, 26), ("matype", I, 0)], [("apo", "f8")]),
    "aroon": (["high", "low"], [("timeperiod", I, 14)], [("aroon_up", "f8"), ("aroon_down", "f8")]),
    "aroonosc": (["high", "low"], [("timeperiod", I, 14)], [("aroonosc", "f8")]),
    "bop": (["open", "high", "low", "close"], [], [("bop", "f8")]),
    "cci": (["high", "low", "close"], [("timeperiod", I, 14)], [("cci", "f8")]),
    "cmo": (["real"], [("timeperiod", I, 14)], [("cmo", "f8")]),
    "dx": (["high", "low", "close"], [("timeperiod", I, 14)], [("dx", "f8")]),
    "macd": (["real"], [("fastperiod", I, 12), ("slowperiod", I, 26), ("signalperiod", I, 9)],
             [("macd", "f8"), ("macd_signal", "f8"), ("macd_hist", "f8")]),
    "macdext": (["real"], [("fastperiod", I, 12), ("fastmatype", I, 0), ("slowperiod", I, 26),
                           ("slowmatype", I, 0), ("signalperiod", I, 9), ("signalmatype", I, 0)],
                [("macd_dif", "f8"), ("macd_dea", "f8"), ("macd_hist", "f8")]),
    "macdfix": (["real"], [("signalperiod", I, 9)], [("macd", "f8"), ("macd_signal", "f8"), ("macd_hist", "f8")]),
    "mfi": (["high", "low", "close", "volume"], [("timeperiod", I, 14)], [("mfi", "f8")]),
    "minus_di": (["high", "low", "close"], [("timeperiod", I, 14)], [("minus_di", "f8")]),
    "minus_dm": (["high", "low"], [("timeperiod", I, 14)], [("minus_dm", "f8")]),
    "mom": (["real"], [("timeperiod", I, 10)], [("mom", "f8")]),
    "plus_di": (["high", "low", "close"], [("timeperiod", I, 14)], [("plus_di", "f8")]),
    "plus_dm": (["high", "low"], [("timeperiod", I, 14)], [("plus_dm", "f8")]),
    "ppo": (["real"], [("fastperiod", I, 12), ("slowperiod", I, 26), ("matype", I, 0)], [("ppo", "f8")]),
    "roc": (["real"], [("timeperiod", I, 10)], [("roc", "f8")]),
    "rocp": (["real"], [("timeperiod", I, 10)], [("rocp", "f8")]),
    "rocr": (["real"], [("timeperiod", I, 10)], [("rocr", "f8")]),
    "rocr100": (["real"], [("timeperiod", I, 10)], [("rocr100", "f8")]),
    "rsi": (["real"], [("timeperiod", I, 14)], [("rsi", "f8")]),
    "stoch": (["high", "low", "close"], [("fastk_period", I, 5), ("slowk_period", I, 3), ("slowk_matype", I, 0),
                                         ("slowd_period", I, 3), ("slowd_matype", I, 0)],
              [("slowk", "f8"), ("slowd", "f8")]),
    "stochf": (["high", "low", "close"], [("fastk_period", I, 5), ("fastd_period", I, 3), ("fastd_matype", I, 0)],
               [("fastk", "f8"), ("fastd", "f8")]),
    "stochrsi": (["real"], [("timeperiod", I, 14), ("fastk_period", I, 5), ("fastd_period", I, 3),
                            ("fastd_matype", I, 0)], [("fastk_rsi", "f8"), ("fastd_rsi", "f8")]),
    "trix": (["real"], [("timeperiod", I, 30)], [("trix", "f8")]),
    "ultosc": (["high", "low", "close"], [("timeperiod1", I, 7), ("timeperiod2", I, 14), ("timeperiod3", I, 28)],
               [("ultosc", "f8")]),
    "willr": (["high", "low", "close"], [("timeperiod", I, 14)], [("willr", "f8")]),
    # volatility.py / volume.py / price.py
    "atr": (["high", "low", "close"], [("timeperiod", I, 14)], [("atr", "f8")]),
    "natr": (["high", "low", "close"], [("timeperiod", I, 14)], [("natr", "f8")]),
    "trange": (["high", "low", "close"], [], [("trange", "f8")]),
    "ad": (["high", "low", "close", "volume"], [], [("ad", "f8")]),
    "adosc": (["high", "low", "close", "volume"], [("fastperiod", I, 3), ("slowperiod", I, 10)], [("adosc", "f8")]),
    "obv": (["real", "volume"], [], [("obv", "f8")]),
    "avgprice": (["open", "high", "low", "close"], [], [("avgprice", "f8")]),
    "medprice": (["high", "low"], [], [("medprice", "f8")]),
    "typprice": (["high", "low", "close"], [], [("typprice", "f8")]),
    "wclprice": (["high", "low", "close"], [], [("wclprice", "f8")]),
    # cycle.py
    "ht_dcperiod": (["real"], [], [("ht_dcperiod", "f8")]),
    "ht_dcphase": (["real"], [], [("ht_dcphase", "f8")]),
    "ht_phasor": (["real"], [], [("inphase", "f8"), ("quadrature", "f8")]),
    "ht_sine": (["real"], [], [("sine", "f8"), ("leadsine", "f8")]),
    "ht_trendline": (["real"], [], [("ht_trendline", "f8")]),
    "ht_trendmode": (["real"], [], [("ht_trendmode", "i4")]),
}

# functions outside talib.* that share the (inputs, params, outputs) calling shape; not part of the indicator suite
EXTRA = {
    "returns": (["real"], [("period", I, 1), ("method", I, 0)], [("return", "f8")]),  # README.md:46-75 (D-13); 0 simple, 1 log
    "rolling_max": (["real"], [("window", I, 20)], [("rolling_max", "f8")]),
    "rolling_min": (["real"], [("window", I, 20)], [("rolling_min", "f8")]),
}

PATTERN_NAMES = [
    "cdl2crows", "cdl3blackcrows", "cdl3inside", "cdl3linestrike", "cdl3outside", "cdl3starsinsouth",
    "cdl3whitesoldiers", "cdlabandonedbaby", "cdladvanceblock", "cdlbelthold", "cdlbreakaway",
    "cdlclosingmarubozu", "cdlconcealbabyswall", "cdlcounterattack", "cdldarkcloudcover", "cdldoji",
    "cdldojistar", "cdldragonflydoji", "cdlengulfing", "cdleveningdojistar", "cdleveningstar",
    "cdlgapsidesidewhite", "cdlgravestonedoji", "cdlhammer", "cdlhangingman", "cdlharami", "cdlharamicross",
    "cdlhighwave", "cdlhikkake", "cdlhikkakemod", "cdlhomingpigeon", "cdlidentical3crows", "cdlinneck",
    "cdlinvertedhammer", "cdlkicking", "cdlkickingbylength", "cdlladderbottom", "cdllongleggeddoji",
    "cdllongline", "cdlmarubozu", "cdlmatchinglow", "cdlmathold", "cdlmorningdojistar", "cdlmorningstar",
    "cdlonneck", "cdlpiercing", "cdlrickshawman", "cdlrisefall3methods", "cdlseparatinglines",
    "cdlshootingstar", "cdlshortline", "cdlspinningtop", "cdlstalledpattern", "cdlsticksandwich", "cdltakuri",
    "cdltasukigap", "cdlthrusting", "cdltristar", "cdlunique3river", "cdlupsidegap2crows",
    "cdlxsidegap3methods"]
# python-wrapper penetration defaults (pattern.py): 0.5 for darkcloudcover / mathold / piercing, else 0.3
PATTERN_PEN_DEFAULT = {n: (0.5 if n in ("cdldarkcloudcover", "cdlmathold", "cdlpiercing") else 0.3)
                       for n in PATTERN_NAMES}

_lib = None


def build(force: bool = False) -> Path:
    so = _HERE / "libpq_oracle.so"
    srcs = list(_HERE.glob("*.c")) + list(_HERE.glob("*.h"))
    if force or not so.exists() or any(s.stat().st_mtime > so.stat().st_mtime for s in srcs):
        subprocess.run(["make", "-C", str(_HERE), "-s"], check=True)
    return so


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        _lib = C.CDLL(str(build()))
        _lib.pqo_suite_bench.restype = C.c_double
    return _lib


def _p(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def _as2d(x, dtype=np.float64):
    a = np.ascontiguousarray(np.asarray(x, dtype=dtype))
    return (a.reshape(1, -1), True) if a.ndim == 1 else (a, False)


def call(name: str, *inputs, **params):
    """Run oracle function `name` on [T] or [N, T] float64 inputs; returns tuple of outputs."""
    if name in PATTERN_NAMES:
        return (pattern(name, *inputs, **params),)
    cols, pspec, outs = SPEC[name] if name in SPEC else EXTRA[name]
    assert len(inputs) == len(cols), f"{name} wants {cols}"
    arrs = [_as2d(x) for x in inputs]
    squeeze = arrs[0][1]
    arrs = [a for a, _ in arrs]
    N, T = arrs[0].shape
    pvals = []
    for pname, kind, default in pspec:
        v = params.pop(pname, default)
        pvals.append(C.c_int64(int(v)) if kind == I else C.c_double(float(v)))
    assert not params, f"unknown params {params}"
    res = [np.empty((N, T), dtype=np.float64 if dt == "f8" else np.int32) for _, dt in outs]
    fn = getattr(lib(), "pqo_" + name)
    fn.restype = None
    for s in range(N):
        args = [_p(a[s]) for a in arrs] + [C.c_int64(T)] + pvals + [_p(r[s]) for r in res]
        fn(*args)
    return tuple(r[0] if squeeze else r for r in res)


def pattern(name: str, o, h, l, c, penetration=None):
    pid = PATTERN_NAMES.index(name)
    pen = PATTERN_PEN_DEFAULT[name] if penetration is None else penetration
    (o2, sq), (h2, _), (l2, _), (c2, _) = _as2d(o), _as2d(h), _as2d(l), _as2d(c)
    N, T = o2.shape
    out = np.empty((N, T), dtype=np.int32)
    fn = lib().pqo_pattern
    fn.restype = None
    for s in range(N):
        fn(C.c_int(pid), _p(o2[s]), _p(h2[s]), _p(l2[s]), _p(c2[s]), C.c_int64(T), C.c_double(pen), _p(out[s]))
    return out[0] if sq else out


class BtParams(C.Structure):
    _fields_ = [(k, C.c_double) for k in ("initial_capital", "buy_slippage", "sell_slippage",
                                          "buy_commission_rate", "sell_commission_rate", "min_commission",
                                          "position_size")]


BT_DEFAULTS = dict(initial_capital=100000.0, buy_slippage=0.0, sell_slippage=0.0, buy_commission_rate=0.0003,
                   sell_commission_rate=0.0003, min_commission=5.0, position_size=1.0)
SUMMARY_KEYS = ["annualized_return", "max_drawdown", "alpha", "beta", "sharpe_ratio", "max_profit", "win_rate",
                "total_trades"]


def backtest(price, buy, sell, benchmark=None, **kw):
    """-> position, cash, equity [N,T], summary [N,8] (key order SUMMARY_KEYS)."""
    prm = BtParams(**{**BT_DEFAULTS, **kw})
    (p2, sq) = _as2d(price)
    b2, _ = _as2d(buy, np.uint8)
    s2, _ = _as2d(sell, np.uint8)
    bm = _as2d(benchmark)[0] if benchmark is not None else None
    N, T = p2.shape
    pos, cash, eq = (np.empty((N, T)) for _ in range(3))
    summ = np.zeros((N, 8))
    fn = lib().pqo_backtest
    fn.restype = None
    for s in range(N):
        fn(_p(p2[s]), _p(b2[s]), _p(s2[s]), _p(bm[s]) if bm is not None else None, C.c_int64(T), C.byref(prm),
           _p(pos[s]), _p(cash[s]), _p(eq[s]), _p(summ[s]))
    if sq:
        return pos[0], cash[0], eq[0], summ[0]
    return pos, cash, eq, summ


class LevParams(C.Structure):
    _fields_ = [(k, C.c_double) for k in ("initial_capital", "position_size", "leverage", "margin_call_threshold",
                                          "interest_rate", "commission_rate", "min_commission", "slippage")]


LEV_DEFAULTS = dict(initial_capital=100000.0, position_size=1.0, leverage=1.0, margin_call_threshold=0.3,
                    interest_rate=0.06, commission_rate=0.0003, min_commission=5.0, slippage=0.0)  # README.md:350-366
TRADE_FIELDS = ("entry_day", "exit_day", "entry_price", "exit_price", "quantity", "pnl", "pnl_pct", "reason")


def backtest_leveraged(price, buy, sell, benchmark=None, max_trades=64, **kw):
    """D-10 (oracle/backtest.c).  price/buy/sell: [N, T]; benchmark: [T] or None.
    -> dict(cash, stock_value, total_value [N,T]; trade_count [N]; trades {field: [N,max_trades]}; summary [N,8])"""
    prm = LevParams(**{**LEV_DEFAULTS, **kw})
    p2, _ = _as2d(price)
    b2, _ = _as2d(buy, np.uint8)
    s2, _ = _as2d(sell, np.uint8)
    bm = np.ascontiguousarray(benchmark, dtype=np.float64) if benchmark is not None else None
    N, T = p2.shape
    cash, sv, tv = (np.empty((N, T)) for _ in range(3))
    cnt = np.zeros(N, np.int32)
    tr = {k: (np.zeros((N, max_trades), np.int32) if k in ("entry_day", "exit_day", "reason") else np.zeros((N, max_trades)))
          for k in TRADE_FIELDS}
    summ = np.zeros((N, 8))
    fn = lib().pqo_backtest_leveraged
    fn.restype = None
    for s in range(N):
        fn(_p(p2[s]), _p(b2[s]), _p(s2[s]), _p(bm) if bm is not None else None, C.c_int64(T), C.byref(prm),
           _p(cash[s]), _p(sv[s]), _p(tv[s]), C.c_int32(max_trades), _p(cnt[s:s + 1]),
           *[_p(tr[k][s]) for k in TRADE_FIELDS], _p(summ[s]))
    return dict(cash=cash, stock_value=sv, total_value=tv, trade_count=cnt, trades=tr, summary=summ)


def portfolio_metrics(total_value, initial_total, benchmark=None):
    tv = np.ascontiguousarray(total_value, dtype=np.float64)
    N, T = tv.shape
    bm = np.ascontiguousarray(benchmark, dtype=np.float64) if benchmark is not None else None
    out = np.zeros((T, 10))
    fn = lib().pqo_portfolio_metrics
    fn.restype = None
    fn(_p(tv), C.c_int64(N), C.c_int64(T), C.c_int64(T), C.c_double(initial_total), _p(bm) if bm is not None else None, _p(out))
    return out


def _signals(fn_name, cols, *scalars):
    arrs = [_as2d(c)[0] for c in cols]
    sq = np.asarray(cols[0]).ndim == 1
    N, T = arrs[0].shape
    buy, sell = np.zeros((N, T), np.uint8), np.zeros((N, T), np.uint8)
    fn = getattr(lib(), fn_name)
    fn.restype = None
    for s in range(N):
        fn(*[_p(a[s]) for a in arrs], C.c_int64(T), *scalars, _p(buy[s]), _p(sell[s]))
    return (buy[0], sell[0]) if sq else (buy, sell)


def cross_signals(a, b):
    return _signals("pqo_cross_signals", [a, b])


def band_signals(x, lower, upper):
    return _signals("pqo_band_signals", [x], C.c_double(lower), C.c_double(upper))


def channel_signals(p, lo, hi, mode):
    return _signals("pqo_channel_signals", [p, lo, hi], C.c_int(mode))


def factor_ic(factor, fwd_return, method=0):
    """D-12: factor, fwd_return [N, T] -> (ic [T], n_valid [T]); method 0 Pearson IC, 1 Spearman Rank-IC"""
    f = np.ascontiguousarray(factor, dtype=np.float64)
    r = np.ascontiguousarray(fwd_return, dtype=np.float64)
    N, T = f.shape
    ic = np.empty(T)
    nv = np.zeros(T, np.int32)
    fn = lib().pqo_factor_ic
    fn.restype = None
    fn(_p(f), _p(r), C.c_int64(N), C.c_int64(T), C.c_int64(T), C.c_int(method), _p(ic), _p(nv))
    return ic, nv


def rolling_ic(ic, window):
    a = np.ascontiguousarray(ic, dtype=np.float64)
    ric, rir = np.empty_like(a), np.empty_like(a)
    fn = lib().pqo_rolling_ic
    fn.restype = None
    fn(_p(a), C.c_int64(len(a)), C.c_int64(window), _p(ric), _p(rir))
    return ric, rir


def summary(equity, benchmark, initial_capital, trades, wins):
    eq = np.ascontiguousarray(equity, dtype=np.float64)
    bm = np.ascontiguousarray(benchmark, dtype=np.float64) if benchmark is not None else None
    out = np.zeros(8)
    fn = lib().pqo_summary
    fn.restype = None
    fn(_p(eq), _p(bm) if bm is not None else None, C.c_int64(len(eq)), C.c_int64(len(bm) if bm is not None else 0),
       C.c_double(initial_capital), C.c_int64(trades), C.c_int64(wins), _p(out))
    return out


def macd_cross_signals(close, fast=12, slow=26, sig=9):
    c2, sq = _as2d(close)
    N, T = c2.shape
    buy = np.zeros((N, T), np.uint8)
    sell = np.zeros((N, T), np.uint8)
    fn = lib().pqo_macd_cross_signals
    fn.restype = None
    for s in range(N):
        fn(_p(c2[s]), C.c_int64(T), C.c_int64(fast), C.c_int64(slow), C.c_int64(sig), _p(buy[s]), _p(sell[s]))
    return (buy[0], sell[0]) if sq else (buy, sell)


def gen_ohlcv(seed: int, n_sym: int, T: int, mode: int = 0):
    """SURVEY 8(d) deterministic OHLCV; returns dict of [N,T] float64 arrays."""
    arrs = {k: np.empty((n_sym, T)) for k in ("open", "high", "low", "close", "volume")}
    fn = lib().pqo_gen_ohlcv
    fn.restype = None
    fn(C.c_uint64(seed), C.c_int64(n_sym), C.c_int64(T), C.c_int(mode), *[_p(arrs[k]) for k in arrs])
    return arrs


_native = None


def native_lib():
    """the same sources compiled -O3 -march=native on THIS host (bench.py's cpu_baseline leg; the portable build is what the
    parity tests use) -> CDLL, or None when it cannot be built here"""
    global _native
    if _native is None:
        so = _HERE / "libpq_oracle_native.so"
        try:
            subprocess.run(["make", "-C", str(_HERE), "-s", "-B", "native"], check=True, capture_output=True)   # -B: never reuse another host's build
            _native = C.CDLL(str(so))
            _native.pqo_suite_bench.restype = C.c_double
        except Exception:  # noqa: BLE001
            _native = False
    return _native or None


def suite_bench(ohlcv: dict, threads: int = 0, native: bool = False) -> float:
    N, T = ohlcv["close"].shape
    L = (native_lib() if native else None) or lib()
    return L.pqo_suite_bench(*[_p(np.ascontiguousarray(ohlcv[k])) for k in ("open", "high", "low", "close", "volume")],
                                 C.c_int64(N), C.c_int64(T), C.c_int(threads))


def input_cols(name):
    if name in PATTERN_NAMES:
        return ["open", "high", "low", "close"]
    return (SPEC[name] if name in SPEC else EXTRA[name])[0]
