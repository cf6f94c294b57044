"""Randomised parity sweep (GPU box): random shapes, parameters and null patterns for every function, GPU vs oracle.
Deterministic per seed.  `python scripts/fuzz_parity.py <seed> <iterations>`; tests/test_gpu_parity.py runs sweep() in-process."""
import sys

sys.path.insert(0, ".")
import numpy as np
import torch

from oracle import pq_oracle as oracle
from polars_quant_amd import api
from polars_quant_amd._spec import I, SPEC

TRANSC = {"ht_dcperiod", "ht_dcphase", "ht_phasor", "ht_sine", "mama"}
NAN_RATE = float(__import__("os").environ.get("PQ_FUZZ_NAN", "0"))   # probability of a NaN per cell in half of the cases (0: none)


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint64 if a.dtype == np.float64 else np.uint32)


def _case(rng, name, log) -> int:
    cols, pspec, outs, fam = SPEC[name]
    N, T = int(rng.integers(1, 140)), int(rng.integers(1, 260))
    d = oracle.gen_ohlcv(int(rng.integers(1, 1 << 30)), N, T, int(rng.integers(0, 2)))
    d["real"] = d["close"]
    d["periods"] = rng.integers(0, 40, size=(N, T)).astype(np.float64)
    if fam in ("N-A", "N-C", "N-0") and name != "stochrsi" and rng.random() < 0.4:
        for k in ("open", "high", "low", "close", "volume", "real"):
            m = rng.random((N, T)) < 0.03
            d[k] = d[k].copy()
            d[k][m] = oracle.NULL
    if NAN_RATE > 0 and rng.random() < 0.5:   # true NaNs (values, not NULLs) are inside every function's domain
        for k in ("open", "high", "low", "close", "volume", "real"):
            m = rng.random((N, T)) < NAN_RATE
            d[k] = d[k].copy()
            d[k][m] = np.nan
    params = {}
    for pname, kind, _default in pspec:
        if kind == I:
            if "matype" in pname:
                params[pname] = int(rng.integers(0, 9))
            else:
                params[pname] = max(int(rng.choice([0, 1, 2, 3, 5, 9, 14, 30, T - 1, T, T + 1, int(rng.integers(1, 60))])), 0)
        elif name == "mama":   # alpha > 1 diverges: chaotic, any tolerance fails
            params[pname] = float(rng.choice([0.0, 0.02, 0.05, 0.2, 0.5, 0.9]))
        else:
            params[pname] = float(rng.choice([0.0, 0.02, 0.2, 0.5, 0.7, 2.0, -50.0, 5.0]))
    if name == "mavp":
        lo = int(rng.integers(0, 20))
        params["minperiod"], params["maxperiod"] = lo, lo + int(rng.integers(0, 40))
    try:
        exp = oracle.call(name, *[d[c] for c in cols], **params)
    except Exception as e:  # noqa: BLE001
        log("oracle error", name, params, e)
        return 1
    # the device layout: dense, or a row pitch larger than the row (even pitch: tiled bodies incl. the pair-mode storer and the ragged
    # tail; odd pitch: gather bodies; a multiple of 16 elements: the 128-byte pitch bench.py uses); the padding holds a poison value
    pitch = T + int(rng.choice([0, 0, 0, 1, 2, 6, 8, 15, 16])) if rng.random() < 0.5 else (T + 15) // 16 * 16
    def dev(a):
        if pitch == T:
            return torch.from_numpy(np.ascontiguousarray(a)).cuda()
        buf = torch.full((N, pitch), 1e300, dtype=torch.float64, device="cuda")
        buf[:, :T] = torch.from_numpy(np.ascontiguousarray(a)).cuda()
        return buf[:, :T]
    got = api.call(name, *[dev(d[c]) for c in cols], **params)
    bad = 0
    for (oname, dt), g, e in zip(outs, got, exp):
        g = g.cpu().numpy()
        if name in TRANSC:
            ok = np.isclose(g, e, rtol=1e-12, atol=1e-12, equal_nan=True) | (_bits(g) == _bits(e))
            if dt == "f8":
                ok = ok & ((_bits(g) == np.uint64(oracle.NULL_BITS)) == (_bits(e) == np.uint64(oracle.NULL_BITS)))
        else:
            ok = _bits(g) == _bits(e)
            if dt == "f8":   # a computed NaN matches a computed NaN; a NULL (itself a NaN pattern) only matches a NULL
                gn, en = _bits(g) == np.uint64(oracle.NULL_BITS), _bits(e) == np.uint64(oracle.NULL_BITS)
                ok = (ok | ((g != g) & (e != e))) & (gn == en)
        if not np.all(ok):
            bad += 1
            idx = np.argwhere(~ok)[:3].tolist()
            log(f"MISMATCH {name}.{oname} N={N} T={T} {params}: {int((~ok).sum())} cells, first {idx} got {g[~ok][:3]} exp {e[~ok][:3]}")
    return bad


def sweep(seed: int, iters: int, log=print) -> int:
    """-> number of mismatching output columns over `iters` random cases (functions in round-robin order)"""
    rng = np.random.default_rng(seed)
    names = sorted(SPEC)
    return sum(_case(rng, names[it % len(names)], log) for it in range(iters))


# ---- long series: the one-symbol-per-wavefront forms (csrc/wt_dev.h: direct calls with 1024 <= len <= 4096; csrc/ops_backtest_wave.h:
# both backtests up to 8192 rows).  Their correctness rests on speculation + a bit test + re-runs, i.e. on paths that only rare data
# reach: flat stretches (no contraction), nulls, short warm-ups, chunk and block boundaries of len.
WT_NAMES = ("ema", "dema", "tema", "trix", "macd", "rsi", "plus_dm", "minus_dm", "plus_di", "minus_di", "dx", "adx", "adxr", "atr", "natr", "midpoint", "midprice")


def _long_prices(rng, N, T, nulls=True):   # nulls: the function's reference accepts NULL rows (families N-A / N-C / N-0; N-B raises)
    d = oracle.gen_ohlcv(int(rng.integers(1, 1 << 30)), N, T, 0)
    d = {k: v.copy() for k, v in d.items()}
    kind = d["_kind"] = np.zeros(N, np.int64)   # what was done to each symbol (for the mismatch report)
    for s in range(N):
        r = rng.random()
        kind[s] = 1 if r < 0.12 else 2 if r < 0.24 else 3 if r < 0.30 else 4 if r < 0.34 else 0
        if r < 0.12:      # flat from some row on: speculative chunks cannot merge
            t0 = int(rng.integers(0, T))
            for k in ("open", "high", "low", "close"):
                d[k][s, t0:] = d[k][s, t0]
        elif r < 0.24:    # a few nulls (the symbol goes to the lane-per-symbol kernel / the general row path)
            if nulls:
                for k in ("open", "high", "low", "close", "volume"):
                    d[k][s, rng.integers(0, T, size=int(rng.integers(1, 4)))] = oracle.NULL
            else:
                kind[s] = 0
        elif r < 0.30:
            for k in ("open", "high", "low", "close", "volume"):
                if rng.random() < 0.5:
                    d[k][s, int(rng.integers(0, T))] = np.nan
        elif r < 0.34:
            d["close"][s, int(rng.integers(0, T))] = -1.0
    d["real"] = d["close"]
    return d


def _long_indicator(rng, name, log) -> int:
    import os
    cols, pspec, outs, fam = SPEC[name]
    N, T = int(rng.integers(1, 150)), int(rng.choice([1024, 1025, 2520, 4095, 4096, int(rng.integers(1024, 4097))]))
    d = _long_prices(rng, N, T, nulls=fam in ("N-A", "N-C", "N-0"))
    params = {}
    for pname, kind, _default in pspec:
        if kind == I:
            params[pname] = int(rng.integers(0, 9)) if "matype" in pname else int(rng.choice([1, 2, 3, 5, 9, 14, 26, 30, 60, 200, int(rng.integers(1, 300)), 0, 1000, 1024, 1025, T - 1, T, T + 1][: 11 if rng.random() < 0.8 else 18]))
        elif name == "mama":
            params[pname] = float(rng.choice([0.02, 0.05, 0.2, 0.5]))
        else:
            params[pname] = float(rng.choice([0.02, 0.2]))
    if name == "mavp":
        lo_ = int(rng.integers(2, 12)); params["minperiod"], params["maxperiod"] = lo_, lo_ + int(rng.integers(0, 40))
        d["periods"] = rng.integers(0, 60, size=(N, T)).astype(np.float64)
    warm = str(rng.choice([10.0, 10.0, 4.0, 1.0]))
    os.environ["PQ_WT_WARM"] = warm   # short warm-ups: chunks that fail the bit test and are re-run
    os.environ["PQ_WT_ALL"] = "1"
    try:
        exp = oracle.call(name, *[d[c] for c in cols], **params)
        pitch = (T + 15) // 16 * 16 if rng.random() < 0.7 else T
        def dev(a):
            buf = torch.full((N, pitch), 1e300, dtype=torch.float64, device="cuda")
            buf[:, :T] = torch.from_numpy(np.ascontiguousarray(a)).cuda()
            return buf[:, :T]
        got = api.call(name, *[dev(d[c]) for c in cols], **params)
    finally:
        os.environ.pop("PQ_WT_WARM", None)
        os.environ.pop("PQ_WT_ALL", None)
    bad = 0
    for (oname, dt), g, e in zip(outs, got, exp):
        g = g.cpu().numpy()
        if dt != "f8":
            ok = g == e
        else:
            ok = (_bits(g) == _bits(e)) | ((g != g) & (e != e) & ((_bits(g) == np.uint64(oracle.NULL_BITS)) == (_bits(e) == np.uint64(oracle.NULL_BITS))))
            if name in TRANSC:
                ok = ok | (np.isclose(g, e, rtol=1e-12, atol=1e-12, equal_nan=True) & ((_bits(g) == np.uint64(oracle.NULL_BITS)) == (_bits(e) == np.uint64(oracle.NULL_BITS))))
        if not np.all(ok):
            bad += 1
            rows = sorted(set(np.argwhere(~ok)[:, 0].tolist()))
            log(f"MISMATCH long {name}.{oname} N={N} T={T} pitch={pitch} warm={warm} {params}: {int((~ok).sum())} cells, first {np.argwhere(~ok)[:3].tolist()} "
                f"symbols {rows[:8]} kinds {[int(d['_kind'][r]) for r in rows[:8]]} got {g[~ok][:2]} exp {e[~ok][:2]}")
    return bad


def _long_backtest(rng, log) -> int:
    import os
    N = int(rng.integers(1, 40))
    T = int(rng.choice([1, 2, 63, 64, 65, 128, 2520, 4096, 4097, 5040, 8191, 8192, int(rng.integers(1, 8193))]))
    d = _long_prices(rng, N, T)
    price = d["close"]
    kw = dict(initial_capital=float(rng.choice([100000.0, 30.0, 1e7, 1.0, 1e12, 0.0])), position_size=float(rng.choice([1.0, 0.5, 1.5, 0.0])),
              buy_slippage=float(rng.choice([0.0, 0.01, -0.01])), sell_slippage=float(rng.choice([0.0, 0.02, 50.0])), min_commission=float(rng.choice([5.0, 0.0, 1.0])),
              buy_commission_rate=float(rng.choice([0.0003, 0.0, 0.01])), sell_commission_rate=float(rng.choice([0.0003, 0.0, 0.5])))
    # (a capital of 1e12 makes share counts exceed 2^28: the fast chain hands the symbol to the block form; 1.0 / 0.0: no buy can afford
    # a share; a sell slippage of 50 makes the execution price negative)
    bad = 0
    waves = str(rng.choice(["", "1", "4"]))   # "": the library's own choice; else the one-wave / four-wave form of the kernels, forced
    if waves:
        os.environ["PQ_BT_WAVES"] = waves
    else:
        os.environ.pop("PQ_BT_WAVES", None)
    def cmp(tag, got, exp):
        nonlocal bad
        tag = f"{tag} waves={waves!r}"
        for nm, g, e in zip(("position", "cash", "equity"), got[:3], exp[:3]):
            g = g.cpu().numpy()
            ok = (_bits(g) == _bits(e)) | ((g != g) & (e != e))
            if not ok.all():
                bad += 1
                log(f"MISMATCH backtest {tag}.{nm} N={N} T={T} {kw}: {int((~ok).sum())} cells, first {np.argwhere(~ok)[:3].tolist()}")
        s, es = got[3].cpu().numpy(), exp[3]
        okr = ~np.isnan(es).any(axis=1)
        exact = all((_bits(s[okr, k]) == _bits(es[okr, k])).all() for k in (1, 5, 6, 7))
        # the ordered sums (mean, variance, covariance -> sharpe, alpha, beta) are summed in another order than the reference's: <= 1e-12
        # of the result on sane parameters; a 50 % commission or a capital of 1e12 makes the daily returns cancel by many orders of
        # magnitude, and the bound is then relative to the terms, not to the result
        sane = kw["sell_commission_rate"] <= 0.01 and kw["initial_capital"] <= 1e7 and kw["sell_slippage"] < 1.0
        if not (exact and np.allclose(s[okr], es[okr], rtol=1e-12 if sane else 1e-9, atol=1e-13)):
            bad += 1
            rel = np.max(np.abs(s[okr] - es[okr]) / np.maximum(np.abs(es[okr]), 1e-300), axis=0) if okr.any() else None
            log(f"MISMATCH backtest {tag}.summary N={N} T={T} {kw}: exact columns {'equal' if exact else 'DIFFER'}, max relative difference per column {rel}")
    if rng.random() < 0.5:
        fast, slow, sig = (12, 26, 9) if rng.random() < 0.5 else (int(rng.integers(1, 40)), int(rng.integers(1, 80)), int(rng.integers(1, 30)))
        if rng.random() < 0.1:
            fast, slow, sig = [int(x) for x in rng.choice([0, 1, T, T + 1, 300, 1500], size=3)]   # dead or very long averages
        w = rng.choice(["", "", "1", "2"])
        if w:
            os.environ["PQ_BT_WARM_CHUNKS2"] = w
            os.environ["PQ_BT_WARM_CHUNKS"] = w
        try:
            eb, es_ = oracle.macd_cross_signals(price, fast, slow, sig)
            cmp(f"macd({fast},{slow},{sig}) warm={w!r}", api.backtest_macd_cross(torch.from_numpy(price).cuda(), fast, slow, sig, **kw), oracle.backtest(price, eb, es_, **kw))
        finally:
            os.environ.pop("PQ_BT_WARM_CHUNKS2", None)
            os.environ.pop("PQ_BT_WARM_CHUNKS", None)
    else:
        dens = float(rng.choice([0.01, 0.05, 0.3, 1.0]))
        buy = (rng.random(price.shape) < dens).astype(np.uint8)
        sell = (rng.random(price.shape) < dens).astype(np.uint8)
        bm = d["open"] if rng.random() < 0.5 else None
        cmp(f"vectorized dens={dens} bench={bm is not None}",
            api.backtest_vectorized(torch.from_numpy(price).cuda(), torch.from_numpy(buy).cuda(), torch.from_numpy(sell).cuda(),
                                    benchmark=None if bm is None else torch.from_numpy(bm).cuda(), **kw),
            oracle.backtest(price, buy, sell, benchmark=bm, **kw))
    os.environ.pop("PQ_BT_WAVES", None)
    return bad


# ---- candlestick recognisers: comparisons of candle parts.  Generic float prices almost never tie; quantised prices make equal
# bodies / shadows / opens the common case (`<=` against `<` then matters), NaN cells make `!(a > b)` differ from `a <= b`.
def _pattern_case(rng, log) -> int:
    from polars_quant_amd._spec import PATTERN_NAMES
    N, T = int(rng.integers(1, 80)), int(rng.integers(1, 400))
    d = oracle.gen_ohlcv(int(rng.integers(1, 1 << 30)), N, T, 0)
    o, h, l, c = (d[k].copy() for k in ("open", "high", "low", "close"))
    mode = int(rng.integers(0, 4))
    if mode >= 1:      # a price grid: ties everywhere (the coarser, the more dojis / marubozus / equal highs)
        q = float(rng.choice([0.01, 0.05, 0.25, 1.0]))
        o, h, l, c = (np.round(x / q) * q for x in (o, h, l, c))
        h = np.maximum(h, np.maximum(o, c)); l = np.minimum(l, np.minimum(o, c))
    if mode == 2:      # NaN cells
        for x in (o, h, l, c):
            x[rng.random(x.shape) < 0.01] = np.nan
    if mode == 3:      # flat stretches (zero ranges: every average of bodies / ranges is 0)
        for s_ in range(N):
            if rng.random() < 0.3 and T > 8:
                t0 = int(rng.integers(0, T - 4)); t1 = min(T, t0 + int(rng.integers(2, 40)))
                o[s_, t0:t1] = h[s_, t0:t1] = l[s_, t0:t1] = c[s_, t0:t1] = c[s_, t0]
    pens = {nm: float(rng.choice([0.0, 0.1, 0.3, 0.5, 0.9])) for nm in PATTERN_NAMES}
    dv = [torch.from_numpy(np.ascontiguousarray(x)).cuda() for x in (o, h, l, c)]
    got = api.cdl_all(*dv, penetrations=pens)
    bad = 0
    single = PATTERN_NAMES[int(rng.integers(0, len(PATTERN_NAMES)))]
    for nm in PATTERN_NAMES:
        exp = oracle.pattern(nm, o, h, l, c, penetration=pens[nm])
        g = got[nm].cpu().numpy()
        if not (g == exp).all():
            bad += 1
            log(f"MISMATCH pattern {nm} N={N} T={T} mode={mode} pen={pens[nm]}: {int((g != exp).sum())} cells, first {np.argwhere(g != exp)[:3].tolist()} "
                f"got {g[g != exp][:3]} exp {exp[g != exp][:3]}")
        if nm == single:
            g1 = api.cdl(nm, *dv, penetration=pens[nm]).cpu().numpy()
            if not (g1 == exp).all():
                bad += 1
                log(f"MISMATCH pattern(single) {nm} N={N} T={T} mode={mode}")
    return bad


def sweep_patterns(seed: int, iters: int, log=print) -> int:
    rng = np.random.default_rng(seed)
    return sum(_pattern_case(rng, log) for _ in range(iters))


# ---- ragged batches (what the Polars plugin's `_over` entry points launch): one long column, groups of any length
def _ragged_case(rng, name, log) -> int:
    cols, pspec, outs, fam = SPEC[name]
    shape = rng.random()
    big = shape < 0.3           # groups averaging >= 1024 rows take the one-symbol-per-wavefront forms where one exists
    panel = shape >= 0.7        # many groups of similar length (>= 16 groups, >= 16 384 rows): re-housed as a regular batch, tiled kernels (round 5)
    if panel:
        L = int(rng.choice([40, 100, 257, 600]))
        ng = int(np.ceil(18000 / (0.9 * L))) + int(rng.integers(0, 12))
        lens = rng.integers(int(0.8 * L), L + 1, size=ng)
        for _ in range(int(rng.integers(0, 4))):   # a few groups shorter than every warm-up
            lens[int(rng.integers(0, ng))] = int(rng.choice([0, 1, 2, 7, 31]))
    else:
        ng = int(rng.integers(1, 12 if big else 60))
        lens = rng.integers(600, 3000, size=ng) if big else rng.choice([0, 1, 2, 7, 31, 32, 33, 64, 65, 100, 255, 256, 300], size=ng)
    lens = np.asarray(lens, dtype=np.int64)
    if big and rng.random() < 0.5:
        lens[int(rng.integers(0, ng))] = int(rng.choice([0, 1, 5, 64]))
    off = np.r_[0, np.cumsum(lens)].astype(np.int64)
    total = int(off[-1])
    if total == 0:
        return 0
    d = {k: np.ascontiguousarray(v[0]) for k, v in oracle.gen_ohlcv(int(rng.integers(1, 1 << 30)), 1, total, 0).items()}
    d["real"] = d["close"]
    d["periods"] = rng.integers(0, 40, size=total).astype(np.float64)
    r = rng.random()
    if r < 0.3 and fam in ("N-A", "N-C", "N-0") and name != "stochrsi":
        for k in ("open", "high", "low", "close", "volume", "real"):
            d[k] = d[k].copy(); d[k][rng.random(total) < 0.01] = oracle.NULL
    elif r < 0.5:
        for k in ("open", "high", "low", "close", "volume", "real"):
            d[k] = d[k].copy(); d[k][rng.random(total) < 0.003] = np.nan
    params = {}
    for pname, kind, _default in pspec:
        if kind == I:
            params[pname] = int(rng.integers(0, 9)) if "matype" in pname else int(rng.choice([1, 2, 3, 5, 9, 14, 30, 64, int(rng.integers(1, 80))]))
        elif name == "mama":
            params[pname] = float(rng.choice([0.02, 0.05, 0.2, 0.5]))
        else:
            params[pname] = float(rng.choice([0.02, 0.2, 0.5, 2.0]))
    if name == "mavp":
        lo_ = int(rng.integers(2, 10)); params["minperiod"], params["maxperiod"] = lo_, lo_ + int(rng.integers(0, 30))
    got = [g.cpu().numpy() for g in api.call(name, *[torch.from_numpy(d[c]).cuda() for c in cols], offsets=off, **params)]
    bad = 0
    for s_ in range(ng):
        lo, hi = int(off[s_]), int(off[s_ + 1])
        if hi == lo:
            continue
        exp = oracle.call(name, *[d[c][lo:hi] for c in cols], **params)
        for (oname, dt), g, e in zip(outs, got, exp):
            g, e = g[lo:hi], np.asarray(e).reshape(-1)
            if name in TRANSC:
                ok = np.isclose(g, e, rtol=1e-12, atol=1e-12, equal_nan=True) | (_bits(g) == _bits(e))
            else:
                ok = (_bits(g) == _bits(e))
                if dt == "f8":
                    gn, en = _bits(g) == np.uint64(oracle.NULL_BITS), _bits(e) == np.uint64(oracle.NULL_BITS)
                    ok = (ok | ((g != g) & (e != e))) & (gn == en)
            if not np.all(ok):
                bad += 1
                log(f"MISMATCH ragged {name}.{oname} group {s_} len {hi - lo} of {lens.tolist()} {params}: {int((~ok).sum())} cells, first {np.argwhere(~ok)[:3].tolist()} got {g[~ok][:2]} exp {e[~ok][:2]}")
                break
    return bad


def sweep_ragged(seed: int, iters: int, log=print) -> int:
    rng = np.random.default_rng(seed)
    names = sorted(SPEC)
    return sum(_ragged_case(rng, names[it % len(names)], log) for it in range(iters))


# ---- the callers either side of the path: leveraged multi-symbol backtest, signal rules, cross-sectional IC, returns / rolling extrema
def _poison(rng, x, p_null=0.0, p_nan=0.0, p_neg=0.0):
    x = x.copy()
    if p_null: x[rng.random(x.shape) < p_null] = oracle.NULL
    if p_nan: x[rng.random(x.shape) < p_nan] = np.nan
    if p_neg: x[rng.random(x.shape) < p_neg] = -1.0
    return x


def _eq(g, e):
    g, e = np.asarray(g), np.asarray(e)
    if g.dtype.kind != "f":
        return bool((g == e).all())
    return bool(((_bits(g) == _bits(e)) | ((g != g) & (e != e) & ((_bits(g) == np.uint64(oracle.NULL_BITS)) == (_bits(e) == np.uint64(oracle.NULL_BITS))))).all())


def _callers_case(rng, it, log) -> int:
    N, T = int(rng.integers(1, 60)), int(rng.choice([1, 2, 30, 64, 65, 200, 700, int(rng.integers(1, 1200))]))
    d = oracle.gen_ohlcv(int(rng.integers(1, 1 << 30)), N, T, 0)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    bad = 0
    kind = it % 5
    if kind == 0:   # leveraged engine
        price = _poison(rng, d["close"], p_null=float(rng.choice([0, 0.01])), p_nan=float(rng.choice([0, 0.005])), p_neg=float(rng.choice([0, 0.005])))
        dens = float(rng.choice([0.01, 0.05, 0.3]))
        buy = (rng.random(price.shape) < dens).astype(np.uint8); sell = (rng.random(price.shape) < dens).astype(np.uint8)
        bench = d["open"][0].copy() if rng.random() < 0.6 else None
        kw = dict(leverage=float(rng.choice([1.0, 2.0, 3.0, 10.0])), slippage=float(rng.choice([0.0, 0.002])), interest_rate=float(rng.choice([0.0, 0.06, 0.5])),
                  margin_call_threshold=float(rng.choice([0.3, 0.6, 0.9])), position_size=float(rng.choice([1.0, 0.5])),
                  min_commission=float(rng.choice([5.0, 0.0, 50.0])), initial_capital=float(rng.choice([100000.0, 5000.0, 300.0])))
        mt = int(rng.choice([1, 4, 64]))
        e = oracle.backtest_leveraged(price, buy, sell, benchmark=bench, max_trades=mt, **kw)
        g = api.backtest_leveraged(dev(price), dev(buy), dev(sell), benchmark=None if bench is None else dev(bench), max_trades=mt, **kw)
        for k in ("cash", "stock_value", "total_value"):
            if not _eq(g[k].cpu().numpy(), e[k]):
                bad += 1; log(f"MISMATCH leveraged.{k} N={N} T={T} {kw}")
        if not (g["trade_count"].cpu().numpy() == e["trade_count"]).all():
            bad += 1; log(f"MISMATCH leveraged.trade_count N={N} T={T} {kw}")
        for k, v in e["trades"].items():
            if not _eq(g["trades"][k].cpu().numpy(), v):
                bad += 1; log(f"MISMATCH leveraged.trades.{k} N={N} T={T} {kw} max_trades={mt}")
        s_, es = g["summary"].cpu().numpy(), e["summary"]
        okr = ~np.isnan(es).any(axis=1)
        if not (all((_bits(s_[okr, k]) == _bits(es[okr, k])).all() for k in (1, 5, 6, 7)) and np.allclose(s_[okr], es[okr], rtol=1e-12, atol=1e-13)):
            bad += 1; log(f"MISMATCH leveraged.summary N={N} T={T} {kw}")
    elif kind == 1:  # signal rules
        a = _poison(rng, d["close"], p_null=0.01, p_nan=0.01); b = _poison(rng, d["open"], p_null=0.01, p_nan=0.01)
        if rng.random() < 0.5:
            q = 0.25; a, b = np.round(a / q) * q, np.round(b / q) * q   # ties: a touch is not a cross
        for tag, g, e in (("cross", api.cross_signals(dev(a), dev(b)), oracle.cross_signals(a, b)),
                          ("band", api.band_signals(dev(a), 9.0, 11.0), oracle.band_signals(a, 9.0, 11.0)),
                          ("channel0", api.channel_signals(dev(a), dev(d["low"]), dev(d["high"]), 0), oracle.channel_signals(a, d["low"], d["high"], 0)),
                          ("channel1", api.channel_signals(dev(a), dev(d["low"]), dev(d["high"]), 1), oracle.channel_signals(a, d["low"], d["high"], 1))):
            for k in range(2):
                if not (g[k].cpu().numpy() == e[k]).all():
                    bad += 1; log(f"MISMATCH rule {tag}[{k}] N={N} T={T}")
    elif kind == 2:  # cross-sectional IC / Rank-IC (days are columns)
        f = rng.standard_normal((N, T)); r = 0.1 * f + rng.standard_normal((N, T))
        if rng.random() < 0.5: f = np.round(f * 4) / 4     # ties in the ranks
        f = _poison(rng, f, p_null=0.02, p_nan=0.01); r = _poison(rng, r, p_null=0.02, p_nan=0.01)
        if rng.random() < 0.3: f[rng.random(f.shape) < 0.01] = np.inf
        for m in (0, 1):
            ic, nv = api.factor_ic(dev(f), dev(r), method=m)
            eic, env = oracle.factor_ic(f, r, method=m)
            if not ((nv.cpu().numpy() == env).all() and _eq(ic.cpu().numpy(), eic)):
                bad += 1; log(f"MISMATCH factor_ic method={m} N={N} T={T}")
    elif kind == 3:  # returns, rolling extrema
        x = _poison(rng, d["close"], p_null=float(rng.choice([0, 0.01])), p_nan=float(rng.choice([0, 0.02])))
        if rng.random() < 0.3: x[rng.random(x.shape) < 0.01] = 0.0
        w = int(rng.choice([0, 1, 2, 5, 20, T, T + 1]))
        for nm, prm in (("rolling_max", dict(window=w)), ("rolling_min", dict(window=w)), ("returns", dict(period=w, method=0)), ("returns", dict(period=w, method=1))):
            e = oracle.call(nm, x, **prm)[0]
            g = api.call(nm, dev(x), **prm)[0].cpu().numpy()
            ok = (_bits(g) == _bits(e)) | ((g != g) & (e != e)) if nm != "returns" or prm["method"] == 0 else np.isclose(g, e, rtol=1e-14, atol=0, equal_nan=True) | (_bits(g) == _bits(e))
            if not ok.all():
                bad += 1; log(f"MISMATCH {nm} {prm} N={N} T={T}: first {np.argwhere(~ok)[:3].tolist()} got {g[~ok][:2]} exp {e[~ok][:2]}")
    else:            # single-asset backtest, signals as inputs, small shapes with poisoned prices and a benchmark
        price = _poison(rng, d["close"], p_null=0.01, p_nan=0.005, p_neg=0.005)
        dens = float(rng.choice([0.02, 0.2, 1.0]))
        buy = (rng.random(price.shape) < dens).astype(np.uint8); sell = (rng.random(price.shape) < dens).astype(np.uint8)
        bm = _poison(rng, d["open"], p_nan=float(rng.choice([0, 0.01]))) if rng.random() < 0.5 else None
        kw = dict(initial_capital=float(rng.choice([100000.0, 30.0])), position_size=float(rng.choice([1.0, 0.3])), min_commission=float(rng.choice([5.0, 0.0])))
        e = oracle.backtest(price, buy, sell, benchmark=bm, **kw)
        g = api.backtest_vectorized(dev(price), dev(buy), dev(sell), benchmark=None if bm is None else dev(bm), **kw)
        for k, nm in enumerate(("position", "cash", "equity")):
            if not _eq(g[k].cpu().numpy(), e[k]):
                bad += 1; log(f"MISMATCH backtest.{nm} N={N} T={T} {kw}")
        s_, es = g[3].cpu().numpy(), e[3]
        okr = ~np.isnan(es).any(axis=1)
        if not (all((_bits(s_[okr, k]) == _bits(es[okr, k])).all() for k in (1, 5, 6, 7)) and np.allclose(s_[okr], es[okr], rtol=1e-12, atol=1e-13)
                and np.isnan(s_[~okr]).any(axis=1).all()):
            bad += 1; log(f"MISMATCH backtest.summary N={N} T={T} {kw} bench={bm is not None}")
    return bad


def sweep_callers(seed: int, iters: int, log=print) -> int:
    rng = np.random.default_rng(seed)
    return sum(_callers_case(rng, it, log) for it in range(iters))


# ---- recorded suites: a random mix of calls recorded into one job grid (pq_suite_begin / end), replayed twice
def _suite_case(rng, log) -> int:
    import ctypes as C
    from polars_quant_amd._lib import Batch, check, lib
    N = int(rng.choice([1, 3, 64, 65, 130, int(rng.integers(1, 200))]))
    T = int(rng.choice([1, 8, 63, 64, 200, 256, 257, 1024, int(rng.integers(1, 700))]))
    pitch = T if N == 1 else int(rng.choice([T, T + (T % 2), (T + 15) // 16 * 16, T + 1, T + 3]))   # (one series: the wrapper passes stride = len)
    d = oracle.gen_ohlcv(int(rng.integers(1, 1 << 30)), N, T, 0)
    d["real"] = d["close"]
    d["periods"] = rng.integers(0, 40, size=(N, T)).astype(np.float64)
    holes = {k: v.copy() for k, v in d.items()}
    mode = int(rng.integers(0, 3))
    for k in ("open", "high", "low", "close", "volume", "real"):
        if mode == 1: holes[k][rng.random((N, T)) < 0.02] = oracle.NULL
        if mode == 2: holes[k][rng.random((N, T)) < 0.01] = np.nan
    clean = d if mode != 2 else holes      # N-B functions reject nulls; NaN values are fine everywhere
    def dev(a):
        buf = torch.full((N, pitch), 1e300, dtype=torch.float64, device="cuda")
        buf[:, :T] = torch.from_numpy(np.ascontiguousarray(a)).cuda()
        return buf[:, :T]
    gd = {id(x): {k: dev(v) for k, v in x.items()} for x in (clean, holes)}
    names = sorted(SPEC)
    picks = [names[i] for i in rng.choice(len(names), size=int(rng.integers(3, 30)), replace=True)]
    L, h, b = lib(), api.ctx(0), Batch(N, T, pitch)
    check(L.pq_suite_begin(h, C.byref(b)))
    rec = []
    try:
        for name in picks:
            cols, pspec, outs, fam = SPEC[name]
            params = {}
            for pname, kind, _default in pspec:
                if kind == I:
                    params[pname] = int(rng.integers(0, 9)) if "matype" in pname else max(int(rng.choice([0, 1, 2, 3, 5, 9, 14, 30, T, int(rng.integers(1, 60))])), 0)
                elif name == "mama":
                    params[pname] = float(rng.choice([0.02, 0.05, 0.2, 0.5]))
                else:
                    params[pname] = float(rng.choice([0.02, 0.2, 0.5, 2.0]))
            if name == "mavp":
                lo_ = int(rng.integers(0, 12)); params["minperiod"], params["maxperiod"] = lo_, lo_ + int(rng.integers(0, 30))
            src = holes if (fam in ("N-A", "N-C", "N-0") and name != "stochrsi") else clean
            rec.append((name, params, src, api.call(name, *[gd[id(src)][c] for c in cols], **params)))
        if rng.random() < 0.5:
            o_, h_, l_, c_ = (gd[id(clean)][k] for k in ("open", "high", "low", "close"))
            rec.append(("cdl_all", {}, clean, api.cdl_all(o_, h_, l_, c_)))
    except Exception:
        L.pq_suite_abort(h)
        raise
    suite = C.c_void_p()
    check(L.pq_suite_end(h, C.byref(suite)))
    bad = 0
    try:
        check(L.pq_suite_run(h, suite)); check(L.pq_suite_run(h, suite))
        torch.cuda.synchronize()
        for name, params, src, got in rec:
            if name == "cdl_all":
                from polars_quant_amd._spec import PATTERN_NAMES
                for nm in PATTERN_NAMES[:: 7]:
                    if not (got[nm].cpu().numpy() == oracle.pattern(nm, src["open"], src["high"], src["low"], src["close"])).all():
                        bad += 1; log(f"MISMATCH suite pattern {nm} N={N} T={T} pitch={pitch}")
                continue
            cols, pspec, outs, fam = SPEC[name]
            exp = oracle.call(name, *[src[c] for c in cols], **params)
            for (oname, dt), g, e in zip(outs, got, exp):
                g = g.cpu().numpy()
                if name in TRANSC:
                    ok = np.isclose(g, e, rtol=1e-12, atol=1e-12, equal_nan=True) | (_bits(g) == _bits(e))
                else:
                    ok = _bits(g) == _bits(e)
                    if dt == "f8":
                        gn, en = _bits(g) == np.uint64(oracle.NULL_BITS), _bits(e) == np.uint64(oracle.NULL_BITS)
                        ok = (ok | ((g != g) & (e != e))) & (gn == en)
                if not np.all(ok):
                    bad += 1
                    log(f"MISMATCH suite {name}.{oname} N={N} T={T} pitch={pitch} mode={mode} {params}: {int((~ok).sum())} cells, first {np.argwhere(~ok)[:3].tolist()} got {g[~ok][:2]} exp {e[~ok][:2]}")
    finally:
        check(L.pq_suite_destroy(h, suite))
    return bad


def sweep_suites(seed: int, iters: int, log=print) -> int:
    rng = np.random.default_rng(seed)
    return sum(_suite_case(rng, log) for _ in range(iters))


def sweep_long(seed: int, iters: int, log=print, names=WT_NAMES) -> int:
    """-> mismatching outputs over `iters` random long-series cases: two indicators (default: those with a wave form), then one backtest,
    in turn.  names=sorted(SPEC): every function at 1 024 .. 4 096 rows -- nulls / NaNs that arrive long after the tiled bodies have
    switched to their straight-line tiles"""
    rng = np.random.default_rng(seed)
    bad = 0
    for it in range(iters):
        bad += _long_backtest(rng, log) if it % 3 == 2 else _long_indicator(rng, names[(it - it // 3) % len(names)], log)
    return bad


if __name__ == "__main__":
    if len(sys.argv) > 3 and sys.argv[3] == "suites":
        print("done, mismatching outputs:", sweep_suites(int(sys.argv[1]), int(sys.argv[2])))
        sys.exit(0)
    if len(sys.argv) > 3 and sys.argv[3] == "callers":
        print("done, mismatching outputs:", sweep_callers(int(sys.argv[1]), int(sys.argv[2])))
        sys.exit(0)
    if len(sys.argv) > 3 and sys.argv[3] == "ragged":
        print("done, mismatching outputs:", sweep_ragged(int(sys.argv[1]), int(sys.argv[2])))
        sys.exit(0)
    if len(sys.argv) > 3 and sys.argv[3] == "patterns":
        print("done, mismatching outputs:", sweep_patterns(int(sys.argv[1]), int(sys.argv[2])))
        sys.exit(0)
    if len(sys.argv) > 3 and sys.argv[3] == "long_all":
        print("done, mismatching outputs:", sweep_long(int(sys.argv[1]), int(sys.argv[2]), names=sorted(SPEC)))
        sys.exit(0)
    if len(sys.argv) > 3 and sys.argv[3] == "long":
        print("done, mismatching outputs:", sweep_long(int(sys.argv[1]), int(sys.argv[2])))
        sys.exit(0)

    n_bad = sweep(int(sys.argv[1]) if len(sys.argv) > 1 else 7, int(sys.argv[2]) if len(sys.argv) > 2 else 400)
    print("done, mismatching outputs:", n_bad)
