"""-m gpu: RAGGED batches (pq_batch.offsets, include/pq_hip.h): long columns sorted by symbol + group offsets -- what the reference
computes with one plugin call per group of whatever length under `.over("symbol")` (python/polars_quant/talib/momentum.py:13-16,
is_elementwise=False).  Every group must equal the oracle run on that group ALONE: lengths from {0, 1 .. 400},
including groups shorter than every warm-up (all-null there, as in the reference), bit for bit (transcendental rows at 1e-12)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
from tolerance import SCALE_OF  # noqa: E402

SEED = 0x5EED0004
TRANSCENDENTAL = {"ht_dcperiod", "ht_dcphase", "ht_phasor", "ht_sine", "mama"}
NULLB = np.uint64(0x7FF80000504E554C)


@pytest.fixture(scope="module")
def pq():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import polars_quant_amd as pq
    from polars_quant_amd._lib import lib
    lib()
    return pq


@pytest.fixture(scope="module")
def groups(oracle):
    rng = np.random.default_rng(21)
    lens = np.r_[0, 1, 2, 3, 5, 8, 13, 25, 26, 27, 31, 32, 33, 34, 35, 63, 64, 65, 127, 128, 129, 400, rng.integers(1, 400, size=70), 0, 7]
    lens = lens.astype(np.int64)
    off = np.r_[0, np.cumsum(lens)]
    d = oracle.gen_ohlcv(SEED, 1, int(off[-1]), 0)       # one long random walk, cut into the groups
    d = {k: np.ascontiguousarray(v[0]) for k, v in d.items()}
    d["real"] = d["close"]
    d["periods"] = rng.integers(0, 40, size=off[-1]).astype(np.float64)
    return d, off, lens


def bits(a):
    return np.ascontiguousarray(a).view(np.uint64 if a.dtype == np.float64 else np.uint32)


def check(name, g, e, exact, price):
    assert g.shape == e.shape and g.dtype == e.dtype, (name, g.shape, e.shape)
    if e.dtype != np.float64:
        assert (g == e).all(), name
        return
    assert ((bits(g) == NULLB) == (bits(e) == NULLB)).all(), f"{name}: null masks differ"
    if exact:
        bad = (bits(g) != bits(e)) & ~(np.isnan(g) & np.isnan(e))
        assert not bad.any(), f"{name}: {bad.sum()} rows not bit-exact, first {np.argwhere(bad)[:3].ravel().tolist()}"
    else:
        sc = SCALE_OF.get(name.split("{")[0], 0.0)
        sc = np.abs(price) if isinstance(sc, str) else sc
        ok = bits(e) != NULLB
        err = np.abs(g[ok] - e[ok]) / np.maximum(np.maximum(np.abs(e[ok]), np.broadcast_to(sc, e.shape)[ok]), 1e-300)
        err[np.isnan(g[ok]) & np.isnan(e[ok])] = 0
        assert (err <= 1e-12).all(), f"{name}: max error {np.nanmax(err):.3e}"


ALL_FUNCS = sorted(__import__("polars_quant_amd._spec", fromlist=["SPEC"]).SPEC)


@pytest.mark.parametrize("name", ALL_FUNCS)
def test_every_function_on_ragged_groups(pq, oracle, groups, name):
    from polars_quant_amd import api
    d, off, lens = groups
    cols = pq.SPEC[name][0]
    got = api.call(name, *[torch.from_numpy(d[c]).cuda() for c in cols], offsets=off)
    got = [g.cpu().numpy() for g in got]
    for s in range(len(lens)):
        lo, hi = off[s], off[s + 1]
        if hi == lo:
            continue
        exp = oracle.call(name, *[d[c][lo:hi] for c in cols])
        for (oname, _), g, e in zip(pq.SPEC[name][2], got, exp):
            check(f"{name}.{oname}{{group {s} len {hi - lo}}}", g[lo:hi], np.asarray(e).reshape(-1), name not in TRANSCENDENTAL, d["close"][lo:hi])


@pytest.fixture(scope="module")
def panel(oracle):
    """What `.over("symbol")` usually sees: a few hundred groups of SIMILAR length (a panel with listing gaps), a handful of very
    short ones among them (shorter than every warm-up: all-null there, as in the reference) -- the shape the re-housed tiled path
    takes (n_series x pitch <= 1.5 x total rows)."""
    rng = np.random.default_rng(22)
    lens = np.r_[256, 255, 1, 0, 5, 31, 32, 33, 64, 129, 200, 254, rng.integers(170, 257, size=300), 2, 256].astype(np.int64)
    off = np.r_[0, np.cumsum(lens)]
    d = oracle.gen_ohlcv(SEED + 20, 1, int(off[-1]), 0)
    d = {k: np.ascontiguousarray(v[0]) for k, v in d.items()}
    d["real"] = d["close"]
    d["periods"] = rng.integers(0, 40, size=off[-1]).astype(np.float64)
    assert len(lens) * 256 <= 1.5 * off[-1] and off[-1] >= 16384
    return d, off, lens


@pytest.mark.parametrize("name", ALL_FUNCS)
def test_every_function_on_a_panel_of_similar_groups_rehoused(pq, oracle, panel, name, monkeypatch):
    """The re-housed path (ragged -> regular padded batch -> the tiled kernel, told each group's own length -> back): every group equals
    the oracle run on it ALONE, and the whole column equals, bit for bit, what the per-lane gather forms write (PQ_NO_RG_PACK=1)."""
    from polars_quant_amd import api
    d, off, lens = panel
    cols = pq.SPEC[name][0]
    ins = [torch.from_numpy(d[c]).cuda() for c in cols]
    api.ragged_rehouse_stats(reset=True)
    api.wt_stats(reset=True)
    got = [g.cpu().numpy() for g in api.call(name, *ins, offsets=off)]
    rehoused, by_wave = api.ragged_rehouse_stats(), api.wt_stats()[0]
    seq = name not in ROW_FUNCS and name not in GATHER_FUNCS
    if seq:
        assert rehoused >= 1 or by_wave > 0, f"{name}: neither the re-housed tiled path nor a wave-per-group form ran"
    monkeypatch.setenv("PQ_NO_RG_PACK", "1")
    ref = [g.cpu().numpy() for g in api.call(name, *ins, offsets=off)]
    monkeypatch.delenv("PQ_NO_RG_PACK")
    for (oname, _), g, r in zip(pq.SPEC[name][2], got, ref):
        if name in TRANSCENDENTAL:
            continue
        same_ = (bits(g) == bits(r)) | ((g != g) & (r != r)) if g.dtype == np.float64 else (g == r)
        assert same_.all(), f"{name}.{oname}: re-housed and gather forms differ at rows {np.argwhere(~same_)[:3].ravel().tolist()}"
    for s in range(len(lens)):
        lo, hi = off[s], off[s + 1]
        if hi == lo:
            continue
        exp = oracle.call(name, *[d[c][lo:hi] for c in cols])
        for (oname, _), g, e in zip(pq.SPEC[name][2], got, exp):
            check(f"{name}.{oname}{{panel group {s} len {hi - lo}}}", g[lo:hi], np.asarray(e).reshape(-1), name not in TRANSCENDENTAL, d["close"][lo:hi])


# the functions that are row-parallel kernels (a pure function of a bounded window): ragged batches are native to them
ROW_FUNCS = {"mom", "roc", "rocp", "rocr", "rocr100", "ht_trendline", "ht_trendmode", "trange", "bop", "avgprice", "medprice", "typprice", "wclprice",
             "aroon", "aroonosc", "willr", "midprice"}


# compute-bound walks whose per-lane form, called alone, beats re-housing + the tiled body (Op::RG_GATHER, profiles/r05_bench_ragged.json)
GATHER_FUNCS = {"sar", "sarext", "stoch", "stochf", "stochrsi", "obv", "mama", "ht_dcperiod", "ht_dcphase", "ht_phasor", "ht_sine"}


def test_patterns_and_parameters_on_ragged_groups(pq, oracle, groups):
    from polars_quant_amd import api
    d, off, lens = groups
    rich = oracle.gen_ohlcv(SEED + 1, 1, int(off[-1]), 1)
    o, h, l, c = (np.ascontiguousarray(rich[k][0]) for k in ("open", "high", "low", "close"))
    T = [torch.from_numpy(x).cuda() for x in (o, h, l, c)]
    for nm in pq.PATTERN_NAMES[::4]:
        g = api.cdl(nm, *T, offsets=off).cpu().numpy()
        for s in range(len(lens)):
            lo, hi = off[s], off[s + 1]
            if hi > lo:
                assert (g[lo:hi] == oracle.pattern(nm, o[lo:hi], h[lo:hi], l[lo:hi], c[lo:hi])).all(), (nm, s)
    for name, prm in (("ema", dict(timeperiod=5)), ("sma", dict(timeperiod=40)), ("macd", dict(fastperiod=3, slowperiod=7, signalperiod=4)),
                      ("rsi", dict(timeperiod=2)), ("stoch", dict(fastk_period=7, slowk_period=4, slowk_matype=1, slowd_period=3, slowd_matype=2)),
                      ("mavp", dict(minperiod=2, maxperiod=12, matype=1)), ("bbands", dict(timeperiod=5, nbdevup=1.5, nbdevdn=2.5))):
        cols = pq.SPEC[name][0]
        got = [g.cpu().numpy() for g in api.call(name, *[torch.from_numpy(d[c]).cuda() for c in cols], offsets=off, **prm)]
        for s in range(len(lens)):
            lo, hi = off[s], off[s + 1]
            if hi == lo:
                continue
            exp = oracle.call(name, *[d[c][lo:hi] for c in cols], **prm)
            for (oname, _), g, e in zip(pq.SPEC[name][2], got, exp):
                check(f"{name}.{oname}{{{prm} group {s}}}", g[lo:hi], np.asarray(e).reshape(-1), True, d["close"][lo:hi])


def same(g, e):
    return (bits(g) == bits(e)) | (np.isnan(g) & np.isnan(e))


def test_backtests_on_ragged_groups(pq, oracle):
    from polars_quant_amd import api
    rng = np.random.default_rng(5)
    lens = np.r_[2520, 1, 0, 64, 65, 700, 2000, 4096, 33, rng.integers(1, 3000, size=12)].astype(np.int64)
    off = np.r_[0, np.cumsum(lens)]
    close = np.ascontiguousarray(oracle.gen_ohlcv(SEED + 2, 1, int(off[-1]), 0)["close"][0])
    bench = np.ascontiguousarray(oracle.gen_ohlcv(SEED + 3, 1, int(off[-1]), 0)["open"][0])
    pos, cash, eq, summ = (t.cpu().numpy() for t in api.backtest_macd_cross(torch.from_numpy(close).cuda(), offsets=off))
    buy = (rng.random(close.shape) < 0.05).astype(np.uint8)
    sell = (rng.random(close.shape) < 0.05).astype(np.uint8)
    vpos, vcash, veq, vsumm = (t.cpu().numpy() for t in api.backtest_vectorized(torch.from_numpy(close).cuda(), torch.from_numpy(buy).cuda(),
                                                                                torch.from_numpy(sell).cuda(), benchmark=torch.from_numpy(bench).cuda(),
                                                                                offsets=off))
    assert summ.shape == (len(lens), 8) and vsumm.shape == (len(lens), 8)
    for s in range(len(lens)):
        lo, hi = off[s], off[s + 1]
        if hi == lo:
            assert (summ[s] == 0).all()
            continue
        eb, es_ = oracle.macd_cross_signals(close[lo:hi])
        for (gp, gc, ge, gs), (ep, ec, ee, es2) in (((pos, cash, eq, summ), oracle.backtest(close[lo:hi], eb, es_)),
                                                     ((vpos, vcash, veq, vsumm), oracle.backtest(close[lo:hi], buy[lo:hi], sell[lo:hi], benchmark=bench[lo:hi]))):
            assert (bits(gp[lo:hi]) == bits(ep)).all() and (bits(gc[lo:hi]) == bits(ec)).all() and (bits(ge[lo:hi]) == bits(ee)).all(), s
            for k in (1, 5, 6, 7):
                assert bits(gs[s, k:k + 1])[0] == bits(es2[k:k + 1])[0], (s, k)
            np.testing.assert_allclose(gs[s], es2, rtol=1e-12, atol=1e-13)


def test_backtests_with_groups_longer_than_4096_rows(pq, oracle):
    """A group of 4097 .. 8192 rows puts the whole batch on chunks of more than 64 rows (two mask words per lane, BtwBits<2>): the
    short groups beside it then use a few lanes only, each with one or two words."""
    from polars_quant_amd import api
    rng = np.random.default_rng(6)
    lens = np.array([5040, 1, 70, 129, 8192, 300, 4097, 64, 2520, 130], dtype=np.int64)
    off = np.r_[0, np.cumsum(lens)]
    close = np.ascontiguousarray(oracle.gen_ohlcv(SEED + 12, 1, int(off[-1]), 0)["close"][0])
    close[off[4] + 5000: off[4] + 5003] = oracle.NULL
    buy = (rng.random(close.shape) < 0.05).astype(np.uint8)
    sell = (rng.random(close.shape) < 0.05).astype(np.uint8)
    api.backtest_wave_stats(reset=True)
    m = [t.cpu().numpy() for t in api.backtest_macd_cross(torch.from_numpy(close).cuda(), offsets=off)]
    v = [t.cpu().numpy() for t in api.backtest_vectorized(torch.from_numpy(close).cuda(), torch.from_numpy(buy).cuda(), torch.from_numpy(sell).cuda(), offsets=off)]
    assert api.backtest_wave_stats()[0] == 2 * len(lens), "the wave form must have run"
    for s in range(len(lens)):
        lo, hi = off[s], off[s + 1]
        eb, es_ = oracle.macd_cross_signals(close[lo:hi])
        for g, e in ((m, oracle.backtest(close[lo:hi], eb, es_)), (v, oracle.backtest(close[lo:hi], buy[lo:hi], sell[lo:hi]))):
            for k in range(3):
                assert (same(g[k][lo:hi], e[k])).all(), (s, k)
            for k in (1, 5, 6, 7):
                assert bits(g[3][s, k:k + 1])[0] == bits(e[3][k:k + 1])[0], (s, k)
            ok = ~np.isnan(e[3])
            np.testing.assert_allclose(g[3][s][ok], e[3][ok], rtol=1e-12, atol=1e-13)


def test_backtest_short_aligned_groups_beside_a_long_one(pq, oracle):
    """A ragged batch sizes the chunk length C for its LONGEST group (3200 rows -> C = 50): a 128- or 130-row group starting on a
    16-byte boundary then owns LDS rows up to 150 that its own staging must fill (round-3 advisor finding: the aligned staging
    stopped at row 128 and the last live lane walked stale LDS -- the previous workgroup's equity row -- into spurious signals).
    Many short groups behind long ones, so that every workgroup inherits poisoned LDS."""
    from polars_quant_amd import api
    lens = np.array([3200, 128, 130, 128, 3200, 130, 128, 2, 128, 130, 64, 128, 3200] + [128, 130] * 40, dtype=np.int64)
    off = np.r_[0, np.cumsum(lens)]
    assert all(o % 2 == 0 for o in off)       # every group starts 16-byte aligned: the double2 staging path
    close = np.ascontiguousarray(oracle.gen_ohlcv(SEED + 7, 1, int(off[-1]), 0)["close"][0])
    for rep in range(2):                       # second pass: LDS holds the first pass's equity / returns rows
        pos, cash, eq, summ = (t.cpu().numpy() for t in api.backtest_macd_cross(torch.from_numpy(close).cuda(), offsets=off))
        for s in range(len(lens)):
            lo, hi = off[s], off[s + 1]
            eb, es_ = oracle.macd_cross_signals(close[lo:hi])
            ep, ec, ee, es2 = oracle.backtest(close[lo:hi], eb, es_)
            assert (bits(pos[lo:hi]) == bits(ep)).all() and (bits(cash[lo:hi]) == bits(ec)).all() and (bits(eq[lo:hi]) == bits(ee)).all(), (rep, s)
            for k in (1, 5, 6, 7):             # max_drawdown, max_profit, win_rate, total_trades
                assert bits(summ[s, k:k + 1])[0] == bits(es2[k:k + 1])[0], (rep, s, k, summ[s], es2)
            np.testing.assert_allclose(summ[s], es2, rtol=1e-12, atol=1e-13)


def test_recorded_suite_on_a_ragged_batch(pq, oracle, groups):
    """pq_suite_begin / end on a ragged batch: sequential jobs (per-lane body: ragged series start at arbitrary rows), the fused
    row-parallel grid, the pattern kernel and the wave-per-symbol backtest replay from one recorded plan; every group equals the
    oracle run on it alone."""
    import ctypes as C
    from polars_quant_amd import api
    from polars_quant_amd._lib import BtParams, check, lib
    from polars_quant_amd._spec import BT_DEFAULTS
    d, off, lens = groups
    dev = torch.device("cuda")
    g = {k: torch.from_numpy(d[k]).to(dev) for k in ("open", "high", "low", "close", "volume")}
    b, keep = api.ragged_batch(off, dev)
    R = int(off[-1])
    f64 = lambda: torch.full((R,), -7.0, dtype=torch.float64, device=dev)
    out = {"ema": f64(), "rsi": f64(), "atr": f64(), "mom": f64(), "typprice": f64(), "macd": [f64(), f64(), f64()], "bt": [f64(), f64(), f64()]}
    pat = torch.full((R,), -7, dtype=torch.int32, device=dev)
    summ = torch.empty((len(lens), 8), dtype=torch.float64, device=dev)
    L, h, prm, vp = lib(), api.ctx(0), BtParams(**BT_DEFAULTS), (lambda t: C.c_void_p(t.data_ptr()))
    check(L.pq_suite_begin(h, C.byref(b)))
    check(L.pq_ema(h, C.byref(b), vp(g["close"]), 9, vp(out["ema"])))
    check(L.pq_rsi(h, C.byref(b), vp(g["close"]), 14, vp(out["rsi"])))
    check(L.pq_atr(h, C.byref(b), vp(g["high"]), vp(g["low"]), vp(g["close"]), 5, vp(out["atr"])))
    check(L.pq_mom(h, C.byref(b), vp(g["close"]), 10, vp(out["mom"])))
    check(L.pq_typprice(h, C.byref(b), vp(g["high"]), vp(g["low"]), vp(g["close"]), vp(out["typprice"])))
    check(L.pq_macd(h, C.byref(b), vp(g["close"]), 12, 26, 9, *[vp(t) for t in out["macd"]]))
    check(L.pq_cdl(h, C.byref(b), pq.PATTERN_NAMES.index("cdlengulfing"), vp(g["open"]), vp(g["high"]), vp(g["low"]), vp(g["close"]), C.c_double(0.3), vp(pat)))
    check(L.pq_backtest_macd_cross(h, C.byref(b), vp(g["close"]), 12, 26, 9, C.byref(prm), *[vp(t) for t in out["bt"]], vp(summ)))
    suite = C.c_void_p()
    check(L.pq_suite_end(h, C.byref(suite)))
    try:
        check(L.pq_suite_run(h, suite))
        check(L.pq_suite_run(h, suite))
        torch.cuda.synchronize()
    finally:
        check(L.pq_suite_destroy(h, suite))
    del keep
    res = {k: ([t.cpu().numpy() for t in v] if isinstance(v, list) else v.cpu().numpy()) for k, v in out.items()}
    pat, summ = pat.cpu().numpy(), summ.cpu().numpy()
    for s in range(len(lens)):
        lo, hi = off[s], off[s + 1]
        if hi == lo:
            continue
        sl = {k: d[k][lo:hi] for k in ("open", "high", "low", "close")}
        check_ = lambda nm, got, exp: check(f"{nm}{{suite, group {s}}}", got[lo:hi], np.asarray(exp).reshape(-1), True, sl["close"])
        for nm, exp in (("ema", oracle.call("ema", sl["close"], timeperiod=9)[0]), ("rsi", oracle.call("rsi", sl["close"], timeperiod=14)[0]),
                        ("atr", oracle.call("atr", sl["high"], sl["low"], sl["close"], timeperiod=5)[0]), ("mom", oracle.call("mom", sl["close"], timeperiod=10)[0]),
                        ("typprice", oracle.call("typprice", sl["high"], sl["low"], sl["close"])[0])):
            g_, e_ = res[nm][lo:hi], np.asarray(exp).reshape(-1)
            assert ((bits(g_) == bits(e_)) | (np.isnan(g_) & np.isnan(e_))).all(), (nm, s)
        for k, e_ in enumerate(oracle.call("macd", sl["close"])):
            g_ = res["macd"][k][lo:hi]
            assert ((bits(g_) == bits(np.asarray(e_).reshape(-1))) | (np.isnan(g_) & np.isnan(np.asarray(e_).reshape(-1)))).all(), ("macd", k, s)
        assert (pat[lo:hi] == oracle.pattern("cdlengulfing", sl["open"], sl["high"], sl["low"], sl["close"]).reshape(-1)).all(), s
        eb, es_ = oracle.macd_cross_signals(sl["close"])
        ep, ec, ee, esum = oracle.backtest(sl["close"], eb, es_)
        for g_, e_ in zip(res["bt"], (ep, ec, ee)):
            assert (bits(g_[lo:hi]) == bits(e_)).all(), ("backtest", s)
        np.testing.assert_allclose(summ[s], esum, rtol=1e-12, atol=1e-13)


def test_offsets_must_start_at_row_zero_and_rule_inputs_must_agree_in_shape(pq, oracle):
    """(round-3 advisor) rows in front of the first group would belong to no group and their outputs stayed unwritten; a rule kernel
    indexes all its columns with ONE batch, so a shorter column was read out of bounds / left output rows unwritten"""
    from polars_quant_amd import api
    from polars_quant_amd._lib import PqError
    x = torch.from_numpy(oracle.gen_ohlcv(SEED + 9, 1, 100, 0)["close"][0]).cuda()
    with pytest.raises(PqError, match="starting at 0"):
        api.call("ema", x, timeperiod=5, offsets=np.array([10, 60, 100]))
    (ok,) = api.call("ema", x, timeperiod=5, offsets=np.array([0, 60, 100]))
    assert ok.shape == (100,)
    a = torch.rand((4, 50), dtype=torch.float64, device="cuda")
    buy = torch.zeros((4, 50), dtype=torch.uint8, device="cuda")
    with pytest.raises(PqError, match="shapes differ"):
        api.gate_signals(buy, buy[:, :40].contiguous(), a, 0, 0.5)
    with pytest.raises(PqError, match="shapes differ"):
        api.zscore(a, a[:3].contiguous(), a)
