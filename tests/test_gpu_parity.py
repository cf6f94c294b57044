"""-m gpu parity: the HIP path (through the C ABI) against the CPU oracle on identical seeded inputs.

Bar (BASELINE.json north_star): bit-exact for integer outputs; f64 indicators <= 1e-12 relative.  The HIP
kernels follow the reference's operation order, so everything that does not pass through a device
transcendental (atan/sin/pow) is in fact compared BIT-FOR-BIT; the HT_*/MAMA family and the summary's
annualised return / sharpe use the stated 1e-12 tolerance.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

# (symbols, days): an odd row pitch runs the 8-byte form of the two-wave LDS-tiled body (rows are not 16-byte aligned:
# run_seq_lds<Op, UNAL> / seq_jobs_kernel<3>; the per-lane gather body run_seq / seq_jobs_kernel<2> is left to ragged batches
# and very long windows, tests/test_ragged_gpu.py), the even ones the body that bench.py times (seq_jobs_kernel<0>) -- with a
# ragged last tile (304 = 38 x 8 = 19 x 16; 312 = 39 x 8 = 19.5 x 16: a 16-row tail for K = 16 ops) and 1 / 2 full + 1 partial
# symbol tiles.  Every default / parameter / null case below runs on all three.
SHAPES = [(70, 301), (70, 304), (130, 312), (66, 306)]  # odd pitch (8-byte tiles); 16-byte tiles: whole tiles, an 8-row tail of the 16-row tiles, a 2-row tail of every tile
SEED = 0x5EED0002
TRANSCENDENTAL = {"ht_dcperiod", "ht_dcphase", "ht_phasor", "ht_sine", "mama"}
RTOL = 1e-12                # north_star tolerance for f64 indicators


@pytest.fixture(scope="module")
def pq():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import polars_quant_amd as pq
    from polars_quant_amd._lib import lib
    lib()  # fail loudly if the HIP library is missing
    return pq


@pytest.fixture(scope="module", params=SHAPES, ids=[f"{n}x{t}" for n, t in SHAPES])
def data(oracle, request):
    n, t = request.param
    d = oracle.gen_ohlcv(SEED, n, t, 0)
    d["real"] = d["close"]
    rng = np.random.default_rng(7)
    d["periods"] = rng.integers(0, 40, size=(n, t)).astype(np.float64)
    return d


@pytest.fixture(scope="module", params=SHAPES[:2], ids=[f"{n}x{t}" for n, t in SHAPES[:2]])
def rich(oracle, request):
    n, t = request.param
    d = oracle.gen_ohlcv(SEED + 1, n, t, 1)
    # the last two symbols carry the hand-built firing sequences of all 60 satisfiable recognisers (tests/pattern_kats.py)
    KAT_MARKS[(n, t)] = {}
    for sym, start in ((n - 1, 0), (n - 2, 44)):
        o, h, l, c, marks = kat_series(t, start)
        d["open"][sym], d["high"][sym], d["low"][sym], d["close"][sym] = o, h, l, c
        KAT_MARKS[(n, t)][sym] = marks
    d["real"] = d["close"]
    return d


def bits(a):
    return np.ascontiguousarray(a).view(np.uint64 if a.dtype == np.float64 else np.uint32)


from pattern_kats import UNSAT, kat_series

KAT_MARKS = {}   # (symbols, days) of a rich data set -> {symbol: [(row, recogniser, hand-derived value)]}
from tolerance import SCALE_OF  # which transcendental outputs are judged against a natural scale, and why


def assert_same(name, got, exp, exact=True, price=None):
    got = np.asarray(got)
    assert got.shape == exp.shape and got.dtype == exp.dtype, (name, got.shape, exp.shape, got.dtype, exp.dtype)
    if exp.dtype != np.float64:
        assert (got == exp).all(), f"{name}: {np.sum(got != exp)} int mismatches"
        return
    NULLB = np.uint64(0x7FF80000504E554C)
    gn, en = bits(got) == NULLB, bits(exp) == NULLB
    assert (gn == en).all(), f"{name}: null masks differ at {np.argwhere(gn != en)[:5].tolist()}"
    if exact:
        bad = (bits(got) != bits(exp)) & ~(np.isnan(got) & np.isnan(exp))
        assert not bad.any(), (f"{name}: {bad.sum()} of {bad.size} not bit-exact; first at {np.argwhere(bad)[:3].tolist()} "
                               f"got {got[bad][:3]} exp {exp[bad][:3]}")
    else:
        key = name.split("{")[0]
        scale = SCALE_OF.get(key, 0.0)
        if isinstance(scale, str):
            assert price is not None, f"{name}: judged against the price level, which the caller must pass"
            scale = np.abs(price)
        ok = ~en
        g, e = got[ok], exp[ok]
        sc = np.broadcast_to(scale, exp.shape)[ok]
        both_nan = np.isnan(g) & np.isnan(e)
        with np.errstate(invalid="ignore"):
            err = np.abs(g - e) / np.maximum(np.maximum(np.abs(e), sc), 1e-300)
        err[both_nan | (g == e)] = 0        # (equal infinities: inf - inf is NaN)
        assert (err <= RTOL).all(), f"{name}: max error {np.nanmax(err):.3e} (relative to max(|expected|, scale))"


def run_gpu(pq, name, d, **params):
    from polars_quant_amd import api
    cols = pq.SPEC[name][0]
    ins = [torch.from_numpy(d[c]).cuda() for c in cols]
    res = api.call(name, *ins, **params)
    torch.cuda.synchronize()
    return [r.cpu().numpy() for r in res]


ALL_FUNCS = sorted(__import__("polars_quant_amd._spec", fromlist=["SPEC"]).SPEC)


@pytest.mark.parametrize("name", ALL_FUNCS)
def test_indicator_defaults(pq, oracle, data, name):
    """every function of SURVEY 8(a) with the Python-wrapper default parameters"""
    cols = pq.SPEC[name][0]
    exp = oracle.call(name, *[data[c] for c in cols])
    got = run_gpu(pq, name, data)
    for (oname, _), g, e in zip(pq.SPEC[name][2], got, exp):
        assert_same(f"{name}.{oname}", g, e, exact=name not in TRANSCENDENTAL, price=data["close"] if "close" in data else None)


PARAM_CASES = [
    ("sma", dict(timeperiod=1)), ("sma", dict(timeperiod=5)), ("sma", dict(timeperiod=0)), ("sma", dict(timeperiod=400)),
    ("ema", dict(timeperiod=2)), ("ema", dict(timeperiod=20)), ("ema", dict(timeperiod=301)),
    ("bbands", dict(timeperiod=5, nbdevup=1.5, nbdevdn=2.5)), ("dema", dict(timeperiod=1)), ("dema", dict(timeperiod=7)),
    ("tema", dict(timeperiod=1)), ("tema", dict(timeperiod=9)), ("t3", dict(timeperiod=1, vfactor=0.7)),
    ("t3", dict(timeperiod=4, vfactor=0.0)), ("t3", dict(timeperiod=10, vfactor=0.3)),
    ("trima", dict(timeperiod=7)), ("trima", dict(timeperiod=8)), ("wma", dict(timeperiod=3)),
    ("kama", dict(timeperiod=1)), ("kama", dict(timeperiod=2)), ("kama", dict(timeperiod=10)),
    ("midpoint", dict(timeperiod=1)), ("midpoint", dict(timeperiod=3)), ("midpoint", dict(timeperiod=0)), ("midprice", dict(timeperiod=0)),
    ("midprice", dict(timeperiod=2)), ("midprice", dict(timeperiod=30)),
    ("mama", dict(fastlimit=0.5, slowlimit=0.05)),
    ("mavp", dict(minperiod=2, maxperiod=12, matype=1)), ("mavp", dict(minperiod=3, maxperiod=9, matype=2)),
    ("mavp", dict(minperiod=5, maxperiod=20, matype=0)),   # one 32-candidate job, lower half only
    ("mavp", dict(minperiod=0, maxperiod=31, matype=0)),   # all 32 candidates, candidate 0 (-> null rows)
    ("mavp", dict(minperiod=7, maxperiod=38, matype=7)),   # 32 candidates from an odd start
    ("mavp", dict(minperiod=2, maxperiod=45, matype=0)),   # 44 candidates: three masked 16-candidate jobs share the column
    ("mavp", dict(minperiod=290, maxperiod=310, matype=0)),  # windows longer than the (short) series
    ("sar", dict(acceleration=0.02, maximum=0.2)), ("sar", dict(acceleration=0.3, maximum=0.2)),
    ("sarext", dict(startvalue=0.0, offsetonreverse=0.01, accelerationinitlong=0.02, accelerationlong=0.02,
                    accelerationmaxlong=0.2, accelerationinitshort=0.03, accelerationshort=0.03, accelerationmaxshort=0.3)),
    ("sarext", dict(startvalue=-50.0, offsetonreverse=0.0, accelerationinitlong=0.02, accelerationlong=0.02,
                    accelerationmaxlong=0.2, accelerationinitshort=0.02, accelerationshort=0.02, accelerationmaxshort=0.2)),
    ("sarext", dict(startvalue=5.0, offsetonreverse=0.0, accelerationinitlong=0.02, accelerationlong=0.02,
                    accelerationmaxlong=0.2, accelerationinitshort=0.02, accelerationshort=0.02, accelerationmaxshort=0.2)),
    *[("ma", dict(timeperiod=6, matype=m)) for m in range(10)],
    ("adx", dict(timeperiod=5)), ("adxr", dict(timeperiod=5)), ("adxr", dict(timeperiod=1)), ("dx", dict(timeperiod=3)),
    ("aroon", dict(timeperiod=5)), ("aroonosc", dict(timeperiod=25)), ("cci", dict(timeperiod=5)), ("cci", dict(timeperiod=20)),
    ("cmo", dict(timeperiod=1)), ("cmo", dict(timeperiod=9)), ("macd", dict(fastperiod=3, slowperiod=7, signalperiod=4)),
    ("macd", dict(fastperiod=5, slowperiod=35, signalperiod=5)), ("mfi", dict(timeperiod=4)),
    ("mom", dict(timeperiod=0)), ("mom", dict(timeperiod=300)), ("roc", dict(timeperiod=1)),
    ("rsi", dict(timeperiod=2)), ("rsi", dict(timeperiod=30)), ("trix", dict(timeperiod=5)),
    ("ultosc", dict(timeperiod1=2, timeperiod2=3, timeperiod3=5)), ("willr", dict(timeperiod=1)), ("willr", dict(timeperiod=40)),
    *[("apo", dict(fastperiod=3, slowperiod=10, matype=m)) for m in (0, 1, 3, 5)],
    ("ppo", dict(fastperiod=3, slowperiod=10, matype=1)),
    ("macdext", dict(fastperiod=4, fastmatype=1, slowperiod=9, slowmatype=2, signalperiod=3, signalmatype=0)),
    ("macdext", dict(fastperiod=4, fastmatype=5, slowperiod=9, slowmatype=4, signalperiod=3, signalmatype=1)),
    # the single-core (SMA / EMA) path in every mix of kinds: shared SMA ring, own rings, EMA seeds
    *[("macdext", dict(fastperiod=4, fastmatype=a, slowperiod=9, slowmatype=b, signalperiod=3, signalmatype=c))
      for a, b, c in ((0, 0, 1), (1, 0, 1), (0, 1, 0), (1, 1, 1), (7, 1, 0))],
    ("macdext", dict(fastperiod=9, fastmatype=0, slowperiod=4, slowmatype=0, signalperiod=1, signalmatype=1)),
    ("stoch", dict(fastk_period=7, slowk_period=4, slowk_matype=1, slowd_period=3, slowd_matype=2)),
    ("stochf", dict(fastk_period=9, fastd_period=5, fastd_matype=1)),
    ("stochrsi", dict(timeperiod=7, fastk_period=6, fastd_period=4, fastd_matype=0)),
    ("atr", dict(timeperiod=1)), ("atr", dict(timeperiod=5)), ("natr", dict(timeperiod=20)),
    ("adosc", dict(fastperiod=2, slowperiod=5)),
]


@pytest.mark.parametrize("name,params", PARAM_CASES, ids=[f"{n}-{i}" for i, (n, _) in enumerate(PARAM_CASES)])
def test_indicator_params(pq, oracle, data, name, params):
    cols = pq.SPEC[name][0]
    exp = oracle.call(name, *[data[c] for c in cols], **params)
    got = run_gpu(pq, name, data, **params)
    for (oname, _), g, e in zip(pq.SPEC[name][2], got, exp):
        assert_same(f"{name}.{oname}{params}", g, e, exact=name not in TRANSCENDENTAL, price=data["close"])


NULL_TOLERANT = [n for n, v in __import__("polars_quant_amd._spec", fromlist=["SPEC"]).SPEC.items()
                 if v[3] in ("N-A", "N-C", "N-0") and n not in ("stochrsi",)]


@pytest.mark.parametrize("name", sorted(NULL_TOLERANT))
def test_null_bearing(pq, oracle, data, name):
    """third data set of SURVEY 8(d): 1 % random nulls + leading nulls (N-A / N-C / N-0 families)"""
    rng = np.random.default_rng(11)
    d = {}
    for k, v in data.items():
        a = v.copy()
        mask = rng.random(a.shape) < 0.01
        mask[:, :3] = True
        mask[5] = False                      # one null-free series
        mask[6, 100:] = True                 # one series that goes fully null
        a[mask] = oracle.NULL
        d[k] = a
    cols = pq.SPEC[name][0]
    params = dict(timeperiod=5) if any(p == "timeperiod" for p, _, _ in pq.SPEC[name][1]) else {}
    exp = oracle.call(name, *[d[c] for c in cols], **params)
    got = run_gpu(pq, name, d, **params)
    for (oname, _), g, e in zip(pq.SPEC[name][2], got, exp):
        assert_same(f"{name}.{oname}", g, e, exact=name not in TRANSCENDENTAL, price=data["close"] if "close" in data else None)


def test_short_and_empty_series(pq, oracle):
    """edge cases: T smaller than every warm-up, T == 1, n_series == 0"""
    from polars_quant_amd import api
    for T_ in (1, 2, 7, 31, 32):
        d = oracle.gen_ohlcv(3, 3, T_, 0)
        for name in ("sma", "ema", "t3", "kama", "macd", "rsi", "ht_dcperiod", "midpoint", "adx"):
            cols = [c if c != "real" else "close" for c in pq.SPEC[name][0]]
            exp = oracle.call(name, *[d[c] for c in cols])
            got = [r.cpu().numpy() for r in api.call(name, *[torch.from_numpy(d[c]).cuda() for c in cols])]
            for g, e in zip(got, exp):
                assert_same(f"{name}@T={T_}", g, e, exact=name != "ht_dcperiod")
    empty = torch.empty((0, 16), dtype=torch.float64, device="cuda")
    (r,) = api.call("sma", empty)
    assert r.shape == (0, 16)


def test_strided_batch(pq, oracle, data):
    """stride > len: series embedded in a wider buffer"""
    from polars_quant_amd import api
    N_SYM, T = data["close"].shape
    for pad in (11, 12):   # odd and even row pitch
        wide = torch.full((N_SYM, T + pad), 123.0, dtype=torch.float64, device="cuda")
        wide[:, :T] = torch.from_numpy(data["close"]).cuda()
        view = wide[:, :T]
        (got,) = api.call("ema", view, timeperiod=10)
        (exp,) = oracle.call("ema", data["close"], timeperiod=10)
        assert_same("ema-strided", got.cpu().numpy(), exp)
        assert (wide[:, T:] == 123.0).all(), "rows beyond len must not be written"


def test_patterns_each(pq, oracle, rich):
    from polars_quant_amd import api
    o, h, l, c = (torch.from_numpy(rich[k]).cuda() for k in ("open", "high", "low", "close"))
    fired = 0
    for name in pq.PATTERN_NAMES:
        exp = oracle.pattern(name, rich["open"], rich["high"], rich["low"], rich["close"])
        got = api.cdl(name, o, h, l, c).cpu().numpy()
        assert (got == exp).all(), f"{name}: {np.sum(got != exp)} mismatches"
        fired += int((exp != 0).any())
        assert (exp != 0).any() == (name not in UNSAT), f"{name} never fires: its parity would be zeros against zeros"
        # the HIP output against the HAND-DERIVED values (no oracle in between)
        for sym, marks in KAT_MARKS[rich["open"].shape].items():
            for row, nm, v in marks:
                if nm == name:
                    assert got[sym, row] == v, (name, sym, row, got[sym, row], v)
    assert fired == 61 - len(UNSAT), f"{fired} of 61 recognisers fire (cdl2crows is unsatisfiable as written, pattern.rs:30-33)"


def test_patterns_fused_equals_single(pq, oracle, rich, data):
    from polars_quant_amd import api
    for d in (rich, data):
        o, h, l, c = (torch.from_numpy(d[k]).cuda() for k in ("open", "high", "low", "close"))
        allp = api.cdl_all(o, h, l, c)
        for name in pq.PATTERN_NAMES:
            exp = oracle.pattern(name, d["open"], d["high"], d["low"], d["close"])
            assert (allp[name].cpu().numpy() == exp).all(), name
    sub = api.cdl_all(o, h, l, c, names=["cdldoji", "cdlengulfing"], penetrations={"cdldoji": 0.1})
    assert sorted(sub) == ["cdldoji", "cdlengulfing"]
    for pen in (0.1, 0.5, 0.9):
        exp = oracle.pattern("cdlpiercing", rich["open"], rich["high"], rich["low"], rich["close"], penetration=pen)
        o, h, l, c = (torch.from_numpy(rich[k]).cuda() for k in ("open", "high", "low", "close"))
        assert (api.cdl("cdlpiercing", o, h, l, c, penetration=pen).cpu().numpy() == exp).all()


def test_backtest_vectorized(pq, oracle, data):
    from polars_quant_amd import api
    rng = np.random.default_rng(5)
    price = data["close"].copy()
    price[3, 50] = np.nan
    price[4, 60] = -1.0
    price[7, 10:20] = oracle.NULL
    buy = (rng.random(price.shape) < 0.05).astype(np.uint8)
    sell = (rng.random(price.shape) < 0.05).astype(np.uint8)
    bench = data["open"]
    for kw in (dict(), dict(buy_slippage=0.01, sell_slippage=0.02, position_size=0.5, min_commission=1.0),
               dict(initial_capital=50.0)):
        epos, ecash, eeq, es = oracle.backtest(np.where(np.isnan(price), np.nan, price), buy, sell, benchmark=bench, **kw)
        pos, cash, eq, s = api.backtest_vectorized(torch.from_numpy(price).cuda(), torch.from_numpy(buy).cuda(),
                                                   torch.from_numpy(sell).cuda(), benchmark=torch.from_numpy(bench).cuda(), **kw)
        for nm, g, e in (("position", pos, epos), ("cash", cash, ecash), ("equity", eq, eeq)):
            g = g.cpu().numpy()
            same = (bits(g) == bits(e)) | (np.isnan(g) & np.isnan(e))
            assert same.all(), f"{nm}: {np.sum(~same)} rows differ"
        s = s.cpu().numpy()
        ok = ~(np.isnan(es).any(axis=1))
        # exact columns: max_drawdown, max_profit, win_rate, total_trades (no transcendental involved)
        for k in (1, 5, 6, 7):
            assert (bits(s[ok, k]) == bits(es[ok, k])).all(), pq.SUMMARY_KEYS[k]
        for k in (0, 2, 3, 4):
            np.testing.assert_allclose(s[ok, k], es[ok, k], rtol=1e-12, atol=1e-13, err_msg=pq.SUMMARY_KEYS[k])
        assert np.isnan(s[~ok]).any(axis=1).all()


def test_backtest_macd_cross_fused(pq, oracle, data):
    from polars_quant_amd import api
    close = data["close"]
    ebuy, esell = oracle.macd_cross_signals(close)
    buy, sell = api.macd_cross_signals(torch.from_numpy(close).cuda())
    assert (buy.cpu().numpy() == ebuy).all() and (sell.cpu().numpy() == esell).all()
    assert ebuy.sum() > 0 and esell.sum() > 0
    epos, ecash, eeq, es = oracle.backtest(close, ebuy, esell)
    pos, cash, eq, s = api.backtest_macd_cross(torch.from_numpy(close).cuda())
    assert (bits(pos.cpu().numpy()) == bits(epos)).all()
    assert (bits(cash.cpu().numpy()) == bits(ecash)).all()
    assert (bits(eq.cpu().numpy()) == bits(eeq)).all()
    s = s.cpu().numpy()
    for k in (1, 5, 6, 7):
        assert (bits(s[:, k]) == bits(es[:, k])).all()
    np.testing.assert_allclose(s, es, rtol=1e-12, atol=1e-13)
    # summary-only mode (curves not materialised by the caller)
    _, _, _, s2 = api.backtest_macd_cross(torch.from_numpy(close).cuda(), want_curves=False)
    assert (bits(s2.cpu().numpy()) == bits(s)).all()


@pytest.mark.parametrize("T", [4097, 5040, 8192, 8200])
def test_backtest_leveraged_long_series(pq, oracle, T):
    """20-year daily series (5040 rows) and up to 8192 rows stay on the wave-per-symbol leveraged kernel (chunks of up to 128 rows
    in LDS); beyond that the lane-per-symbol form takes over.  Same results either way."""
    from polars_quant_amd import api
    rng = np.random.default_rng(T)
    d = oracle.gen_ohlcv(0x5EED0031, 9, T, 0)
    price = d["close"].copy()
    price[1, T - 50] = np.nan
    price[2, 4090:4100] = oracle.NULL
    buy = (rng.random(price.shape) < 0.02).astype(np.uint8)
    sell = (rng.random(price.shape) < 0.02).astype(np.uint8)
    bench = d["open"][0].copy()
    kw = dict(leverage=3.0, slippage=0.002, interest_rate=0.5, margin_call_threshold=0.6)
    e = oracle.backtest_leveraged(price, buy, sell, benchmark=bench, max_trades=64, **kw)
    g = api.backtest_leveraged(torch.from_numpy(price).cuda(), torch.from_numpy(buy).cuda(), torch.from_numpy(sell).cuda(),
                               benchmark=torch.from_numpy(bench).cuda(), max_trades=64, **kw)
    for k in ("cash", "stock_value", "total_value"):
        ge, ee = g[k].cpu().numpy(), e[k]
        assert ((bits(ge) == bits(ee)) | (np.isnan(ge) & np.isnan(ee))).all(), k
    assert (g["trade_count"].cpu().numpy() == e["trade_count"]).all() and e["trade_count"].sum() > 0
    s, es = g["summary"].cpu().numpy(), e["summary"]
    ok = ~np.isnan(es).any(axis=1)
    for k in (1, 5, 6, 7):
        assert (bits(s[ok, k]) == bits(es[ok, k])).all(), pq.SUMMARY_KEYS[k]
    np.testing.assert_allclose(s[ok], es[ok], rtol=1e-12, atol=1e-13)


def test_backtest_leveraged_matches_oracle(pq, oracle, data):
    """SURVEY 8(f) rank 1 (decision D-10): leveraged multi-symbol engine, bit-exact against the oracle."""
    from polars_quant_amd import api
    rng = np.random.default_rng(11)
    price = data["close"].copy()
    price[2, 40] = np.nan
    price[5, 70:75] = oracle.NULL
    price[6, 90] = -3.0
    buy = (rng.random(price.shape) < 0.04).astype(np.uint8)
    sell = (rng.random(price.shape) < 0.04).astype(np.uint8)
    bench = data["open"][0].copy()
    for kw in (dict(), dict(leverage=3.0, slippage=0.002, interest_rate=0.5, margin_call_threshold=0.6),
               dict(leverage=2.0, position_size=0.5, min_commission=50.0, initial_capital=5000.0)):
        e = oracle.backtest_leveraged(np.where(np.isnan(price), np.nan, price), buy, sell, benchmark=bench, max_trades=16, **kw)
        g = api.backtest_leveraged(torch.from_numpy(price).cuda(), torch.from_numpy(buy).cuda(), torch.from_numpy(sell).cuda(),
                                   benchmark=torch.from_numpy(bench).cuda(), max_trades=16, **kw)
        for k in ("cash", "stock_value", "total_value"):
            assert (bits(g[k].cpu().numpy()) == bits(e[k])).all(), k
        assert (g["trade_count"].cpu().numpy() == e["trade_count"]).all() and e["trade_count"].sum() > 0
        for k, v in e["trades"].items():
            gv = g["trades"][k].cpu().numpy()
            assert (gv == v).all() if v.dtype == np.int32 else (bits(gv) == bits(v)).all(), k
        s, es = g["summary"].cpu().numpy(), e["summary"]
        for k in (1, 5, 6, 7):
            assert (bits(s[:, k]) == bits(es[:, k])).all(), pq.SUMMARY_KEYS[k]
        np.testing.assert_allclose(s, es, rtol=1e-12, atol=1e-13)
        init_total = kw.get("initial_capital", 100000.0) * price.shape[0]
        em = oracle.portfolio_metrics(e["total_value"], init_total, bench)
        gm = api.portfolio_metrics(g["total_value"], init_total, torch.from_numpy(bench).cuda()).cpu().numpy()
        assert (bits(gm) == bits(em)).all()
    if any(kw.get("leverage", 1.0) > 1.0 for kw in (dict(leverage=3.0),)):
        e = oracle.backtest_leveraged(price, buy, sell, leverage=3.0, interest_rate=0.5, margin_call_threshold=0.6, max_trades=16)
        assert (e["trades"]["reason"] == 2).any(), "the margin-call path must be exercised"


def test_backtest_class_tables(pq, oracle, data):
    """README.md:416-477 table shapes of the `Backtest` class."""
    close = data["close"][:3, :200]
    ebuy, esell = oracle.macd_cross_signals(close)
    dates = [f"d{t:03d}" for t in range(close.shape[1])]
    syms = ["AAA", "BBB", "CCC"]
    wide = lambda a: {"date": dates, **{s: a[i] for i, s in enumerate(syms)}}
    bt = pq.Backtest(wide(close), wide(ebuy), wide(esell), leverage=2.0, benchmark={"date": dates, "IDX": data["open"][0, :200]})
    bt.run()
    daily = bt.get_daily_records()
    assert len(daily["symbol"]) == 3 * 200 and set(daily) == {"symbol", "date", "cash", "stock_value", "total_value"}
    e = oracle.backtest_leveraged(close, ebuy, esell, benchmark=data["open"][0, :200], leverage=2.0)
    assert (bits(np.asarray(daily["total_value"])) == bits(e["total_value"].reshape(-1))).all()
    pos = bt.get_position_records()
    assert len(pos["symbol"]) == int(e["trade_count"].sum()) and (np.asarray(pos["holding_days"]) > 0).all()
    one = bt.get_stock_positions("BBB")
    assert len(one["symbol"]) == int(e["trade_count"][1])
    perf = bt.get_performance_metrics()
    assert list(perf)[:3] == ["date", "portfolio_value", "daily_pnl"] and "beta" in perf
    assert "sharpe_ratio" in bt.get_stock_summary("AAA")


def test_signal_rules_and_strategies(pq, oracle, data):
    """SURVEY 8(f) rank 2 (decision D-11): rule kernels bit-exact, strategies = indicator + rule compositions."""
    from polars_quant_amd import api
    close, high, low = data["close"].copy(), data["high"], data["low"]
    close[1, 30:33] = oracle.NULL
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    (f,), (s_,) = oracle.call("sma", close, timeperiod=5), oracle.call("sma", close, timeperiod=12)
    eb, es = oracle.cross_signals(f, s_)
    gb, gs = api.cross_signals(dev(f), dev(s_))
    assert (gb.cpu().numpy() == eb).all() and (gs.cpu().numpy() == es).all() and eb.sum() > 0 and es.sum() > 0
    (r,) = oracle.call("rsi", data["close"], timeperiod=14)
    eb, es = oracle.band_signals(r, 40.0, 60.0)
    gb, gs = api.band_signals(dev(r), 40.0, 60.0)
    assert (gb.cpu().numpy() == eb).all() and (gs.cpu().numpy() == es).all() and eb.sum() > 0
    up, mid, lo = oracle.call("bbands", data["close"], timeperiod=10, nbdevup=1.0, nbdevdn=1.0)
    for mode in (0, 1):
        eb, es = oracle.channel_signals(data["close"], lo, up, mode)
        gb, gs = api.channel_signals(dev(data["close"]), dev(lo), dev(up), mode)
        assert (gb.cpu().numpy() == eb).all() and (gs.cpu().numpy() == es).all() and eb.sum() > 0, mode
    # strategies: the same compositions on both sides
    st = pq.Strategy()
    df = {k: dev(v) for k, v in data.items()}
    sig = st.ma(df, fast_period=5, slow_period=12)
    (f,), (s_,) = oracle.call("sma", data["close"], timeperiod=5), oracle.call("sma", data["close"], timeperiod=12)
    eb, es = oracle.cross_signals(f, s_)
    assert (sig["buy_signal"].cpu().numpy() == eb).all() and (sig["sell_signal"].cpu().numpy() == es).all()
    sig = st.macd(df)
    eb, es = oracle.macd_cross_signals(data["close"])
    assert (sig["buy_signal"].cpu().numpy() == eb).all() and (sig["sell_signal"].cpu().numpy() == es).all()
    sig = st.rsi(df, oversold=40.0, overbought=60.0)
    eb, es = oracle.band_signals(r, 40.0, 60.0)
    assert (sig["buy_signal"].cpu().numpy() == eb).all() and (sig["sell_signal"].cpu().numpy() == es).all()
    sig = st.bband(df, period=10, nbdev=1.0)
    eb, es = oracle.channel_signals(data["close"], lo, up, 0)
    assert (sig["buy_signal"].cpu().numpy() == eb).all() and (sig["sell_signal"].cpu().numpy() == es).all()
    sig = st.stoch(df)
    k, d = oracle.call("stoch", high, low, data["close"])
    eb, es = oracle.cross_signals(k, d)
    with np.errstate(invalid="ignore"):
        eb = eb & (k < 20.0); es = es & (k > 80.0)
    assert (sig["buy_signal"].cpu().numpy() == eb).all() and (sig["sell_signal"].cpu().numpy() == es).all()
    sig = st.cci(df)
    (c,) = oracle.call("cci", high, low, data["close"])
    eb, es = oracle.band_signals(c, -100.0, 100.0)
    assert (sig["buy_signal"].cpu().numpy() == eb).all() and (sig["sell_signal"].cpu().numpy() == es).all()
    # signals feed the leveraged engine end to end
    sig = st.ma(df, fast_period=5, slow_period=12, trend_period=30, trend_filter=True)
    out = api.backtest_leveraged(df["close"], sig["buy_signal"], sig["sell_signal"], leverage=2.0)
    assert int(out["trade_count"].sum()) > 0


def test_factor_ic_matches_oracle(pq, oracle, data):
    """SURVEY 8(f) rank 3 (decision D-12): IC, Rank-IC and rolling IC bit-exact against the oracle."""
    from polars_quant_amd import api
    rng = np.random.default_rng(3)
    for N, T in ((37, 50), (300, 131), (1, 5), (2, 3)):
        f = rng.normal(size=(N, T))
        r = 0.2 * f + rng.normal(size=(N, T))
        if N > 10:
            f[rng.random((N, T)) < 0.05] = oracle.NULL
            r[rng.random((N, T)) < 0.03] = np.nan
            f[:, 3] = np.round(f[:, 3] * 2.0)          # heavy ties
            r[:, 4] = np.round(r[:, 4])
            f[:, 5] = 2.5                                # zero variance
            f[1:, 6] = oracle.NULL                      # one pair only
            f[0, 7] = np.inf; r[1, 7] = -np.inf
        for method in (0, 1):
            eic, env = oracle.factor_ic(f, r, method)
            gic, gnv = api.factor_ic(torch.from_numpy(f).cuda(), torch.from_numpy(r).cuda(), method)
            gic, gnv = gic.cpu().numpy(), gnv.cpu().numpy()
            assert (gnv == env).all(), (N, T, method)
            same = (bits(gic) == bits(eic)) | (np.isnan(gic) & np.isnan(eic))
            assert same.all(), f"N={N} T={T} method={method}: {np.flatnonzero(~same)[:5]}"
            em, eir = oracle.rolling_ic(eic, 4)
            gm, gir = api.rolling_ic(torch.from_numpy(eic).cuda(), 4)
            for g, e in ((gm.cpu().numpy(), em), (gir.cpu().numpy(), eir)):
                assert ((bits(g) == bits(e)) | (np.isnan(g) & np.isnan(e))).all()
    fac = pq.Factor()
    out = fac.rolling_ic(torch.from_numpy(f).cuda(), torch.from_numpy(r).cuda(), window=2, rank=True)
    assert set(out) == {"rolling_ic", "rolling_ir"}


def test_randomised_parity_sweep(pq, oracle):
    """800 random (function, shape, parameters, null pattern) cases with a fixed seed (scripts/fuzz_parity.py, in-process).
    Shapes 1..139 x 1..259 (ragged tiles, series shorter than the windows), periods in {0, 1, .., T-1, T, T+1}, all matypes."""
    import importlib.util
    from pathlib import Path
    spec = importlib.util.spec_from_file_location("fuzz_parity", Path(__file__).resolve().parent.parent / "scripts" / "fuzz_parity.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    msgs = []
    n_bad = mod.sweep(21, 800, log=lambda *a: msgs.append(" ".join(str(x) for x in a)))
    assert n_bad == 0 and not msgs, "\n".join(msgs[:10])
    # the same with NaN VALUES (not NULLs) in half of the cases: they are inside every function's domain, and the reference's answer
    # is an artefact of its loops more often than a rule (monotonic deques that a NaN blocks: tests/test_oracle_nan.py)
    mod.NAN_RATE = 0.03
    n_bad = mod.sweep(22, 600, log=lambda *a: msgs.append(" ".join(str(x) for x in a)))
    assert n_bad == 0 and not msgs, "\n".join(msgs[:10])


@pytest.mark.parametrize("which, seed, iters", [("sweep_patterns", 41, 40), ("sweep_ragged", 42, 250), ("sweep_callers", 43, 200), ("sweep_suites", 44, 80)])
def test_randomised_parity_sweeps_beside_the_indicators(pq, oracle, which, seed, iters):
    """scripts/fuzz_parity.py: the 61 recognisers on quantised / flat / NaN-holed candles with random penetrations; every function on
    random ragged batches (nulls where the reference accepts them, NaNs everywhere); the leveraged engine, the signal rules, IC /
    Rank-IC with ties, NaN, inf and nulls, returns / rolling extrema, and the single-asset backtest on poisoned prices; random
    mixes of 3 .. 30 calls (+ the fused recognisers) recorded into one job grid on random shapes and row pitches, replayed twice."""
    import importlib.util
    from pathlib import Path
    spec = importlib.util.spec_from_file_location("fuzz_parity", Path(__file__).resolve().parent.parent / "scripts" / "fuzz_parity.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    msgs = []
    n_bad = getattr(mod, which)(seed, iters, log=lambda *a: msgs.append(" ".join(str(x) for x in a)))
    assert n_bad == 0 and not msgs, "\n".join(msgs[:10])


def test_randomised_parity_sweep_long_series(pq, oracle):
    """150 random long cases (scripts/fuzz_parity.py sweep_long): the one-symbol-per-wavefront indicator forms at 1024 .. 4096 rows with
    shortened warm-ups, flat stretches, NaNs and (where the reference accepts them) nulls; both backtests at 1 .. 8192 rows."""
    import importlib.util
    from pathlib import Path
    spec = importlib.util.spec_from_file_location("fuzz_parity", Path(__file__).resolve().parent.parent / "scripts" / "fuzz_parity.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    msgs = []
    n_bad = mod.sweep_long(31, 150, log=lambda *a: msgs.append(" ".join(str(x) for x in a)))
    assert n_bad == 0 and not msgs, "\n".join(msgs[:10])
    # every function (not only those with a wave form) at 1 024 .. 4 096 rows: nulls / NaNs that arrive long after a tiled body
    # has switched to its straight-line tiles, windows of up to 300 rows, all matypes
    n_bad = mod.sweep_long(32, 250, log=lambda *a: msgs.append(" ".join(str(x) for x in a)), names=sorted(mod.SPEC))
    assert n_bad == 0 and not msgs, "\n".join(msgs[:10])


def test_error_paths_leave_the_context_usable(pq, oracle, data):
    """Argument errors come back as status codes with a message (PqError), never as a launch; the context keeps working."""
    import ctypes as C
    from polars_quant_amd import api
    from polars_quant_amd._lib import Batch, LevParams, check, lib
    x = torch.from_numpy(data["close"]).cuda()
    n, TT = x.shape
    L, h = lib(), api.ctx(0)
    b = Batch(n, TT, TT)
    out = torch.empty_like(x)
    vp = lambda t: C.c_void_p(t.data_ptr())
    with pytest.raises(pq.PqError, match="null pointer"):
        check(L.pq_sma(h, C.byref(b), None, C.c_int64(5), vp(out)))
    with pytest.raises(pq.PqError):
        check(L.pq_sma(h, C.byref(Batch(n, TT, TT - 1)), vp(x), C.c_int64(5), vp(out)))      # stride < len
    ic = torch.empty(TT, dtype=torch.float64, device="cuda")
    with pytest.raises(pq.PqError, match="method"):
        check(L.pq_factor_ic(h, C.byref(b), vp(x), vp(x), 7, vp(ic), None))
    u8 = torch.zeros((n, TT), dtype=torch.uint8, device="cuda")
    with pytest.raises(pq.PqError, match="mode"):
        check(L.pq_channel_signals(h, C.byref(b), vp(x), vp(x), vp(x), 5, vp(u8), vp(u8)))
    prm = LevParams(100000.0, 1.0, 1.0, 0.3, 0.06, 0.0003, 5.0, 0.0)
    i32 = torch.zeros((n, 4), dtype=torch.int32, device="cuda")
    with pytest.raises(pq.PqError, match="all eight"):
        check(L.pq_backtest_leveraged(h, C.byref(b), vp(x), vp(u8), vp(u8), None, C.byref(prm), vp(out), vp(out), vp(out), 4, None,
                                      vp(i32), None, None, None, None, None, None, None, None))
    # a call that cannot be recorded fails inside a suite; after the abort the context computes again
    check(L.pq_suite_begin(h, C.byref(b)))
    with pytest.raises(pq.PqError, match="recorded"):
        check(L.pq_factor_ic(h, C.byref(b), vp(x), vp(x), 0, vp(ic), None))
    check(L.pq_suite_abort(h))
    with pytest.raises(TypeError):
        api.call("sma", x, timeperiod=5, bogus=1)
    (got,) = api.call("sma", x, timeperiod=5)
    (exp,) = oracle.call("sma", data["close"], timeperiod=5)
    assert_same("sma-after-errors", got.cpu().numpy(), exp)


def test_numpy_and_arrow_roundtrip(pq, oracle, data):
    import pyarrow as pa
    x = data["close"][0].copy()
    (exp,) = oracle.call("sma", x, timeperiod=4)
    got = pq.SMA(x, 4)
    assert isinstance(got, np.ndarray)
    assert_same("sma-numpy", got, exp)
    arr = pa.array(x, mask=np.arange(len(x)) % 17 == 3)
    xn = x.copy(); xn[np.arange(len(x)) % 17 == 3] = oracle.NULL
    (exp,) = oracle.call("ema", xn, timeperiod=4)
    got = pq.EMA(arr, 4)
    assert isinstance(got, pa.Array) and got.null_count == int(np.sum(bits(exp) == np.uint64(oracle.NULL_BITS)))
    vals = got.to_numpy(zero_copy_only=False)
    ok = ~np.asarray(got.is_null())
    assert (vals[ok] == exp[ok]).all()
    with pytest.raises(pq.NullsNotAllowed):
        pq.RSI(arr, 14)            # N-B family rejects nulls like the reference (momentum.rs:12-13)
    up, mid, lo = pq.BBANDS(x)
    assert up.shape == x.shape


def test_suite_replay_matches_direct_calls_and_oracle(pq, oracle, data):
    """pq_suite_*: the recorded job grid must reproduce the direct calls (and therefore the oracle) bit for bit"""
    _check_suite_replay(pq, oracle, data, None)


@pytest.mark.parametrize("pad", [3, 8, 16])
def test_suite_replay_padded_row_pitch(pq, oracle, data, pad):
    """The same with a row pitch larger than the row length (bench.py's layout: pitch = a multiple of 128 B): rows between
    the series are never read or written; an odd length under an even pitch runs the tiled bodies with a ragged tail."""
    T = data["close"].shape[1]
    stride = T + pad if (T + pad) % 2 == 0 else T + pad + 1
    if pad == 16: stride = (T + 15) // 16 * 16       # the 128-byte pitch itself
    _check_suite_replay(pq, oracle, data, stride)


def _check_suite_replay(pq, oracle, data, stride):
    from polars_quant_amd.suite import Suite
    N_SYM, T = data["close"].shape
    if stride is None:
        g = {k: torch.from_numpy(v).cuda() for k, v in data.items() if k in ("open", "high", "low", "close", "volume")}
    else:                                   # inputs re-housed with the padded pitch; the padding holds a poison value
        g = {}
        for k in ("open", "high", "low", "close", "volume"):
            buf = torch.full((N_SYM, stride), 1e300, dtype=torch.float64, device="cuda")
            buf[:, :T] = torch.from_numpy(data[k]).cuda()
            g[k] = buf[:, :T]
    st = Suite(N_SYM, T, "cuda", stride=stride, exact_layout=True)   # the caller's layout as it is (the 8-byte forms at an odd pitch)
    pitch = T if stride is None else stride
    st.record(g)
    info = st.info()
    # tiled path: the multi-output forms make ~24 single-phase jobs (the gather path -- ragged batches, windows whose rings exceed
    # 64 KB -- would run every composite as a chain through scratch)
    assert info["seq_jobs"] >= 24 and info["phases"] >= 1, info
    kernels = {gs["kernel"] for gs in st.grid_stats()}
    # an even row pitch must run the tiled bodies bench.py times; an odd one (rows only 8-byte aligned) their 8-byte form
    # seq_jobs_kernel<3> -- until round 4 it fell to the per-lane gather bodies seq_jobs_kernel<2>, 2.6 x slower at full size.
    # (no job of the suite needs the register-heavy kernel seq_jobs_kernel<1> since the Hilbert pipeline keeps its delay lines in LDS)
    aligned = "seq_jobs_kernel<0>"
    assert "seq_jobs_kernel<2>" not in kernels and (aligned if pitch % 2 == 0 else "seq_jobs_kernel<3>") in kernels, kernels
    for t in [x for ts in st.out.values() for x in ts] + list(st.pat.values()) + st.bt:
        t.fill_(-7)                      # poison: every row must be produced by the replay
    st.run()
    st.run()                             # replays are idempotent
    torch.cuda.synchronize()
    d = dict(data)
    d["periods"] = st.periods.cpu().numpy()
    for name in sorted(pq.SPEC):
        cols = pq.SPEC[name][0]
        exp = oracle.call(name, *[d[c] for c in cols])
        for (oname, _), got, e in zip(pq.SPEC[name][2], st.out[name], exp):
            assert_same(f"{name}.{oname}{{suite}}", got.cpu().numpy(), e, exact=name not in TRANSCENDENTAL, price=data["close"])
    for nm in pq.PATTERN_NAMES:
        exp = oracle.pattern(nm, data["open"], data["high"], data["low"], data["close"])
        assert (st.pat[nm].cpu().numpy() == exp).all(), nm
    ebuy, esell = oracle.macd_cross_signals(data["close"])
    epos, ecash, eeq, es = oracle.backtest(data["close"], ebuy, esell)
    assert (bits(st.bt[2].cpu().numpy()) == bits(eeq)).all()
    np.testing.assert_allclose(st.summary.cpu().numpy(), es, rtol=1e-12, atol=1e-13)
    st.close()


def test_fused_row_grid_slices_more_than_65535_series(pq, oracle):
    """The suite's fused ROW grid (row_jobs_kernel) and the tiled SEQ grids with more series than one grid dimension holds
    (grid.y <= 65535): 66 000 short series, a task list of ROW ops plus two sequential jobs, against the oracle."""
    from polars_quant_amd.suite import Suite
    N, T_ = 66000, 48
    d = oracle.gen_ohlcv(0x5EED0009, N, T_, 0)
    g = {k: torch.from_numpy(v).cuda() for k, v in d.items()}
    st = Suite(N, T_, "cuda")
    tasks = ["bop", "mom", "roc", "willr", "trange", "avgprice", "typprice", "aroon_all", "ht_trendmode", "sma", "rsi"]
    st.record(g, tasks)
    assert st.info()["row_launches"] >= 9
    st.run(); torch.cuda.synchronize()
    d = dict(d); d["real"] = d["close"]
    for name in ("bop", "mom", "roc", "willr", "trange", "avgprice", "typprice", "aroon", "aroonosc", "ht_trendmode", "sma", "rsi"):
        exp = oracle.call(name, *[d[c] for c in pq.SPEC[name][0]])
        for (oname, _), got, e in zip(pq.SPEC[name][2], st.out[name], exp):
            assert_same(f"wide:{name}.{oname}", got.cpu().numpy(), e, exact=True)
    st.close()


def _pitched(d, stride, dev="cuda"):
    """host [N, T] columns -> device columns with a row pitch of `stride` elements (the layout bench.py times)"""
    out = {}
    for k, v in d.items():
        buf = torch.zeros((v.shape[0], stride), dtype=torch.float64, device=dev)
        buf[:, : v.shape[1]] = torch.from_numpy(v).to(dev)
        out[k] = buf[:, : v.shape[1]]
    return out


@pytest.mark.parametrize("N", [5000, 1000], ids=["config3-5000", "config2-1000"])
def test_full_size_suite_all_symbols_parity_and_properties(pq, oracle, N):
    """BASELINE full size (config 2: 1000 symbols, configs 3 / bench: 5000 symbols, x 2520 days) through the bench's own Suite
    object ON THE BENCH'S LAYOUT (row pitch 2528 elements): (1) EVERY output of EVERY symbol against the oracle, in chunks of
    500 symbols on the host's cores; (2) size-independent properties: exact scaling by a power of two, equity = cash +
    position * price, replay idempotence."""
    from concurrent.futures import ThreadPoolExecutor
    from polars_quant_amd import api
    from polars_quant_amd.suite import Suite
    TT, STRIDE = 2520, 2528
    d = oracle.gen_ohlcv(0x5EED0002, N, TT, 0)
    g = _pitched(d, STRIDE)
    st = Suite(N, TT, "cuda", stride=STRIDE)
    st.record(g)
    st.run(); st.run()
    torch.cuda.synchronize()
    periods = st.periods[:1].cpu().numpy()
    CH = 500

    def expect(lo):
        """the oracle on symbols [lo, lo + CH): {key: array}; ctypes releases the GIL, so chunks run on separate cores"""
        sub = {k: np.ascontiguousarray(v[lo:lo + CH]) for k, v in d.items()}
        sub["periods"] = np.repeat(periods, sub["close"].shape[0], axis=0)
        sub["real"] = sub["close"]
        res = {}
        for name in pq.SPEC:
            for (oname, _), e in zip(pq.SPEC[name][2], oracle.call(name, *[sub[c] for c in pq.SPEC[name][0]])):
                res[(name, oname)] = e
        for nm in pq.PATTERN_NAMES:
            res[("pattern", nm)] = oracle.pattern(nm, sub["open"], sub["high"], sub["low"], sub["close"])
        ebuy, esell = oracle.macd_cross_signals(sub["close"])
        res["bt"] = oracle.backtest(sub["close"], ebuy, esell)
        return lo, sub, res

    compared = 0
    with ThreadPoolExecutor(max_workers=8) as pool:
        for lo, sub, res in pool.map(expect, range(0, N, CH)):
            hi = lo + sub["close"].shape[0]
            for name in pq.SPEC:
                for (oname, _), got in zip(pq.SPEC[name][2], st.out[name]):
                    assert_same(f"{name}.{oname}{{{lo}:{hi}}}", got[lo:hi].cpu().numpy(), res[(name, oname)],
                                exact=name not in TRANSCENDENTAL, price=sub["close"])
                    compared += 1
            for nm in pq.PATTERN_NAMES:
                assert (st.pat[nm][lo:hi].cpu().numpy() == res[("pattern", nm)]).all(), (nm, lo)
            epos, ecash, eeq, es = res["bt"]
            pos, cash, eq = (t[lo:hi].cpu().numpy() for t in st.bt)
            assert (bits(eq) == bits(eeq)).all() and (bits(pos) == bits(epos)).all() and (bits(cash) == bits(ecash)).all(), lo
            sm = st.summary[lo:hi].cpu().numpy()
            for k in (1, 5, 6, 7):   # max_drawdown, max_profit, win_rate, total_trades: exact
                assert (bits(sm[:, k]) == bits(es[:, k])).all(), (pq.SUMMARY_KEYS[k], lo)
            np.testing.assert_allclose(sm, es, rtol=1e-12, atol=1e-13)
    assert compared == (N + CH - 1) // CH * sum(len(v[2]) for v in pq.SPEC.values())
    # properties over ALL rows
    pos, cash, eq = (t.cpu().numpy() for t in st.bt)
    assert (bits(eq) == bits(cash + pos * d["close"])).all()                       # vectorized.rs:177
    (ema1,) = api.call("ema", g["close"], timeperiod=30)
    (ema2,) = api.call("ema", g["close"] * 2.0, timeperiod=30)
    a, b = ema1.cpu().numpy(), ema2.cpu().numpy()
    assert ((bits(a * 2.0) == bits(b)) | (np.isnan(a) & np.isnan(b))).all()         # every op of the recurrence scales exactly
    assert (bits(st.out["ema"][0].cpu().numpy()) == bits(a)).all()                  # suite replay == direct call
    st.close()


def test_fused_multi_output_calls(pq, oracle, data):
    """pq_dmi_all / pq_ht_all evaluate a shared core once; every column must equal the single-output function"""
    import ctypes as C
    from polars_quant_amd import api
    from polars_quant_amd._lib import Batch, check, lib
    g = {k: torch.from_numpy(data[k]).cuda() for k in ("high", "low", "close")}
    N_SYM, T = data["close"].shape
    b = Batch(N_SYM, T, T)
    outs = [torch.empty((N_SYM, T), dtype=torch.float64, device="cuda") for _ in range(5)]
    for p in (14, 1, 5):
        check(lib().pq_dmi_all(api.ctx(0), C.byref(b), *[C.c_void_p(g[k].data_ptr()) for k in ("high", "low", "close")], p,
                               *[C.c_void_p(t.data_ptr()) for t in outs]))
        for nm, t in zip(("dx", "plus_di", "minus_di", "adx", "adxr"), outs):
            (exp,) = oracle.call(nm, data["high"], data["low"], data["close"], timeperiod=p)
            assert_same(f"dmi_all.{nm}(p={p})", t.cpu().numpy(), exp)
    outs = [torch.empty((N_SYM, T), dtype=torch.float64, device="cuda") for _ in range(6)]
    check(lib().pq_ht_all(api.ctx(0), C.byref(b), C.c_void_p(g["close"].data_ptr()), *[C.c_void_p(t.data_ptr()) for t in outs]))
    exp = (oracle.call("ht_dcperiod", data["close"]) + oracle.call("ht_dcphase", data["close"]) +
           oracle.call("ht_phasor", data["close"]) + oracle.call("ht_sine", data["close"]))
    names = ("ht_dcperiod.ht_dcperiod", "ht_dcphase.ht_dcphase", "ht_phasor.inphase", "ht_phasor.quadrature", "ht_sine.sine", "ht_sine.leadsine")
    for nm_, t, e in zip(names, outs, exp):
        assert_same(nm_ + "{ht_all}", t.cpu().numpy(), e, exact=False, price=data["close"])
    # Fuse2 forms: every column equals the single function's
    g2 = {k: torch.from_numpy(data[k]).cuda() for k in ("open", "high", "low", "close", "volume")}
    P = lambda k: C.c_void_p(g2[k].data_ptr())
    mk = lambda n: [torch.empty((N_SYM, T), dtype=torch.float64, device="cuda") for _ in range(n)]
    V = lambda ts: [C.c_void_p(t.data_ptr()) for t in ts]
    cases = []
    for p in (30, 3, 1):
        o = mk(4); check(lib().pq_ema_all(api.ctx(0), C.byref(b), P("close"), p, *V(o)))
        cases.append((f"ema_all(p={p})", o, [oracle.call(n, data["close"], timeperiod=p)[0] for n in ("ema", "dema", "tema", "trix")]))
    o = mk(2); check(lib().pq_atr_all(api.ctx(0), C.byref(b), P("high"), P("low"), P("close"), 14, *V(o)))
    cases.append(("atr_all", o, [oracle.call(n, data["high"], data["low"], data["close"], timeperiod=14)[0] for n in ("atr", "natr")]))
    o = mk(2); check(lib().pq_dm_pair(api.ctx(0), C.byref(b), P("high"), P("low"), 14, *V(o)))
    cases.append(("dm_pair", o, [oracle.call(n, data["high"], data["low"], timeperiod=14)[0] for n in ("plus_dm", "minus_dm")]))
    o = mk(2); check(lib().pq_ad_all(api.ctx(0), C.byref(b), P("high"), P("low"), P("close"), P("volume"), 3, 10, *V(o)))
    cases.append(("ad_all", o, [oracle.call("ad", data["high"], data["low"], data["close"], data["volume"])[0],
                                oracle.call("adosc", data["high"], data["low"], data["close"], data["volume"], fastperiod=3, slowperiod=10)[0]]))
    o = mk(6); check(lib().pq_macd_pair(api.ctx(0), C.byref(b), P("close"), 5, 13, 4, 9, *V(o)))
    cases.append(("macd_pair", o, list(oracle.call("macd", data["close"], fastperiod=5, slowperiod=13, signalperiod=4)) +
                  list(oracle.call("macdfix", data["close"], signalperiod=9))))
    for mt in (0, 1):
        o = mk(2); check(lib().pq_apo_ppo(api.ctx(0), C.byref(b), P("close"), 12, 26, mt, *V(o)))
        cases.append((f"apo_ppo(mt={mt})", o, [oracle.call(n, data["close"], fastperiod=12, slowperiod=26, matype=mt)[0] for n in ("apo", "ppo")]))
    o = mk(2); check(lib().pq_sar_pair(api.ctx(0), C.byref(b), P("high"), P("low"), C.c_double(0.02), C.c_double(0.2), C.c_double(0.0),
                                         C.c_double(0.01), C.c_double(0.02), C.c_double(0.02), C.c_double(0.2), C.c_double(0.03),
                                         C.c_double(0.03), C.c_double(0.3), *V(o)))
    cases.append(("sar_pair", o, [oracle.call("sar", data["high"], data["low"], acceleration=0.02, maximum=0.2)[0],
                                  oracle.call("sarext", data["high"], data["low"], startvalue=0.0, offsetonreverse=0.01, accelerationinitlong=0.02,
                                              accelerationlong=0.02, accelerationmaxlong=0.2, accelerationinitshort=0.03,
                                              accelerationshort=0.03, accelerationmaxshort=0.3)[0]]))
    o = mk(7); check(lib().pq_dm_system_all(api.ctx(0), C.byref(b), P("high"), P("low"), P("close"), 14, *V(o)))
    cases.append(("dm_system_all", o, [oracle.call(n, data["high"], data["low"], data["close"], timeperiod=14)[0]
                                       for n in ("dx", "plus_di", "minus_di", "adx", "adxr", "atr", "natr")]))
    o = mk(2); check(lib().pq_cmo_rsi(api.ctx(0), C.byref(b), P("close"), 14, *V(o)))
    cases.append(("cmo_rsi", o, [oracle.call(n, data["close"], timeperiod=14)[0] for n in ("cmo", "rsi")]))
    for p_ in (30, 7):
        o = mk(2); check(lib().pq_sma_ma(api.ctx(0), C.byref(b), P("close"), p_, *V(o)))
        cases.append((f"sma_ma({p_})", o, [oracle.call("sma", data["close"], timeperiod=p_)[0], oracle.call("ma", data["close"], timeperiod=p_, matype=0)[0]]))
    o = mk(4); check(lib().pq_volume_all(api.ctx(0), C.byref(b), P("high"), P("low"), P("close"), P("volume"), 14, 3, 10, *V(o)))
    hlcv = [data[k] for k in ("high", "low", "close", "volume")]
    cases.append(("volume_all", o, [oracle.call("mfi", *hlcv, timeperiod=14)[0], oracle.call("ad", *hlcv)[0],
                                    oracle.call("adosc", *hlcv, fastperiod=3, slowperiod=10)[0], oracle.call("obv", data["close"], data["volume"])[0]]))
    for fk, sk, skm, sd, sdm, fd, fdm in ((5, 3, 0, 3, 0, 3, 0), (14, 3, 1, 5, 0, 4, 1)):
        o = mk(4); check(lib().pq_stoch_all(api.ctx(0), C.byref(b), P("high"), P("low"), P("close"), fk, sk, skm, sd, sdm, fd, fdm, *V(o)))
        cases.append((f"stoch_all({fk},{sk},{skm},{sd},{sdm},{fd},{fdm})", o,
                      list(oracle.call("stoch", data["high"], data["low"], data["close"], fastk_period=fk, slowk_period=sk, slowk_matype=skm,
                                       slowd_period=sd, slowd_matype=sdm)) +
                      list(oracle.call("stochf", data["high"], data["low"], data["close"], fastk_period=fk, fastd_period=fd, fastd_matype=fdm))))
    for label, got, exp in cases:
        for i, (t_, e) in enumerate(zip(got, exp)):
            assert_same(f"{label}[{i}]", t_.cpu().numpy(), e)
    outs = [torch.empty((N_SYM, T), dtype=torch.float64, device="cuda") for _ in range(3)]
    for p in (14, 3, 0):
        check(lib().pq_aroon_all(api.ctx(0), C.byref(b), C.c_void_p(g["high"].data_ptr()), C.c_void_p(g["low"].data_ptr()), p,
                                 *[C.c_void_p(t.data_ptr()) for t in outs]))
        exp = oracle.call("aroon", data["high"], data["low"], timeperiod=p) + oracle.call("aroonosc", data["high"], data["low"], timeperiod=p)
        for i, (t, e) in enumerate(zip(outs, exp)):
            assert_same(f"aroon_all[{i}](p={p})", t.cpu().numpy(), e)


def test_recorded_suite_nulls_and_parameters_on_the_tiled_bodies(pq, oracle):
    """A recorded job grid on an EVEN row pitch (seq_jobs_kernel<0>/<1>: the two-wave tiled bodies with rings, steady-state
    fast paths, barrier hand-off and epilogues that bench.py times) with null-bearing inputs and non-default parameters,
    against the oracle.  N-B functions get the null-free copy (the reference rejects nulls there)."""
    import ctypes as C
    from polars_quant_amd import api
    from polars_quant_amd._lib import Batch, check, lib
    n, T_ = 200, 344
    clean = oracle.gen_ohlcv(SEED + 5, n, T_, 0)
    clean["real"] = clean["close"]
    rng = np.random.default_rng(3)
    holes = {}
    for k, v in clean.items():
        a = v.copy()
        m = rng.random(a.shape) < 0.01
        m[:, :2] = True            # leading nulls
        m[4] = False               # one null-free series
        m[9, 150:] = True          # one series that goes null for good
        m[:, 200:260] = False      # a null-free stretch: tiles that take the fast path again after the general one
        a[m] = oracle.NULL
        holes[k] = a
    cases = [("sma", dict(timeperiod=7)), ("sma", dict(timeperiod=40)), ("ema", dict(timeperiod=5)), ("bbands", dict(timeperiod=9, nbdevup=1.0, nbdevdn=3.0)),
             ("dema", dict(timeperiod=4)), ("tema", dict(timeperiod=6)), ("t3", dict(timeperiod=3, vfactor=0.5)), ("wma", dict(timeperiod=11)),
             ("kama", dict(timeperiod=6)), ("trima", dict(timeperiod=9)), ("trima", dict(timeperiod=20)), ("midpoint", dict(timeperiod=5)),
             ("midprice", dict(timeperiod=9)), ("ma", dict(timeperiod=8, matype=1)), ("apo", dict(fastperiod=4, slowperiod=9, matype=0)),
             ("ppo", dict(fastperiod=4, slowperiod=9, matype=1)), ("macdext", dict(fastperiod=5, fastmatype=0, slowperiod=11, slowmatype=0, signalperiod=8, signalmatype=0)),
             ("macdext", dict(fastperiod=5, fastmatype=1, slowperiod=11, slowmatype=0, signalperiod=4, signalmatype=1)),
             ("stoch", dict(fastk_period=6, slowk_period=3, slowk_matype=0, slowd_period=4, slowd_matype=1)),
             ("stochf", dict(fastk_period=4, fastd_period=3, fastd_matype=0)), ("atr", dict(timeperiod=6)), ("natr", dict(timeperiod=9)),
             ("ad", {}), ("adosc", dict(fastperiod=2, slowperiod=7)), ("obv", {}), ("sar", dict(acceleration=0.02, maximum=0.2)),
             ("mavp", dict(minperiod=3, maxperiod=25, matype=0)), ("mama", dict(fastlimit=0.5, slowlimit=0.05)),
             # N-B family: clean inputs
             ("rsi", dict(timeperiod=9)), ("cmo", dict(timeperiod=6)), ("macd", dict(fastperiod=4, slowperiod=10, signalperiod=5)),
             ("trix", dict(timeperiod=4)), ("ultosc", dict(timeperiod1=3, timeperiod2=6, timeperiod3=12)), ("mfi", dict(timeperiod=9)),
             ("cci", dict(timeperiod=9)), ("cci", dict(timeperiod=21)), ("adx", dict(timeperiod=8)), ("adxr", dict(timeperiod=6)),
             ("dx", dict(timeperiod=5)), ("minus_di", dict(timeperiod=7)), ("plus_dm", dict(timeperiod=5)), ("stochrsi", dict(timeperiod=9, fastk_period=5, fastd_period=3, fastd_matype=0)),
             ("ht_dcperiod", {}), ("ht_sine", {})]
    clean["periods"] = holes["periods"] = rng.integers(0, 40, size=(n, T_)).astype(np.float64)
    dev = {id(d): {k: torch.from_numpy(v).cuda() for k, v in d.items()} for d in (clean, holes)}
    L, h, b = lib(), api.ctx(0), Batch(n, T_, T_)
    check(L.pq_suite_begin(h, C.byref(b)))
    recorded = []
    try:
        for name, prm in cases:
            src = clean if pq.SPEC[name][3] == "N-B" else holes
            outs = api.call(name, *[dev[id(src)][c] for c in pq.SPEC[name][0]], **prm)   # recorded, not launched
            recorded.append((name, prm, src, outs))
    except Exception:
        L.pq_suite_abort(h)
        raise
    suite = C.c_void_p()
    check(L.pq_suite_end(h, C.byref(suite)))
    try:
        check(L.pq_suite_run(h, suite))
        check(L.pq_suite_run(h, suite))
        torch.cuda.synchronize()
        k, kernels = 0, set()
        while True:
            var = C.c_int32()
            if L.pq_suite_grid_variant(suite, k, C.byref(var)) != 0:
                break
            kernels.add(var.value); k += 1
        assert 0 in kernels and 2 not in kernels, f"expected only the tiled job kernels, got variants {kernels}"
        for name, prm, src, outs in recorded:
            exp = oracle.call(name, *[src[c] for c in pq.SPEC[name][0]], **prm)
            for (oname, _), g, e in zip(pq.SPEC[name][2], outs, exp):
                assert_same(f"{name}.{oname}{{recorded {prm}}}", g.cpu().numpy(), e, exact=name not in TRANSCENDENTAL, price=src["close"])
    finally:
        check(L.pq_suite_destroy(h, suite))


@pytest.mark.parametrize("shape", [(70, 300, 300), (9, 127, 127), (130, 2520, 2528), (66, 1024, 1024), (5, 1501, 1501), (64, 64, 80)],
                         ids=lambda s: f"{s[0]}x{s[1]}p{s[2]}")
def test_nan_values_in_rolling_extrema(pq, oracle, shape):
    """NaN VALUES (not NULLs) in MIDPOINT / MIDPRICE follow the reference's monotonic deques (a NaN is never popped and shields
    everything older: tests/test_oracle_nan.py has the hand-derived cases), on every body that computes them: gather, tiled + its
    straight-line tiles, the one-symbol-per-wavefront form (which hands such a symbol to the lane-per-symbol kernel), MIDPRICE's
    row-parallel launch, and the recorded job grid."""
    import ctypes as C
    from polars_quant_amd import api
    from polars_quant_amd._lib import Batch, check, lib
    N, TT, PITCH = shape
    rng = np.random.default_rng(N * 7919 + TT)
    d = oracle.gen_ohlcv(0x5EED0041, N, TT, 0)
    d = {k: v.copy() for k, v in d.items()}
    for s in range(N):
        r = s % 8
        cols = ("high", "low", "close")
        if r == 0:                      # a NaN now and then
            for k in cols: d[k][s, rng.random(TT) < 0.01] = np.nan
        elif r == 1:                    # the very first value
            for k in cols: d[k][s, 0] = np.nan
        elif r == 2 and TT > 40:        # a run of NaNs, then one alone within the next window
            for k in cols: d[k][s, 20:23] = np.nan; d[k][s, 30] = np.nan
        elif r == 3:                    # the last row
            for k in cols: d[k][s, TT - 1] = np.nan
        elif r == 4 and TT > 50:        # NaNs and NULLs mixed (MIDPOINT skips NULL rows, a NULL in either MIDPRICE input gives a NULL row)
            d["close"][s, 10] = np.nan; d["close"][s, 12] = oracle.NULL; d["close"][s, 40] = np.nan
            d["high"][s, 15] = np.nan; d["low"][s, 17] = oracle.NULL
        elif r == 5:                    # only one of the two MIDPRICE inputs
            d["high"][s, TT // 2] = np.nan
    def dev(a):
        buf = torch.full((N, PITCH), 1e300, dtype=torch.float64, device="cuda")
        buf[:, :TT] = torch.from_numpy(np.ascontiguousarray(a)).cuda()
        return buf[:, :TT]
    g = {k: dev(v) for k, v in d.items()}
    periods = [1, 2, 3, 14, 30, min(100, TT), TT + 3]
    def check_all(tag, call):
        for p_ in periods:
            (got,) = call("midpoint", g["close"], timeperiod=p_)
            (exp,) = oracle.call("midpoint", d["close"], timeperiod=p_)
            assert_same(f"{tag} midpoint p={p_}", got.cpu().numpy(), exp, price=d["close"])
            (got,) = call("midprice", g["high"], g["low"], timeperiod=p_)
            (exp,) = oracle.call("midprice", d["high"], d["low"], timeperiod=p_)
            assert_same(f"{tag} midprice p={p_}", got.cpu().numpy(), exp, price=d["close"])
    check_all("direct", api.call)
    for env in ("PQ_NO_WT", "PQ_MIDPRICE_SEQ"):   # the lane-per-symbol bodies without the wave form in front / instead of the row-parallel launch
        os.environ[env] = "1"
        try:
            check_all(env, api.call)
        finally:
            del os.environ[env]
    # the same calls recorded into one job grid
    L, h, b = lib(), api.ctx(0), Batch(N, TT, PITCH)
    check(L.pq_suite_begin(h, C.byref(b)))
    rec = []
    try:
        for p_ in periods[:5]:
            rec.append(("midpoint", p_, api.call("midpoint", g["close"], timeperiod=p_)[0]))
            rec.append(("midprice", p_, api.call("midprice", g["high"], g["low"], timeperiod=p_)[0]))
    except Exception:
        L.pq_suite_abort(h)
        raise
    suite = C.c_void_p()
    check(L.pq_suite_end(h, C.byref(suite)))
    try:
        check(L.pq_suite_run(h, suite)); check(L.pq_suite_run(h, suite))
        torch.cuda.synchronize()
        for name, p_, out in rec:
            exp = oracle.call(name, *([d["close"]] if name == "midpoint" else [d["high"], d["low"]]), timeperiod=p_)[0]
            assert_same(f"recorded {name} p={p_}", out.cpu().numpy(), exp, price=d["close"])
    finally:
        check(L.pq_suite_destroy(h, suite))


def test_full_size_config4_factor_ic(pq, oracle):
    """BASELINE config 4 at full size (10 000 symbols x 5 040 days): IC and Rank-IC of every day on the GPU; a sample of days
    against the oracle (days are independent cross-sections), rolling IC of the whole series, and the n_valid census."""
    from polars_quant_amd import api
    N, TT = 10000, 5040
    rng = np.random.default_rng(4)
    f = rng.standard_normal((N, TT))
    r = 0.05 * f + rng.standard_normal((N, TT))
    f[rng.random((N, TT)) < 0.02] = oracle.NULL
    r[:, -1] = oracle.NULL                       # forward return of the last day does not exist
    r[rng.random((N, TT)) < 0.01] = np.nan
    fg, rg = torch.from_numpy(f).cuda(), torch.from_numpy(r).cuda()
    days = np.array([0, 1, 7, 63, 64, 1000, 2519, 2520, 4000, 5038, 5039])
    fs, rs = np.ascontiguousarray(f[:, days]), np.ascontiguousarray(r[:, days])
    ics = {}
    for method in (0, 1):
        ic, nv = api.factor_ic(fg, rg, method=method)
        ic, nv = ic.cpu().numpy(), nv.cpu().numpy()
        eic, env = oracle.factor_ic(fs, rs, method=method)
        assert (nv[days] == env).all()
        g = ic[days]
        assert ((bits(g) == bits(eic)) | (np.isnan(g) & np.isnan(eic))).all(), f"method {method}: {g} vs {eic}"
        assert np.isnan(ic[-1]) and nv[-1] == 0
        valid = (bits(f) != np.uint64(oracle.NULL_BITS)) & np.isfinite(f) & (bits(r) != np.uint64(oracle.NULL_BITS)) & np.isfinite(r)
        assert (nv == valid.sum(axis=0)).all()
        assert np.nanmax(np.abs(ic[:-1])) <= 1.0 + 1e-12 and abs(np.nanmean(ic[:-1]) - 0.05) < 0.01
        ics[method] = ic
    em, eir = oracle.rolling_ic(ics[0], 20)
    gm, gir = api.rolling_ic(torch.from_numpy(ics[0]).cuda(), 20)
    for g, e in ((gm.cpu().numpy(), em), (gir.cpu().numpy(), eir)):
        assert ((bits(g) == bits(e)) | (np.isnan(g) & np.isnan(e))).all()


def test_full_size_config5_leveraged_backtest(pq, oracle):
    """BASELINE config 5 at full size (5 000 x 2 520, leverage 2, commission + slippage): EVERY symbol against the oracle
    (capital pools are independent), total = cash_net + stock_value on all 12.6 M rows, and the portfolio table.  (Dense
    layout: this engine is not part of the timed step.)"""
    from polars_quant_amd import api
    N, TT = 5000, 2520
    d = oracle.gen_ohlcv(0x5EED0005, N, TT, 0)
    close = d["close"]
    cg = torch.from_numpy(close).cuda()
    buy, sell = api.macd_cross_signals(cg)
    kw = dict(leverage=2.0, slippage=0.001, commission_rate=0.0005, interest_rate=0.08, margin_call_threshold=0.5)
    bench = d["open"][0].copy()
    g = api.backtest_leveraged(cg, buy, sell, benchmark=torch.from_numpy(bench).cuda(), max_trades=32, **kw)
    torch.cuda.synchronize()
    ebuy, esell = oracle.macd_cross_signals(close)
    assert (buy.cpu().numpy() == ebuy).all() and (sell.cpu().numpy() == esell).all()
    e = oracle.backtest_leveraged(close, ebuy, esell, benchmark=bench, max_trades=32, **kw)
    for k in ("cash", "stock_value", "total_value"):
        assert (bits(g[k].cpu().numpy()) == bits(e[k])).all(), k
    assert (g["trade_count"].cpu().numpy() == e["trade_count"]).all() and e["trade_count"].sum() > 0
    for k, v in e["trades"].items():
        gv = g["trades"][k].cpu().numpy()
        assert (gv == v).all() if v.dtype == np.int32 else (bits(gv) == bits(v)).all(), k
    np.testing.assert_allclose(g["summary"].cpu().numpy(), e["summary"], rtol=1e-12, atol=1e-13)
    cash, sv, tv = (g[k].cpu().numpy() for k in ("cash", "stock_value", "total_value"))
    assert (bits(tv) == bits(cash + sv)).all()                       # D-10: total_value = (cash - debt) + stock_value, every row
    assert (sv >= 0).all() and np.isfinite(tv).all()
    pm = api.portfolio_metrics(g["total_value"], 100000.0 * N, torch.from_numpy(bench).cuda()).cpu().numpy()
    assert pm.shape == (TT, 10) and np.isfinite(pm[:, 0]).all()
    np.testing.assert_allclose(pm[:, 0], tv.sum(axis=0), rtol=1e-12)  # portfolio_value = sum over symbols (blocked order)


def test_rank_ic_discrete_factor_long_tie_runs(pq, oracle):
    """Rank-IC with a discrete factor (signals in {-1, 0, 1}) and coarsely rounded returns: tie runs of ~1000 entries per day,
    found by binary search instead of a linear walk (which would be O(n^2) loads per day)."""
    from polars_quant_amd import api
    rng = np.random.default_rng(9)
    N, T_ = 3000, 24
    f = rng.integers(-1, 2, size=(N, T_)).astype(np.float64)
    r = np.round(0.5 * f + rng.normal(size=(N, T_)), 1)
    f[:, 0] = 0.0                                    # a constant day: zero variance -> null
    f[rng.random((N, T_)) < 0.02] = oracle.NULL
    eic, env = oracle.factor_ic(f, r, method=1)
    ic, nv = api.factor_ic(torch.from_numpy(f).cuda(), torch.from_numpy(r).cuda(), method=1)
    g = ic.cpu().numpy()
    assert (nv.cpu().numpy() == env).all()
    assert ((bits(g) == bits(eic)) | (np.isnan(g) & np.isnan(eic))).all(), (g, eic)
    assert bits(g)[0] == np.uint64(oracle.NULL_BITS) and np.isfinite(g[1:]).all()


@pytest.mark.parametrize("N,T_", [(16, 9), (17, 9), (255, 7), (1024, 6), (1025, 6), (4097, 5), (10000, 4), (16384, 4), (16385, 3), (20000, 3)])
def test_rank_ic_cross_section_sizes(pq, oracle, N, T_):
    """Rank-IC on both sides of every size boundary of the per-day LDS sort (P = 16 .. 16384 keys, 64 .. 1024 threads) and beyond
    it (N > 16384: the segmented-sort path).  Day 0: every key equal (one tie run of all N); day 1: two-valued factor; day 2: nulls."""
    from polars_quant_amd import api
    rng = np.random.default_rng(1000 + N)
    f = rng.normal(size=(N, T_))
    r = 0.3 * f + rng.normal(size=(N, T_))
    f[:, 0] = 1.25
    f[:, 1] = (rng.random(N) < 0.5).astype(np.float64)
    r[:, 1] = np.round(r[:, 1], 1)
    f[rng.random(N) < 0.1, 2] = oracle.NULL
    r[rng.random(N) < 0.1, 2] = np.nan
    u = rng.random(N)
    r[:, -1] = np.where(u < 0.3, 0.0, np.where(u < 0.6, -0.0, r[:, -1]))      # +0.0 and -0.0 are one tie run
    eic, env = oracle.factor_ic(f, r, method=1)
    ic, nv = api.factor_ic(torch.from_numpy(f).cuda(), torch.from_numpy(r).cuda(), method=1)
    g = ic.cpu().numpy()
    assert (nv.cpu().numpy() == env).all()
    assert ((bits(g) == bits(eic)) | (np.isnan(g) & np.isnan(eic))).all(), (N, g, eic)
    assert np.isfinite(g[1:]).all()


def test_remaining_readme_strategies(pq, oracle, rich):
    """The README names without documented parameters (README.md:946-953; decision D-11b): each strategy against the same
    composition built on the host from the ORACLE's indicator columns and signal rules (numpy for the element-wise glue)."""
    from polars_quant_amd import api
    d = rich
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    df = {k: dev(v) for k, v in d.items()}
    st = pq.Strategy()
    NULLB = np.uint64(oracle.NULL_BITS)
    isn = lambda a: bits(a) == NULLB
    chk = lambda sig, eb, es, nm: (np.testing.assert_array_equal(sig["buy_signal"].cpu().numpy().astype(bool), eb.astype(bool), err_msg=nm + ".buy"),
                                   np.testing.assert_array_equal(sig["sell_signal"].cpu().numpy().astype(bool), es.astype(bool), err_msg=nm + ".sell"))
    h, l, c, o, v = d["high"], d["low"], d["close"], d["open"], d["volume"]
    with np.errstate(invalid="ignore"):
        # adx
        (pdm,), (mdm,), (adx,) = oracle.call("plus_dm", h, l, timeperiod=7), oracle.call("minus_dm", h, l, timeperiod=7), oracle.call("adx", h, l, c, timeperiod=7)
        eb, es = oracle.cross_signals(pdm, mdm)
        strong = np.where(isn(adx), False, adx > 20.0)
        chk(st.adx(df, period=7, threshold=20.0), eb & strong, es & strong, "adx")
        assert (eb & strong).sum() > 0
        # breakout (rolling extrema kernels + rule channel mode 1)
        (hi,), (lo,) = oracle.call("rolling_max", h, window=10), oracle.call("rolling_min", l, window=10)
        (ghi,) = api.call("rolling_max", df["high"], window=10)
        assert (bits(ghi.cpu().numpy()) == bits(hi)).all()
        eb, es = oracle.channel_signals(c, lo, hi, 1)
        chk(st.breakout(df, period=10), eb, es, "breakout")
        assert eb.sum() > 0 and es.sum() > 0
        # reversion: z = (x - mid) / (up - mid) with nbdev 1
        up, mid, _ = oracle.call("bbands", c, timeperiod=10, nbdevup=1.0, nbdevdn=1.0)
        z = (c - mid) / (up - mid)
        z[isn(up)] = oracle.NULL
        eb, es = oracle.band_signals(z, -1.5, 1.5)
        chk(st.reversion(df, period=10, threshold=1.5), eb, es, "reversion")
        assert eb.sum() > 0
        # volume
        (sv,) = oracle.call("sma", v, timeperiod=10)
        surge = np.where(isn(sv), False, v > 1.2 * sv)
        upd = np.zeros_like(surge); dnd = np.zeros_like(surge)
        upd[:, 1:] = c[:, 1:] > c[:, :-1]; dnd[:, 1:] = c[:, 1:] < c[:, :-1]
        chk(st.volume(df, period=10, multiplier=1.2), surge & upd, surge & dnd, "volume")
        assert (surge & upd).sum() > 0
        # grid
        (base,) = oracle.call("sma", c, timeperiod=10)
        glo, ghi_ = np.where(isn(base), base, base * (1.0 - 0.02)), np.where(isn(base), base, base * (1.0 + 0.02))
        glo[isn(base)] = oracle.NULL; ghi_[isn(base)] = oracle.NULL
        eb, es = oracle.channel_signals(c, glo, ghi_, 0)
        chk(st.grid(df, base_period=10, grid_pct=2.0), eb, es, "grid")
        assert eb.sum() > 0
        # gap
        eb = np.zeros(c.shape, bool); es = np.zeros(c.shape, bool)
        eb[:, 1:] = o[:, 1:] > h[:, :-1] * (1.0 + 0.5 / 100.0); es[:, 1:] = o[:, 1:] < l[:, :-1] * (1.0 - 0.5 / 100.0)
        chk(st.gap(df, gap_pct=0.5), eb, es, "gap")
        # pattern
        bull, bear = ("cdlhammer", "cdlengulfing", "cdlpiercing"), ("cdlhangingman", "cdlengulfing", "cdldarkcloudcover")
        eb = np.any([oracle.pattern(n, o, h, l, c) == 100 for n in bull], axis=0)
        es = np.any([oracle.pattern(n, o, h, l, c) == -100 for n in bear], axis=0)
        chk(st.pattern(df, bullish=bull, bearish=bear), eb, es, "pattern")
        assert eb.sum() > 0 and es.sum() > 0
        # trend
        mas = [oracle.call("sma", c, timeperiod=p)[0] for p in (3, 6, 12)]
        ok = ~np.any([isn(m) for m in mas], axis=0)
        bullm = ok & (mas[0] > mas[1]) & (mas[1] > mas[2]); bearm = ok & (mas[0] < mas[1]) & (mas[1] < mas[2])
        first = lambda m: np.concatenate([np.zeros_like(m[:, :1]), m[:, 1:] & ~m[:, :-1]], axis=1)
        chk(st.trend(df, periods=(3, 6, 12)), first(bullm), first(bearm), "trend")
        assert first(bullm).sum() > 0
        # ma with the README's slope / distance filters
        (f,), (s_,) = oracle.call("ema", c, timeperiod=4), oracle.call("ema", c, timeperiod=9)
        eb, es = oracle.cross_signals(f, s_)
        rising = np.zeros(c.shape, bool); rising[:, 1:] = np.where(isn(s_[:, 1:]) | isn(s_[:, :-1]), False, s_[:, 1:] > s_[:, :-1])
        far = np.where(isn(s_), False, (c - s_) > np.abs(s_) * 0.001)
        chk(st.ma(df, fast_period=4, slow_period=9, ma_type="ema", slope_filter=True, distance_pct=0.1), eb & rising & far, es, "ma+filters")


def test_backtest_get_stock_performance(pq, oracle):
    """README.md:554-589: per-symbol daily performance table of the `Backtest` class"""
    d = oracle.gen_ohlcv(0x5EED000C, 3, 160, 0)
    close = d["close"]
    ebuy, esell = oracle.macd_cross_signals(close)
    dates = [f"d{t:03d}" for t in range(close.shape[1])]
    syms = ["AAA", "BBB", "CCC"]
    wide = lambda a: {"date": dates, **{s: a[i] for i, s in enumerate(syms)}}
    bench = d["open"][0]
    bt = pq.Backtest(wide(close), wide(ebuy), wide(esell), leverage=2.0, benchmark={"date": dates, "IDX": bench})
    bt.run()
    perf = bt.get_stock_performance("BBB")
    assert list(perf) == ["symbol", "date", "stock_value", "daily_pnl", "daily_return_pct", "cumulative_pnl", "cumulative_return_pct",
                          "benchmark_return_pct", "alpha_pct", "relative_return_pct"]
    e = oracle.backtest_leveraged(close, ebuy, esell, benchmark=bench, leverage=2.0)
    em = oracle.portfolio_metrics(e["total_value"][1:2], 100000.0, bench)
    assert (bits(np.asarray(perf["stock_value"])) == bits(em[:, 0])).all()
    assert (bits(np.asarray(perf["cumulative_return_pct"])) == bits(em[:, 4])).all()
    assert (bits(np.asarray(perf["alpha_pct"])) == bits(em[:, 6])).all()
    assert set(bt.get_stock_daily("AAA")) == {"symbol", "date", "cash", "stock_value", "total_value"}
    bt2 = pq.Backtest(wide(close), wide(ebuy), wide(esell))
    bt2.run()
    assert "alpha_pct" not in bt2.get_stock_performance("AAA")


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_recorded_suite_random_task_subsets(pq, oracle, seed):
    """The launch plan of a recorded suite depends on WHAT was recorded (which chains exist, whether the backtest and a fused ROW grid
    share the tail of the SHORT chain, whether a ROW launch is hoisted behind its producers): random subsets of the benchmark suite's
    tasks in random order, on the tiled bodies, every produced column against the oracle and every other column untouched."""
    from polars_quant_amd.suite import Suite
    rng = np.random.default_rng(seed)
    n, T_ = int(rng.choice([70, 200, 333])), int(rng.choice([304, 312, 344]))
    d = oracle.gen_ohlcv(SEED + 40 + seed, n, T_, 0)
    d["real"] = d["close"]
    g = {k: torch.from_numpy(d[k]).cuda() for k in ("open", "high", "low", "close", "volume")}
    st = Suite(n, T_, "cuda")
    all_tasks = st.tasks(fused=True)
    k = int(rng.integers(1, len(all_tasks)))
    tasks = [all_tasks[i] for i in rng.permutation(len(all_tasks))[:k]]
    if seed % 2 == 0 and "backtest_macd_cross" not in tasks: tasks.append("backtest_macd_cross")
    for t in [x for ts in st.out.values() for x in ts] + list(st.pat.values()) + st.bt:
        t.fill_(-7)
    st.record(g, tasks)
    st.run(); st.run()
    torch.cuda.synchronize()
    covered = set()
    for t in tasks:
        covered |= set({"ht_all": ("ht_dcperiod", "ht_dcphase", "ht_phasor", "ht_sine")}.get(t) or Suite.FUSED.get(t) or (t,))
    dd = dict(d); dd["periods"] = st.periods.cpu().numpy()
    for name in sorted(pq.SPEC):
        if name in covered:
            exp = oracle.call(name, *[dd[c] for c in pq.SPEC[name][0]])
            for (oname, _), got, e in zip(pq.SPEC[name][2], st.out[name], exp):
                assert_same(f"{name}.{oname}{{subset {seed}}}", got.cpu().numpy(), e, exact=name not in TRANSCENDENTAL, price=d["close"])
        else:
            for got in st.out[name]:
                assert (got == -7).all(), f"{name} was not recorded but its column changed (seed {seed})"
    if "cdl_all" in tasks:
        for nm in pq.PATTERN_NAMES:
            assert (st.pat[nm].cpu().numpy() == oracle.pattern(nm, d["open"], d["high"], d["low"], d["close"])).all(), nm
    if "backtest_macd_cross" in tasks:
        ebuy, esell = oracle.macd_cross_signals(d["close"])
        _, _, eeq, es = oracle.backtest(d["close"], ebuy, esell)
        assert (bits(st.bt[2].cpu().numpy()) == bits(eeq)).all()
        np.testing.assert_allclose(st.summary.cpu().numpy(), es, rtol=1e-12, atol=1e-13)
    st.close()
