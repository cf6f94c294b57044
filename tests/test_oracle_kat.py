"""Known-answer tests that pin the CPU oracle to the reference source by hand-derivable cases.

The reference has no tests / fixtures, so these KATs -- each derived by hand from the
cited reference lines -- are what pins the oracle ("parity unpinned" otherwise).
"""
import math

import numpy as np
import pytest


def isnull(o, a):
    return np.asarray(a).view(np.uint64) == np.uint64(o.NULL_BITS)


def vals(o, a):
    """list with None for nulls"""
    a = np.asarray(a)
    m = isnull(o, a)
    return [None if mm else float(x) for x, mm in zip(a, m)]


def test_sma_ramp(oracle):
    # overlap.rs:871-937: null until count==p, then sum*(1/p)
    (out,) = oracle.call("sma", np.arange(1.0, 11.0), timeperiod=3)
    v = vals(oracle, out)
    assert v[:2] == [None, None]
    exp = []
    s = 0.0
    x = np.arange(1.0, 11.0)
    for i in range(10):
        s += x[i]
        if i >= 3:
            s -= x[i - 3]
        exp.append(s * (1.0 / 3.0))
    assert v[2:] == exp[2:]
    assert np.allclose(v[2:], np.arange(2.0, 10.0), rtol=0, atol=1e-15)


def test_sma_short_and_zero_period(oracle):
    # overlap.rs:874-876: p == 0 or n < p -> all null
    for p in (0, 5):
        (out,) = oracle.call("sma", [1.0, 2.0, 3.0], timeperiod=p)
        assert isnull(oracle, out).all()


def test_sma_null_transparent(oracle):
    # overlap.rs:892-895 (N-A): null row -> null out, state not advanced
    x = np.array([1.0, oracle.NULL, 2.0, 3.0, oracle.NULL, 4.0])
    (out,) = oracle.call("sma", x, timeperiod=2)
    assert vals(oracle, out) == [None, None, 1.5, 2.5, None, 3.5]


def test_ema_ramp(oracle):
    # overlap.rs:692-698: seed = mean(first p) = 2.0, alpha = 0.5 -> 3, 4, ...
    (out,) = oracle.call("ema", np.arange(1.0, 11.0), timeperiod=3)
    assert vals(oracle, out) == [None, None, 2.0, 3.0, 4.0, 5.0, 6.0, 7.0, 8.0, 9.0]


def test_bbands_constant(oracle):
    up, mid, lo = oracle.call("bbands", np.full(8, 5.0), timeperiod=4)
    assert vals(oracle, up)[3:] == [5.0] * 5 and vals(oracle, mid)[3:] == [5.0] * 5 and vals(oracle, lo)[3:] == [5.0] * 5
    assert vals(oracle, up)[:3] == [None] * 3


def test_bbands_population_variance(oracle):
    # overlap.rs:101-103: var = sum_sq/p - mean^2 (population)
    up, mid, lo = oracle.call("bbands", [1.0, 2.0, 3.0, 4.0], timeperiod=4, nbdevup=2.0, nbdevdn=1.0)
    mean, var = 2.5, 30.0 / 4 - 2.5 * 2.5
    assert vals(oracle, mid)[3] == mean
    assert vals(oracle, up)[3] == mean + 2.0 * math.sqrt(var)
    assert vals(oracle, lo)[3] == mean - 1.0 * math.sqrt(var)


def test_wma_quirk(oracle):
    # overlap.rs:1356,1365 (Q-WMA): [a,b,c], p=2 -> (a+2b)/3, (-a+2b+3c)/3
    a, b, c = 3.0, 5.0, 11.0
    (out,) = oracle.call("wma", [a, b, c], timeperiod=2)
    assert vals(oracle, out) == [None, (a + 2 * b) / 3.0, (-a + 2 * b + 3 * c) / 3.0]


def test_dema_bitmap_branch(oracle):
    # overlap.rs:561-598 (D-2); p=2: alpha=2/3; first value at count 2p = 4 (row 3)
    x = [1.0, 2.0, 3.0, 4.0, 5.0]
    (out,) = oracle.call("dema", x, timeperiod=2)
    al = 2.0 / 3.0
    e0 = 1.5                      # count 2: seed
    s1 = e0
    e0 = math.fma(al, 3.0 - e0, e0) if hasattr(math, "fma") else al * (3.0 - e0) + e0  # count 3 == 2p-1
    s1 += e0
    e1 = s1 / 2.0
    v = vals(oracle, out)
    assert v[:3] == [None, None, None]
    e0b = al * (4.0 - e0) + e0
    e1b = al * (e0b - e1) + e1
    assert v[3] == pytest.approx(2 * e0b - e1b, rel=1e-15)


def test_t3_first_row_and_unseeded_e5(oracle):
    # overlap.rs:942: n < 6p-5 -> all null; first value at count 6p-5
    x = np.arange(1.0, 30.0)
    (out,) = oracle.call("t3", x, timeperiod=3, vfactor=0.7)
    m = isnull(oracle, out)
    assert m[:12].all() and not m[12:].any()          # 6*3-6 = 12
    (short,) = oracle.call("t3", x[:12], timeperiod=3, vfactor=0.7)
    assert isnull(oracle, short).all()


def test_tema_first_row(oracle):
    (out,) = oracle.call("tema", np.arange(1.0, 20.0), timeperiod=3)
    m = isnull(oracle, out)
    assert m[:6].all() and not m[6:].any()            # 3p-3
    # a ramp is reproduced exactly by TEMA once all stages are seeded on a ramp
    assert np.allclose(out[6:], np.arange(7.0, 20.0), atol=1e-12)


def test_trima(oracle):
    # overlap.rs:1313-1326: odd p -> sma(sma(x,k),k), k = p/2+1
    x = np.array([1.0, 4.0, 2.0, 8.0, 5.0, 7.0])
    (out,) = oracle.call("trima", x, timeperiod=3)
    (inner,) = oracle.call("sma", x, timeperiod=2)
    (exp,) = oracle.call("sma", inner, timeperiod=2)
    assert vals(oracle, out) == vals(oracle, exp)
    assert vals(oracle, out)[:2] == [None, None] and vals(oracle, out)[2] is not None


def test_kama_first_row(oracle):
    # overlap.rs:836-845: pass 2 restarts its count at row p -> first value at row 2p = mean(x[p..2p))
    x = np.arange(1.0, 13.0) ** 1.5
    (out,) = oracle.call("kama", x, timeperiod=3)
    v = vals(oracle, out)
    assert v[:6] == [None] * 6
    assert v[6] == (x[3] + x[4] + x[5]) / 3.0


def test_midpoint_quirk(oracle):
    # overlap.rs:227-231 (Q-MID): min deque never expires -> cumulative min; no warm-up nulls
    (out,) = oracle.call("midpoint", [5.0, 1.0, 3.0, 2.0, 4.0, 6.0], timeperiod=2)
    assert vals(oracle, out) == [5.0, 3.0, 2.0, 2.0, 2.5, 3.5]


def test_midprice(oracle):
    (out,) = oracle.call("midprice", [5.0, 1.0, 3.0, 2.0], [4.0, 0.5, 2.0, 1.0], timeperiod=2)
    assert vals(oracle, out) == [4.5, (5 + 0.5) / 2, (3 + 0.5) / 2, (3 + 1.0) / 2]


def test_rma_d1(oracle):
    # D-1: seed = mean(x[0..p)) at i = p-1; then (prev*(p-1)+x)/p
    import ctypes as C
    x = np.array([0.0, 1.0, 2.0, 3.0, 4.0])
    out = np.empty(5)
    oracle.lib().pqo_rma(x.ctypes.data_as(C.c_void_p), C.c_int64(5), C.c_int64(3), out.ctypes.data_as(C.c_void_p))
    assert vals(oracle, out) == [None, None, 1.0, (1.0 * 2 + 3.0) / 3, (((1.0 * 2 + 3.0) / 3) * 2 + 4.0) / 3]


def test_macd_zero_fill_quirk(oracle):
    # momentum.rs:268-271 (Q-MACD): signal = EMA(dif with None->0) -> Some from sig-1, before macd exists
    x = np.arange(1.0, 9.0)
    macd, sig, hist = oracle.call("macd", x, fastperiod=2, slowperiod=3, signalperiod=2)
    vm, vs, vh = vals(oracle, macd), vals(oracle, sig), vals(oracle, hist)
    assert vm[:2] == [None, None] and vm[2] == 0.5
    assert vs[0] is None and vs[1] == 0.0
    assert vs[2] == pytest.approx(1.0 / 3.0, rel=1e-15)
    assert vh[:2] == [None, None] and vh[2] == pytest.approx(0.5 - 1.0 / 3.0, rel=1e-15)


def test_rsi_monotone(oracle):
    # momentum.rs:532-533: avg_down == 0 -> 100
    (out,) = oracle.call("rsi", np.arange(1.0, 12.0), timeperiod=3)
    v = vals(oracle, out)
    assert v[:2] == [None, None] and all(z == 100.0 for z in v[2:])


def test_obv_sign_inverted(oracle):
    # volume.rs:78-87 (Q-OBV): d = prev - close; d>0 -> +v
    (out,) = oracle.call("obv", [10.0, 11.0, 10.5], [100.0, 200.0, 300.0])
    assert vals(oracle, out) == [None, -200.0, 100.0]


def test_ad_flat_bar_emits_zero(oracle):
    # volume.rs:115-116 (Q-AD)
    h = [10.0, 10.0, 12.0]
    l = [8.0, 10.0, 10.0]
    c = [9.5, 10.0, 10.5]
    v = [100.0, 100.0, 100.0]
    (out,) = oracle.call("ad", h, l, c, v)
    first = (2 * 9.5 - 8 - 10) / 2.0 * 100.0
    assert vals(oracle, out) == [first, 0.0, first + (2 * 10.5 - 10 - 12) / 2.0 * 100.0]


def test_trange_atr(oracle):
    h = np.array([10.0, 11.0, 12.0, 11.5, 13.0])
    l = np.array([9.0, 10.0, 10.5, 10.0, 11.0])
    c = np.array([9.5, 10.5, 11.0, 11.0, 12.5])
    (tr,) = oracle.call("trange", h, l, c)
    assert vals(oracle, tr) == [None, 1.5, 1.5, 1.5, 2.0]
    # volatility.rs:30: calc_ema(trange, 2p-1): p=2 -> EMA(3) over non-null TR -> first at row 3
    (atr,) = oracle.call("atr", h, l, c, timeperiod=2)
    v = vals(oracle, atr)
    assert v[:3] == [None] * 3 and v[3] == 1.5 and v[4] == 0.5 * (2.0 - 1.5) + 1.5


def test_price_transforms(oracle):
    o, h, l, c = [1.0], [4.0], [0.5], [2.0]
    assert oracle.call("avgprice", o, h, l, c)[0][0] == (1 + 4 + 0.5 + 2) * 0.25
    assert oracle.call("medprice", h, l)[0][0] == 2.25
    assert oracle.call("typprice", h, l, c)[0][0] == (4 + 0.5 + 2) / 3.0
    assert oracle.call("wclprice", h, l, c)[0][0] == (4 + 0.5 + 4) / 4.0


def test_roc_family_and_mom(oracle):
    x = [2.0, 0.0, 4.0, 5.0]
    assert vals(oracle, oracle.call("mom", x, timeperiod=2)[0]) == [None, None, 2.0, 5.0]
    assert vals(oracle, oracle.call("roc", x, timeperiod=2)[0]) == [None, None, 100.0, None]  # prev == 0 -> None
    assert vals(oracle, oracle.call("rocr", x, timeperiod=2)[0]) == [None, None, 2.0, None]


def test_aroon_ties_latest(oracle):
    # momentum.rs:90,96: >= / <= -> latest index wins
    up, dn = oracle.call("aroon", [3.0, 3.0, 1.0], [1.0, 0.5, 0.5], timeperiod=2)
    assert vals(oracle, up) == [None, None, 50.0] and vals(oracle, dn) == [None, None, 100.0]


def test_willr_and_stochf_ramp(oracle):
    h = np.arange(2.0, 12.0)
    l = h - 1.0
    c = h.copy()
    (w,) = oracle.call("willr", h, l, c, timeperiod=3)
    assert vals(oracle, w)[:2] == [None, None] and all(z == 0.0 for z in vals(oracle, w)[2:])  # -100*(0)/d = -0.0 == 0.0
    fk, fd = oracle.call("stochf", h, l, c, fastk_period=3, fastd_period=2, fastd_matype=0)
    assert vals(oracle, fk)[:2] == [None, None] and all(z == 100.0 for z in vals(oracle, fk)[2:])
    assert vals(oracle, fd)[:3] == [None] * 3 and all(abs(z - 100.0) < 1e-12 for z in vals(oracle, fd)[3:])


def test_ht_short_and_warmup(oracle):
    (out,) = oracle.call("ht_dcperiod", np.arange(1.0, 32.0))
    assert isnull(oracle, out).all()                                   # n < 32 (cycle.rs:16)
    x = 50 + 5 * np.sin(np.arange(64) * 2 * np.pi / 20.0)
    (out,) = oracle.call("ht_dcperiod", x)
    m = isnull(oracle, out)
    assert m[:31].all() and not m[31:].any()
    assert ((out[31:] >= 6.0 * 0.33) & (out[31:] <= 50.0)).all()
    (tm,) = oracle.call("ht_trendmode", x)
    assert (tm[:31] == oracle.NULL_I32).all() and set(np.unique(tm[31:])) <= {0, 1}
    (tl,) = oracle.call("ht_trendline", x)
    assert tl[40] == (x[40] + x[39] + x[38] + x[37]) * 0.25


def test_patterns_hand_candles(oracle):
    # bodies >= 6 % so long_body fires (pattern.rs:2097-2099)
    # engulfing (pattern.rs:635-662): bear candle then bull candle engulfing it
    o = np.array([10.0, 9.4]); c = np.array([9.5, 10.2]); h = np.array([10.1, 10.3]); l = np.array([9.4, 9.3])
    assert list(oracle.pattern("cdlengulfing", o, h, l, c)) == [0, 100]
    assert list(oracle.pattern("cdlengulfing", c, h, l, o)) == [0, -100]
    # doji (:553-575): |o-c| <= 0.5 % of mid
    assert list(oracle.pattern("cdldoji", [10.0, 10.0], [10.5, 10.5], [9.5, 9.5], [10.04, 10.06])) == [100, 0]
    # 3 white soldiers (:234-265)
    o = np.array([10.0, 10.5, 11.2]); c = np.array([10.8, 11.5, 12.3]); h = c + 0.05; l = o - 0.05
    assert list(oracle.pattern("cdl3whitesoldiers", o, h, l, c)) == [0, 0, 100]
    assert list(oracle.pattern("cdl3blackcrows", c, h, l, o)) == [0, 0, 0]  # opens must be inside prior body
    # dark cloud cover (:519-550), penetration default python 0.5 vs rust 0.3
    o = np.array([10.0, 11.2]); c = np.array([11.0, 10.6]); h = np.array([11.05, 11.3]); l = np.array([9.9, 10.5])
    assert list(oracle.pattern("cdldarkcloudcover", o, h, l, c, penetration=0.3)) == [0, -100]
    assert list(oracle.pattern("cdldarkcloudcover", o, h, l, c, penetration=0.5)) == [0, 0]
    # hammer (:802-829): small body, long lower shadow, after a bear candle
    o = np.array([10.5, 10.0]); c = np.array([10.0, 10.05]); h = np.array([10.6, 10.052]); l = np.array([9.9, 9.5])
    assert list(oracle.pattern("cdlhammer", o, h, l, c)) == [0, 100]
    # morning star (:1454-1487)
    o = np.array([11.0, 9.8, 9.9]); c = np.array([10.0, 9.75, 10.8]); h = np.maximum(o, c) + 0.02; l = np.minimum(o, c) - 0.02
    assert list(oracle.pattern("cdlmorningstar", o, h, l, c, penetration=0.3)) == [0, 0, 100]


def test_backtest_script(oracle):
    # vectorized.rs:130-194, hand-computed
    price = np.array([10.0, 11.0, 12.0, 11.0, 13.0, 14.0])
    buy = np.array([1, 0, 0, 0, 1, 0], np.uint8)
    sell = np.array([0, 0, 1, 0, 0, 0], np.uint8)
    pos, cash, eq, s = oracle.backtest(price, buy, sell)
    assert list(pos) == [10000.0, 10000.0, 0.0, 0.0, 9225.0, 9225.0]
    assert cash[0] == 100000.0 - (100000.0 + 30.0)
    assert eq[0] == 99970.0 and eq[1] == 109970.0
    assert cash[2] == -30.0 + (120000.0 - 36.0) and eq[3] == 119934.0
    assert cash[4] == pytest.approx(119934.0 - (9225 * 13.0 + 9225 * 13.0 * 0.0003), abs=1e-9)
    assert eq[5] == pytest.approx(-26.9775 + 9225 * 14.0, abs=1e-9)
    assert s[7] == 2.0 and s[6] == 0.5           # 2 trades, 1 closed winner -> win_rate = wins/trades
    assert s[5] == pytest.approx((eq[5] - 1e5) / 1e5)


def test_backtest_skips_bad_price_rows(oracle):
    # vectorized.rs:141-144: NaN / <= 0 price -> state untouched, equity = cash + pos*price
    price = np.array([10.0, np.nan, -1.0, 12.0])
    buy = np.array([1, 1, 1, 0], np.uint8)
    sell = np.array([0, 1, 1, 1], np.uint8)
    pos, cash, eq, s = oracle.backtest(price, buy, sell)
    assert list(pos) == [10000.0, 10000.0, 10000.0, 0.0]
    assert math.isnan(eq[1]) and eq[2] == cash[2] + pos[2] * -1.0


def test_leveraged_backtest_lots_and_fees(oracle):
    # D-10 (oracle/backtest.c; README.md:346-366, :443): 100-share lots, the lot count shrinks until cost + fee fits
    price = np.array([[10.0, 11.0, 12.0]])
    buy = np.array([[1, 0, 0]], np.uint8)
    sell = np.array([[0, 0, 1]], np.uint8)
    r = oracle.backtest_leveraged(price, buy, sell)
    cost = 99 * 100.0 * 10.0                       # 100 lots would need 100000 + 30 > 100000
    outlay = cost + cost * 0.0003
    assert r["stock_value"][0, 0] == 9900 * 10.0 and r["cash"][0, 0] == 100000.0 - outlay
    assert r["total_value"][0, 1] == (100000.0 - outlay) + 9900 * 11.0
    rev = 9900 * 12.0
    net = rev - rev * 0.0003
    assert r["total_value"][0, 2] == (100000.0 - outlay) + net and r["stock_value"][0, 2] == 0.0
    t = r["trades"]
    assert r["trade_count"][0] == 1 and t["entry_day"][0, 0] == 0 and t["exit_day"][0, 0] == 2 and t["reason"][0, 0] == 1
    assert t["quantity"][0, 0] == 9900.0 and t["pnl"][0, 0] == net - outlay and t["pnl_pct"][0, 0] == (net - outlay) / outlay * 100.0
    assert r["summary"][0, 7] == 1.0 and r["summary"][0, 6] == 1.0


def test_leveraged_backtest_interest_and_margin_call(oracle):
    # D-10 steps 1-2: the debt compounds daily at rate/252; equity below threshold * position value forces a sale
    price = np.array([[10.0, 10.0, 6.0, 7.0]])
    buy = np.array([[1, 0, 0, 0]], np.uint8)
    sell = np.zeros((1, 4), np.uint8)
    r = oracle.backtest_leveraged(price, buy, sell, leverage=2.0)
    cost = 199 * 100.0 * 10.0                      # 200 lots: 200000 + 60 > 2 * 100000
    outlay = cost + cost * 0.0003
    debt = outlay - 100000.0
    assert r["cash"][0, 0] == 0.0 - debt and r["stock_value"][0, 0] == 199000.0
    debt += debt * (0.06 / 252.0)
    assert r["cash"][0, 1] == 0.0 - debt
    debt += debt * (0.06 / 252.0)
    assert 0.0 + 19900 * 6.0 - debt < 0.3 * (19900 * 6.0)       # margin call on day 2
    rev = 19900 * 6.0
    net = rev - rev * 0.0003
    assert r["total_value"][0, 2] == (0.0 + net - debt) and r["stock_value"][0, 2] == 0.0
    assert r["trades"]["reason"][0, 0] == 2 and r["trades"]["pnl"][0, 0] == net - outlay
    assert r["total_value"][0, 3] == r["total_value"][0, 2]      # flat afterwards, no interest without debt


def test_leveraged_backtest_invalid_prices_and_portfolio(oracle):
    # D-10 step 0: no trading on a null / non-positive price, valuation at the last valid price
    price = np.array([[10.0, np.nan, -1.0, 12.0], [20.0, 21.0, 22.0, 23.0]])
    buy = np.array([[1, 1, 1, 0], [0, 0, 0, 0]], np.uint8)
    sell = np.array([[0, 1, 1, 0], [0, 0, 0, 0]], np.uint8)
    r = oracle.backtest_leveraged(price, buy, sell)
    assert list(r["stock_value"][0]) == [99000.0, 99000.0, 99000.0, 9900 * 12.0]
    assert list(r["total_value"][1]) == [100000.0] * 4 and r["trade_count"][1] == 0
    bench = np.array([100.0, 101.0, 99.0, 102.0])
    m = oracle.portfolio_metrics(r["total_value"], 200000.0, bench)
    pv = r["total_value"][0] + r["total_value"][1]
    assert list(m[:, 0]) == list(pv)
    assert m[0, 1] == pv[0] - 200000.0 and m[2, 2] == (pv[2] - pv[1]) / pv[1] * 100.0
    assert m[3, 3] == pv[3] - 200000.0 and m[3, 4] == (pv[3] - 200000.0) / 200000.0 * 100.0
    assert m[0, 5] == 0.0 and m[2, 5] == (99.0 - 101.0) / 101.0 * 100.0 and m[2, 6] == m[2, 2] - m[2, 5]
    assert m[3, 7] == m[3, 4] - (102.0 - 100.0) / 100.0 * 100.0
    rr, br = m[:, 2], m[:, 5]
    cov = sum((rr - sum(rr) / 4) * (br - sum(br) / 4)) / 3.0
    var = sum((br - sum(br) / 4) ** 2) / 3.0
    assert m[0, 8] == pytest.approx(cov / var, rel=1e-12) and (m[:, 8] == m[0, 8]).all()


def test_signal_rules(oracle):
    # D-11 (oracle/backtest.c): cross / band / channel rules, null-safe, row 0 never fires
    a = np.array([1.0, 2.0, 3.0, 2.0, 1.0, oracle.NULL, 3.0])
    b = np.array([2.0, 2.0, 2.0, 2.0, 2.0, 2.0, 2.0])
    buy, sell = oracle.cross_signals(a, b)
    assert list(buy) == [0, 0, 1, 0, 0, 0, 0]      # 2<=2 then 3>2 at row 2 (row 1: 2>2 is false)
    assert list(sell) == [0, 0, 0, 0, 1, 0, 0]     # 2>=2 then 1<2 at row 4; rows touching the null never fire
    x = np.array([25.0, 35.0, 75.0, 65.0, 28.0, 30.0])
    buy, sell = oracle.band_signals(x, 30.0, 70.0)
    assert list(buy) == [0, 1, 0, 0, 0, 1] and list(sell) == [0, 0, 0, 1, 0, 0]
    p = np.array([10.0, 8.0, 9.0, 13.0, 12.0])
    lo = np.array([9.0, 9.0, 9.0, 9.0, 9.0]); hi = np.array([12.0, 12.0, 12.0, 12.0, 12.0])
    buy, sell = oracle.channel_signals(p, lo, hi, 0)
    assert list(buy) == [0, 1, 0, 0, 0] and list(sell) == [0, 0, 0, 1, 0]
    buy, sell = oracle.channel_signals(p, lo, hi, 1)
    assert list(buy) == [0, 0, 0, 1, 0] and list(sell) == [0, 1, 0, 0, 0]


def test_factor_ic_against_scipy(oracle):
    # D-12 (oracle/backtest.c): an independent pin -- scipy.stats.pearsonr / spearmanr on the same cross-sections
    from scipy import stats
    rng = np.random.default_rng(0)
    f = rng.normal(size=(60, 9))
    r = 0.3 * f + rng.normal(size=(60, 9))
    f[3, 2] = oracle.NULL; r[7, 2] = np.nan; f[5, 2] = np.inf     # pairwise deletion
    f[:, 5] = np.round(f[:, 5], 0)                                 # ties -> average ranks
    f[:, 6] = 1.0                                                  # zero variance -> null
    f[2:, 7] = oracle.NULL                                         # fewer than 2 pairs -> null
    ic, nv = oracle.factor_ic(f, r, 0)
    ric, nv2 = oracle.factor_ic(f, r, 1)
    assert list(nv) == [60, 60, 57, 60, 60, 60, 60, 2, 60] and list(nv2) == list(nv)
    for t in (0, 1, 2, 3, 4, 5, 8):
        m = np.isfinite(f[:, t]) & np.isfinite(r[:, t])
        assert ic[t] == pytest.approx(stats.pearsonr(f[m, t], r[m, t])[0], abs=1e-14)
        assert ric[t] == pytest.approx(stats.spearmanr(f[m, t], r[m, t])[0], abs=1e-14)
    assert math.isnan(ic[6]) and math.isnan(ric[6])
    assert ic[7] == pytest.approx(1.0 if (f[1, 7] - f[0, 7]) * (r[1, 7] - r[0, 7]) > 0 else -1.0)   # two points
    m3, ir3 = oracle.rolling_ic(ic, 3)
    assert math.isnan(m3[1]) and m3[2] == (ic[0] + ic[1] + ic[2]) / 3.0
    assert math.isnan(m3[6]) and math.isnan(m3[8])                 # a null IC inside the window
    sd = math.sqrt(((ic[0] - m3[2]) ** 2 + (ic[1] - m3[2]) ** 2 + (ic[2] - m3[2]) ** 2) / 2.0)
    assert ir3[2] == pytest.approx(m3[2] / sd, rel=1e-14)


def test_summary_small(oracle):
    # metrics.rs:7-152 on [100k, 101k, 99k]
    eq = np.array([100000.0, 101000.0, 99000.0])
    s = oracle.summary(eq, None, 100000.0, 0, 0)
    r = [0.0, 0.01, (99000.0 - 101000.0) / 101000.0]
    mean = (r[0] + r[1] + r[2]) / 3.0
    var = sum((x - mean) ** 2 for x in r) / 2.0
    ann = (1.0 + (-0.01)) ** (252.0 / 3.0) - 1.0
    vol = math.sqrt(var) * math.sqrt(252.0)
    assert s[0] == pytest.approx(ann, rel=1e-14)
    assert s[1] == (101000.0 - 99000.0) / 101000.0
    assert s[4] == pytest.approx((ann - 0.03) / vol, rel=1e-13)
    assert s[2] == 0.0 and s[3] == 0.0 and s[5] == 0.0 and s[6] == 0.0 and s[7] == 0.0
    # with benchmark == equity: beta 1, alpha = ann - (rf + (bench_ann - rf)) where bench starts at eq[0]
    s2 = oracle.summary(eq, eq, 100000.0, 0, 0)
    assert s2[3] == pytest.approx(1.0, rel=1e-12)


def test_sar_basic(oracle):
    # D-4 (TA-Lib algorithm): rising market starts long with sar = low[0]
    h = np.array([10.0, 11.0, 12.0, 13.0, 12.0, 9.0])
    l = np.array([9.0, 10.0, 11.0, 12.0, 11.0, 8.0])
    (out,) = oracle.call("sar", h, l, acceleration=0.02, maximum=0.2)
    v = vals(oracle, out)
    assert v[0] is None and v[1] == 9.0
    assert v[2] == pytest.approx(9.0 + 0.02 * (11.0 - 9.0))
    assert v[5] == 13.0  # reversal on the crash bar: sar jumps to the extreme point


def test_product_generator_matches_oracle_generator(oracle):
    # bench.py takes its inputs from polars_quant_amd.synthetic (no oracle on the measured path): same bits as pqo_gen_ohlcv
    from polars_quant_amd.synthetic import gen_ohlcv
    for mode in (0, 1):
        a, b = oracle.gen_ohlcv(0x5EED0002, 97, 311, mode), gen_ohlcv(0x5EED0002, 97, 311, mode)
        for k in a:
            assert (a[k].view(np.uint64) == b[k].view(np.uint64)).all(), (mode, k)


def test_generator_reproducible(oracle):
    a = oracle.gen_ohlcv(0x5EED0001, 3, 50)
    b = oracle.gen_ohlcv(0x5EED0001, 3, 50)
    for k in a:
        assert (a[k] == b[k]).all()
    assert (a["high"] >= np.maximum(a["open"], a["close"])).all()
    assert (a["low"] <= np.minimum(a["open"], a["close"])).all() and (a["low"] > 0).all()
    assert a["close"][1, 0] != a["close"][0, 0]
