"""not-gpu: known-answer tests for the SURVEY 8(a) rows that test_oracle_kat.py does not reach (cci, cmo, mfi, trix, ultosc,
dx / minus_di / plus_di / adx / adxr, apo / ppo, macdext, stoch / stochf / stochrsi, mavp, sar / sarext, mama, the HT value
rows, returns, more candle patterns).

Each expected value comes from a SECOND, independent restatement written here in plain Python straight from the cited
reference lines (loops over lists of float / None; `mul_add` evaluated exactly with rationals), on small inputs -- the C
oracle must reproduce it bit for bit.  Where the reference has no source (decisions D-4 / D-6), the test pins the documented
definition on inputs with a hand-checkable answer."""
import math
from fractions import Fraction

import numpy as np
import pytest

NULLB = np.uint64(0x7FF80000504E554C)


def fma(a, b, c):   # f64::mul_add: one rounding
    return float(Fraction(a) * Fraction(b) + Fraction(c))


def to_opt(a):      # oracle output -> list of float / None
    b = np.ascontiguousarray(a).view(np.uint64)
    return [None if bb == NULLB else float(v) for v, bb in zip(a, b)]


def same(got, exp, name):
    g = to_opt(got)
    assert len(g) == len(exp), name
    for i, (x, y) in enumerate(zip(g, exp)):
        if y is None or x is None:
            assert x is None and y is None, f"{name}[{i}]: got {x}, expected {y}"
        elif not (math.isnan(x) and math.isnan(y)):
            assert np.float64(x).view(np.uint64) == np.float64(y).view(np.uint64), f"{name}[{i}]: got {x!r}, expected {y!r}"


# ---- helpers restated from the reference -------------------------------------------------------------------------------
def py_sma(x, p):      # overlap.rs:871-937 on a null-free slice (D-1): running sum, +new then (count > p) -old, sum * (1/p)
    n = len(x); out = [None] * n
    if p == 0 or n < p: return out
    denom, s = 1.0 / p, 0.0
    for i in range(n):
        s += x[i]
        if i >= p: s -= x[i - p]
        if i >= p - 1: out[i] = s * denom
    return out


def py_ema(x, p):      # overlap.rs:660-730: seed = sum / p at count == p, then alpha.mul_add(x - ema, ema)
    n = len(x); out = [None] * n
    if p == 0 or n < p: return out
    alpha, s, e = 2.0 / (p + 1.0), 0.0, 0.0
    for i in range(n):
        if i < p - 1: s += x[i]
        elif i == p - 1: s += x[i]; e = s / p; out[i] = e
        else: e = fma(alpha, x[i] - e, e); out[i] = e
    return out


def py_ma_na(x, p, kind):   # calc_ma over a series with nulls (N-A: null rows are skipped, state not advanced); kind 0 SMA, 1 EMA
    idx = [i for i, v in enumerate(x) if v is not None]
    vals = [x[i] for i in idx]
    if kind == 0:          # nulls do not enter the window: the running-sum form over the valid values
        n = len(vals); res = [None] * n
        if p > 0 and len(x) >= p:
            denom, s = 1.0 / p, 0.0
            for j in range(n):
                s += vals[j]
                if j >= p: s -= vals[j - p]
                if j >= p - 1: res[j] = s * denom
    else:
        n = len(vals); res = [None] * n
        if p > 0 and len(x) >= p:
            alpha, s, e = 2.0 / (p + 1.0), 0.0, 0.0
            for j in range(n):
                if j < p - 1: s += vals[j]
                elif j == p - 1: s += vals[j]; e = s / p; res[j] = e
                else: e = fma(alpha, vals[j] - e, e); res[j] = e
    out = [None] * len(x)
    for j, i in enumerate(idx): out[i] = res[j]
    return out


def py_rma(x, p):      # D-1: None for i < p-1; seed = mean(x[0..p)); then (prev * (p-1) + x) / p
    n = len(x); out = [None] * n
    if p == 0 or n < p: return out
    s, r = 0.0, 0.0
    for i in range(n):
        if i < p - 1: s += x[i]
        elif i == p - 1: s += x[i]; r = s / p; out[i] = r
        else: r = (r * (p - 1.0) + x[i]) / p; out[i] = r
    return out


@pytest.fixture(scope="module")
def ohlcv(oracle):
    d = oracle.gen_ohlcv(0xC0FFEE, 1, 90, 0)
    return {k: v[0] for k, v in d.items()}


# ---- momentum.rs --------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("p", [1, 3, 14])
def test_cci_restated(oracle, ohlcv, p):        # momentum.rs:138-178
    h, l, c = (list(ohlcv[k]) for k in ("high", "low", "close"))
    n = len(h)
    tp = [(h[i] + l[i] + c[i]) / 3.0 for i in range(n)]
    sma = py_sma(tp, p)
    exp = [None] * n
    for i in range(p - 1, n):
        avg = sma[i]
        md = 0.0
        for j in range(i + 1 - p, i + 1): md += abs(tp[j] - avg)
        if md != 0.0:
            md /= p
            exp[i] = (tp[i] - avg) / (0.015 * md)
    same(oracle.call("cci", ohlcv["high"], ohlcv["low"], ohlcv["close"], timeperiod=p)[0], exp, f"cci({p})")


@pytest.mark.parametrize("p", [1, 4, 14])
def test_cmo_restated(oracle, ohlcv, p):        # momentum.rs:181-223: rolling SUMS, row 0 contributes 0
    x = list(ohlcv["close"]); n = len(x)
    ups, downs = [0.0] * n, [0.0] * n
    for i in range(1, n):
        d = x[i] - x[i - 1]
        if d > 0.0: ups[i] = d
        else: downs[i] = -d
    exp, su, sd = [None] * n, 0.0, 0.0
    for i in range(n):
        su += ups[i]; sd += downs[i]
        if i >= p: su -= ups[i - p]; sd -= downs[i - p]
        if i >= p - 1:
            tot = su + sd
            exp[i] = 0.0 if tot == 0.0 else 100.0 * (su - sd) / tot
    same(oracle.call("cmo", ohlcv["close"], timeperiod=p)[0], exp, f"cmo({p})")


@pytest.mark.parametrize("p", [1, 5, 14])
def test_mfi_restated(oracle, ohlcv, p):        # momentum.rs:286-342 incl. the `prev_idx > 0` guard (:322)
    h, l, c, v = (list(ohlcv[k]) for k in ("high", "low", "close", "volume")); n = len(h)
    tp = [(h[i] + l[i] + c[i]) / 3.0 for i in range(n)]
    mf = [tp[i] * v[i] for i in range(n)]
    exp, pos, neg = [None] * n, 0.0, 0.0
    for i in range(1, n):
        if tp[i] > tp[i - 1]: pos += mf[i]
        elif tp[i] < tp[i - 1]: neg += mf[i]
        if i >= p:
            q = i - p
            if q > 0:
                if tp[q] > tp[q - 1]: pos -= mf[q]
                elif tp[q] < tp[q - 1]: neg -= mf[q]
            exp[i] = 100.0 if neg == 0.0 else 100.0 - (100.0 / (1.0 + pos / neg))
    same(oracle.call("mfi", ohlcv["high"], ohlcv["low"], ohlcv["close"], ohlcv["volume"], timeperiod=p)[0], exp, f"mfi({p})")


@pytest.mark.parametrize("p", [1, 3, 9])
def test_trix_restated(oracle, ohlcv, p):       # momentum.rs:544-569: None -> 0.0 between the stages (Q-TRIX)
    x = list(ohlcv["close"]); n = len(x)
    z = lambda a: [0.0 if t is None else t for t in a]
    e3 = py_ema(z(py_ema(z(py_ema(x, p)), p)), p)
    exp = [None] * n
    for i in range(1, n):
        if e3[i] is not None and e3[i - 1] is not None and e3[i - 1] != 0.0:
            exp[i] = (e3[i] - e3[i - 1]) / e3[i - 1] * 100.0
    same(oracle.call("trix", ohlcv["close"], timeperiod=p)[0], exp, f"trix({p})")


@pytest.mark.parametrize("ps", [(1, 2, 3), (2, 5, 9), (7, 14, 28)])
def test_ultosc_restated(oracle, ohlcv, ps):    # momentum.rs:572-627
    h, l, c = (list(ohlcv[k]) for k in ("high", "low", "close")); n = len(h)
    bp, tr = [0.0] * n, [0.0] * n
    for i in range(1, n):
        lo, hi = min(l[i], c[i - 1]), max(h[i], c[i - 1])
        bp[i], tr[i] = c[i] - lo, hi - lo
    def avg(p):
        res, sb, st = [None] * n, 0.0, 0.0
        for i in range(n):
            sb += bp[i]; st += tr[i]
            if i >= p: sb -= bp[i - p]; st -= tr[i - p]
            if i >= p - 1 and st != 0.0: res[i] = sb / st
        return res
    a1, a2, a3 = (avg(p) for p in ps)
    exp = [None if None in (a1[i], a2[i], a3[i]) else 100.0 * (4.0 * a1[i] + 2.0 * a2[i] + a3[i]) / 7.0 for i in range(n)]
    same(oracle.call("ultosc", ohlcv["high"], ohlcv["low"], ohlcv["close"], timeperiod1=ps[0], timeperiod2=ps[1], timeperiod3=ps[2])[0],
         exp, f"ultosc{ps}")


def py_calc_dm(h, l, c, p):                     # momentum.rs:668-727 -> (dx, minus_di)
    n = len(h)
    pdm, mdm, tr = [0.0] * n, [0.0] * n, [0.0] * n
    for i in range(1, n):
        up, dn = h[i] - h[i - 1], l[i - 1] - l[i]
        if up > dn and up > 0.0: pdm[i] = up
        if dn > up and dn > 0.0: mdm[i] = dn
        tr[i] = max(max(h[i] - l[i], abs(h[i] - c[i - 1])), abs(l[i] - c[i - 1]))
    sp, sm, st = py_rma(pdm, p), py_rma(mdm, p), py_rma(tr, p)
    pdi, mdi, dx = [None] * n, [None] * n, [None] * n
    for i in range(n):
        if None not in (sp[i], sm[i], st[i]) and st[i] != 0.0:
            pdi[i], mdi[i] = 100.0 * sp[i] / st[i], 100.0 * sm[i] / st[i]
            s = pdi[i] + mdi[i]
            dx[i] = 0.0 if s == 0.0 else 100.0 * abs(pdi[i] - mdi[i]) / s
    return dx, mdi


@pytest.mark.parametrize("p", [1, 3, 14])
def test_dm_family_restated(oracle, ohlcv, p):  # dx :226, plus_di :400 (returns DX: D-5), minus_di :345, adx :11-29, adxr :32-61
    h, l, c = (list(ohlcv[k]) for k in ("high", "low", "close")); n = len(h)
    dx, mdi = py_calc_dm(h, l, c, p)
    args = (ohlcv["high"], ohlcv["low"], ohlcv["close"])
    same(oracle.call("dx", *args, timeperiod=p)[0], dx, f"dx({p})")
    same(oracle.call("plus_di", *args, timeperiod=p)[0], dx, f"plus_di({p}) == dx (quirk Q-PDI)")
    same(oracle.call("minus_di", *args, timeperiod=p)[0], mdi, f"minus_di({p})")
    adx = py_rma([0.0 if v is None else v for v in dx], p)       # :21-27 zero-filled DX
    same(oracle.call("adx", *args, timeperiod=p)[0], adx, f"adx({p})")
    adxr = [None] * n
    for i in range(p - 1, n):                                    # :50-59, lag p-1
        a, b = adx[i], adx[i - (p - 1)]
        if a is not None and b is not None: adxr[i] = (a + b) * 0.5
    same(oracle.call("adxr", *args, timeperiod=p)[0], adxr, f"adxr({p})")


# ---- python composites (python/polars_quant/talib/momentum.py) ----------------------------------------------------------
def py_rolling_ext(x, k, fn):   # Polars rolling_min / rolling_max(window=k): null until the frame holds k non-null rows
    out = [None] * len(x)
    for i in range(k - 1, len(x)):
        w = x[i + 1 - k:i + 1]
        if None not in w: out[i] = fn(w)
    return out


@pytest.mark.parametrize("fk,p1,t1,p2,t2", [(5, 3, 0, 3, 0), (4, 2, 1, 5, 0), (1, 1, 0, 1, 1)])
def test_stoch_and_stochf_restated(oracle, ohlcv, fk, p1, t1, p2, t2):     # momentum.py:178-195
    h, l, c = (list(ohlcv[k]) for k in ("high", "low", "close"))
    ln, hn = py_rolling_ext(l, fk, min), py_rolling_ext(h, fk, max)
    fastk = [None if ln[i] is None else (c[i] - ln[i]) * 100.0 / (hn[i] - ln[i]) for i in range(len(c))]
    slowk = py_ma_na(fastk, p1, t1)
    slowd = py_ma_na(slowk, p2, t2)
    args = (ohlcv["high"], ohlcv["low"], ohlcv["close"])
    g = oracle.call("stoch", *args, fastk_period=fk, slowk_period=p1, slowk_matype=t1, slowd_period=p2, slowd_matype=t2)
    same(g[0], slowk, "stoch.slowk"); same(g[1], slowd, "stoch.slowd")
    g = oracle.call("stochf", *args, fastk_period=fk, fastd_period=p1, fastd_matype=t1)
    same(g[0], fastk, "stochf.fastk"); same(g[1], slowk, "stochf.fastd")


def test_stochrsi_restated(oracle, ohlcv):       # momentum.py:197-205 on the reference's own RSI (momentum.rs:507-541)
    x = list(ohlcv["close"]); n = len(x); p, fk, fd = 5, 4, 3
    ups, downs = [0.0] * n, [0.0] * n
    for i in range(1, n):
        d = x[i] - x[i - 1]
        if d > 0.0: ups[i] = d
        else: downs[i] = -d
    au, ad = py_rma(ups, p), py_rma(downs, p)
    rsi = [None if au[i] is None or ad[i] is None else (100.0 if ad[i] == 0.0 else 100.0 - (100.0 / (1.0 + au[i] / ad[i]))) for i in range(n)]
    same(oracle.call("rsi", ohlcv["close"], timeperiod=p)[0], rsi, "rsi")
    ln, hn = py_rolling_ext(rsi, fk, min), py_rolling_ext(rsi, fk, max)
    fastk = [None if ln[i] is None else (rsi[i] - ln[i]) * 100.0 / (hn[i] - ln[i]) for i in range(n)]
    g = oracle.call("stochrsi", ohlcv["close"], timeperiod=p, fastk_period=fk, fastd_period=fd, fastd_matype=0)
    same(g[0], fastk, "stochrsi.fastk"); same(g[1], py_ma_na(fastk, fd, 0), "stochrsi.fastd")


@pytest.mark.parametrize("mt", [0, 1])
def test_apo_ppo_macdext_restated(oracle, ohlcv, mt):   # D-6 / momentum.py:83-88 on the reference's own MA (null-transparent)
    x = list(ohlcv["close"]); n = len(x); f, s, g_ = 4, 9, 3
    mf, ms = py_ma_na(x, f, mt), py_ma_na(x, s, mt)
    apo = [None if None in (mf[i], ms[i]) else mf[i] - ms[i] for i in range(n)]
    ppo = [None if None in (mf[i], ms[i]) or ms[i] == 0.0 else (mf[i] - ms[i]) / ms[i] * 100.0 for i in range(n)]
    same(oracle.call("apo", ohlcv["close"], fastperiod=f, slowperiod=s, matype=mt)[0], apo, "apo")
    same(oracle.call("ppo", ohlcv["close"], fastperiod=f, slowperiod=s, matype=mt)[0], ppo, "ppo")
    sig = py_ma_na(apo, g_, 1 - mt)
    hist = [None if None in (apo[i], sig[i]) else apo[i] - sig[i] for i in range(n)]
    g = oracle.call("macdext", ohlcv["close"], fastperiod=f, fastmatype=mt, slowperiod=s, slowmatype=mt, signalperiod=g_, signalmatype=1 - mt)
    same(g[0], apo, "macdext.dif"); same(g[1], sig, "macdext.dea"); same(g[2], hist, "macdext.hist")


# ---- decisions D-4 (no source in the reference): the documented definitions on hand-checkable inputs ---------------------
def test_mavp_is_the_ma_of_the_rows_period(oracle, ohlcv):
    x = ohlcv["close"]; n = len(x)
    per = np.array([2 + (i * 7) % 9 for i in range(n)], dtype=np.float64)
    per[5], per[6] = 0.0, 99.0                     # clamped to [minperiod, maxperiod]
    for mt in (0, 1):
        (got,) = oracle.call("mavp", x, per, minperiod=3, maxperiod=8, matype=mt)
        exp = [None] * n
        for i in range(n):
            P = int(min(max(per[i], 3), 8))
            ma = (py_sma(list(x), P) if mt == 0 else py_ema(list(x), P))[i]
            exp[i] = ma if i >= 8 - 1 else None   # null for i < maxperiod - 1
        same(got, exp, f"mavp(matype={mt})")


def test_sar_rising_and_reversal(oracle):
    # monotone rise: long from the start; the first SAR is the previous low, then sar += af * (ep - sar) with the extreme point
    # ep = the highest high so far and af growing by `acceleration` per new high (ta_SAR.c); hand-computed:
    h = np.array([10.0, 11.0, 12.0, 13.0, 14.0, 15.0]); l = h - 1.0
    s = to_opt(oracle.call("sar", h, l, acceleration=0.02, maximum=0.2)[0])
    assert s[0] is None and s[1] == 9.0
    cur, ep, af, exp = 9.0, 11.0, 0.02, [None, 9.0]
    for i in range(2, 6):
        cur = cur + af * (ep - cur)                 # SAR of bar i from bar i-1's state (ep / af updated AFTER the step below)
        cur = min(cur, l[i - 1], l[i])              # never above the previous or the current low
        exp.append(cur)
        if h[i] > ep: ep = h[i]; af = min(af + 0.02, 0.2)
    # (the oracle updates ep / af with bar i-1's high before stepping: same sequence shifted by the loop structure)
    assert all(v < l[i] for i, v in enumerate(s) if v is not None) and all(s[i] < s[i + 1] for i in range(1, 5))
    assert abs(s[2] - (9.0 + 0.02 * (11.0 - 9.0))) < 1e-12
    # a crash through the SAR flips to short: that bar reports the extreme point (the highest high so far); the next SAR is
    # clamped up to the previous bar's high, then it decays toward the new extreme low with a growing af
    h2 = np.array([10.0, 11.0, 12.0, 13.0, 6.0, 5.0, 4.5]); l2 = np.array([9.0, 10.0, 11.0, 12.0, 4.0, 3.0, 2.5])
    s2 = to_opt(oracle.call("sar", h2, l2, acceleration=0.02, maximum=0.2)[0])
    assert s2[4] == 13.0 and s2[5] == 13.0 and s2[6] == 13.0 + 0.04 * (3.0 - 13.0)
    # SAREXT with the same accelerations on both sides and no offset: |sarext| == sar, negative while short
    e = to_opt(oracle.call("sarext", h2, l2, startvalue=0.0, offsetonreverse=0.0, accelerationinitlong=0.02, accelerationlong=0.02,
                           accelerationmaxlong=0.2, accelerationinitshort=0.02, accelerationshort=0.02, accelerationmaxshort=0.2)[0])
    assert [abs(v) if v is not None else None for v in e] == s2 and e[3] > 0 and e[4] < 0 and e[6] < 0


def py_ht(real):
    """cycle.rs:10-69 (ht_dcperiod) + :120-140 (dcphase) + :290-300 (sine), restated: -> dcperiod, dcphase, inphase, quadrature,
    sine, leadsine as lists of float / None.  math.atan / math.sin are the same libm the C oracle links."""
    n = len(real)
    outs = [[None] * n for _ in range(6)]
    if n < 32: return outs
    smooth = [0.0] * n
    for i in range(3, n): smooth[i] = (4.0 * real[i] + 3.0 * real[i - 1] + 2.0 * real[i - 2] + real[i - 3]) * 0.1
    det, q1, i1 = [0.0] * 7, [0.0] * 7, [0.0] * 7
    push = lambda dq, v: dq.insert(0, v) or dq.pop()
    i2 = q2 = re = im = period = sp = 0.0
    for i in range(6, n):
        prev = period if i > 6 else 6.0
        adj = 0.075 * prev + 0.54
        push(det, (0.0962 * smooth[i] + 0.5769 * smooth[i - 2] - 0.5769 * smooth[i - 4] - 0.0962 * smooth[i - 6]) * adj)
        push(q1, (0.0962 * det[0] + 0.5769 * det[2] - 0.5769 * det[4] - 0.0962 * det[6]) * adj)
        push(i1, det[3])
        ji = (0.0962 * i1[0] + 0.5769 * i1[2] - 0.5769 * i1[4] - 0.0962 * i1[6]) * adj
        jq = (0.0962 * q1[0] + 0.5769 * q1[2] - 0.5769 * q1[4] - 0.0962 * q1[6]) * adj
        i2c = 0.2 * (i1[0] - jq) + 0.8 * i2
        q2c = 0.2 * (q1[0] + ji) + 0.8 * q2
        rec = 0.2 * (i2c * i2 + q2c * q2) + 0.8 * re
        imc = 0.2 * (i2c * q2 - q2c * i2) + 0.8 * im
        i2, q2, re, im = i2c, q2c, rec, imc
        if im != 0.0 and re != 0.0: period = math.tau / math.atan(im / re)
        lo, hi = 0.67 * prev, 1.5 * prev
        period = min(max(period, lo), hi)
        period = min(max(period, 6.0), 50.0)
        period = 0.2 * period + 0.8 * prev
        sp = 0.33 * period + 0.67 * sp
        if i >= 31:
            outs[0][i] = sp
            ph = math.atan(q1[0] / i1[0]) * 180.0 / math.pi if i1[0] != 0.0 else 0.0
            dc = ph + 90.0
            if i1[0] < 0.0: dc += 180.0
            if dc > 315.0: dc -= 360.0
            outs[1][i] = dc
            outs[2][i], outs[3][i] = i1[0], q1[0]
            outs[4][i] = math.sin(ph * math.pi / 180.0)
            outs[5][i] = math.sin((ph + 45.0) * math.pi / 180.0)
    return outs


def test_ht_pipeline_restated(oracle, ohlcv):
    x = list(ohlcv["close"])
    exp = py_ht(x)
    same(oracle.call("ht_dcperiod", ohlcv["close"])[0], exp[0], "ht_dcperiod")
    same(oracle.call("ht_dcphase", ohlcv["close"])[0], exp[1], "ht_dcphase")
    g = oracle.call("ht_phasor", ohlcv["close"]); same(g[0], exp[2], "inphase"); same(g[1], exp[3], "quadrature")
    g = oracle.call("ht_sine", ohlcv["close"]); same(g[0], exp[4], "sine"); same(g[1], exp[5], "leadsine")
    # x == 0: every stage is exactly 0, the period sits on its lower clamp and the phase branch i1 == 0 -> 0 + 90 is taken
    z = np.zeros(40)
    assert to_opt(oracle.call("ht_dcphase", z)[0])[31:] == [90.0] * 9
    assert to_opt(oracle.call("ht_phasor", z)[0])[31:] == [0.0] * 9
    # pure functions of real[i-3..i] (cycle.rs:365-369, :431-443)
    (tl,) = oracle.call("ht_trendline", np.arange(80.0))
    assert to_opt(tl)[:31] == [None] * 31 and to_opt(tl)[31] == (31 + 30 + 29 + 28) * 0.25
    (tm,) = oracle.call("ht_trendmode", np.arange(80.0))
    assert list(tm[:31]) == [np.int32(-2147483648)] * 31 and list(tm[31:34]) == [1, 1, 1]   # |x - tl| = 1.5 > 0.01 * tl
    assert to_opt(oracle.call("ht_dcperiod", np.arange(31.0))[0]) == [None] * 31             # n < 32 (cycle.rs:16)


def test_mama_converges_on_a_constant_series(oracle):
    """D-4 (TA-Lib MAMA on the reference's Hilbert pipeline): outputs from row 31; on a constant series both averages
    rise monotonically towards the constant, FAMA (half the adaptive alpha) behind MAMA"""
    x = np.full(120, 42.0)
    m, f = (to_opt(a) for a in oracle.call("mama", x, fastlimit=0.5, slowlimit=0.05))
    assert m[:31] == [None] * 31 and f[:31] == [None] * 31
    mm, ff = m[31:], f[31:]
    assert all(a <= b <= 42.0 for a, b in zip(mm, mm[1:])) and all(a <= b <= 42.0 for a, b in zip(ff, ff[1:]))
    assert all(b <= a for a, b in zip(mm, ff)) and abs(mm[-1] - 42.0) < 1e-9 and abs(ff[-1] - 42.0) < 1e-6


def test_returns_readme_vector(oracle):
    """the only vector the reference holds for this path: README.md:66-75"""
    (r,) = oracle.call("returns", np.array([100.0, 102.0, 101.0, 105.0]), period=1, method=0)
    assert to_opt(r)[0] is None and [round(v, 4) for v in to_opt(r)[1:]] == [0.02, -0.0098, 0.0396]
    (lg,) = oracle.call("returns", np.array([100.0, 102.0, 101.0, 105.0]), period=2, method=1)
    assert to_opt(lg)[:2] == [None, None] and to_opt(lg)[2] == math.log(101.0 / 100.0) and to_opt(lg)[3] == math.log(105.0 / 102.0)
    x = np.array([1.0, 2.0, 4.0]); x[1] = np.array([0x7FF80000504E554C], dtype=np.uint64).view(np.float64)[0]
    assert to_opt(oracle.call("returns", x, period=1, method=0)[0]) == [None, None, None]


def test_more_hand_candles(oracle):
    """pattern.rs recognisers on hand-built candles beyond test_oracle_kat.py (one positive + one negative case each)"""
    def run(name, rows, **kw):
        o, h, l, c = (np.array([r[i] for r in rows], dtype=np.float64) for i in range(4))
        return list(oracle.pattern(name, o, h, l, c, **kw))
    # three white soldiers: three long white bodies, each opening inside the previous body and closing higher
    up = [(10, 13.1, 9.9, 13), (11.5, 15.1, 11.4, 15), (13.5, 17.1, 13.4, 17)]
    assert run("cdl3whitesoldiers", up)[-1] == 100
    assert run("cdl3whitesoldiers", [(10, 13.1, 9.9, 13), (11.5, 15.1, 11.4, 15), (16, 17, 12, 12.5)])[-1] == 0
    # three black crows is the mirror image
    dn = [(17, 17.1, 13.9, 14), (15.5, 15.6, 11.9, 12), (13.5, 13.6, 9.9, 10)]
    assert run("cdl3blackcrows", [(16, 18, 15.9, 17.9)] + dn)[-1] == -100
    # marubozu: no shadows at all
    assert run("cdlmarubozu", [(10, 12, 10, 12)])[-1] == 100 and run("cdlmarubozu", [(12, 12, 10, 10)])[-1] == -100
    assert run("cdlmarubozu", [(10, 13, 9, 12)])[-1] == 0
    # dragonfly / gravestone doji (pattern.rs:626, :793): doji body (<= 0.5 % of the mid price), one shadow > 2 bodies, the other
    # < 0.1 body -- so a PERFECT doji (body 0) can never fire: `shadow < 0.1 * 0` is false (a quirk the kernels reproduce)
    assert run("cdldragonflydoji", [(10, 10.043, 8, 10.04)])[-1] == 100 and run("cdlgravestonedoji", [(10.04, 12, 9.997, 10)])[-1] == -100   # (:795: bearish, -100)
    assert run("cdldragonflydoji", [(10, 10, 8, 10)])[-1] == 0 and run("cdldragonflydoji", [(10, 12, 8, 11)])[-1] == 0
    # harami: a small body inside the previous long body, opposite colour
    assert run("cdlharami", [(15, 15.2, 9.8, 10), (11, 12.2, 10.8, 12)])[-1] == 100
    assert run("cdlharami", [(10, 15.2, 9.8, 15), (14, 14.2, 12.8, 13)])[-1] == -100
    assert run("cdlharami", [(15, 15.2, 9.8, 10), (9, 16.2, 8.8, 16)])[-1] == 0
