"""Known-answer tests for ALL 61 recognisers: the C oracle against the hand-built candles of tests/pattern_kats.py, which were
derived from the Rust source (src/talib/pattern.rs, lines cited there) -- one firing case per sign a recogniser can emit and at
least one near miss that differs from a firing case in ONE condition.  Runs on CPU."""
import numpy as np
import pytest

from pattern_kats import KAT_PENETRATION, KATS, PEN_CASES, UNSAT, kat_series, positives


def run(oracle, name, candles, pen=KAT_PENETRATION):
    a = np.asarray(candles, dtype=np.float64)
    return oracle.pattern(name, a[:, 0].copy(), a[:, 1].copy(), a[:, 2].copy(), a[:, 3].copy(), penetration=pen)


def test_table_covers_every_recogniser(oracle):
    assert sorted(KATS) == sorted(oracle.PATTERN_NAMES) and len(KATS) == 61
    for name, cases in KATS.items():
        vals = {v for _, v, _ in cases}
        assert 0 in vals, f"{name}: no near miss"
        if name in UNSAT:
            assert vals == {0}
        else:
            assert vals - {0}, f"{name}: no firing case"


@pytest.mark.parametrize("name", sorted(KATS))
def test_pattern_kat(oracle, name):
    for candles, expect, why in KATS[name]:
        out = run(oracle, name, candles)
        assert out[-1] == expect, f"{name}: {why}: got {out[-1]}, hand-derived {expect}"
        assert (out[:-1] == 0).all() or len(candles) == 1, f"{name}: fired before the look-back was complete: {out}"
        # candles are valid OHLC bars
        a = np.asarray(candles)
        assert (a[:, 1] >= a[:, [0, 3]].max(axis=1) - 1e-12).all() and (a[:, 2] <= a[:, [0, 3]].min(axis=1) + 1e-12).all(), name


def test_signs_a_recogniser_can_emit(oracle):
    """both signs are pinned wherever the source has an `else if ... -100` arm (pattern.rs: the 25 two-sided recognisers)"""
    two_sided = {"cdl3inside", "cdl3linestrike", "cdl3outside", "cdlabandonedbaby", "cdlbelthold", "cdlbreakaway", "cdlclosingmarubozu",
                 "cdlcounterattack", "cdldojistar", "cdlengulfing", "cdlgapsidesidewhite", "cdlharami", "cdlharamicross", "cdlhighwave",
                 "cdlhikkake", "cdlhikkakemod", "cdlkicking", "cdlkickingbylength", "cdllongline", "cdlmarubozu", "cdlrisefall3methods",
                 "cdlseparatinglines", "cdlshortline", "cdlspinningtop", "cdltasukigap", "cdltristar", "cdlxsidegap3methods"}
    for name, cases in KATS.items():
        signs = {v for _, v, _ in cases} - {0}
        assert signs == ({100, -100} if name in two_sided else signs) and len(signs) == (2 if name in two_sided else (0 if name in UNSAT else 1)), name


@pytest.mark.parametrize("name,candles,pen,expect", PEN_CASES, ids=[f"{c[0]}@{c[2]}" for c in PEN_CASES])
def test_penetration_cases(oracle, name, candles, pen, expect):
    assert run(oracle, name, candles, pen)[-1] == expect


def test_cdl2crows_is_unsatisfiable(oracle):
    """pattern.rs:30-33: bear2 (c2 < o2) contradicts open_in2 = (o > o2) && (o < c2): no input fires, including inputs built to
    satisfy every OTHER condition of the recogniser"""
    rng = np.random.default_rng(2)
    n = 20000
    o1 = rng.uniform(5, 50, n); c1 = o1 * rng.uniform(1.06, 1.3, n)                 # long bull
    o2 = c1 * rng.uniform(1.001, 1.1, n); c2 = o2 * rng.uniform(0.9, 0.999, n)      # gapped bear
    o3 = rng.uniform(np.minimum(o2, c2) * 0.98, np.maximum(o2, c2) * 1.02)          # anywhere around body 2
    c3 = rng.uniform(o1, c1)                                                        # inside body 1
    O = np.stack([o1, o2, o3], 1); Cc = np.stack([c1, c2, c3], 1)
    H = np.maximum(O, Cc) * 1.001; L = np.minimum(O, Cc) * 0.999
    out = oracle.pattern("cdl2crows", O.copy(), H.copy(), L.copy(), Cc.copy())
    assert not out.any()
    big = oracle.gen_ohlcv(0x5EED0009, 64, 2000, 1)
    assert not oracle.pattern("cdl2crows", big["open"], big["high"], big["low"], big["close"]).any()


def test_kat_series_fires_every_satisfiable_recogniser(oracle):
    """the concatenated positives (the series the GPU parity tests append to their pattern-rich input): each marked row carries
    its hand-derived value whatever precedes the sequence, and 60 of 61 recognisers fire"""
    need = sum(len(cs) + 1 for _, cs, _ in positives())
    o, h, l, c, marks = kat_series(need)
    assert {m[1] for m in marks} == set(KATS) - set(UNSAT)
    for name in KATS:
        for pen in (0.3, None):    # the Rust default everywhere / each Python wrapper's own default (0.5 for darkcloudcover, piercing)
            out = oracle.pattern(name, o[None].copy(), h[None].copy(), l[None].copy(), c[None].copy(), penetration=pen)[0]
            for row, nm, v in marks:
                if nm == name:
                    assert out[row] == v, (name, row, out[row], v)
            assert out.any() == (name not in UNSAT), name
