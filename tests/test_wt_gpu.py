"""-m gpu: the one-symbol-per-wavefront, time-parallel indicator kernels (csrc/wt_dev.h, ops_wt.h) against the oracle, bit for bit.

They apply to regular batches with 1024 <= len <= 4096; every test asserts through pq_wt_stats that the wave form really ran
(or, for the NULL-bearing inputs, that the symbols were handed to the gated lane-per-symbol path).  The speculative chunks of the
contractive recurrences are verified on the device against their predecessors' states; here the RESULT is compared with the
oracle's serial walk: EMA / DEMA / TEMA / TRIX (overlap.rs:660-730, :543-598, :1177-1311, momentum.rs:544-569), MACD / MACDFIX
(momentum.rs:250-283), RSI (:507-541), +DM / -DM (:359-436), DX / DI / ADX / ADXR (:668-727, :11-61), ATR / NATR
(volatility.rs:18-48), MIDPOINT / MIDPRICE (overlap.rs:180-404)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

SEED = 0x5EED0011
NULLB = np.uint64(0x7FF80000504E554C)
SHAPES = [(70, 1024), (130, 2520), (66, 4096), (5, 1501), (3, 1027)]   # C = 16 / 40 / 64 / 24 (odd length) / 17


@pytest.fixture(autouse=True)
def every_wave_form(monkeypatch):
    """the library uses by default only the wave forms that beat the lane-per-symbol body they replace (csrc/wt.hip); here all"""
    monkeypatch.setenv("PQ_WT_ALL", "1")
    monkeypatch.setenv("PQ_MIDPRICE_SEQ", "1")     # (a direct MIDPRICE call is a ROW launch by default: csrc/overlap.hip)


@pytest.fixture(scope="module")
def pq():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import polars_quant_amd as pq
    from polars_quant_amd._lib import lib
    lib()
    return pq


def bits(a):
    return np.ascontiguousarray(a).view(np.uint64)


def same(name, g, e):
    g = np.asarray(g)
    assert g.shape == e.shape, (name, g.shape, e.shape)
    bad = (bits(g) != bits(e)) & ~(np.isnan(g) & np.isnan(e) & (bits(g) != NULLB) & (bits(e) != NULLB))
    assert not bad.any(), f"{name}: {bad.sum()} of {bad.size} values differ; first at {np.argwhere(bad)[:3].tolist()}: got {g[bad][:3]} expected {e[bad][:3]}"


CASES = [
    ("ema", ("close",), dict(timeperiod=30)), ("ema", ("close",), dict(timeperiod=1)), ("ema", ("close",), dict(timeperiod=2)),
    ("ema", ("close",), dict(timeperiod=200)), ("dema", ("close",), dict(timeperiod=30)), ("dema", ("close",), dict(timeperiod=3)),
    ("tema", ("close",), dict(timeperiod=30)), ("tema", ("close",), dict(timeperiod=7)), ("tema", ("close",), dict(timeperiod=350)),
    ("trix", ("close",), dict(timeperiod=30)), ("trix", ("close",), dict(timeperiod=4)),
    ("macd", ("close",), dict()), ("macd", ("close",), dict(fastperiod=5, slowperiod=13, signalperiod=4)),
    ("macd", ("close",), dict(fastperiod=26, slowperiod=12, signalperiod=40)), ("macdfix", ("close",), dict(signalperiod=9)),
    ("macdfix", ("close",), dict(signalperiod=3)),
    ("rsi", ("close",), dict(timeperiod=14)), ("rsi", ("close",), dict(timeperiod=1)), ("rsi", ("close",), dict(timeperiod=60)),
    ("plus_dm", ("high", "low"), dict(timeperiod=14)), ("minus_dm", ("high", "low"), dict(timeperiod=5)),
    ("dx", ("high", "low", "close"), dict(timeperiod=14)), ("plus_di", ("high", "low", "close"), dict(timeperiod=14)),
    ("minus_di", ("high", "low", "close"), dict(timeperiod=9)), ("adx", ("high", "low", "close"), dict(timeperiod=14)),
    ("adx", ("high", "low", "close"), dict(timeperiod=2)), ("adxr", ("high", "low", "close"), dict(timeperiod=14)),
    ("adxr", ("high", "low", "close"), dict(timeperiod=1)),
    ("atr", ("high", "low", "close"), dict(timeperiod=14)), ("natr", ("high", "low", "close"), dict(timeperiod=3)),
    ("midpoint", ("close",), dict(timeperiod=14)), ("midpoint", ("close",), dict(timeperiod=1)),
    ("midprice", ("high", "low"), dict(timeperiod=14)), ("midprice", ("high", "low"), dict(timeperiod=100)),
]


@pytest.fixture(scope="module", params=SHAPES, ids=[f"{n}x{t}" for n, t in SHAPES])
def data(oracle, request):
    n, t = request.param
    return oracle.gen_ohlcv(SEED, n, t, 0)


@pytest.mark.parametrize("name,cols,prm", CASES, ids=[f"{c[0]}-{'-'.join(str(v) for v in c[2].values())}" for c in CASES])
def test_single_functions(pq, oracle, data, name, cols, prm):
    from polars_quant_amd import api
    api.wt_stats(reset=True)
    got = api.call(name, *[torch.from_numpy(data[c]).cuda() for c in cols], **prm)
    st = api.wt_stats()
    n = data["close"].shape[0]
    # outside the form's scope (the gated general path must be the tiled body): rings above 64 KB, an odd row pitch
    outside = (name == "midprice" and prm.get("timeperiod", 0) > 30) or data["close"].shape[1] % 2 == 1   # (an odd pitch runs the 8-byte tiled body)
    assert (st == (0, 0, 0, 0)) if outside else (st[0] == n and st[3] == 0), f"the wave form did not run: {st}"
    exp = oracle.call(name, *[data[c] for c in cols], **prm)
    for k, (g, e) in enumerate(zip(got, exp)):
        same(f"{name}[{k}]{prm}", g.cpu().numpy(), e)


@pytest.mark.parametrize("name", ["dema", "tema", "ema", "trix", "rsi", "atr", "natr", "adx", "macd", "midpoint"])
def test_timeperiod_one(pq, oracle, name):
    """Period 1 is where the reference's if-chains degenerate: every seeding branch of DEMA / TEMA (overlap.rs:590-597, :1238-1240)
    falls on the first row and only the first is taken -- the later averages start from 0.0 and TEMA's first row is null -- so
    those two leave p = 1 to the lane-per-symbol ops that restate the chain literally (found by the ragged fuzz, seed 602); every
    other form handles it itself.  Regular and ragged batches."""
    from polars_quant_amd import api
    cols, pspec = pq.SPEC[name][0], pq.SPEC[name][1]
    prm = {pn: 1 for pn, kind, _ in pspec if kind == "i"}
    d = oracle.gen_ohlcv(SEED + 21, 70, 1500, 0)
    d["real"] = d["close"]
    d["close"][3, 700] *= 5.0   # a jump of more than 2 x: (x - e) + e is no longer exactly x
    d["real"] = d["close"]
    got = api.call(name, *[torch.from_numpy(d[c]).cuda() for c in cols], **prm)
    exp = oracle.call(name, *[d[c] for c in cols], **prm)
    for k, (g, e) in enumerate(zip(got, exp)):
        same(f"{name}[{k}] p=1", g.cpu().numpy(), e)
    lens = np.array([1500, 1100, 1, 0, 2000], dtype=np.int64)
    off = np.r_[0, np.cumsum(lens)]
    long = {k: np.ascontiguousarray(v[0]) for k, v in oracle.gen_ohlcv(SEED + 22, 1, int(off[-1]), 0).items()}
    long["real"] = long["close"]
    got = [g.cpu().numpy() for g in api.call(name, *[torch.from_numpy(long[c]).cuda() for c in cols], offsets=off, **prm)]
    for s_ in range(len(lens)):
        lo, hi = int(off[s_]), int(off[s_ + 1])
        if hi == lo:
            continue
        exp = oracle.call(name, *[long[c][lo:hi] for c in cols], **prm)
        for k, (g, e) in enumerate(zip(got, exp)):
            same(f"{name}[{k}] p=1 group {s_}", g[lo:hi], np.asarray(e).reshape(-1))


def test_fused_forms_and_pitched_columns(pq, oracle):
    """the multi-output entry points the suite records (pq_ema_all, pq_macd_pair, pq_dm_system_all, pq_dm_pair, pq_cmo_rsi) on a row
    pitch that is not the row count"""
    import ctypes as C
    from polars_quant_amd import api
    from polars_quant_amd._lib import Batch, check, lib
    n, T, S = 130, 2520, 2528
    d = oracle.gen_ohlcv(SEED + 1, n, T, 0)
    dev = {}
    for k, v in d.items():
        buf = torch.full((n, S), 777.0, dtype=torch.float64, device="cuda")
        buf[:, :T] = torch.from_numpy(v).cuda()
        dev[k] = buf
    b = Batch(n, T, S)
    P = lambda k: C.c_void_p(dev[k].data_ptr())
    mk = lambda m: [torch.full((n, S), 555.0, dtype=torch.float64, device="cuda") for _ in range(m)]
    V = lambda ts: [C.c_void_p(t.data_ptr()) for t in ts]
    h = api.ctx(0)
    L = lib()
    api.wt_stats(reset=True)
    o = mk(4); check(L.pq_ema_all(h, C.byref(b), P("close"), 30, *V(o)))
    cases = [("ema_all", o, [oracle.call(nm, d["close"], timeperiod=30)[0] for nm in ("ema", "dema", "tema", "trix")])]
    o = mk(6); check(L.pq_macd_pair(h, C.byref(b), P("close"), 12, 26, 9, 9, *V(o)))
    cases.append(("macd_pair", o, list(oracle.call("macd", d["close"])) + list(oracle.call("macdfix", d["close"], signalperiod=9))))
    o = mk(6); check(L.pq_macd_pair(h, C.byref(b), P("close"), 12, 26, 9, 5, *V(o)))
    cases.append(("macd_pair(9,5)", o, list(oracle.call("macd", d["close"])) + list(oracle.call("macdfix", d["close"], signalperiod=5))))
    o = mk(7); check(L.pq_dm_system_all(h, C.byref(b), P("high"), P("low"), P("close"), 14, *V(o)))
    cases.append(("dm_system_all", o, [oracle.call(nm, d["high"], d["low"], d["close"], timeperiod=14)[0]
                                       for nm in ("dx", "plus_di", "minus_di", "adx", "adxr", "atr", "natr")]))
    o = mk(2); check(L.pq_dm_pair(h, C.byref(b), P("high"), P("low"), 14, *V(o)))
    cases.append(("dm_pair", o, [oracle.call(nm, d["high"], d["low"], timeperiod=14)[0] for nm in ("plus_dm", "minus_dm")]))
    o = mk(2); check(L.pq_cmo_rsi(h, C.byref(b), P("close"), 14, *V(o)))
    cases.append(("cmo_rsi", o, [oracle.call(nm, d["close"], timeperiod=14)[0] for nm in ("cmo", "rsi")]))
    st = api.wt_stats()
    assert st[0] == n * 7 and st[3] == 0, st     # ema_all, macd_pair x 2, dmi + atr, dm_pair, rsi
    for nm, outs, exps in cases:
        for k, (t, e) in enumerate(zip(outs, exps)):
            g = t.cpu().numpy()
            same(f"{nm}[{k}]", g[:, :T], e)
            assert (g[:, T:] == 555.0).all(), f"{nm}[{k}]: rows beyond len were written"


def test_nulls_and_nans_take_the_gated_general_path(pq, oracle):
    """a symbol with a NULL or NaN input is left to the lane-per-symbol kernel launched behind; the other tiles stay in the wave form"""
    from polars_quant_amd import api
    n, T = 200, 2000
    d = oracle.gen_ohlcv(SEED + 2, n, T, 0)
    c = d["close"].copy()
    c[3, 100] = oracle.NULL
    c[3, 0] = oracle.NULL
    c[70, 1999] = oracle.NULL
    c[199, 500:520] = oracle.NULL
    h, l = d["high"].copy(), d["low"].copy()
    h[130, 7] = oracle.NULL
    for rep in range(2):     # second pass: the gate flags of the first have been consumed
        api.wt_stats(reset=True)
        for name, cols, prm in (("ema", (c,), dict(timeperiod=30)), ("tema", (c,), dict(timeperiod=10)), ("midpoint", (c,), dict(timeperiod=14)),
                                ("midprice", (h, l), dict(timeperiod=14)), ("atr", (h, l, c), dict(timeperiod=14))):
            got = api.call(name, *[torch.from_numpy(x).cuda() for x in cols], **prm)
            exp = oracle.call(name, *cols, **prm)
            same(f"{name} with nulls (pass {rep})", got[0].cpu().numpy(), exp[0])
        st = api.wt_stats()
        assert st[3] == 3 + 3 + 3 + 1 + 4 and st[0] == 5 * n - st[3], st
    # momentum.rs functions reject nulls at the API (N-B); a NaN that is not the NULL pattern flows through the general path
    c2 = d["close"].copy()
    c2[5, 300] = np.nan
    api.wt_stats(reset=True)
    (got,) = api.call("rsi", torch.from_numpy(c2).cuda(), timeperiod=14)
    (exp,) = oracle.call("rsi", c2, timeperiod=14)
    same("rsi with a NaN", got.cpu().numpy(), exp)
    assert api.wt_stats()[3] == 1


def test_flat_and_degenerate_series(pq, oracle):
    """constant prices (a recurrence that holds its value: every speculative chunk merges at once or is re-run), zero ranges (st == 0
    in calc_dm -> nulls), a step function, huge and tiny magnitudes"""
    from polars_quant_amd import api
    n, T = 8, 2520
    d = oracle.gen_ohlcv(SEED + 3, n, T, 0)
    for k in ("open", "high", "low", "close"):
        d[k][0] = 10.0
        d[k][1, :1200] = 5.0; d[k][1, 1200:] = 50.0
        d[k][2] *= 1e150
        d[k][3] *= 1e-150
        d[k][4, 1000:] = d[k][4, 1000:1001]
    for name, cols, prm in (("ema", ("close",), dict(timeperiod=30)), ("trix", ("close",), dict(timeperiod=30)), ("macd", ("close",), dict()),
                            ("rsi", ("close",), dict(timeperiod=14)), ("adx", ("high", "low", "close"), dict(timeperiod=14)),
                            ("adxr", ("high", "low", "close"), dict(timeperiod=14)), ("natr", ("high", "low", "close"), dict(timeperiod=14)),
                            ("plus_dm", ("high", "low"), dict(timeperiod=14)), ("midpoint", ("close",), dict(timeperiod=14))):
        got = api.call(name, *[torch.from_numpy(d[c]).cuda() for c in cols], **prm)
        exp = oracle.call(name, *[d[c] for c in cols], **prm)
        for k, (g, e) in enumerate(zip(got, exp)):
            same(f"{name}[{k}] degenerate", g.cpu().numpy(), e)


def test_short_warm_up_forces_re_runs_and_stays_exact(pq, oracle, monkeypatch):
    """PQ_WT_WARM shortens the warm-up: chunks fail the bit test and are re-run from their predecessors -- a speed knob only"""
    from polars_quant_amd import api
    d = oracle.gen_ohlcv(SEED + 4, 64, 2520, 0)
    monkeypatch.setenv("PQ_WT_WARM", "0.5")
    api.wt_stats(reset=True)
    for name, cols, prm in (("tema", ("close",), dict(timeperiod=30)), ("macd", ("close",), dict()), ("adx", ("high", "low", "close"), dict(timeperiod=14))):
        got = api.call(name, *[torch.from_numpy(d[c]).cuda() for c in cols], **prm)
        exp = oracle.call(name, *[d[c] for c in cols], **prm)
        for k, (g, e) in enumerate(zip(got, exp)):
            same(f"{name}[{k}] short warm-up", g.cpu().numpy(), e)
    st = api.wt_stats()
    assert st[1] > 0 and st[2] >= st[1], f"expected failing chunks at this warm-up: {st}"
    monkeypatch.delenv("PQ_WT_WARM")
    api.wt_stats(reset=True)
    api.call("tema", torch.from_numpy(d["close"]).cuda(), timeperiod=30)
    st = api.wt_stats()
    assert st[1] <= 64 * 3, f"too many speculative chunks fail at the default warm-up: {st}"


def test_ragged_groups_one_wavefront_each(pq, oracle):
    """RAGGED batches (pq_batch.offsets: the groups of `.over("symbol")`, any lengths, any starts) in the wave form: every group
    equals the oracle run on that group alone; groups with a NULL go to the gated gather body; empty and one-row groups are fine"""
    from polars_quant_amd import api
    rng = np.random.default_rng(31)
    lens = np.r_[2520, 1024, 3001, 1, 0, 4096, 700, 2000, 1501, 33, 2520, rng.integers(900, 3500, size=20)].astype(np.int64)
    off = np.r_[0, np.cumsum(lens)]
    d = oracle.gen_ohlcv(SEED + 8, 1, int(off[-1]), 0)
    d = {k: np.ascontiguousarray(v[0]) for k, v in d.items()}
    c_null = d["close"].copy()
    c_null[off[2] + 100] = oracle.NULL               # group 2 and group 7 carry a NULL
    c_null[off[7] + 5] = oracle.NULL
    for name, cols, prm, nulls in (("ema", ("close",), dict(timeperiod=30), True), ("trix", ("close",), dict(timeperiod=10), False),
                                   ("rsi", ("close",), dict(timeperiod=14), False), ("atr", ("high", "low", "close"), dict(timeperiod=14), True),
                                   ("natr", ("high", "low", "close"), dict(timeperiod=5), False), ("plus_dm", ("high", "low"), dict(timeperiod=14), False),
                                   ("midpoint", ("close",), dict(timeperiod=14), True), ("macd", ("close",), dict(), False),
                                   ("adx", ("high", "low", "close"), dict(timeperiod=14), False), ("tema", ("close",), dict(timeperiod=9), True)):
        src = dict(d)
        if nulls:
            src["close"] = c_null
        api.wt_stats(reset=True)
        got = [t.cpu().numpy() for t in api.call(name, *[torch.from_numpy(src[c]).cuda() for c in cols], offsets=off, **prm)]
        st = api.wt_stats()
        live = int((lens > 0).sum())
        assert st[0] + st[3] == live and st[3] == (2 if nulls else 0), (name, st)
        for gi in range(len(lens)):
            lo, hi = off[gi], off[gi + 1]
            if hi == lo:
                continue
            exp = oracle.call(name, *[src[c][lo:hi] for c in cols], **prm)
            for k, (g, e) in enumerate(zip(got, exp)):
                same(f"{name}[{k}] group {gi} (len {hi - lo})", g[lo:hi], np.asarray(e).reshape(-1))


def test_default_policy_uses_the_forms_that_win(pq, oracle, monkeypatch):
    from polars_quant_amd import api
    monkeypatch.delenv("PQ_WT_ALL")
    monkeypatch.delenv("PQ_MIDPRICE_SEQ")
    d = oracle.gen_ohlcv(SEED + 6, 64, 2520, 0)
    g = {k: torch.from_numpy(v).cuda() for k, v in d.items()}
    for nm, cols, used in (("ema", ("close",), True), ("trix", ("close",), True), ("rsi", ("close",), True), ("atr", ("high", "low", "close"), True),
                           ("midpoint", ("close",), True), ("midprice", ("high", "low"), False), ("tema", ("close",), False), ("macd", ("close",), False), ("adx", ("high", "low", "close"), False)):
        api.wt_stats(reset=True)
        got = api.call(nm, *[g[c] for c in cols])
        assert (api.wt_stats()[0] == 64) == used, nm
        for a, e in zip(got, oracle.call(nm, *[d[c] for c in cols])):
            same(nm, a.cpu().numpy(), e)


def test_wave_form_equals_lane_form(pq, oracle, monkeypatch):
    from polars_quant_amd import api
    d = oracle.gen_ohlcv(SEED + 5, 100, 2520, 0)
    g = {k: torch.from_numpy(v).cuda() for k, v in d.items()}
    calls = (("ema", ("close",), {}), ("dema", ("close",), {}), ("tema", ("close",), {}), ("trix", ("close",), {}), ("macd", ("close",), {}),
             ("rsi", ("close",), {}), ("adx", ("high", "low", "close"), {}), ("adxr", ("high", "low", "close"), {}), ("natr", ("high", "low", "close"), {}),
             ("minus_dm", ("high", "low"), {}), ("midpoint", ("close",), {}), ("midprice", ("high", "low"), {}))
    wave = {nm: [t.cpu().numpy() for t in api.call(nm, *[g[c] for c in cols], **prm)] for nm, cols, prm in calls}
    monkeypatch.setenv("PQ_NO_WT", "1")
    api.wt_stats(reset=True)
    lane = {nm: [t.cpu().numpy() for t in api.call(nm, *[g[c] for c in cols], **prm)] for nm, cols, prm in calls}
    assert api.wt_stats()[0] == 0
    for nm in wave:
        for a, b_ in zip(wave[nm], lane[nm]):
            same(nm, a, b_)
