"""-m gpu: the wave-per-symbol backtest (csrc/ops_backtest_wave.h) against the CPU oracle.

Reference semantics: src/backtest/vectorized.rs:124-194 (scan), src/backtest/metrics.rs:7-152 (summary), momentum.rs:250-283 + D-8
(MACD-cross signals).  position / cash / equity, max_drawdown, max_profit, win_rate and total_trades are compared BIT FOR BIT;
annualized_return / alpha / beta / sharpe (pow, ordered sums) at the north-star tolerance of 1e-12.

What these cases reach that the 300-row cases of test_gpu_parity.py cannot: series long enough for lanes to start their chunk
speculatively (len >= ~15 chunks), chunks that FAIL the bit test and are re-run (forced with PQ_BT_WARM_CHUNKS, and naturally by
flat prices, whose EMA trajectories never contract), nulls / NaN / non-positive prices inside and in front of a series, pools that
cannot afford one share (a buy signal that is not an event), every block / chunk boundary of len, and the lane-form fallback.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
SEED = 0x5EED0003
EXACT, TOL = (1, 5, 6, 7), (0, 2, 3, 4)


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


NULLB = np.uint64(0x7FF80000504E554C)


def same_bits(g, e):   # equal bits; two NaNs match whatever their payload, except that a NULL only matches a NULL
    return (bits(g) == bits(e)) | (np.isnan(g) & np.isnan(e) & ((bits(g) == NULLB) == (bits(e) == NULLB)))


def check_summary(s, es, tag):
    ok = ~np.isnan(es).any(axis=1)
    for k in EXACT:
        assert (bits(s[ok, k]) == bits(es[ok, k])).all(), (tag, "exact column", k, np.argwhere(bits(s[ok, k]) != bits(es[ok, k]))[:3].tolist())
    for k in TOL:
        np.testing.assert_allclose(s[ok, k], es[ok, k], rtol=1e-12, atol=1e-13, err_msg=f"{tag} column {k}")
    assert np.isnan(s[~ok]).any(axis=1).all(), tag


def special_prices(oracle, n, T):
    d = oracle.gen_ohlcv(SEED, n, T, 0)
    close = d["close"].copy()
    if T >= 8 and n >= 8:
        close[1, :] = 50.0                                # flat: the speculative EMA never merges -> every chunk is re-run
        close[2, T // 3: T // 3 + 5] = oracle.NULL        # interior nulls
        close[3, : min(40, T // 2)] = oracle.NULL         # leading nulls: seeds shift
        close[4, T // 2] = np.nan                         # a NaN poisons the averages from there on
        close[5, min(7, T - 1)] = -1.0                    # invalid price: state untouched
        close[6, T // 2:] = close[6, T // 2]              # goes flat half way
        close[7, ::3] = oracle.NULL                       # every third row null
    return d, close


@pytest.fixture(params=["auto", "1"], ids=["waves-auto", "one-wave"])
def waves(request, monkeypatch):
    """The kernels share a symbol among four wavefronts (staging, fill, summary) except on batches of 4-5 x #CUs symbols and on
    series beyond 4096 rows: "1" forces the one-wave form, which small test batches would otherwise never run."""
    if request.param == "auto":
        monkeypatch.delenv("PQ_BT_WAVES", raising=False)
    else:
        monkeypatch.setenv("PQ_BT_WAVES", request.param)
    return request.param


@pytest.mark.parametrize("T", [2520, 4096, 1000, 641, 640, 129, 65, 64, 63, 2, 1, 4097, 5040, 8191, 8192])  # > 4096: two mask words per lane
def test_macd_cross_wave_form(pq, oracle, T, monkeypatch, waves):
    from polars_quant_amd import api
    n = 24
    _, close = special_prices(oracle, n, T)
    ebuy, esell = oracle.macd_cross_signals(close)
    cases = [dict(), dict(initial_capital=30.0), dict(buy_slippage=0.01, sell_slippage=0.02, position_size=0.5, min_commission=1.0)]
    for warm in (None, "1", "3"):
        if warm is None:
            monkeypatch.delenv("PQ_BT_WARM_CHUNKS", raising=False)
        else:
            monkeypatch.setenv("PQ_BT_WARM_CHUNKS", warm)
        api.backtest_wave_stats(reset=True)
        for kw in cases:
            epos, ecash, eeq, es = oracle.backtest(close, ebuy, esell, **kw)
            pos, cash, eq, s = api.backtest_macd_cross(torch.from_numpy(close).cuda(), **kw)
            for nm, g, e in (("position", pos, epos), ("cash", cash, ecash), ("equity", eq, eeq)):
                ok = same_bits(g.cpu().numpy(), e)
                assert ok.all(), (T, warm, kw, nm, np.argwhere(~ok)[:4].tolist())
            check_summary(s.cpu().numpy(), es, (T, warm, tuple(kw)))
            _, _, _, s2 = api.backtest_macd_cross(torch.from_numpy(close).cuda(), want_curves=False, **kw)
            assert same_bits(s2.cpu().numpy(), s.cpu().numpy()).all()
        st = api.backtest_wave_stats()
        assert st[0] == 2 * len(cases) * n, "the wave form must have run"
        if warm == "1" and T >= 1000:
            assert st[1] > 0 and st[2] >= st[1], f"a one-chunk warm-up must fail the bit test somewhere: {st}"
    monkeypatch.delenv("PQ_BT_WARM_CHUNKS", raising=False)
    if T >= 2520:  # at the default warm-up failures are rare on generic series
        g = oracle.gen_ohlcv(SEED + 9, 64, T, 0)["close"]
        api.backtest_wave_stats(reset=True)
        api.backtest_macd_cross(torch.from_numpy(g).cuda())
        st = api.backtest_wave_stats()
        assert st[0] == 64 and st[1] <= 64, st


@pytest.mark.parametrize("T", [2520, 4096, 777, 65, 1, 4097, 5040, 8192])
def test_vectorized_wave_form(pq, oracle, T, waves):
    from polars_quant_amd import api
    n = 20
    d, price = special_prices(oracle, n, T)
    rng = np.random.default_rng(T)
    for dens in (0.05, 0.6, 1.0):  # 1.0: a buy and a sell signal on every row -> an event on every row
        buy = (rng.random(price.shape) < dens).astype(np.uint8)
        sell = (rng.random(price.shape) < dens).astype(np.uint8)
        bench = d["open"].copy()
        for kw, bm in ((dict(), bench), (dict(initial_capital=50.0), None),
                       (dict(buy_slippage=0.01, sell_slippage=0.02, position_size=0.5, min_commission=1.0), bench)):
            epos, ecash, eeq, es = oracle.backtest(price, buy, sell, benchmark=bm, **kw)
            api.backtest_wave_stats(reset=True)
            pos, cash, eq, s = api.backtest_vectorized(torch.from_numpy(price).cuda(), torch.from_numpy(buy).cuda(), torch.from_numpy(sell).cuda(),
                                                       benchmark=None if bm is None else torch.from_numpy(bm).cuda(), **kw)
            assert api.backtest_wave_stats()[0] == n
            for nm, g, e in (("position", pos, epos), ("cash", cash, ecash), ("equity", eq, eeq)):
                ok = same_bits(g.cpu().numpy(), e)
                assert ok.all(), (T, dens, kw, nm, np.argwhere(~ok)[:4].tolist())
            check_summary(s.cpu().numpy(), es, (T, dens, tuple(kw)))


def test_wave_form_equals_lane_form_and_long_series_fall_back(pq, oracle, monkeypatch):
    """len > 8192 keeps the lane-per-symbol kernels; PQ_BT_LANE_FORM forces them for an A/B on the same inputs."""
    from polars_quant_amd import api
    close = oracle.gen_ohlcv(SEED + 1, 70, 2520, 0)["close"]
    w = [t.cpu().numpy() for t in api.backtest_macd_cross(torch.from_numpy(close).cuda())]
    monkeypatch.setenv("PQ_BT_LANE_FORM", "1")
    api.backtest_wave_stats(reset=True)
    l = [t.cpu().numpy() for t in api.backtest_macd_cross(torch.from_numpy(close).cuda())]
    assert api.backtest_wave_stats()[0] == 0
    monkeypatch.delenv("PQ_BT_LANE_FORM")
    for k in range(3):
        assert (bits(w[k]) == bits(l[k])).all()
    for k in EXACT:
        assert (bits(w[3][:, k]) == bits(l[3][:, k])).all()
    np.testing.assert_allclose(w[3], l[3], rtol=1e-12, atol=1e-13)
    long = oracle.gen_ohlcv(SEED + 2, 6, 8200, 0)["close"]
    ebuy, esell = oracle.macd_cross_signals(long)
    epos, ecash, eeq, es = oracle.backtest(long, ebuy, esell)
    api.backtest_wave_stats(reset=True)
    pos, cash, eq, s = api.backtest_macd_cross(torch.from_numpy(long).cuda())
    assert api.backtest_wave_stats()[0] == 0
    assert (bits(eq.cpu().numpy()) == bits(eeq)).all() and (bits(pos.cpu().numpy()) == bits(epos)).all()


def test_full_size_config3_wave_backtest(pq, oracle, waves):
    """BASELINE config 3 (5000 x 2520) on the bench's layout (row pitch 2528): EVERY symbol against the oracle, plus the
    size-independent properties."""
    from polars_quant_amd import api
    from polars_quant_amd.synthetic import gen_ohlcv
    N, T, STRIDE = 5000, 2520, 2528
    close = gen_ohlcv(0x5EED0002, N, T, 0)["close"]
    buf = torch.zeros((N, STRIDE), dtype=torch.float64, device="cuda")
    buf[:, :T] = torch.from_numpy(close).cuda()
    api.backtest_wave_stats(reset=True)
    pos, cash, eq, s = api.backtest_macd_cross(buf[:, :T])
    st = api.backtest_wave_stats()
    assert st[0] == N
    assert st[1] <= N // 4, f"too many speculative chunks fail at the default warm-up: {st}"
    pos, cash, eq, s = (t.cpu().numpy() for t in (pos, cash, eq, s))
    ebuy, esell = oracle.macd_cross_signals(close)
    epos, ecash, eeq, es = oracle.backtest(close, ebuy, esell)
    assert (bits(pos) == bits(epos)).all() and (bits(cash) == bits(ecash)).all() and (bits(eq) == bits(eeq)).all()
    check_summary(s, es, "config3")
    # properties over every symbol: equity identity row by row, whole shares, trades counted = position changes / 2 (rounded up)
    assert (bits(eq) == bits(cash + pos * close)).all()
    assert (pos == np.floor(pos)).all() and (pos >= 0).all()
    opens = ((pos[:, 1:] > 0) & (pos[:, :-1] == 0)).sum(axis=1) + (pos[:, 0] > 0)
    assert (opens == s[:, 7]).all()
    assert (s[:, 1] >= 0).all() and (s[:, 1] <= 1).all() and (s[:, 6] >= 0).all() and (s[:, 6] <= 1).all()
