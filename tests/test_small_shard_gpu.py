"""Recorded suites on SMALL shards (one rank's share of BASELINE's 5 000 symbols: 625 at 8 GPUs, 1 250 at 4): the recording takes the
multi-output forms apart, orders its chains by data dependencies instead of phase barriers (csrc/suite.hip suite_launch_small), splits
the Hilbert job in time (csrc/fused.hip) and writes MAMA(fastlimit = 0) without walking the pipeline (csrc/misc.hip).  Every output of
every symbol must still be the oracle's -- bit for bit where the full-size suite is, within 1e-12 for the transcendental columns."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from test_gpu_parity import TRANSCENDENTAL, _pitched, assert_same, bits, pq  # noqa: E402,F401  (pq: the module fixture)

TT, STRIDE = 2520, 2528
NULL = np.frombuffer(np.uint64(0x7FF80000504E554C).tobytes(), dtype=np.float64)[0]


def _compare_everything(pq, oracle, d, st, ch=320):
    from concurrent.futures import ThreadPoolExecutor
    N = d["close"].shape[0]
    periods = st.periods[:1].cpu().numpy()

    def expect(lo):
        sub = {k: np.ascontiguousarray(v[lo:lo + ch]) for k, v in d.items()}
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
        for lo, sub, res in pool.map(expect, range(0, N, ch)):
            hi = lo + sub["close"].shape[0]
            for name in pq.SPEC:
                for (oname, _), got in zip(pq.SPEC[name][2], st.out[name]):
                    # (the price level is the scale of the phasor components: rows whose price is NaN / inf are judged against |expected| alone --
                    #  the pipeline's outputs lag its input by three rows, so such rows can still hold finite values)
                    assert_same(f"{name}.{oname}{{{lo}:{hi}}}", got[lo:hi].cpu().numpy(), res[(name, oname)],
                                exact=name not in TRANSCENDENTAL, price=np.nan_to_num(sub["close"], nan=0.0, posinf=0.0, neginf=0.0))
                    compared += 1
            for nm in pq.PATTERN_NAMES:
                assert (st.pat[nm][lo:hi].cpu().numpy() == res[("pattern", nm)]).all(), (nm, lo)
            epos, ecash, eeq, es = res["bt"]
            pos, cash, eq = (t[lo:hi].cpu().numpy() for t in st.bt)
            same = lambda a, b: ((bits(a) == bits(b)) | (np.isnan(a) & np.isnan(b))).all()
            assert same(eq, eeq) and same(pos, epos) and same(cash, ecash), lo
            np.testing.assert_allclose(st.summary[lo:hi].cpu().numpy(), es, rtol=1e-12, atol=1e-13)
    assert compared == (N + ch - 1) // ch * sum(len(v[2]) for v in pq.SPEC.values())


@pytest.mark.parametrize("N", [625, 1250], ids=["8gpu-shard-625", "4gpu-shard-1250"])
def test_small_shard_suite_every_output_of_every_symbol(pq, oracle, N):
    from polars_quant_amd.suite import Suite
    d = oracle.gen_ohlcv(0x5EED0002, N, TT, 0)
    g = _pitched(d, STRIDE)
    st = Suite(N, TT, "cuda:0", stride=STRIDE)
    st.record(g)
    info = st.info()
    assert info["phases"] >= 3, info      # the composites were taken apart into dependent links (else this is not the small-shard plan)
    assert info["seq_jobs"] >= 40, info   # ... and the Hilbert job into chunks
    st.run(); st.run()
    torch.cuda.synchronize()
    _compare_everything(pq, oracle, d, st)
    # replay idempotence, and the same columns as the full-chip plan of the same step (PQ_SMALL_SHARD_TILES=0), bit for bit except the
    # time-split Hilbert columns (a few ulp by construction)
    import os
    first = {name: [t.clone() for t in ts] for name, ts in st.out.items()}
    st.run(); torch.cuda.synchronize()
    for name, ts in st.out.items():
        for a, b in zip(first[name], ts):
            assert torch.equal(a.view(torch.int64), b.view(torch.int64)), name
    st.close()
    os.environ["PQ_SMALL_SHARD_TILES"] = "0"
    try:
        big = Suite(N, TT, "cuda:0", stride=STRIDE)
        big.record(g)
        assert big.info()["phases"] == 1
        big.run(); torch.cuda.synchronize()
    finally:
        del os.environ["PQ_SMALL_SHARD_TILES"]
    for name, ts in big.out.items():
        for (oname, _), a, b in zip(pq.SPEC[name][2], first[name], ts):
            if name in ("ht_dcperiod", "ht_dcphase", "ht_phasor", "ht_sine"):
                assert_same(f"{name}.{oname}{{small against full plan}}", a.cpu().numpy(), b.cpu().numpy(), exact=False, price=d["close"])
            else:
                assert torch.equal(a.view(torch.int64), b.view(torch.int64)), (name, oname)
    big.close()


def test_small_shard_fallbacks_nulls_nans_infinities_and_flat_series(pq, oracle):
    """series that the fast forms must hand to their gated general paths: a NaN that a later Hilbert chunk never sees (row 100), one
    inside a warm-up (row 1 000), an infinity, a run of NaNs as long as a chunk, values beyond MAMA's finiteness bound, a flat series (the
    period recurrence sits on its clamps), a series that is flat until row 1 300 -- on tiles 0, 1, 2 and the last, partial one; tile 3
    stays clean.  (NaN VALUES, not NULLs: the momentum / cycle family of the reference refuses a column with nulls -- `cont_slice()?`,
    momentum.rs:141-143 -- so a NULL is outside the domain of half of the suite; MAMA's null rule has its own test below.)"""
    from polars_quant_amd.suite import Suite
    N = 330
    d = oracle.gen_ohlcv(0x5EED0601, N, TT, 0)

    def poke(sym, row, val, cols=("close",)):
        for c in cols:
            d[c][sym, row] = val
    poke(3, 100, np.nan)
    poke(70, 1000, np.nan)
    poke(71, 1500, np.inf)
    poke(130, 900, np.nan, cols=("high",))
    poke(131, 2519, np.nan)
    for c in ("open", "high", "low", "close"):
        d[c][140] *= 1e200          # beyond MAMA's finiteness bound: the pipeline overflows, the walk decides
        d[c][5] = 100.0             # flat
        d[c][329, :1300] = 50.0     # flat, then moving
    d["close"][328, 640:1280] = np.nan  # a chunk's worth of NaN rows
    g = _pitched(d, STRIDE)
    st = Suite(N, TT, "cuda:0", stride=STRIDE)
    st.record(g)
    assert st.info()["phases"] >= 3
    st.run(); st.run()
    torch.cuda.synchronize()
    _compare_everything(pq, oracle, d, st, ch=110)
    st.close()


def test_mama_at_the_wrappers_default_limits_is_the_walks_result(pq, oracle):
    """pq_mama(fastlimit = 0): the row-parallel form + gated walk against the oracle's walk and against the device walk (PQ_MAMA_WALK=1),
    bit for bit, on clean series, on series with NaN / inf / NULL / huge values, on short series, direct and ragged"""
    import os
    from polars_quant_amd import api
    for n, T in ((130, 400), (70, 31), (70, 32), (3, 2520)):
        d = oracle.gen_ohlcv(0x5EED0602 + T, n, T, 0)
        x = d["close"].copy()
        if T >= 400 and n > 66:
            x[1, 50] = np.nan; x[64, 399] = np.inf; x[65, 0] = NULL; x[2] *= 1e150; x[66] *= 1e139
        for fl, sl in ((0.0, 0.0), (0.0, 0.05), (-0.0, 0.0), (0.0, float("nan"))):
            exp = oracle.call("mama", x, fastlimit=fl, slowlimit=sl)
            got = api.call("mama", torch.from_numpy(x).cuda(), fastlimit=fl, slowlimit=sl)
            os.environ["PQ_MAMA_WALK"] = "1"
            try:
                walk = api.call("mama", torch.from_numpy(x).cuda(), fastlimit=fl, slowlimit=sl)
            finally:
                del os.environ["PQ_MAMA_WALK"]
            for k, nm in enumerate(("mama", "fama")):
                a, w, e = got[k].cpu().numpy(), walk[k].cpu().numpy(), exp[k]
                assert ((bits(a) == bits(w)) | (np.isnan(a) & np.isnan(w))).all(), (nm, n, T, fl, sl)
                assert_same(f"{nm}{{{n}x{T}}}", a, e, exact=False, price=x)
                clean = np.isfinite(x).all(axis=1) & (np.abs(np.where(bits(x) == bits(np.array([NULL]))[0], 0.0, x)).max(axis=1) <= 1e140)
                if T >= 32:
                    assert (bits(a[clean][:, 31:]) == 0).all()      # +0.0, not -0.0


def test_small_shard_schedule_variants_write_the_same_columns(pq, oracle):
    """The small-shard schedule orders its launches by data dependencies (csrc/suite.hip small_deps): whatever ORDER a stream issues its
    parts in -- the links' job grid before or after their ROW launches, RSI under STOCHRSI as a wave-per-symbol launch or as a job -- every
    column must come out bit-identical (a missing dependency would show as a column computed from stale inputs under one of the orders)."""
    import os
    from polars_quant_amd.suite import Suite
    N = 330
    d = oracle.gen_ohlcv(0x5EED0603, N, TT, 0)
    g = _pitched(d, STRIDE)

    def run(env):
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            st = Suite(N, TT, "cuda:0", stride=STRIDE)
            st.record(g)
            assert st.info()["phases"] >= 3
            for t in [x for ts in st.out.values() for x in ts]:
                t.fill_(-7)                       # poison: every row must be produced by the replay
            st.run(); st.run(); st.run()
            torch.cuda.synchronize()
            cols = {name: [t.clone() for t in ts] for name, ts in st.out.items()}
            cols["__bt"] = [t.clone() for t in st.bt] + [st.summary.clone()]
            st.close()
            return cols
        finally:
            for k, v in old.items():
                if v is None: os.environ.pop(k, None)
                else: os.environ[k] = v
    base = run({})
    for env in ({"PQ_SMALL_GRIDS_FIRST": "0"}, {"PQ_SMALL_GRIDS_FIRST": "1"}, {"PQ_NO_WT_SMALL": "1"}, {"PQ_NO_WT_SMALL": "1", "PQ_SMALL_GRIDS_FIRST": "0"}):
        other = run(env)
        for name, ts in base.items():
            for k, (a, b) in enumerate(zip(ts, other[name])):
                assert torch.equal(a.view(torch.int64), b.view(torch.int64)), (env, name, k)
