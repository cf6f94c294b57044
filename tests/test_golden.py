"""Committed fixtures tests/golden/*.npz (written by scripts/make_golden.py from the CPU oracle; the reference holds no
vectors -- see that script's header).  not-gpu: the oracle still reproduces them bit for bit, so an edit that shifts its
semantics shows up as a changed fixture in review.  gpu: the HIP path is checked against the FILES (not the live oracle)."""
import importlib.util
from pathlib import Path

import numpy as np
import pytest
from tolerance import SCALE_OF

ROOT = Path(__file__).resolve().parent.parent
GOLD = sorted((ROOT / "tests" / "golden").glob("oracle_*_4x200.npz"))
TRANSC = ("ht_dcperiod", "ht_dcphase", "ht_phasor", "ht_sine", "mama", "returns@alt", "backtest.summary", "leveraged.summary")


def _mk():
    spec = importlib.util.spec_from_file_location("make_golden", ROOT / "scripts" / "make_golden.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _same(name, g, e, exact, scale=0.0):
    g, e = np.asarray(g), np.asarray(e)
    assert g.shape == e.shape and g.dtype == e.dtype, (name, g.shape, e.shape, g.dtype, e.dtype)
    if e.dtype != np.float64:
        assert (g == e).all(), name
        return
    gb, eb = g.view(np.uint64), e.view(np.uint64)
    nullb = np.uint64(0x7FF80000504E554C)
    assert ((gb == nullb) == (eb == nullb)).all(), f"{name}: null masks differ"
    if exact:
        assert ((gb == eb) | (np.isnan(g) & np.isnan(e))).all(), f"{name}: not bit-identical"
    else:  # transcendental rows: 1e-12 relative to max(|expected|, scale) -- the scale table of tests/test_gpu_parity.py
        ok = eb != nullb
        sc = np.broadcast_to(scale, e.shape)[ok]
        err = np.abs(g[ok] - e[ok]) / np.maximum(np.maximum(np.abs(e[ok]), sc), 1e-300)
        err[np.isnan(g[ok]) & np.isnan(e[ok])] = 0
        assert (err <= 1e-12).all(), f"{name}: max error {np.nanmax(err):.3e}"


def test_fixtures_exist():
    assert len(GOLD) == 3, "run python scripts/make_golden.py"


@pytest.mark.parametrize("path", GOLD, ids=[p.stem for p in GOLD])
def test_oracle_reproduces_the_committed_fixtures(oracle, path):
    mk = _mk()
    z = np.load(path)
    d = {k[3:]: z[k] for k in z.files if k.startswith("in.")}
    res = mk.compute(d, null_bearing="nulls" in path.stem)
    want = {k[4:] for k in z.files if k.startswith("out.")}
    assert set(res) == want, sorted(set(res) ^ want)[:10]
    for k, v in res.items():
        _same(k, v, z["out." + k], exact=True)     # same libm, same compiler flags: bit for bit, transcendental rows included
    gen = mk.datasets()[path.stem.split("_")[1]]    # the inputs are reproducible from the seeds
    for k, v in d.items():
        assert (np.asarray(gen[k]).view(np.uint64) == v.view(np.uint64)).all(), k


@pytest.mark.gpu
@pytest.mark.parametrize("path", GOLD, ids=[p.stem for p in GOLD])
def test_hip_path_against_the_committed_fixtures(path):
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from polars_quant_amd import api
    from polars_quant_amd._spec import EXTRA, PATTERN_NAMES, SPEC
    mk = _mk()
    z = np.load(path)
    d = {k[3:]: torch.from_numpy(z[k]).cuda() for k in z.files if k.startswith("in.")}
    spec = {**SPEC, **EXTRA}
    checked = 0
    for key in (k[4:] for k in z.files if k.startswith("out.")):
        fn, oname = key.split(".", 1)
        name, _, tag = fn.partition("@")
        if name in spec:
            cols, pspec, outs, _fam = spec[name]
            prm = {p: mk.ALT[p] for p, _, _ in pspec if p in mk.ALT} if tag else {}
            got = api.call(name, *[d[c] for c in cols], **prm)
            g = got[[o for o, _ in outs].index(oname)].cpu().numpy()
            sc = SCALE_OF.get(f"{name}.{oname}", 0.0)
            if isinstance(sc, str):
                sc = np.abs(z["in.close"])
            _same(key, g, z["out." + key], exact=not any(fn.startswith(t.split("@")[0]) and (("@" not in t) or fn == t) for t in TRANSC), scale=sc)
            checked += 1
        elif name in PATTERN_NAMES:
            g = api.cdl(name, d["open"], d["high"], d["low"], d["close"]).cpu().numpy()
            assert (g == z["out." + key]).all(), key
            checked += 1
    assert checked >= (60 if "nulls" in path.stem else 200)
    if "nulls" not in path.stem:
        buy, sell = api.macd_cross_signals(d["close"])
        assert (buy.cpu().numpy() == z["out.macd_cross.buy"]).all() and (sell.cpu().numpy() == z["out.macd_cross.sell"]).all()
        pos, cash, eq, summ = api.backtest_vectorized(d["close"], buy, sell, benchmark=d["open"], buy_slippage=0.01, position_size=0.5)
        for k, g in (("position", pos), ("cash", cash), ("equity", eq)):
            _same("backtest." + k, g.cpu().numpy(), z["out.backtest." + k], exact=True)
        np.testing.assert_allclose(summ.cpu().numpy(), z["out.backtest.summary"], rtol=1e-12, atol=1e-13)
        lev = api.backtest_leveraged(d["close"], buy, sell, benchmark=d["open"][0], max_trades=8, leverage=2.0, slippage=0.001)
        for k in ("cash", "stock_value", "total_value"):
            _same("leveraged." + k, lev[k].cpu().numpy(), z["out.leveraged." + k], exact=True)
        assert (lev["trade_count"].cpu().numpy() == z["out.leveraged.trade_count"]).all()
