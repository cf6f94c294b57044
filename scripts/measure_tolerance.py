"""Measured error of every output that passes through a device transcendental (atan / sin / pow / log) against the oracle, on the
three SURVEY 8(d) data sets and at a larger shape: max relative error |g - e| / |e| and, for the outputs the parity tests judge
against their natural scale (tests/test_gpu_parity.py SCALE_OF), max |g - e| / scale.  Prints one JSON object
(profiles/r03_tolerance.json).  Everything NOT listed here is compared bit for bit by the tests."""
import json
import sys

sys.path.insert(0, ".")
import numpy as np
import torch

from oracle import pq_oracle as o
from polars_quant_amd import api

SEED = 0x5EED0002
NULLB = np.uint64(0x7FF80000504E554C)


def err(g, e, scale=None):
    g, e = np.asarray(g, dtype=np.float64), np.asarray(e, dtype=np.float64)
    ok = (e.view(np.uint64) != NULLB) & np.isfinite(e)
    if not ok.any():
        return {"rel": 0.0, "abs": 0.0}
    d = np.abs(g[ok] - e[ok])
    out = {"rel": float(np.max(d / np.maximum(np.abs(e[ok]), 1e-300))), "abs": float(d.max())}
    if scale is not None:
        sc = np.broadcast_to(scale, e.shape)[ok] if not np.isscalar(scale) else scale
        out["scaled"] = float(np.max(d / np.maximum(np.maximum(np.abs(e[ok]), sc), 1e-300)))
    return out


def run(name, d, **kw):
    cols = api.SPEC[name][0] if hasattr(api, "SPEC") else __import__("polars_quant_amd").SPEC[name][0]
    ins = [torch.from_numpy(d[c]).cuda() for c in cols]
    res = api.call(name, *ins, **kw)
    torch.cuda.synchronize()
    return [r.cpu().numpy() for r in res]


def main():
    import polars_quant_amd as pq
    out = {}
    sets = {"clean_70x304": o.gen_ohlcv(SEED, 70, 304, 0), "rich_70x304": o.gen_ohlcv(SEED + 1, 70, 304, 1),
            "clean_256x2520": o.gen_ohlcv(SEED, 256, 2520, 0)}
    nul = o.gen_ohlcv(SEED + 2, 70, 304, 0)
    rng = np.random.default_rng(11)
    for k in nul:
        a = nul[k]
        m = rng.random(a.shape) < 0.01
        m[:, :3] = True
        a[m] = o.NULL
    sets["nulls_70x304"] = nul
    for tag, d in sets.items():
        d["real"] = d["close"]
        price = np.abs(d["close"])
        res = {}
        for name, scales in (("ht_dcperiod", [50.0]), ("ht_dcphase", [360.0]), ("ht_phasor", [price, price]), ("ht_sine", [1.0, 1.0]),
                             ("mama", [price, price])):
            if "nulls" in tag and pq.SPEC[name][3] == "N-B":
                continue
            exp = o.call(name, *[d[c] for c in pq.SPEC[name][0]])
            got = run(name, d)
            for (oname, _), g, e, sc in zip(pq.SPEC[name][2], got, exp, scales):
                res[f"{name}.{oname}"] = err(g, e, sc)
        if "nulls" not in tag:
            close = d["close"]
            eb, es_ = o.macd_cross_signals(close)
            _, _, _, esum = o.backtest(close, eb, es_, benchmark=d["open"])
            _, _, _, gsum = api.backtest_vectorized(torch.from_numpy(close).cuda(), torch.from_numpy(eb).cuda(), torch.from_numpy(es_).cuda(),
                                                    benchmark=torch.from_numpy(d["open"]).cuda())
            gsum = gsum.cpu().numpy()
            for k, nm in ((0, "annualized_return"), (2, "alpha"), (3, "beta"), (4, "sharpe_ratio")):
                res[f"summary.{nm}"] = err(gsum[:, k], esum[:, k], 1.0)
            if hasattr(api, "returns"):
                g = api.returns(torch.from_numpy(close).cuda(), 1, "log")
                g = (g[0] if isinstance(g, (tuple, list)) else g).cpu().numpy()
                e = o.call("returns", close, period=1, method=1)[0] if "returns" in getattr(o, "EXTRA", ()) else None
                if e is not None:
                    res["returns.log"] = err(g, e, 1.0)
        out[tag] = res
    worst = {}
    for tag, res in out.items():
        for k, v in res.items():
            w = worst.setdefault(k, {"rel": 0.0, "abs": 0.0, "scaled": 0.0})
            for f in w:
                w[f] = max(w[f], v.get(f, 0.0))
    print(json.dumps({"worst_over_data_sets": worst, "per_data_set": out}, indent=1))


if __name__ == "__main__":
    main()
