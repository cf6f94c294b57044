"""Randomised parity sweep (GPU box): random shapes, parameters and null patterns for every function, GPU vs oracle.
Deterministic per seed.  `python scripts/fuzz_parity.py <seed> <iterations>`; tests/test_gpu_parity.py runs sweep() in-process."""
import sys

sys.path.insert(0, ".")
import numpy as np
import torch

from oracle import pq_oracle as oracle
from polars_quant_amd import api
from polars_quant_amd._spec import I, SPEC

TRANSC = {"ht_dcperiod", "ht_dcphase", "ht_phasor", "ht_sine", "mama"}


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint64 if a.dtype == np.float64 else np.uint32)


def _case(rng, name, log) -> int:
    cols, pspec, outs, fam = SPEC[name]
    N, T = int(rng.integers(1, 140)), int(rng.integers(1, 260))
    d = oracle.gen_ohlcv(int(rng.integers(1, 1 << 30)), N, T, int(rng.integers(0, 2)))
    d["real"] = d["close"]
    d["periods"] = rng.integers(0, 40, size=(N, T)).astype(np.float64)
    if fam in ("N-A", "N-C", "N-0") and name != "stochrsi" and rng.random() < 0.4:
        for k in ("open", "high", "low", "close", "volume", "real"):
            m = rng.random((N, T)) < 0.03
            d[k] = d[k].copy()
            d[k][m] = oracle.NULL
    params = {}
    for pname, kind, _default in pspec:
        if kind == I:
            if "matype" in pname:
                params[pname] = int(rng.integers(0, 9))
            else:
                params[pname] = max(int(rng.choice([0, 1, 2, 3, 5, 9, 14, 30, T - 1, T, T + 1, int(rng.integers(1, 60))])), 0)
        elif name == "mama":   # alpha > 1 diverges: chaotic, any tolerance fails
            params[pname] = float(rng.choice([0.0, 0.02, 0.05, 0.2, 0.5, 0.9]))
        else:
            params[pname] = float(rng.choice([0.0, 0.02, 0.2, 0.5, 0.7, 2.0, -50.0, 5.0]))
    if name == "mavp":
        lo = int(rng.integers(0, 20))
        params["minperiod"], params["maxperiod"] = lo, lo + int(rng.integers(0, 40))
    try:
        exp = oracle.call(name, *[d[c] for c in cols], **params)
    except Exception as e:  # noqa: BLE001
        log("oracle error", name, params, e)
        return 1
    # the device layout: dense, or a row pitch larger than the row (even pitch: tiled bodies incl. the pair-mode storer and the ragged
    # tail; odd pitch: gather bodies; a multiple of 16 elements: the 128-byte pitch bench.py uses); the padding holds a poison value
    pitch = T + int(rng.choice([0, 0, 0, 1, 2, 6, 8, 15, 16])) if rng.random() < 0.5 else (T + 15) // 16 * 16
    def dev(a):
        if pitch == T:
            return torch.from_numpy(np.ascontiguousarray(a)).cuda()
        buf = torch.full((N, pitch), 1e300, dtype=torch.float64, device="cuda")
        buf[:, :T] = torch.from_numpy(np.ascontiguousarray(a)).cuda()
        return buf[:, :T]
    got = api.call(name, *[dev(d[c]) for c in cols], **params)
    bad = 0
    for (oname, dt), g, e in zip(outs, got, exp):
        g = g.cpu().numpy()
        if name in TRANSC:
            ok = np.isclose(g, e, rtol=1e-12, atol=1e-12, equal_nan=True) | (_bits(g) == _bits(e))
            if dt == "f8":
                ok = ok & ((_bits(g) == np.uint64(oracle.NULL_BITS)) == (_bits(e) == np.uint64(oracle.NULL_BITS)))
        else:
            ok = _bits(g) == _bits(e)
            if dt == "f8":   # a computed NaN matches a computed NaN; a NULL (itself a NaN pattern) only matches a NULL
                gn, en = _bits(g) == np.uint64(oracle.NULL_BITS), _bits(e) == np.uint64(oracle.NULL_BITS)
                ok = (ok | ((g != g) & (e != e))) & (gn == en)
        if not np.all(ok):
            bad += 1
            idx = np.argwhere(~ok)[:3].tolist()
            log(f"MISMATCH {name}.{oname} N={N} T={T} {params}: {int((~ok).sum())} cells, first {idx} got {g[~ok][:3]} exp {e[~ok][:3]}")
    return bad


def sweep(seed: int, iters: int, log=print) -> int:
    """-> number of mismatching output columns over `iters` random cases (functions in round-robin order)"""
    rng = np.random.default_rng(seed)
    names = sorted(SPEC)
    return sum(_case(rng, names[it % len(names)], log) for it in range(iters))


if __name__ == "__main__":
    n_bad = sweep(int(sys.argv[1]) if len(sys.argv) > 1 else 7, int(sys.argv[2]) if len(sys.argv) > 2 else 400)
    print("done, mismatching outputs:", n_bad)
