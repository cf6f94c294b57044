"""Synthetic OHLCV of SURVEY 8(d): splitmix64-driven, transcendental-free, bit-reproducible.  numpy restatement of the
generator the tests use (oracle/backtest.c pqo_gen_ohlcv); `tests/test_oracle_kat.py` checks the two are bit-identical.
It lives in the product package so that bench.py's measured path does not touch `oracle/`."""
from __future__ import annotations

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return x ^ (x >> np.uint64(31))


def _u01(seed: int, k: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        return (_splitmix64(np.uint64(seed) + k) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def gen_ohlcv(seed: int, n_sym: int, T: int, mode: int = 0) -> dict:
    """-> {"open","high","low","close","volume"}: float64 [n_sym, T] (mode 1: wider ranges, pattern-rich)"""
    ret_rng, gap_rng, sh_rng = (0.16, 0.04, 0.08) if mode else (0.04, 0.01, 0.01)
    s = np.arange(n_sym, dtype=np.uint64)[:, None]
    t = np.arange(T, dtype=np.uint64)[None, :]
    with np.errstate(over="ignore"):
        k = np.uint64(5) * (s * np.uint64(T) + t)
    ret = (_u01(seed, k) - 0.5) * ret_rng
    start = 10.0 + (np.arange(n_sym) % 90).astype(np.float64)
    fac = 1.0 + ret
    fac[:, 0] = start * fac[:, 0] if T else fac[:, 0]          # c_0 = start * (1 + ret_0); then c_t = c_{t-1} * (1 + ret_t)
    close = np.cumprod(fac, axis=1)
    prev = np.empty_like(close)
    prev[:, 0] = start
    prev[:, 1:] = close[:, :-1]
    with np.errstate(over="ignore"):
        o = prev * (1.0 + (_u01(seed, k + np.uint64(1)) - 0.5) * gap_rng)
        h = np.maximum(o, close) * (1.0 + _u01(seed, k + np.uint64(2)) * sh_rng)
        l = np.minimum(o, close) * (1.0 - _u01(seed, k + np.uint64(3)) * sh_rng)
        v = np.floor(1e5 + _u01(seed, k + np.uint64(4)) * 9e5)
    return {"open": o, "high": h, "low": l, "close": close, "volume": v}
