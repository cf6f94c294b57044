"""`returns(df, price_col, period, method, return_col)` -- README.md:46-75 of the reference (README-only, no source in the
tree; semantics = decision D-13, pinned by the README's own example vector).  The arithmetic runs in pq_returns (HIP).

`df` may be a polars DataFrame (returned with the new column, as the README shows), a pyarrow Table, or a dict of columns;
a bare array / tensor [T] or [N, T] is treated as the price column and the return column alone comes back.
"""
from __future__ import annotations

from . import api

_METHODS = {"simple": 0, "log": 1}


def returns(df, price_col: str = "close", period: int = 1, method: str = "simple", return_col: str = "return"):
    if method not in _METHODS:
        raise ValueError(f"method must be 'simple' or 'log', got {method!r}")
    mod = type(df).__module__.split(".")[0]
    if mod == "polars":
        import polars as pl
        (r,) = api.call("returns", df[price_col], period=period, method=_METHODS[method])
        return df.with_columns(pl.Series(return_col, r))
    if mod == "pyarrow" and hasattr(df, "append_column"):
        col = df.column(price_col)
        (r,) = api.call("returns", col, period=period, method=_METHODS[method])
        return df.append_column(return_col, r)
    if isinstance(df, dict):
        (r,) = api.call("returns", df[price_col], period=period, method=_METHODS[method])
        out = dict(df)
        out[return_col] = r
        return out
    (r,) = api.call("returns", df, period=period, method=_METHODS[method])
    return r
