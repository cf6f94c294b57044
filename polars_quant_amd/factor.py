"""Factor -- the evaluation half of the README's `Factor` class (README.md:1429-1430, :1480-1482, :1626-1634): per-day
cross-sectional IC, Rank-IC and their rolling mean / information ratio.  README-only in the reference; semantics =
decision D-12 (oracle/backtest.c).  Inputs are [N, T] arrays (symbol-major, like every other column of this package): the
factor and the forward return of every symbol on every day.
"""
from __future__ import annotations

from . import api as _api


class Factor:
    def ic(self, factor, next_return):
        """-> (ic [T], n_valid [T]): Pearson correlation across symbols, per day"""
        return _api.factor_ic(factor, next_return, 0)

    def rank_ic(self, factor, next_return):
        """-> (rank_ic [T], n_valid [T]): Spearman correlation (average ranks) across symbols, per day"""
        return _api.factor_ic(factor, next_return, 1)

    def rolling_ic(self, factor, next_return, window=60, rank=False):
        """-> {"rolling_ic": [T], "rolling_ir": [T]} of the (rank) IC series"""
        ic, _ = _api.factor_ic(factor, next_return, 1 if rank else 0)
        m, ir = _api.rolling_ic(ic, window)
        return {"rolling_ic": m, "rolling_ir": ir}
