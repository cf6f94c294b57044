"""VectorizedBacktester -- same constructor signature, defaults and result shape as the reference pyclass
(src/backtest/vectorized.rs:37-66, .run() :69-224), backed by the fused HIP scan+summary kernel.

The reference is single-asset; here `price` may also be [N, T] (N independent capital pools, one per
symbol), in which case run() returns batched curves and a list of N summary dicts.
"""
from __future__ import annotations

from . import api as _api
from ._spec import BT_DEFAULTS, SUMMARY_KEYS


class VectorizedBacktester:
    def __init__(self, price, buy_signal, sell_signal, benchmark=None, initial_capital=100_000.0, buy_slippage=0.0,
                 sell_slippage=0.0, buy_commission_rate=0.0003, sell_commission_rate=0.0003, min_commission=5.0,
                 position_size=1.0):
        self.price_data, self.buy_signals, self.sell_signals, self.benchmark = price, buy_signal, sell_signal, benchmark
        self.params = dict(initial_capital=initial_capital, buy_slippage=buy_slippage, sell_slippage=sell_slippage,
                           buy_commission_rate=buy_commission_rate, sell_commission_rate=sell_commission_rate,
                           min_commission=min_commission, position_size=position_size)

    def run(self):
        """-> (positions {"position"}, capital {"cash","equity"}, summary dict[8])  (vectorized.rs:204-223).

        With polars installed the first two are pl.DataFrame like the reference; otherwise dicts of arrays."""
        pos, cash, eq, summ = _api.backtest_vectorized(self.price_data, self.buy_signals, self.sell_signals,
                                                       self.benchmark, **self.params)
        s = summ.cpu().numpy() if hasattr(summ, "cpu") else summ
        if s.ndim == 1:
            summary = dict(zip(SUMMARY_KEYS, (float(v) for v in s)))
        else:
            summary = [dict(zip(SUMMARY_KEYS, (float(v) for v in row))) for row in s]
        positions, capital = {"position": pos}, {"cash": cash, "equity": eq}
        try:
            import polars as pl  # optional
            if getattr(pos, "ndim", 2) == 1:
                to = lambda a: a.cpu().numpy() if hasattr(a, "cpu") else a
                positions = pl.DataFrame({"position": to(pos)})
                capital = pl.DataFrame({"cash": to(cash), "equity": to(eq)})
        except ImportError:
            pass
        return positions, capital, summary
