"""VectorizedBacktester -- same constructor signature, defaults and result shape as the reference pyclass
(src/backtest/vectorized.rs:37-66, .run() :69-224), backed by the fused HIP scan+summary kernel.

The reference is single-asset; here `price` may also be [N, T] (N independent capital pools, one per
symbol), in which case run() returns batched curves and a list of N summary dicts.
"""
from __future__ import annotations

import numpy as np

from . import api as _api
from ._spec import BT_DEFAULTS, PORTFOLIO_COLS, SUMMARY_KEYS, TRADE_FIELDS


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


def _wide(frame, dtype):
    """A wide table (first column = date, one column per symbol; README.md:355-357) -> (dates, symbols, [N, T] array).
    Accepts a polars/pandas DataFrame, a pyarrow Table or a dict of columns."""
    if hasattr(frame, "to_dict") and not isinstance(frame, dict):           # polars / pandas
        cols = frame.to_dict(as_series=False) if "as_series" in frame.to_dict.__code__.co_varnames else frame.to_dict("list")
    elif hasattr(frame, "to_pydict"):                                        # pyarrow
        cols = frame.to_pydict()
    else:
        cols = dict(frame)
    names = list(cols)
    dates = list(cols[names[0]])
    arr = np.ascontiguousarray(np.stack([np.asarray(cols[k], dtype=dtype) for k in names[1:]]))
    return dates, names[1:], arr


class Backtest:
    """The README's multi-symbol engine (README.md:346-640): independent capital pool per symbol, 100-share lots, leverage /
    margin call / interest, commission with a minimum, slippage.  README-only in the reference; semantics = decision D-10
    (DESIGN.md).  Tables are returned as dicts of columns (pl.DataFrame when polars is installed)."""

    def __init__(self, prices, buy_signals, sell_signals, initial_capital=100_000.0, position_size=1.0, leverage=1.0,
                 margin_call_threshold=0.3, interest_rate=0.06, commission_rate=0.0003, min_commission=5.0, slippage=0.0,
                 benchmark=None, max_trades=64):
        self.dates, self.symbols, self._price = _wide(prices, np.float64)
        _, s_buy, self._buy = _wide(buy_signals, np.uint8)
        _, s_sell, self._sell = _wide(sell_signals, np.uint8)
        if s_buy != self.symbols or s_sell != self.symbols:
            raise ValueError("prices, buy_signals and sell_signals must have the same symbol columns")
        self._bench = None
        if benchmark is not None:
            self._bench = _wide(benchmark, np.float64)[2][0]
        self.params = dict(initial_capital=initial_capital, position_size=position_size, leverage=leverage,
                           margin_call_threshold=margin_call_threshold, interest_rate=interest_rate,
                           commission_rate=commission_rate, min_commission=min_commission, slippage=slippage)
        self.max_trades = max_trades
        self._r = None

    @staticmethod
    def _table(cols: dict):
        try:
            import polars as pl  # optional
            return pl.DataFrame(cols)
        except ImportError:
            return cols

    def run(self) -> None:
        r = _api.backtest_leveraged(self._price, self._buy, self._sell, self._bench, self.max_trades, **self.params)
        self._metrics = _api.portfolio_metrics(r["total_value"], self.params["initial_capital"] * len(self.symbols), self._bench)
        self._r = {k: (v.cpu().numpy() if hasattr(v, "cpu") else v) for k, v in r.items() if k != "trades"}
        self._r["trades"] = {k: v.cpu().numpy() for k, v in r["trades"].items()}
        self._metrics = self._metrics.cpu().numpy()

    def _need(self):
        if self._r is None:
            raise RuntimeError("call run() first")

    def get_daily_records(self, symbol=None):
        self._need()
        idx = range(len(self.symbols)) if symbol is None else [self.symbols.index(symbol)]
        T = len(self.dates)
        return self._table({"symbol": [self.symbols[i] for i in idx for _ in range(T)], "date": [d for _ in idx for d in self.dates],
                            "cash": np.concatenate([self._r["cash"][i] for i in idx]),
                            "stock_value": np.concatenate([self._r["stock_value"][i] for i in idx]),
                            "total_value": np.concatenate([self._r["total_value"][i] for i in idx])})

    def get_position_records(self, symbol=None):
        self._need()
        idx = range(len(self.symbols)) if symbol is None else [self.symbols.index(symbol)]
        tr, cnt = self._r["trades"], self._r["trade_count"]
        rows = [(i, k) for i in idx for k in range(min(int(cnt[i]), self.max_trades))]
        col = lambda f: np.array([tr[f][i, k] for i, k in rows])
        ed, xd = col("entry_day").astype(int), col("exit_day").astype(int)
        return self._table({"symbol": [self.symbols[i] for i, _ in rows], "entry_date": [self.dates[d] for d in ed],
                            "entry_price": col("entry_price"), "quantity": col("quantity"), "exit_date": [self.dates[d] for d in xd],
                            "exit_price": col("exit_price"), "pnl": col("pnl"), "pnl_pct": col("pnl_pct"),
                            "holding_days": xd - ed, "reason": col("reason")})

    def get_performance_metrics(self):
        self._need()
        ncol = len(PORTFOLIO_COLS) if self._bench is not None else 5
        return self._table({"date": self.dates, **{PORTFOLIO_COLS[k]: self._metrics[:, k] for k in range(ncol)}})

    def get_stock_daily(self, symbol):
        return self.get_daily_records(symbol)

    def get_stock_positions(self, symbol):
        return self.get_position_records(symbol)

    def get_stock_performance(self, symbol):
        """README.md:554-589: one symbol's daily performance -- stock_value (its total assets), daily / cumulative P&L and
        returns, and with a benchmark the benchmark return, alpha and relative return: the portfolio-metrics kernel on this
        symbol's own capital pool."""
        self._need()
        i = self.symbols.index(symbol)
        tv = np.ascontiguousarray(self._r["total_value"][i:i + 1])
        m = _api.portfolio_metrics(tv, self.params["initial_capital"], self._bench).cpu().numpy()
        cols = {"symbol": [symbol] * len(self.dates), "date": self.dates, "stock_value": m[:, 0], "daily_pnl": m[:, 1],
                "daily_return_pct": m[:, 2], "cumulative_pnl": m[:, 3], "cumulative_return_pct": m[:, 4]}
        if self._bench is not None:
            cols.update({"benchmark_return_pct": m[:, 5], "alpha_pct": m[:, 6], "relative_return_pct": m[:, 7]})
        return self._table(cols)

    def get_stock_summary(self, symbol) -> str:
        self._need()
        i = self.symbols.index(symbol)
        return "\n".join(f"{k}: {v:.6g}" for k, v in zip(SUMMARY_KEYS, self._r["summary"][i]))

    def summary(self) -> None:
        self._need()
        m = self._metrics
        print(f"symbols: {len(self.symbols)}  days: {len(self.dates)}  trades: {int(self._r['trade_count'].sum())}")
        print(f"final portfolio value: {m[-1, 0]:.2f}  cumulative return: {m[-1, 4]:.4f} %")
