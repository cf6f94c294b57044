"""Strategy -- the README's signal generators (README.md:862-994) on the HIP indicator kernels + the three signal rules of
decision D-11 (oracle/backtest.c).  README-only in the reference: the rules are this build's definition.

Every method takes a dict / DataFrame of [N, T] (or [T]) columns named open/high/low/close/volume and returns
{"buy_signal": uint8 [N, T], "sell_signal": uint8 [N, T]} (device tensors), ready for `VectorizedBacktester`,
`api.backtest_vectorized` or `Backtest`.  Implemented: ma, macd, rsi, bband, stoch, cci (the other README strategies need
indicators outside the reference's talib set, e.g. rolling max/min for the Donchian breakout: `api.channel_signals(price,
lo, hi, 1)` takes such bounds directly).
"""
from __future__ import annotations

import torch

from . import api as _api

_MA = {"sma": "sma", "ema": "ema", "wma": "wma", "dema": "dema", "tema": "tema"}


def _col(df, name):
    return df[name]


class Strategy:
    def ma(self, df, price_col="close", fast_period=10, slow_period=20, ma_type="sma", trend_period=0, trend_filter=False):
        """golden / dead cross of MA(fast) and MA(slow); with trend_filter, buys only while price > MA(trend_period)"""
        if ma_type not in _MA:
            raise ValueError(f"ma_type must be one of {sorted(_MA)}")
        x = _col(df, price_col)
        (fast,) = _api.call(_MA[ma_type], x, timeperiod=fast_period)
        (slow,) = _api.call(_MA[ma_type], x, timeperiod=slow_period)
        buy, sell = _api.cross_signals(fast, slow)
        if trend_filter and trend_period > 0:
            (trend,) = _api.call(_MA[ma_type], x, timeperiod=trend_period)
            p = _api._to_device(x)[0]
            t = _api._to_device(trend)[0]
            buy = buy & (p > t).to(torch.uint8)      # a null trend value compares False
        return {"buy_signal": buy, "sell_signal": sell}

    def macd(self, df, price_col="close", fast_period=12, slow_period=26, signal_period=9):
        """MACD line crossing its signal line (identical to api.macd_cross_signals)"""
        m, s, _h = _api.call("macd", _col(df, price_col), fastperiod=fast_period, slowperiod=slow_period, signalperiod=signal_period)
        buy, sell = _api.cross_signals(m, s)
        return {"buy_signal": buy, "sell_signal": sell}

    def rsi(self, df, price_col="close", period=14, oversold=30.0, overbought=70.0):
        (r,) = _api.call("rsi", _col(df, price_col), timeperiod=period)
        buy, sell = _api.band_signals(r, oversold, overbought)
        return {"buy_signal": buy, "sell_signal": sell}

    def bband(self, df, price_col="close", period=20, nbdev=2.0):
        """mean reversion at the Bollinger bands"""
        x = _col(df, price_col)
        up, _mid, lo = _api.call("bbands", x, timeperiod=period, nbdevup=nbdev, nbdevdn=nbdev)
        buy, sell = _api.channel_signals(x, lo, up, 0)
        return {"buy_signal": buy, "sell_signal": sell}

    def stoch(self, df, fastk_period=5, slowk_period=3, slowd_period=3, oversold=20.0, overbought=80.0):
        """%K crossing %D, buys only in the oversold zone and sells only in the overbought zone"""
        k, d = _api.call("stoch", _col(df, "high"), _col(df, "low"), _col(df, "close"), fastk_period=fastk_period,
                         slowk_period=slowk_period, slowd_period=slowd_period)
        buy, sell = _api.cross_signals(k, d)
        kk = _api._to_device(k)[0]
        return {"buy_signal": buy & (kk < oversold).to(torch.uint8), "sell_signal": sell & (kk > overbought).to(torch.uint8)}

    def cci(self, df, period=14, oversold=-100.0, overbought=100.0):
        (c,) = _api.call("cci", _col(df, "high"), _col(df, "low"), _col(df, "close"), timeperiod=period)
        buy, sell = _api.band_signals(c, oversold, overbought)
        return {"buy_signal": buy, "sell_signal": sell}
