"""Strategy -- the README's signal generators (README.md:862-994) on the HIP indicator kernels + the three signal rules of
decision D-11 (oracle/backtest.c).  README-only in the reference: the rules are this build's definition.

Every method takes a dict / DataFrame of [N, T] (or [T]) columns named open/high/low/close/volume and returns
{"buy_signal": uint8 [N, T], "sell_signal": uint8 [N, T]} (device tensors), ready for `VectorizedBacktester`,
`api.backtest_vectorized` or `Backtest`.  All fourteen README names are here: ma, macd, rsi, bband, stoch, cci, adx, breakout,
reversion, volume, grid, gap, pattern, trend.  The README documents parameters only for ma / macd / rsi (its `.pyi` has no
`Strategy` class), so for the others the definitions in the docstrings below are this build's (decision D-11b).  Every
indicator value comes from the HIP indicator kernels and every comparison that turns them into signals from the rule kernels of the
C ABI (pq_cross / band / channel_signals, pq_gate_signals, pq_zscore, pq_scale_band, pq_volume_surge_signals, pq_gap_signals,
pq_pattern_any_signals, pq_ma_stack_signals: csrc/backtest.hip, csrc/strategy.hip) -- this module holds no arithmetic of its own,
so a non-Python host composes the same strategies from the same calls, and a recorded suite runs them in front of
pq_backtest_vectorized without the signal columns leaving the device.
"""
from __future__ import annotations

from . import api as _api

_MA = {"sma": "sma", "ema": "ema", "wma": "wma", "dema": "dema", "tema": "tema"}


def _col(df, name):
    return df[name]


class Strategy:
    def ma(self, df, price_col="close", fast_period=10, slow_period=20, ma_type="sma", trend_period=0, trend_filter=False,
           slope_filter=False, distance_pct=0.0):
        """golden / dead cross of MA(fast) and MA(slow) (README.md:877-905); trend_filter: buys only while price > MA(trend_period);
        slope_filter: buys only while the slow MA rises (slow[i] > slow[i-1]); distance_pct: buys only while the price is at
        least that many percent above the slow MA"""
        if ma_type not in _MA:
            raise ValueError(f"ma_type must be one of {sorted(_MA)}")
        x = _col(df, price_col)
        (fast,) = _api.call(_MA[ma_type], x, timeperiod=fast_period)
        (slow,) = _api.call(_MA[ma_type], x, timeperiod=slow_period)
        buy, sell = _api.cross_signals(fast, slow)
        if trend_filter and trend_period > 0:
            (trend,) = _api.call(_MA[ma_type], x, timeperiod=trend_period)
            buy, sell = _api.gate_signals(buy, sell, x, 2, c=trend)              # buys only while price > MA(trend_period)
        if slope_filter:
            buy, sell = _api.gate_signals(buy, sell, slow, 3)                    # buys only while the slow MA rises
        if distance_pct > 0.0:
            buy, sell = _api.gate_signals(buy, sell, x, 4, k0=distance_pct / 100.0, c=slow)
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
        buy, sell = _api.gate_signals(*_api.cross_signals(k, d), k, 0, k0=oversold, k1=overbought)
        return {"buy_signal": buy, "sell_signal": sell}

    def cci(self, df, period=14, oversold=-100.0, overbought=100.0):
        (c,) = _api.call("cci", _col(df, "high"), _col(df, "low"), _col(df, "close"), timeperiod=period)
        buy, sell = _api.band_signals(c, oversold, overbought)
        return {"buy_signal": buy, "sell_signal": sell}

    # ---- the README names without documented parameters (README.md:946-953): definitions = decision D-11b --------------
    def adx(self, df, period=14, threshold=25.0):
        """trend strength: +DM crossing above -DM (the reference's smoothed directional movements) while ADX > threshold buys,
        the opposite cross while ADX > threshold sells"""
        h, l, c = _col(df, "high"), _col(df, "low"), _col(df, "close")
        (pdm,) = _api.call("plus_dm", h, l, timeperiod=period)
        (mdm,) = _api.call("minus_dm", h, l, timeperiod=period)
        (adx,) = _api.call("adx", h, l, c, timeperiod=period)
        buy, sell = _api.gate_signals(*_api.cross_signals(pdm, mdm), adx, 1, k0=threshold)      # a null ADX compares false
        return {"buy_signal": buy, "sell_signal": sell}

    def breakout(self, df, period=20):
        """Donchian channel: buy when the close exceeds the highest high of the previous `period` bars, sell when it falls
        below their lowest low (rule `channel`, mode 1)"""
        (hi,) = _api.call("rolling_max", _col(df, "high"), window=period)
        (lo,) = _api.call("rolling_min", _col(df, "low"), window=period)
        buy, sell = _api.channel_signals(_col(df, "close"), lo, hi, 1)
        return {"buy_signal": buy, "sell_signal": sell}

    def reversion(self, df, price_col="close", period=20, threshold=2.0):
        """mean reversion on the z-score (price - SMA) / population std (the BBANDS arithmetic): buy when z comes back up
        through -threshold, sell when it comes back down through +threshold (rule `band`)"""
        x = _col(df, price_col)
        up, mid, _lo = _api.call("bbands", x, timeperiod=period, nbdevup=1.0, nbdevdn=1.0)
        z = _api.zscore(x, up, mid)                   # up - mid = 1.0 * sd; a null band row stays null: the rule is false there
        buy, sell = _api.band_signals(z, -threshold, threshold)
        return {"buy_signal": buy, "sell_signal": sell}

    def volume(self, df, period=20, multiplier=2.0):
        """volume breakout: volume above multiplier x SMA(volume, period) on an up day buys, on a down day sells"""
        (sv,) = _api.call("sma", _col(df, "volume"), timeperiod=period)
        buy, sell = _api.volume_surge_signals(_col(df, "volume"), sv, _col(df, "close"), multiplier)
        return {"buy_signal": buy, "sell_signal": sell}

    def grid(self, df, price_col="close", base_period=20, grid_pct=5.0):
        """one grid level around a moving base line: buy when the price drops through SMA x (1 - grid_pct %), sell when it rises
        through SMA x (1 + grid_pct %) (rule `channel`, mode 0)"""
        x = _col(df, price_col)
        (base,) = _api.call("sma", x, timeperiod=base_period)
        lo, hi = _api.scale_band(base, 1.0 - grid_pct / 100.0, 1.0 + grid_pct / 100.0)
        buy, sell = _api.channel_signals(x, lo, hi, 0)
        return {"buy_signal": buy, "sell_signal": sell}

    def gap(self, df, gap_pct=2.0):
        """opening gaps: an open above the previous high by gap_pct % buys, below the previous low by gap_pct % sells"""
        buy, sell = _api.gap_signals(_col(df, "open"), _col(df, "high"), _col(df, "low"), 1.0 + gap_pct / 100.0, 1.0 - gap_pct / 100.0)
        return {"buy_signal": buy, "sell_signal": sell}

    def pattern(self, df, bullish=("cdlhammer", "cdlengulfing", "cdlmorningstar", "cdlpiercing", "cdl3whitesoldiers"),
                bearish=("cdlhangingman", "cdlengulfing", "cdleveningstar", "cdldarkcloudcover", "cdl3blackcrows")):
        """candlestick patterns (one fused pass over OHLC): any of `bullish` at +100 buys, any of `bearish` at -100 sells"""
        names = sorted(set(bullish) | set(bearish))
        pats = _api.cdl_all(_col(df, "open"), _col(df, "high"), _col(df, "low"), _col(df, "close"), names=names)
        buy, sell = _api.pattern_any_signals([pats[n] for n in bullish], [pats[n] for n in bearish])
        return {"buy_signal": buy, "sell_signal": sell}

    def trend(self, df, price_col="close", periods=(5, 10, 20, 60), ma_type="sma"):
        """multi-MA alignment: buy on the first bar where MA(p1) > MA(p2) > ... holds, sell on the first bar of the reverse order"""
        x = _col(df, price_col)
        buy, sell = _api.ma_stack_signals([_api.call(_MA[ma_type], x, timeperiod=p)[0] for p in periods])   # nulls compare false
        return {"buy_signal": buy, "sell_signal": sell}
