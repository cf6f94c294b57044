"""Function table of the C ABI (include/pq_hip.h): name -> (inputs, params, outputs).

Argument order of every entry point is (ctx, batch, inputs..., params..., outputs...).  Names, parameter
order and defaults follow the reference's Python wrappers (python/polars_quant/talib/*.py; SURVEY.md
Appendix A); output names are the reference's struct field names (overlap.rs:30-44, momentum.rs:63-67,
:239-247, cycle.rs:149-156, :229-233).
"""
I, F = "i", "f"  # int64 / double parameter

# null policy of the reference family (SURVEY.md 8a legend)
NA, NB, NC, N0 = "N-A", "N-B", "N-C", "N-0"

SPEC = {
    # ---- overlap.py -------------------------------------------------------------------------------
    "bbands": (["real"], [("timeperiod", I, 20), ("nbdevup", F, 2.0), ("nbdevdn", F, 2.0)],
               [("bb_upper", "f8"), ("bb_middle", "f8"), ("bb_lower", "f8")], NA),
    "dema": (["real"], [("timeperiod", I, 30)], [("dema", "f8")], NA),
    "ema": (["real"], [("timeperiod", I, 30)], [("ema", "f8")], NA),
    "kama": (["real"], [("timeperiod", I, 30)], [("kama", "f8")], NA),
    "ma": (["real"], [("timeperiod", I, 30), ("matype", I, 0)], [("ma", "f8")], NA),
    "mama": (["real"], [("fastlimit", F, 0.0), ("slowlimit", F, 0.0)], [("mama", "f8"), ("fama", "f8")], N0),
    "mavp": (["real", "periods"], [("minperiod", I, 2), ("maxperiod", I, 30), ("matype", I, 0)], [("mavp", "f8")], N0),
    "midpoint": (["real"], [("timeperiod", I, 14)], [("midpoint", "f8")], NA),
    "midprice": (["high", "low"], [("timeperiod", I, 14)], [("midprice", "f8")], NA),
    "sar": (["high", "low"], [("acceleration", F, 0.0), ("maximum", F, 0.0)], [("sar", "f8")], N0),
    "sarext": (["high", "low"], [("startvalue", F, 0.0), ("offsetonreverse", F, 0.0),
                                ("accelerationinitlong", F, 0.0), ("accelerationlong", F, 0.0),
                                ("accelerationmaxlong", F, 0.0), ("accelerationinitshort", F, 0.0),
                                ("accelerationshort", F, 0.0), ("accelerationmaxshort", F, 0.0)],
               [("sarext", "f8")], N0),
    "sma": (["real"], [("timeperiod", I, 30)], [("sma", "f8")], NA),
    "t3": (["real"], [("timeperiod", I, 5), ("vfactor", F, 0.7)], [("t3", "f8")], NA),
    "tema": (["real"], [("timeperiod", I, 30)], [("tema", "f8")], NA),
    "trima": (["real"], [("timeperiod", I, 30)], [("trima", "f8")], NA),
    "wma": (["real"], [("timeperiod", I, 30)], [("wma", "f8")], NA),
    # ---- momentum.py ------------------------------------------------------------------------------
    "adx": (["high", "low", "close"], [("timeperiod", I, 14)], [("adx", "f8")], NB),
    "adxr": (["high", "low", "close"], [("timeperiod", I, 14)], [("adxr", "f8")], NB),
    "apo": (["real"], [("fastperiod", I, 12), ("slowperiod", I, 26), ("matype", I, 0)], [("apo", "f8")], NA),
    "aroon": (["high", "low"], [("timeperiod", I, 14)], [("aroon_up", "f8"), ("aroon_down", "f8")], NB),
    "aroonosc": (["high", "low"], [("timeperiod", I, 14)], [("aroonosc", "f8")], NB),
    "bop": (["open", "high", "low", "close"], [], [("bop", "f8")], NB),
    "cci": (["high", "low", "close"], [("timeperiod", I, 14)], [("cci", "f8")], NB),
    "cmo": (["real"], [("timeperiod", I, 14)], [("cmo", "f8")], NB),
    "dx": (["high", "low", "close"], [("timeperiod", I, 14)], [("dx", "f8")], NB),
    "macd": (["real"], [("fastperiod", I, 12), ("slowperiod", I, 26), ("signalperiod", I, 9)],
             [("macd", "f8"), ("macd_signal", "f8"), ("macd_hist", "f8")], NB),
    "macdext": (["real"], [("fastperiod", I, 12), ("fastmatype", I, 0), ("slowperiod", I, 26),
                           ("slowmatype", I, 0), ("signalperiod", I, 9), ("signalmatype", I, 0)],
                [("macd_dif", "f8"), ("macd_dea", "f8"), ("macd_hist", "f8")], NA),
    "macdfix": (["real"], [("signalperiod", I, 9)], [("macd", "f8"), ("macd_signal", "f8"), ("macd_hist", "f8")], NB),
    "mfi": (["high", "low", "close", "volume"], [("timeperiod", I, 14)], [("mfi", "f8")], NB),
    "minus_di": (["high", "low", "close"], [("timeperiod", I, 14)], [("minus_di", "f8")], NB),
    "minus_dm": (["high", "low"], [("timeperiod", I, 14)], [("minus_dm", "f8")], NB),
    "mom": (["real"], [("timeperiod", I, 10)], [("mom", "f8")], NB),
    "plus_di": (["high", "low", "close"], [("timeperiod", I, 14)], [("plus_di", "f8")], NB),
    "plus_dm": (["high", "low"], [("timeperiod", I, 14)], [("plus_dm", "f8")], NB),
    "ppo": (["real"], [("fastperiod", I, 12), ("slowperiod", I, 26), ("matype", I, 0)], [("ppo", "f8")], NA),
    "roc": (["real"], [("timeperiod", I, 10)], [("roc", "f8")], NB),
    "rocp": (["real"], [("timeperiod", I, 10)], [("rocp", "f8")], NB),
    "rocr": (["real"], [("timeperiod", I, 10)], [("rocr", "f8")], NB),
    "rocr100": (["real"], [("timeperiod", I, 10)], [("rocr100", "f8")], NB),
    "rsi": (["real"], [("timeperiod", I, 14)], [("rsi", "f8")], NB),
    "stoch": (["high", "low", "close"], [("fastk_period", I, 5), ("slowk_period", I, 3), ("slowk_matype", I, 0),
                                         ("slowd_period", I, 3), ("slowd_matype", I, 0)],
              [("slowk", "f8"), ("slowd", "f8")], NA),
    "stochf": (["high", "low", "close"], [("fastk_period", I, 5), ("fastd_period", I, 3), ("fastd_matype", I, 0)],
               [("fastk", "f8"), ("fastd", "f8")], NA),
    "stochrsi": (["real"], [("timeperiod", I, 14), ("fastk_period", I, 5), ("fastd_period", I, 3),
                            ("fastd_matype", I, 0)], [("fastk_rsi", "f8"), ("fastd_rsi", "f8")], NB),
    "trix": (["real"], [("timeperiod", I, 30)], [("trix", "f8")], NB),
    "ultosc": (["high", "low", "close"], [("timeperiod1", I, 7), ("timeperiod2", I, 14), ("timeperiod3", I, 28)],
               [("ultosc", "f8")], NB),
    "willr": (["high", "low", "close"], [("timeperiod", I, 14)], [("willr", "f8")], NB),
    # ---- volatility.py / volume.py / price.py -----------------------------------------------------
    "atr": (["high", "low", "close"], [("timeperiod", I, 14)], [("atr", "f8")], NC),
    "natr": (["high", "low", "close"], [("timeperiod", I, 14)], [("natr", "f8")], NC),
    "trange": (["high", "low", "close"], [], [("trange", "f8")], NC),
    "ad": (["high", "low", "close", "volume"], [], [("ad", "f8")], NC),
    "adosc": (["high", "low", "close", "volume"], [("fastperiod", I, 3), ("slowperiod", I, 10)], [("adosc", "f8")], NC),
    "obv": (["real", "volume"], [], [("obv", "f8")], NC),
    "avgprice": (["open", "high", "low", "close"], [], [("avgprice", "f8")], NC),
    "medprice": (["high", "low"], [], [("medprice", "f8")], NC),
    "typprice": (["high", "low", "close"], [], [("typprice", "f8")], NC),
    "wclprice": (["high", "low", "close"], [], [("wclprice", "f8")], NC),
    # ---- cycle.py ---------------------------------------------------------------------------------
    "ht_dcperiod": (["real"], [], [("ht_dcperiod", "f8")], NB),
    "ht_dcphase": (["real"], [], [("ht_dcphase", "f8")], NB),
    "ht_phasor": (["real"], [], [("inphase", "f8"), ("quadrature", "f8")], NB),
    "ht_sine": (["real"], [], [("sine", "f8"), ("leadsine", "f8")], NB),
    "ht_trendline": (["real"], [], [("ht_trendline", "f8")], NB),
    "ht_trendmode": (["real"], [], [("ht_trendmode", "i4")], NB),
}

# functions outside talib.* with the same calling shape (not part of the indicator suite)
EXTRA = {
    # README.md:46-75 returns(df, price_col, period, method): method 0 "simple", 1 "log" (decision D-13)
    "returns": (["real"], [("period", I, 1), ("method", I, 0)], [("return", "f8")], NC),
    "rolling_max": (["real"], [("window", I, 20)], [("rolling_max", "f8")], NA),   # Polars rolling_max (momentum.py:182)
    "rolling_min": (["real"], [("window", I, 20)], [("rolling_min", "f8")], NA),
}

PATTERN_NAMES = [
    "cdl2crows", "cdl3blackcrows", "cdl3inside", "cdl3linestrike", "cdl3outside", "cdl3starsinsouth",
    "cdl3whitesoldiers", "cdlabandonedbaby", "cdladvanceblock", "cdlbelthold", "cdlbreakaway",
    "cdlclosingmarubozu", "cdlconcealbabyswall", "cdlcounterattack", "cdldarkcloudcover", "cdldoji",
    "cdldojistar", "cdldragonflydoji", "cdlengulfing", "cdleveningdojistar", "cdleveningstar",
    "cdlgapsidesidewhite", "cdlgravestonedoji", "cdlhammer", "cdlhangingman", "cdlharami", "cdlharamicross",
    "cdlhighwave", "cdlhikkake", "cdlhikkakemod", "cdlhomingpigeon", "cdlidentical3crows", "cdlinneck",
    "cdlinvertedhammer", "cdlkicking", "cdlkickingbylength", "cdlladderbottom", "cdllongleggeddoji",
    "cdllongline", "cdlmarubozu", "cdlmatchinglow", "cdlmathold", "cdlmorningdojistar", "cdlmorningstar",
    "cdlonneck", "cdlpiercing", "cdlrickshawman", "cdlrisefall3methods", "cdlseparatinglines",
    "cdlshootingstar", "cdlshortline", "cdlspinningtop", "cdlstalledpattern", "cdlsticksandwich", "cdltakuri",
    "cdltasukigap", "cdlthrusting", "cdltristar", "cdlunique3river", "cdlupsidegap2crows",
    "cdlxsidegap3methods"]
# Python-wrapper penetration defaults (pattern.py:109, :321, :353): 0.5 for these three, 0.3 elsewhere.
# Patterns whose wrapper takes a `penetration` argument (the Rust ignores it for abandonedbaby/mathold/thrusting).
PATTERN_PEN_DEFAULT = {n: (0.5 if n in ("cdldarkcloudcover", "cdlmathold", "cdlpiercing") else 0.3) for n in PATTERN_NAMES}
PATTERNS_WITH_PEN_ARG = ("cdlabandonedbaby", "cdldarkcloudcover", "cdleveningdojistar", "cdleveningstar", "cdlmathold",
                         "cdlmorningdojistar", "cdlmorningstar", "cdlpiercing", "cdlthrusting")

SUMMARY_KEYS = ["annualized_return", "max_drawdown", "alpha", "beta", "sharpe_ratio", "max_profit", "win_rate",
                "total_trades"]  # metrics.rs:142-149
BT_DEFAULTS = dict(initial_capital=100000.0, buy_slippage=0.0, sell_slippage=0.0, buy_commission_rate=0.0003,
                   sell_commission_rate=0.0003, min_commission=5.0, position_size=1.0)  # vectorized.rs:38

# README.md:350-366 `Backtest(...)` defaults (decision D-10)
LEV_DEFAULTS = dict(initial_capital=100000.0, position_size=1.0, leverage=1.0, margin_call_threshold=0.3,
                    interest_rate=0.06, commission_rate=0.0003, min_commission=5.0, slippage=0.0)
TRADE_FIELDS = ("entry_day", "exit_day", "entry_price", "exit_price", "quantity", "pnl", "pnl_pct", "reason")
PORTFOLIO_COLS = ("portfolio_value", "daily_pnl", "daily_return_pct", "cumulative_pnl", "cumulative_return_pct",
                  "benchmark_return_pct", "alpha_pct", "relative_return_pct", "beta")
