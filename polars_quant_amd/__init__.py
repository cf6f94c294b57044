"""polars_quant_amd -- MI355X (gfx950) execution path for polars-quant's indicator + backtest hot path.

Mirrors the reference's Python surface (python/polars_quant/__init__.py:1-203): the UPPER-CASE talib
functions (same argument order and defaults), `VectorizedBacktester`, plus batched [N, T] entry points.
All compute goes through libpolars_quant_hip.so (hand-written HIP kernels); there is no CPU fallback.
"""
from . import talib
from ._lib import NullsNotAllowed, PqError
from ._spec import PATTERN_NAMES, SPEC, SUMMARY_KEYS
from .backtest import Backtest, VectorizedBacktester
from .factor import Factor
from .returns import returns
from .strategy import Strategy
from .talib import *  # noqa: F401,F403

__version__ = "0.1.0"
