"""ctypes binding of libpolars_quant_hip.so (the C ABI declared in include/pq_hip.h).

There is NO CPU fallback: if the HIP library is missing or fails to load, importing the compute API raises.
Build it with `python -c "import __graft_entry__ as g; g.build()"` or `make -C polars_quant_amd/csrc`.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

from ._spec import EXTRA, I, SPEC

_HERE = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ["PQ_LIB_PATH"]) if os.environ.get("PQ_LIB_PATH") else _HERE / "libpolars_quant_hip.so"  # override: A/B builds

PQ_OK = 0
NULL_BITS = 0x7FF80000504E554C
NULL_I32 = -2147483648


class PqError(RuntimeError):
    """Raised when a C-ABI call returns a non-zero status (message from pq_last_error)."""


class NullsNotAllowed(PqError):
    """The reference function rejects nulls (N-B family: rechunk().cont_slice()? fails, momentum.rs:12-13)."""


class Batch(C.Structure):
    # offsets: device pointer to n_series + 1 int64 row indices of a RAGGED batch (then len = the longest series, stride = the
    # total row count), or None for the regular [n_series][stride] layout (include/pq_hip.h)
    _fields_ = [("n_series", C.c_int64), ("len", C.c_int64), ("stride", C.c_int64), ("offsets", C.c_void_p)]


class BtParams(C.Structure):
    _fields_ = [(k, C.c_double) for k in ("initial_capital", "buy_slippage", "sell_slippage", "buy_commission_rate",
                                          "sell_commission_rate", "min_commission", "position_size")]


class LevParams(C.Structure):
    _fields_ = [(k, C.c_double) for k in ("initial_capital", "position_size", "leverage", "margin_call_threshold",
                                          "interest_rate", "commission_rate", "min_commission", "slippage")]


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise ImportError(
                f"{LIB_PATH} not found: the MI355X HIP library is not built. "
                "Run `make -C polars_quant_amd/csrc` (hipcc --offload-arch=gfx950). There is no CPU fallback.")
        L = C.CDLL(str(LIB_PATH))
        L.pq_last_error.restype = C.c_char_p
        L.pq_pattern_name.restype = C.c_char_p
        L.pq_pattern_name.argtypes = [C.c_int32]
        L.pq_pattern_id.argtypes = [C.c_char_p]
        vp = C.c_void_p
        for name, (cols, params, outs, _fam) in {**SPEC, **EXTRA}.items():
            fn = getattr(L, "pq_" + name)
            fn.restype = C.c_int32
            fn.argtypes = [vp, C.POINTER(Batch)] + [vp] * len(cols) + \
                          [C.c_int64 if k == I else C.c_double for _, k, _ in params] + [vp] * len(outs)
        L.pq_dmi_all.restype = C.c_int32
        L.pq_dmi_all.argtypes = [vp, C.POINTER(Batch), vp, vp, vp, C.c_int64, vp, vp, vp, vp, vp]
        for nm, args in (("pq_ema_all", [vp, C.c_int64, vp, vp, vp, vp]), ("pq_atr_all", [vp, vp, vp, C.c_int64, vp, vp]),
                         ("pq_dm_pair", [vp, vp, C.c_int64, vp, vp]), ("pq_ad_all", [vp, vp, vp, vp, C.c_int64, C.c_int64, vp, vp]),
                         ("pq_macd_pair", [vp, C.c_int64, C.c_int64, C.c_int64, C.c_int64, vp, vp, vp, vp, vp, vp]),
                         ("pq_apo_ppo", [vp, C.c_int64, C.c_int64, C.c_int64, vp, vp]),
                         ("pq_stoch_all", [vp, vp, vp] + [C.c_int64] * 7 + [vp] * 4),
                         ("pq_sar_pair", [vp, vp] + [C.c_double] * 10 + [vp, vp]),
                         ("pq_dm_system_all", [vp, vp, vp, C.c_int64] + [vp] * 7), ("pq_cmo_rsi", [vp, C.c_int64, vp, vp]), ("pq_sma_ma", [vp, C.c_int64, vp, vp]),
                         ("pq_volume_all", [vp, vp, vp, vp, C.c_int64, C.c_int64, C.c_int64, vp, vp, vp, vp])):
            getattr(L, nm).restype = C.c_int32
            getattr(L, nm).argtypes = [vp, C.POINTER(Batch)] + args
        L.pq_aroon_all.restype = C.c_int32
        L.pq_aroon_all.argtypes = [vp, C.POINTER(Batch), vp, vp, C.c_int64, vp, vp, vp]
        L.pq_ht_all.restype = C.c_int32
        L.pq_ht_all.argtypes = [vp, C.POINTER(Batch), vp, vp, vp, vp, vp, vp, vp]
        L.pq_cdl.restype = C.c_int32
        L.pq_cdl.argtypes = [vp, C.POINTER(Batch), C.c_int32, vp, vp, vp, vp, C.c_double, vp]
        L.pq_cdl_all.restype = C.c_int32
        L.pq_cdl_all.argtypes = [vp, C.POINTER(Batch), vp, vp, vp, vp, C.POINTER(C.c_double), C.POINTER(vp)]
        L.pq_backtest_vectorized.restype = C.c_int32
        L.pq_backtest_vectorized.argtypes = [vp, C.POINTER(Batch), vp, vp, vp, vp, C.POINTER(BtParams), vp, vp, vp, vp]
        L.pq_backtest_macd_cross.restype = C.c_int32
        L.pq_backtest_macd_cross.argtypes = [vp, C.POINTER(Batch), vp, C.c_int64, C.c_int64, C.c_int64,
                                             C.POINTER(BtParams), vp, vp, vp, vp]
        L.pq_comm_unique_id.restype = C.c_int32
        L.pq_comm_unique_id.argtypes = [vp]
        L.pq_comm_init.restype = C.c_int32
        L.pq_comm_init.argtypes = [vp, C.c_int32, C.c_int32, vp]
        L.pq_comm_destroy.restype = C.c_int32
        L.pq_comm_destroy.argtypes = [vp]
        L.pq_shard_range.restype = C.c_int32
        L.pq_shard_range.argtypes = [C.c_int64, C.c_int32, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L.pq_gather_summaries.restype = C.c_int32
        L.pq_gather_summaries.argtypes = [vp, vp, C.c_int64, vp]
        L.pq_gather_summaries_begin.restype = C.c_int32
        L.pq_gather_summaries_begin.argtypes = [vp, vp, C.c_int64, vp, C.c_int32]
        L.pq_gather_summaries_end.restype = C.c_int32
        L.pq_gather_summaries_end.argtypes = [vp, C.c_int32]
        L.pq_comm_sync.restype = C.c_int32
        L.pq_comm_sync.argtypes = [vp]
        L.pq_backtest_wave_stats.restype = C.c_int32
        L.pq_backtest_wave_stats.argtypes = [vp, C.POINTER(C.c_int64), C.c_int32]
        L.pq_wt_stats.restype = C.c_int32
        L.pq_wt_stats.argtypes = [vp, C.POINTER(C.c_int64), C.c_int32]
        L.pq_ragged_rehouse_stats.restype = C.c_int32
        L.pq_ragged_rehouse_stats.argtypes = [vp, C.POINTER(C.c_int64), C.c_int32]
        L.pq_macd_cross_signals.restype = C.c_int32
        L.pq_macd_cross_signals.argtypes = [vp, C.POINTER(Batch), vp, C.c_int64, C.c_int64, C.c_int64, vp, vp]
        L.pq_cross_signals.restype = C.c_int32
        L.pq_cross_signals.argtypes = [vp, C.POINTER(Batch), vp, vp, vp, vp]
        L.pq_band_signals.restype = C.c_int32
        L.pq_band_signals.argtypes = [vp, C.POINTER(Batch), vp, C.c_double, C.c_double, vp, vp]
        L.pq_channel_signals.restype = C.c_int32
        L.pq_channel_signals.argtypes = [vp, C.POINTER(Batch), vp, vp, vp, C.c_int32, vp, vp]
        L.pq_gate_signals.restype = C.c_int32
        L.pq_gate_signals.argtypes = [vp, C.POINTER(Batch), vp, vp, C.c_int32, C.c_double, C.c_double, vp, vp, vp, vp]
        L.pq_zscore.restype = C.c_int32
        L.pq_zscore.argtypes = [vp, C.POINTER(Batch), vp, vp, vp, vp]
        L.pq_scale_band.restype = C.c_int32
        L.pq_scale_band.argtypes = [vp, C.POINTER(Batch), vp, C.c_double, C.c_double, vp, vp]
        L.pq_volume_surge_signals.restype = C.c_int32
        L.pq_volume_surge_signals.argtypes = [vp, C.POINTER(Batch), vp, vp, vp, C.c_double, vp, vp]
        L.pq_gap_signals.restype = C.c_int32
        L.pq_gap_signals.argtypes = [vp, C.POINTER(Batch), vp, vp, vp, C.c_double, C.c_double, vp, vp]
        L.pq_pattern_any_signals.restype = C.c_int32
        L.pq_pattern_any_signals.argtypes = [vp, C.POINTER(Batch), vp, C.c_int32, vp, C.c_int32, vp, vp]
        L.pq_ma_stack_signals.restype = C.c_int32
        L.pq_ma_stack_signals.argtypes = [vp, C.POINTER(Batch), vp, C.c_int32, vp, vp]
        L.pq_factor_ic.restype = C.c_int32
        L.pq_factor_ic.argtypes = [vp, C.POINTER(Batch), vp, vp, C.c_int32, vp, vp]
        L.pq_rolling_ic.restype = C.c_int32
        L.pq_rolling_ic.argtypes = [vp, vp, C.c_int64, C.c_int64, vp, vp]
        L.pq_backtest_leveraged.restype = C.c_int32
        L.pq_backtest_leveraged.argtypes = [vp, C.POINTER(Batch), vp, vp, vp, vp, C.POINTER(LevParams), vp, vp, vp, C.c_int32] + [vp] * 10
        L.pq_portfolio_metrics.restype = C.c_int32
        L.pq_portfolio_metrics.argtypes = [vp, C.POINTER(Batch), vp, C.c_double, vp, vp]
        L.pq_recommended_stride.restype = C.c_int64
        L.pq_recommended_stride.argtypes = [C.c_int64]
        L.pq_layout_check.restype = C.c_int32
        L.pq_layout_check.argtypes = [C.POINTER(Batch), C.POINTER(vp), C.c_int32]
        L.pq_ctx_create.argtypes = [C.c_int32, vp, C.POINTER(vp)]
        L.pq_ctx_destroy.argtypes = [vp]
        L.pq_ctx_set_stream.argtypes = [vp, vp]
        L.pq_ctx_sync.argtypes = [vp]
        L.pq_count_nulls.argtypes = [vp, C.POINTER(Batch), vp, C.POINTER(C.c_int64)]
        L.pq_nulls_from_arrow.argtypes = [vp, vp, vp, C.c_int64, C.c_int64]
        L.pq_validity_to_arrow.argtypes = [vp, vp, C.c_int64, vp, vp]
        L.pq_malloc.argtypes = [vp, C.c_size_t, C.POINTER(vp)]
        L.pq_free.argtypes = [vp, vp]
        L.pq_host_register.argtypes = [vp, C.c_size_t]
        L.pq_host_unregister.argtypes = [vp]
        L.pq_memcpy_h2d.argtypes = [vp, vp, vp, C.c_size_t]
        L.pq_memcpy_d2h.argtypes = [vp, vp, vp, C.c_size_t]
        L.pq_memcpy_h2d_pitched.argtypes = [vp, vp, C.c_size_t, vp, C.c_size_t, C.c_size_t, C.c_size_t]
        L.pq_memcpy_d2h_pitched.argtypes = [vp, vp, C.c_size_t, vp, C.c_size_t, C.c_size_t, C.c_size_t]
        L.pq_device_count.argtypes = [C.POINTER(C.c_int32)]
        L.pq_suite_begin.argtypes = [vp, C.POINTER(Batch)]
        L.pq_suite_end.argtypes = [vp, C.POINTER(vp)]
        L.pq_suite_abort.argtypes = [vp]
        L.pq_suite_run.argtypes = [vp, vp]
        L.pq_suite_destroy.argtypes = [vp, vp]
        L.pq_suite_info.argtypes = [vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.pq_suite_set_timing.argtypes = [vp, C.c_int32]
        L.pq_suite_grid_stats.argtypes = [vp, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int32),
                                          C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.pq_suite_grid_variant.argtypes = [vp, C.c_int32, C.POINTER(C.c_int32)]
        L.pq_suite_span_stats.argtypes = [vp, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        _lib = L
    return _lib


def check(status: int) -> None:
    if status != PQ_OK:
        msg = lib().pq_last_error().decode("utf-8", "replace")
        raise (NullsNotAllowed if status == 3 else PqError)(f"pq status {status}: {msg}")
