"""Host side of the drop-in: batches of series in, batches out, computed by the HIP library.

Inputs may be torch CUDA tensors (zero-copy), numpy arrays / pyarrow arrays / polars Series (copied to the
device through torch); shapes [T] (one series) or [N, T] (N symbols, symbol-major = long format sorted by
symbol, date).  Outputs come back in the container kind of the first input.  Nulls travel as the NaN bit
pattern NULL_BITS on the device and are converted from/to Arrow validity at the edge.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import Batch, BtParams, NullsNotAllowed, PqError, check, lib
from ._spec import BT_DEFAULTS, EXTRA, I, NB, PATTERN_NAMES, PATTERN_PEN_DEFAULT, SPEC, SUMMARY_KEYS

NULL = np.array([_lib.NULL_BITS], dtype=np.uint64).view(np.float64)[0]

_ctx_cache = {}


def _require_gpu():
    if not torch.cuda.is_available():
        raise PqError("polars_quant_amd needs an MI355X (HIP device); there is no CPU fallback")


def ctx(device: int | None = None):
    """pq_ctx bound to torch's current stream on `device` (cached per device+stream)."""
    _require_gpu()
    dev = torch.cuda.current_device() if device is None else device
    stream = torch.cuda.current_stream(dev).cuda_stream
    key = (dev, stream)
    h = _ctx_cache.get(key)
    if h is None:
        out = C.c_void_p()
        check(lib().pq_ctx_create(dev, C.c_void_p(stream) if stream else None, C.byref(out)))
        h = out
        _ctx_cache[key] = h
    return h


class _Kind:
    TORCH, NUMPY, ARROW, POLARS = range(4)


def _to_device(x, dtype=torch.float64):
    """-> (tensor [N,T] on cuda, kind, squeeze)"""
    kind = _Kind.TORCH
    if isinstance(x, torch.Tensor):
        t = x
    else:
        mod = type(x).__module__.split(".")[0]
        if mod == "polars":
            kind = _Kind.POLARS
            x = x.to_arrow()
            mod = "pyarrow"
        if mod == "pyarrow":
            if kind != _Kind.POLARS:
                kind = _Kind.ARROW
            import pyarrow as pa
            if isinstance(x, pa.ChunkedArray):
                x = x.combine_chunks()
            arr = x.cast(pa.float64()) if dtype == torch.float64 else x
            vals = arr.to_numpy(zero_copy_only=False).astype(np.float64 if dtype == torch.float64 else np.uint8, copy=True)
            if arr.null_count:
                mask = np.asarray(arr.is_null())
                if dtype == torch.float64:
                    vals[mask] = NULL
                else:
                    vals[mask] = 0
            t = torch.from_numpy(vals)
        else:
            kind = _Kind.NUMPY
            t = torch.from_numpy(np.ascontiguousarray(np.asarray(x)))
    if t.dtype != dtype:
        t = t.to(torch.uint8 if dtype == torch.uint8 else dtype)
    squeeze = t.dim() == 1
    if squeeze:
        t = t.unsqueeze(0)
    if t.dim() != 2:
        raise PqError("inputs must be [T] or [N, T]")
    if not t.is_cuda:
        _require_gpu()
        n, T = t.shape
        pitch = recommended_stride(T)
        if n > 1 and pitch != T and dtype == torch.float64:
            # a host column is uploaded into a device copy this library owns: place it at the 128-byte row pitch straight away
            # (pq_recommended_stride; one strided H2D copy instead of a dense one -- a dense ODD-T copy would run the 8-byte forms)
            buf = torch.empty((n, pitch), dtype=dtype, device="cuda")
            buf[:, :T].copy_(t)
            t = buf[:, :T]
        else:
            t = t.cuda()
    elif t.dim() == 2 and t.shape[0] > 1 and t.dtype == torch.float64 and t.stride(1) == 1 and (t.stride(0) % 2 or t.data_ptr() % 16):
        _warn_slow_layout(t)
    if t.stride(1) != 1 and t.shape[1] > 1:
        t = t.contiguous()
    return t, kind, squeeze


def recommended_stride(T: int) -> int:
    """pq_recommended_stride: the row pitch (elements) to allocate [N, T] device columns with -- the smallest multiple of 128 B >= T"""
    return (int(T) + 15) // 16 * 16


class PqLayoutWarning(UserWarning):
    """A device tensor handed to the library sits on a layout whose rows are only 8-byte aligned (odd row pitch, or a base 8 bytes off a
    16-byte boundary): every kernel then runs its 8-byte form, about 1.5 x slower (pq_layout_check returns PQ_WARN_SLOW_LAYOUT).  Results
    are identical.  Allocate [N, recommended_stride(T)] and pass the [:, :T] view, or hand over host data (uploaded pitched), or use
    Suite / loader.DeviceFrame, which re-house such inputs once."""


_warned_layout = False


def _warn_slow_layout(t: torch.Tensor) -> None:
    global _warned_layout
    if _warned_layout:
        return
    _warned_layout = True
    import warnings
    warnings.warn(f"polars_quant_amd: device tensor of shape {tuple(t.shape)} has a row pitch of {t.stride(0)} elements / a base that is not "
                  f"16-byte aligned: the 8-byte kernel forms run (~1.5x slower, same results); use a row pitch of "
                  f"{recommended_stride(t.shape[1])} (api.recommended_stride) -- warned once per process", PqLayoutWarning, stacklevel=4)


def _from_device(t: torch.Tensor, kind: int, squeeze: bool, name: str = ""):
    if squeeze:
        t = t[0]
    if kind == _Kind.TORCH:
        return t
    a = np.ascontiguousarray(t.cpu().numpy())   # (a pitched device column comes back dense)
    if kind == _Kind.NUMPY:
        return a
    import pyarrow as pa
    if a.dtype == np.float64:
        mask = a.view(np.uint64) == np.uint64(_lib.NULL_BITS)
    else:
        mask = a == np.int32(_lib.NULL_I32)
    arr = pa.array(a.ravel(), mask=mask.ravel())
    if kind == _Kind.POLARS:
        import polars as pl
        return pl.Series(name, arr)
    return arr


def _batch_of(t: torch.Tensor) -> Batch:
    n, T = t.shape
    return Batch(n, T, t.stride(0) if n > 1 else max(T, t.stride(0) if t.stride(0) >= T else T))


def _same_layout(ts):
    n, T = ts[0].shape
    s0 = ts[0].stride(0)
    out = []
    for t in ts:
        if t.shape != (n, T):
            raise PqError(f"input shapes differ: {tuple(t.shape)} vs {(n, T)}")
        if t.stride(0) != s0 or t.stride(1) != 1:
            t = t.contiguous()
            if s0 != T:
                ts0 = ts[0].contiguous()
                return _same_layout([ts0] + [x.contiguous() for x in ts[1:]])
        out.append(t)
    return out


def count_nulls(t: torch.Tensor) -> int:
    t2, _, _ = _to_device(t)
    b = _batch_of(t2)
    n = C.c_int64(0)
    check(lib().pq_count_nulls(ctx(t2.device.index), C.byref(b), C.c_void_p(t2.data_ptr()), C.byref(n)))
    return n.value


def ragged_batch(offsets, device):
    """offsets: n + 1 ascending row indices (series s = rows [offsets[s], offsets[s + 1]) of the long columns, sorted by symbol:
    the groups of `.over("symbol")`) -> (Batch, the device copy it points at)."""
    off = torch.as_tensor(np.asarray(offsets.cpu() if isinstance(offsets, torch.Tensor) else offsets), dtype=torch.int64).reshape(-1)
    if off.numel() < 1 or (off.numel() > 1 and bool((off[1:] < off[:-1]).any())) or int(off[0]) != 0:
        raise PqError("offsets must be n + 1 ascending row indices starting at 0 (rows in front of the first group would belong to no "
                      "group and their outputs would stay unwritten)")
    n = off.numel() - 1
    longest = int((off[1:] - off[:-1]).max()) if n else 0
    total = int(off[-1])
    dev_off = off.to(device)
    return Batch(n, longest, total, C.c_void_p(dev_off.data_ptr())), dev_off


def call(name: str, *inputs, check_nulls: bool = False, offsets=None, **params):
    """Run indicator `name` (lower-case plugin name, e.g. "ema") -> tuple of outputs.
    offsets: run it over the groups of LONG columns ([rows], sorted by symbol) instead of [N, T] matrices -- every group as if it
    were passed alone, one launch for all of them (what the reference does with one plugin call per group under .over("symbol"))."""
    if name in PATTERN_NAMES:
        return (cdl(name, *inputs, offsets=offsets, **params),)
    cols, pspec, outs, fam = SPEC[name] if name in SPEC else EXTRA[name]
    if len(inputs) != len(cols):
        raise TypeError(f"{name}() takes inputs {cols}")
    conv = [_to_device(x) for x in inputs]
    kind, squeeze = conv[0][1], conv[0][2]
    ts = _same_layout([c[0] for c in conv])
    if offsets is not None:
        if not squeeze:
            raise PqError("with offsets the inputs are long columns [rows]")
        ts = [t.contiguous() for t in ts]
        dev = ts[0].device
        b, keep = ragged_batch(offsets, dev)
        if b.stride != ts[0].shape[1]:
            raise PqError(f"offsets end at row {b.stride} but the columns have {ts[0].shape[1]} rows")
        pvals = []
        for pname, k, default in pspec:
            v = params.pop(pname, default)
            pvals.append(C.c_int64(int(v)) if k == I else C.c_double(float(v)))
        if params:
            raise TypeError(f"{name}() got unexpected parameters {sorted(params)}")
        res = [torch.empty((1, b.stride), dtype=torch.float64 if dt == "f8" else torch.int32, device=dev) for _, dt in outs]
        with torch.cuda.device(dev):
            if b.n_series and b.stride:
                check(getattr(lib(), "pq_" + name)(ctx(dev.index), C.byref(b), *[C.c_void_p(t.data_ptr()) for t in ts], *pvals,
                                                   *[C.c_void_p(r.data_ptr()) for r in res]))
        del keep
        return tuple(_from_device(r, kind, True, oname) for r, (oname, _) in zip(res, outs))
    if fam == NB and (check_nulls or kind in (_Kind.ARROW, _Kind.POLARS)):
        for t in ts:
            if count_nulls(t):
                raise NullsNotAllowed(f"{name}: input contains nulls (the reference's cont_slice() rejects them)")
    dev = ts[0].device
    n, T = ts[0].shape
    b = Batch(n, T, ts[0].stride(0) if n > 1 else T)
    pvals = []
    for pname, k, default in pspec:
        v = params.pop(pname, default)
        pvals.append(C.c_int64(int(v)) if k == I else C.c_double(float(v)))
    if params:
        raise TypeError(f"{name}() got unexpected parameters {sorted(params)}")
    res = [torch.empty_strided((n, T), (b.stride, 1), dtype=torch.float64 if dt == "f8" else torch.int32, device=dev)
           for _, dt in outs]
    with torch.cuda.device(dev):
        fn = getattr(lib(), "pq_" + name)
        if n * T > 0:  # an empty batch has no device storage to point at
            check(fn(ctx(dev.index), C.byref(b), *[C.c_void_p(t.data_ptr()) for t in ts], *pvals,
                     *[C.c_void_p(r.data_ptr()) for r in res]))
    return tuple(_from_device(r, kind, squeeze, oname) for r, (oname, _) in zip(res, outs))


def cdl(name: str, open, high, low, close, penetration: float | None = None, offsets=None):
    pid = PATTERN_NAMES.index(name)
    pen = PATTERN_PEN_DEFAULT[name] if penetration is None else float(penetration)
    conv = [_to_device(x) for x in (open, high, low, close)]
    kind, squeeze = conv[0][1], conv[0][2]
    ts = _same_layout([c[0] for c in conv])
    if offsets is not None:
        ts = [t.contiguous() for t in ts]
        dev = ts[0].device
        b, keep = ragged_batch(offsets, dev)
        out = torch.zeros((1, b.stride), dtype=torch.int32, device=dev)
        with torch.cuda.device(dev):
            if b.n_series and b.stride:
                check(lib().pq_cdl(ctx(dev.index), C.byref(b), pid, *[C.c_void_p(t.data_ptr()) for t in ts], C.c_double(pen),
                                   C.c_void_p(out.data_ptr())))
        del keep
        return _from_device(out, kind, True, name)
    if kind in (_Kind.ARROW, _Kind.POLARS):
        for t in ts:
            if count_nulls(t):
                raise NullsNotAllowed(f"{name}: input contains nulls")
    dev = ts[0].device
    n, T = ts[0].shape
    b = Batch(n, T, ts[0].stride(0) if n > 1 else T)
    out = torch.empty_strided((n, T), (b.stride, 1), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        check(lib().pq_cdl(ctx(dev.index), C.byref(b), pid, *[C.c_void_p(t.data_ptr()) for t in ts], C.c_double(pen),
                           C.c_void_p(out.data_ptr())))
    return _from_device(out, kind, squeeze, name)


def cdl_all(open, high, low, close, penetrations: dict | None = None, names=None):
    """All (or `names`) candlestick recognisers in one pass over OHLC -> dict name -> int32 [N,T]."""
    conv = [_to_device(x) for x in (open, high, low, close)]
    kind, squeeze = conv[0][1], conv[0][2]
    ts = _same_layout([c[0] for c in conv])
    dev = ts[0].device
    n, T = ts[0].shape
    b = Batch(n, T, ts[0].stride(0) if n > 1 else T)
    want = PATTERN_NAMES if names is None else list(names)
    outs = {nm: torch.empty_strided((n, T), (b.stride, 1), dtype=torch.int32, device=dev) for nm in want}
    pens = (C.c_double * 61)(*[(penetrations or {}).get(nm, PATTERN_PEN_DEFAULT[nm]) for nm in PATTERN_NAMES])
    ptrs = (C.c_void_p * 61)(*[outs[nm].data_ptr() if nm in outs else None for nm in PATTERN_NAMES])
    with torch.cuda.device(dev):
        check(lib().pq_cdl_all(ctx(dev.index), C.byref(b), *[C.c_void_p(t.data_ptr()) for t in ts], pens, ptrs))
    return {nm: _from_device(o, kind, squeeze, nm) for nm, o in outs.items()}


def backtest_vectorized(price, buy, sell, benchmark=None, want_curves: bool = True, offsets=None, **kw):
    """Batched VectorizedBacktester.run(): -> (position, cash, equity, summary[N,8]) (curves None if not wanted).
    offsets: the columns are long columns of ragged groups (see call()); summary [n_groups, 8]."""
    prm = BtParams(**{**BT_DEFAULTS, **kw})
    p, kind, squeeze = _to_device(price)
    bu, _, _ = _to_device(buy, torch.uint8)
    se, _, _ = _to_device(sell, torch.uint8)
    p = p.contiguous(); bu = bu.contiguous(); se = se.contiguous()
    dev = p.device
    n, T = p.shape
    if offsets is not None:
        b, keep = ragged_batch(offsets, dev)
        bm = _to_device(benchmark)[0].contiguous() if benchmark is not None else None
        for nm, t in (("buy", bu), ("sell", se), ("benchmark", bm)):
            if t is not None and t.shape != p.shape:
                raise PqError(f"backtest_vectorized: `{nm}` must be a long column like `price`")
        mk = lambda: torch.empty((1, b.stride), dtype=torch.float64, device=dev)
        pos, cash, eq = (mk(), mk(), mk()) if want_curves else (None, None, None)
        summ = torch.empty((b.n_series, 8), dtype=torch.float64, device=dev)
        vp = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        with torch.cuda.device(dev):
            if b.n_series:
                check(lib().pq_backtest_vectorized(ctx(dev.index), C.byref(b), vp(p), vp(bu), vp(se), vp(bm), C.byref(prm),
                                                   vp(pos), vp(cash), vp(eq), vp(summ)))
        del keep
        f = lambda t, sq: _from_device(t, _Kind.TORCH if kind == _Kind.TORCH else _Kind.NUMPY, sq) if t is not None else None
        return f(pos, True), f(cash, True), f(eq, True), f(summ, False)
    # the kernel indexes every column as base + s * stride: all of them must be [N, T] like the prices
    for nm, t in (("buy", bu), ("sell", se)):
        if t.shape == (1, T) and n > 1:
            raise PqError(f"backtest_vectorized: `{nm}` is one series but `price` has {n}; signals are per series ([N, T])")
        if t.shape != (n, T):
            raise PqError(f"backtest_vectorized: `{nm}` has shape {tuple(t.shape)}, expected {(n, T)}")
    bm = None
    if benchmark is not None:
        bm = _to_device(benchmark)[0]
        if bm.shape == (1, T) and n > 1:
            bm = bm.expand(n, T)          # one benchmark series shared by all symbols (the reference is single-asset)
        if bm.shape != (n, T):
            raise PqError(f"backtest_vectorized: `benchmark` has shape {tuple(bm.shape)}, expected {(T,)} or {(n, T)}")
        bm = bm.contiguous()
    b = Batch(n, T, T)
    mk = lambda: torch.empty((n, T), dtype=torch.float64, device=dev)
    pos, cash, eq = (mk(), mk(), mk()) if want_curves else (None, None, None)
    summ = torch.empty((n, 8), dtype=torch.float64, device=dev)
    vp = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    with torch.cuda.device(dev):
        check(lib().pq_backtest_vectorized(ctx(dev.index), C.byref(b), vp(p), vp(bu), vp(se), vp(bm), C.byref(prm),
                                           vp(pos), vp(cash), vp(eq), vp(summ)))
    f = lambda t: _from_device(t, kind if kind != _Kind.ARROW and kind != _Kind.POLARS else _Kind.NUMPY, squeeze) if t is not None else None
    return f(pos), f(cash), f(eq), f(summ)


def backtest_macd_cross(close, fastperiod=12, slowperiod=26, signalperiod=9, want_curves: bool = True, offsets=None, **kw):
    """Fused MACD-cross strategy + per-symbol backtest + summary (one kernel).  offsets: `close` is a long column of ragged
    groups (see call()); the curves come back as long columns, the summary as [n_groups, 8]."""
    prm = BtParams(**{**BT_DEFAULTS, **kw})
    p, kind, squeeze = _to_device(close)
    p = p.contiguous()
    dev = p.device
    n, T = p.shape
    b = Batch(n, T, T)
    keep = None
    if offsets is not None:
        b, keep = ragged_batch(offsets, dev)
        n = b.n_series
    mk = lambda: torch.empty(tuple(p.shape), dtype=torch.float64, device=dev)
    pos, cash, eq = (mk(), mk(), mk()) if want_curves else (None, None, None)
    summ = torch.empty((n, 8), dtype=torch.float64, device=dev)
    vp = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    with torch.cuda.device(dev):
        if n:
            check(lib().pq_backtest_macd_cross(ctx(dev.index), C.byref(b), vp(p), fastperiod, slowperiod, signalperiod,
                                               C.byref(prm), vp(pos), vp(cash), vp(eq), vp(summ)))
    del keep
    f = lambda t, sq=squeeze: _from_device(t, kind if kind in (_Kind.TORCH, _Kind.NUMPY) else _Kind.NUMPY, sq) if t is not None else None
    return f(pos), f(cash), f(eq), f(summ, squeeze and offsets is None)


def backtest_wave_stats(reset: bool = False, device=None):
    """(symbols run by the wave-per-symbol backtest, speculative chunks that failed the bit test, chunk re-runs) since the last
    reset -- pq_backtest_wave_stats (csrc/ops_backtest_wave.h).  Synchronises the stream."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    out = (C.c_int64 * 3)()
    with torch.cuda.device(dev):
        check(lib().pq_backtest_wave_stats(ctx(dev.index), out, 1 if reset else 0))
    return tuple(int(v) for v in out)


def ragged_rehouse_stats(reset: bool = False, device=None) -> int:
    """launches on a ragged batch that took the re-housed tiled path (pq_ragged_rehouse_stats) since the last reset"""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    out = C.c_int64(0)
    with torch.cuda.device(dev):
        check(lib().pq_ragged_rehouse_stats(ctx(dev.index), C.byref(out), 1 if reset else 0))
    return int(out.value)


def wt_stats(reset: bool = False, device=None):
    """(symbols computed by the wave-per-symbol indicator kernels, speculative chunks that failed the bit test, chunk re-runs,
    symbols handed to the gated lane-per-symbol path) since the last reset -- pq_wt_stats (csrc/wt_dev.h).  Synchronises."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    out = (C.c_int64 * 4)()
    with torch.cuda.device(dev):
        check(lib().pq_wt_stats(ctx(dev.index), out, 1 if reset else 0))
    return tuple(int(v) for v in out)


def factor_ic(factor, fwd_return, method: int = 0):
    """D-12: per-day cross-sectional IC of factor vs forward return, both [N, T] -> (ic [T], n_valid [T]) device tensors.
    method 0 = Pearson IC, 1 = Spearman Rank-IC"""
    f = _to_device(factor)[0].contiguous()
    r = _to_device(fwd_return)[0].contiguous()
    if f.shape != r.shape:
        raise ValueError("factor and fwd_return must have the same shape")
    dev = f.device
    n, T = f.shape
    ic = torch.empty(T, dtype=torch.float64, device=dev)
    nv = torch.zeros(T, dtype=torch.int32, device=dev)
    b = Batch(n, T, T)
    if T:
        with torch.cuda.device(dev):
            check(lib().pq_factor_ic(ctx(dev.index), C.byref(b), C.c_void_p(f.data_ptr()) if n else None,
                                     C.c_void_p(r.data_ptr()) if n else None, int(method), C.c_void_p(ic.data_ptr()),
                                     C.c_void_p(nv.data_ptr())) if n else 0)
        if n == 0:
            ic.fill_(float("nan"))
    return ic, nv


def rolling_ic(ic, window: int):
    """D-12: rolling mean of the IC series and its information ratio -> (rolling_ic, rolling_ir) device tensors [T]"""
    t = _to_device(ic)[0].contiguous().reshape(-1)
    dev = t.device
    ric, rir = torch.empty_like(t), torch.empty_like(t)
    if t.numel():
        with torch.cuda.device(dev):
            check(lib().pq_rolling_ic(ctx(dev.index), C.c_void_p(t.data_ptr()), t.numel(), int(window), C.c_void_p(ric.data_ptr()),
                                      C.c_void_p(rir.data_ptr())))
    return ric, rir


def _signal_call(fn_name, cols, *scalars):
    ts = [_to_device(c)[0].contiguous() for c in cols]
    dev = ts[0].device
    n, T = ts[0].shape
    for t in ts:
        if t.shape != (n, T):
            raise ValueError("signal rule inputs must have the same shape")
    b = Batch(n, T, T)
    buy = torch.zeros((n, T), dtype=torch.uint8, device=dev)
    sell = torch.zeros((n, T), dtype=torch.uint8, device=dev)
    if n * T:
        with torch.cuda.device(dev):
            check(getattr(lib(), fn_name)(ctx(dev.index), C.byref(b), *[C.c_void_p(t.data_ptr()) for t in ts], *scalars,
                                          C.c_void_p(buy.data_ptr()), C.c_void_p(sell.data_ptr())))
    return buy, sell


def cross_signals(a, b):
    """D-11: buy = a crosses above b, sell = a crosses below b -> (buy, sell) uint8 device tensors [N, T]"""
    return _signal_call("pq_cross_signals", [a, b])


def band_signals(x, lower: float, upper: float):
    """D-11: buy = x comes back up through `lower`, sell = x comes back down through `upper`"""
    return _signal_call("pq_band_signals", [x], float(lower), float(upper))


def channel_signals(price, lo, hi, mode: int):
    """D-11: mode 0 = reversion at the bands, mode 1 = breakout of the previous bar's channel"""
    return _signal_call("pq_channel_signals", [price, lo, hi], int(mode))


def _dev2(x, dtype=torch.float64):
    return _to_device(x, dtype)[0].contiguous()


def _rule(fn_name, shape_like, args, outs, same=()):
    """one of the strategy rule kernels (csrc/strategy.hip): args = ctypes values, outs = output tensors (returned); `same`:
    the other tensor arguments, which must have shape_like's shape (the kernel indexes all of them with one batch)"""
    for t in same:
        if t is not None and tuple(t.shape) != tuple(shape_like.shape):
            raise PqError(f"{fn_name}: input shapes differ: {tuple(t.shape)} vs {tuple(shape_like.shape)}")
    n, T = shape_like.shape
    b = Batch(n, T, T)
    if n * T:
        with torch.cuda.device(shape_like.device):
            check(getattr(lib(), fn_name)(ctx(shape_like.device.index), C.byref(b), *args, *[C.c_void_p(o.data_ptr()) for o in outs]))
    return outs


def gate_signals(buy, sell, a, mode: int, k0: float = 0.0, k1: float = 0.0, c=None):
    """pq_gate_signals: filter existing signals by a comparison on column `a` (see include/pq_hip.h for the modes)"""
    a_, bu, se = _dev2(a), _dev2(buy, torch.uint8), _dev2(sell, torch.uint8)
    c_ = _dev2(c) if c is not None else None
    outs = [torch.empty_like(bu), torch.empty_like(se)]
    return tuple(_rule("pq_gate_signals", a_, [C.c_void_p(a_.data_ptr()), C.c_void_p(c_.data_ptr()) if c_ is not None else None, int(mode),
                                               C.c_double(float(k0)), C.c_double(float(k1)), C.c_void_p(bu.data_ptr()), C.c_void_p(se.data_ptr())], outs, same=(bu, se, c_)))


def zscore(price, upper, mid):
    p, u, m = _dev2(price), _dev2(upper), _dev2(mid)
    return _rule("pq_zscore", p, [C.c_void_p(t.data_ptr()) for t in (p, u, m)], [torch.empty_like(p)], same=(u, m))[0]


def scale_band(base, f_lo: float, f_hi: float):
    b_ = _dev2(base)
    return tuple(_rule("pq_scale_band", b_, [C.c_void_p(b_.data_ptr()), C.c_double(float(f_lo)), C.c_double(float(f_hi))],
                       [torch.empty_like(b_), torch.empty_like(b_)]))


def _u8_pair(like):
    return [torch.zeros(like.shape, dtype=torch.uint8, device=like.device), torch.zeros(like.shape, dtype=torch.uint8, device=like.device)]


def volume_surge_signals(volume, avg_volume, close, multiplier: float):
    v, sv, c = _dev2(volume), _dev2(avg_volume), _dev2(close)
    return tuple(_rule("pq_volume_surge_signals", v, [C.c_void_p(t.data_ptr()) for t in (v, sv, c)] + [C.c_double(float(multiplier))], _u8_pair(v), same=(sv, c)))


def gap_signals(open, high, low, f_up: float, f_dn: float):
    o, h, l = _dev2(open), _dev2(high), _dev2(low)
    return tuple(_rule("pq_gap_signals", o, [C.c_void_p(t.data_ptr()) for t in (o, h, l)] + [C.c_double(float(f_up)), C.c_double(float(f_dn))], _u8_pair(o), same=(h, l)))


def pattern_any_signals(bullish, bearish):
    """bullish / bearish: lists of int32 recogniser columns (device) -> (buy, sell)"""
    bu = [_dev2(t, torch.int32) for t in bullish]
    be = [_dev2(t, torch.int32) for t in bearish]
    like = (bu + be)[0]
    pb = (C.c_void_p * max(len(bu), 1))(*[t.data_ptr() for t in bu])
    ps = (C.c_void_p * max(len(be), 1))(*[t.data_ptr() for t in be])
    return tuple(_rule("pq_pattern_any_signals", like, [pb, len(bu), ps, len(be)], _u8_pair(like), same=bu + be))


def ma_stack_signals(mas):
    ms = [_dev2(t) for t in mas]
    ptrs = (C.c_void_p * len(ms))(*[t.data_ptr() for t in ms])
    return tuple(_rule("pq_ma_stack_signals", ms[0], [ptrs, len(ms)], _u8_pair(ms[0]), same=ms))


def backtest_leveraged(price, buy, sell, benchmark=None, max_trades: int = 64, **kw):
    """README `Backtest` engine (decision D-10): price/buy/sell [N, T], benchmark [T] or None.
    -> dict of device tensors: cash, stock_value, total_value [N,T]; trade_count [N]; trades {field: [N,max_trades]};
    summary [N,8]"""
    from ._lib import LevParams
    from ._spec import LEV_DEFAULTS, TRADE_FIELDS
    prm = LevParams(**{**LEV_DEFAULTS, **kw})
    p, _, _ = _to_device(price)
    bu, _, _ = _to_device(buy, torch.uint8)
    se, _, _ = _to_device(sell, torch.uint8)
    p = p.contiguous(); bu = bu.contiguous(); se = se.contiguous()
    dev = p.device
    n, T = p.shape
    bm = None
    if benchmark is not None:
        bm = _to_device(benchmark)[0].contiguous().reshape(-1)
        if bm.numel() != T:
            raise ValueError("benchmark must be one series with as many rows as the prices")
    b = Batch(n, T, T)
    mk = lambda: torch.empty((n, T), dtype=torch.float64, device=dev)
    cash, sv, tv = mk(), mk(), mk()
    cnt = torch.zeros(n, dtype=torch.int32, device=dev)
    tr = {k: torch.zeros((n, max_trades), dtype=torch.int32 if k in ("entry_day", "exit_day", "reason") else torch.float64, device=dev)
          for k in TRADE_FIELDS}
    summ = torch.empty((n, 8), dtype=torch.float64, device=dev)
    vp = lambda t: C.c_void_p(t.data_ptr()) if t is not None and t.numel() else None
    if n * T:
        with torch.cuda.device(dev):
            check(lib().pq_backtest_leveraged(ctx(dev.index), C.byref(b), vp(p), vp(bu), vp(se), vp(bm), C.byref(prm), vp(cash), vp(sv),
                                              vp(tv), max_trades, vp(cnt),
                                              *[vp(tr[k]) if max_trades > 0 else None for k in ("entry_day", "exit_day", "entry_price", "exit_price",
                                                                          "quantity", "pnl", "pnl_pct", "reason")], vp(summ)))
    return dict(cash=cash, stock_value=sv, total_value=tv, trade_count=cnt, trades=tr, summary=summ)


def portfolio_metrics(total_value, initial_total: float, benchmark=None):
    """get_performance_metrics: total_value [N, T] -> [T, 10] device tensor (columns: _spec.PORTFOLIO_COLS + reserved)"""
    tv, _, _ = _to_device(total_value)
    tv = tv.contiguous()
    dev = tv.device
    n, T = tv.shape
    bm = _to_device(benchmark)[0].contiguous().reshape(-1) if benchmark is not None else None
    out = torch.zeros((T, 10), dtype=torch.float64, device=dev)
    b = Batch(n, T, T)
    if n * T:
        with torch.cuda.device(dev):
            check(lib().pq_portfolio_metrics(ctx(dev.index), C.byref(b), C.c_void_p(tv.data_ptr()), float(initial_total),
                                             C.c_void_p(bm.data_ptr()) if bm is not None else None, C.c_void_p(out.data_ptr())))
    return out


def macd_cross_signals(close, fastperiod=12, slowperiod=26, signalperiod=9):
    p, kind, squeeze = _to_device(close)
    p = p.contiguous()
    dev = p.device
    n, T = p.shape
    b = Batch(n, T, T)
    bu = torch.empty((n, T), dtype=torch.uint8, device=dev)
    se = torch.empty((n, T), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        check(lib().pq_macd_cross_signals(ctx(dev.index), C.byref(b), C.c_void_p(p.data_ptr()), fastperiod, slowperiod,
                                          signalperiod, C.c_void_p(bu.data_ptr()), C.c_void_p(se.data_ptr())))
    k = kind if kind in (_Kind.TORCH, _Kind.NUMPY) else _Kind.NUMPY
    return _from_device(bu, k, squeeze), _from_device(se, k, squeeze)
