"""UPPER-CASE TA-Lib style wrappers, one per reference wrapper (python/polars_quant/talib/*.py).

Same names, positional order and defaults as the reference.  A single-output function returns one
array, a multi-output one a tuple (the reference returns a tuple of Series / struct fields).
"""
from __future__ import annotations

from .. import api as _api
from .._spec import PATTERN_NAMES as _PN
from .._spec import PATTERNS_WITH_PEN_ARG as _PEN
from .._spec import SPEC as _SPEC

__all__ = []


def _make(name):
    cols, pspec, outs, _ = _SPEC[name]
    pnames = [p for p, _, _ in pspec]

    def fn(*args, **kwargs):
        if len(args) < len(cols):
            raise TypeError(f"{name.upper()}() missing inputs {cols[len(args):]}")
        inputs, extra = args[:len(cols)], args[len(cols):]
        if len(extra) > len(pnames):
            raise TypeError(f"{name.upper()}() takes at most {len(cols) + len(pnames)} positional arguments")
        params = dict(zip(pnames, extra))
        for k, v in kwargs.items():
            if k in params:
                raise TypeError(f"{name.upper()}() got multiple values for {k}")
            params[k] = v
        res = _api.call(name, *inputs, **params)
        return res[0] if len(res) == 1 else res

    fn.__name__ = name.upper()
    fn.__doc__ = f"{name.upper()}({', '.join(cols + [f'{p}={d!r}' for p, _, d in pspec])}) -> {[o for o, _ in outs]}"
    return fn


def _make_cdl(name):
    has_pen = name in _PEN

    def fn(open, high, low, close, penetration=None):
        return _api.cdl(name, open, high, low, close, penetration)

    def fn_nopen(open, high, low, close):
        return _api.cdl(name, open, high, low, close, None)

    f = fn if has_pen else fn_nopen
    f.__name__ = name.upper()
    return f


for _n in _SPEC:
    globals()[_n.upper()] = _make(_n)
    __all__.append(_n.upper())
for _n in _PN:
    globals()[_n.upper()] = _make_cdl(_n)
    __all__.append(_n.upper())
CDL_ALL = _api.cdl_all
__all__.append("CDL_ALL")
