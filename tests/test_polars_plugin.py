"""The Polars expression-plugin exporter spike (include/pq_polars_plugin.h).  `polars` is not installed in
this image, so the caller's side -- what Polars does after dlsym -- is played by pyarrow: Series are exported through the
Arrow C Data Interface into hand-built SeriesExport structs.  The struct layout is the published polars-ffi one and is NOT
verified against the pinned polars 0.53."""
import ctypes as C
import pickle
import re
from pathlib import Path

import numpy as np
import pytest

pa = pytest.importorskip("pyarrow")
ROOT = Path(__file__).resolve().parent.parent


class ArrowSchema(C.Structure):
    _fields_ = [("format", C.c_char_p), ("name", C.c_char_p), ("metadata", C.c_char_p), ("flags", C.c_int64), ("n_children", C.c_int64),
                ("children", C.c_void_p), ("dictionary", C.c_void_p), ("release", C.c_void_p), ("private_data", C.c_void_p)]


class ArrowArray(C.Structure):
    _fields_ = [("length", C.c_int64), ("null_count", C.c_int64), ("offset", C.c_int64), ("n_buffers", C.c_int64), ("n_children", C.c_int64),
                ("buffers", C.c_void_p), ("children", C.c_void_p), ("dictionary", C.c_void_p), ("release", C.c_void_p), ("private_data", C.c_void_p)]


class SeriesExport(C.Structure):
    _fields_ = [("field", C.POINTER(ArrowSchema)), ("arrays", C.POINTER(C.POINTER(ArrowArray))), ("len", C.c_size_t), ("release", C.c_void_p),
                ("private_data", C.c_void_p)]


# every reference function of the shape (1..4 Float64 columns[, timeperiod]) -> Float64 has a plugin symbol pair
PLUGIN_FUNCS = ["sma", "ema", "wma", "dema", "tema", "trima", "kama", "midpoint", "rsi", "cmo", "mom", "roc", "rocp", "rocr", "rocr100",
                "trix", "ht_dcperiod", "ht_dcphase", "ht_trendline", "midprice", "plus_dm", "minus_dm", "aroonosc", "medprice", "obv",
                "adx", "adxr", "dx", "plus_di", "minus_di", "cci", "willr", "atr", "natr", "trange", "typprice", "wclprice", "mfi", "bop",
                "ad", "avgprice"]
# ... and the functions with other scalar parameters: name -> (input columns, [(parameter, reference default, a test value)])
MULTI_FUNCS = {"ma": (["real"], [("timeperiod", 30, 12), ("matype", 0, 1)]),
               "t3": (["real"], [("timeperiod", 5, 7), ("vfactor", 0.0, 0.7)]),
               "ultosc": (["high", "low", "close"], [("timeperiod1", 7, 5), ("timeperiod2", 14, 9), ("timeperiod3", 28, 20)]),
               "adosc": (["high", "low", "close", "volume"], [("fastperiod", 3, 4), ("slowperiod", 10, 13)]),
               "sar": (["high", "low"], [("acceleration", 0.0, 0.02), ("maximum", 0.0, 0.2)]),
               "sarext": (["high", "low"], [("startvalue", 0.0, 0.0), ("offsetonreverse", 0.0, 0.01), ("accelerationinitlong", 0.0, 0.02),
                                            ("accelerationlong", 0.0, 0.02), ("accelerationmaxlong", 0.0, 0.2),
                                            ("accelerationinitshort", 0.0, 0.03), ("accelerationshort", 0.0, 0.03),
                                            ("accelerationmaxshort", 0.0, 0.3)]),
               "ht_trendmode": (["real"], []),
               "apo": (["real"], [("fastperiod", 12, 5), ("slowperiod", 26, 13), ("matype", 0, 1)]),
               "ppo": (["real"], [("fastperiod", 12, 5), ("slowperiod", 26, 13), ("matype", 0, 1)]),
               "mavp": (["real", "periods"], [("minperiod", 2, 3), ("maxperiod", 30, 20), ("matype", 0, 1)])}
# ... and the Struct-valued ones: name -> (struct name, input columns, [(parameter, reference default, a test value)], field names)
STRUCT_FUNCS = {"bbands": ("bbands", ["real"], [("timeperiod", 20, 10), ("nbdevup", 2.0, 1.5), ("nbdevdn", 2.0, 2.5)], ["bb_upper", "bb_middle", "bb_lower"]),
                "mama": ("mama", ["real"], [("fastlimit", 0.0, 0.5), ("slowlimit", 0.0, 0.05)], ["mama", "fama"]),
                "aroon": ("aroon", ["high", "low"], [("timeperiod", 14, 9)], ["aroon_up", "aroon_down"]),
                "macd": ("macd_res", ["real"], [("fastperiod", 12, 5), ("slowperiod", 26, 13), ("signalperiod", 9, 4)], ["macd", "macd_signal", "macd_hist"]),
                "ht_phasor": ("ht_phasor", ["real"], [], ["inphase", "quadrature"]),
                "ht_sine": ("ht_sine", ["real"], [], ["sine", "leadsine"])}
TP_FUNCS = PLUGIN_FUNCS + list(MULTI_FUNCS) + list(STRUCT_FUNCS)   # (every symbol pair with the common signature: the argtypes loops below)


def _lib():
    so = ROOT / "polars_quant_amd" / "libpolars_quant_hip.so"
    if not so.exists():
        import __graft_entry__ as g
        g.build()
    try:
        import torch  # noqa: F401  (its bundled HIP runtime must be the first one in the process, as in polars_quant_amd._lib)
    except ImportError:
        pass
    L = C.CDLL(str(so))
    L._polars_plugin_get_version.restype = C.c_uint32
    L._polars_plugin_get_last_error_message.restype = C.c_char_p
    L.pq_plugin_kwargs_i64.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.POINTER(C.c_int64)]
    from polars_quant_amd._spec import PATTERN_NAMES
    both = [n + sfx for n in list(TP_FUNCS) + list(PATTERN_NAMES) for sfx in ("", "_over")]
    for f in ("_polars_plugin_" + n for n in both):
        getattr(L, f).argtypes = [C.POINTER(SeriesExport), C.c_size_t, C.c_char_p, C.c_size_t, C.POINTER(SeriesExport), C.c_void_p]
        getattr(L, f).restype = None
    for f in ("_polars_plugin_field_" + n for n in both):
        getattr(L, f).argtypes = [C.POINTER(ArrowSchema), C.c_size_t, C.POINTER(ArrowSchema), C.c_char_p, C.c_size_t]
        getattr(L, f).restype = None
    return L


def test_plugin_symbols_and_version():
    txt = (ROOT / "include" / "pq_polars_plugin.h").read_text()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    declared = set(re.findall(r"\b(_polars_plugin_[a-z0-9_]+|pq_plugin_[a-z0-9_]+)\s*\(", txt))
    for n in re.findall(r"\bPQ_PLUGIN_DECL\((\w+)\)", txt):      # the per-function pairs are declared through a macro
        if n != "NAME":                                           # (the macro's own parameter)
            declared |= {"_polars_plugin_" + n, "_polars_plugin_field_" + n, "_polars_plugin_" + n + "_over", "_polars_plugin_field_" + n + "_over"}
    declared = {n for n in declared if "##" not in n and not n.endswith("_")}
    from polars_quant_amd._spec import PATTERN_NAMES
    assert {n for n in declared if n.startswith("_polars_plugin_field_")} == {"_polars_plugin_field_" + n + sfx for n in list(TP_FUNCS) + list(PATTERN_NAMES)
                                                                                for sfx in ("", "_over")}
    assert len(TP_FUNCS) + len(PATTERN_NAMES) == 118, "115 Rust names + aroonosc + apo + ppo"
    declared = sorted(declared)
    assert {"_polars_plugin_get_version", "_polars_plugin_get_last_error_message", "_polars_plugin_ema", "_polars_plugin_field_ema"} <= set(declared)
    L = _lib()
    assert not [s for s in declared if not hasattr(L, s)]
    assert L._polars_plugin_get_version() == (0 << 16) | 1


def test_pickled_kwargs_reader():
    """what polars.plugins passes for `kwargs=`: pickle.dumps(dict) (serde-pickle on the Rust side, overlap.rs:18-22)"""
    L = _lib()
    for proto in (2, 3, 4, 5):
        for d, want in (({"timeperiod": 20}, (1, 20)), ({"timeperiod": 300, "matype": None}, (1, 300)), ({"matype": 1, "timeperiod": 70000}, (1, 70000)),
                        ({"timeperiod": 2 ** 40}, (1, 2 ** 40)), ({"timeperiod": -3, "flag": True}, (1, -3)), ({}, (0, None)),
                        ({"timeperiod": None}, (0, None)), ({"vfactor": 0.7}, (0, None))):
            b, v = pickle.dumps(d, protocol=proto), C.c_int64(-999)
            r = L.pq_plugin_kwargs_i64(b, len(b), b"timeperiod", C.byref(v))
            assert r == want[0] and (want[1] is None or v.value == want[1]), (proto, d, r, v.value)
    v = C.c_int64()
    assert L.pq_plugin_kwargs_i64(b"\x80\x05garbage", 9, b"timeperiod", C.byref(v)) == -1
    assert L.pq_plugin_kwargs_i64(pickle.dumps([1, 2]), len(pickle.dumps([1, 2])), b"timeperiod", C.byref(v)) == -1


def _export(chunks, name):
    """pyarrow chunks -> SeriesExport (keeps the ctypes objects alive through the returned tuple)"""
    field = ArrowSchema()
    arrs = [ArrowArray() for _ in chunks]
    ptrs = (C.POINTER(ArrowArray) * len(chunks))(*[C.pointer(a) for a in arrs])
    for i, (ch, a) in enumerate(zip(chunks, arrs)):
        sch = ArrowSchema()
        ch._export_to_c(C.addressof(a), C.addressof(sch) if i else C.addressof(field))
    field_named = pa.field(name, chunks[0].type)
    field2 = ArrowSchema()
    field_named._export_to_c(C.addressof(field2))
    se = SeriesExport(C.pointer(field2), ptrs, len(chunks), None, None)
    return se, (field, field2, arrs, ptrs)


def _import(ret):
    assert ret.release, "the plugin left return_value empty"
    assert ret.len == 1
    return pa.Array._import_from_c(C.addressof(ret.arrays[0].contents), C.addressof(ret.field.contents))


@pytest.mark.gpu
def test_plugin_calls_match_the_oracle(oracle):
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    L = _lib()
    d = oracle.gen_ohlcv(0x5EED000B, 1, 500, 0)
    x = d["close"][0]
    mask = np.zeros(500, bool); mask[[0, 1, 77, 300]] = True
    xn = x.copy(); xn[mask] = oracle.NULL
    whole = pa.array(x, mask=mask)
    # (a) three chunks (one of them a slice with a non-zero offset), timeperiod as pickled kwargs
    chunks = [whole.slice(0, 100), pa.concat_arrays([pa.array([1.0, 2.0, 3.0]), whole.slice(100, 250)]).slice(3), whole.slice(350)]
    assert chunks[1].offset == 3
    se, keep = _export(chunks, "close")
    ret = SeriesExport()
    kw = pickle.dumps({"timeperiod": 20})
    L._polars_plugin_ema(C.byref(se), 1, kw, len(kw), C.byref(ret), None)
    assert ret.release, L._polars_plugin_get_last_error_message()
    assert ret.field.contents.name == b"close" and ret.field.contents.format == b"g"
    got = _import(ret)
    (exp,) = oracle.call("ema", xn, timeperiod=20)
    en = exp.view(np.uint64) == np.uint64(oracle.NULL_BITS)
    assert got.null_count == int(en.sum()) and (np.asarray(got.is_null()) == en).all()
    vals = got.to_numpy(zero_copy_only=False)
    assert (vals[~en].view(np.uint64) == exp[~en].view(np.uint64)).all()
    # (b) the Python wrapper's convention: timeperiod as a trailing literal input (overlap.py:36-43), no kwargs
    se0, keep0 = _export([whole], "px")
    se1, keep1 = _export([pa.array([7], type=pa.int64())], "literal")
    ins = (SeriesExport * 2)(se0, se1)
    ret = SeriesExport()
    L._polars_plugin_sma(ins, 2, None, 0, C.byref(ret), None)
    got = _import(ret)
    (exp,) = oracle.call("sma", xn, timeperiod=7)
    en = exp.view(np.uint64) == np.uint64(oracle.NULL_BITS)
    assert (np.asarray(got.is_null()) == en).all()
    assert (got.to_numpy(zero_copy_only=False)[~en].view(np.uint64) == exp[~en].view(np.uint64)).all()
    # (c) output field + error path (a NON-NUMERIC input is refused with a message, return_value stays empty)
    out_field = ArrowSchema()
    L._polars_plugin_field_ema(se0.field, 1, C.byref(out_field), None, 0)
    assert out_field.format == b"g" and out_field.name == b"px"
    sei, keepi = _export([pa.array(["1", "2", "3"], type=pa.string())], "strs")
    ret = SeriesExport()
    L._polars_plugin_ema(C.byref(sei), 1, None, 0, C.byref(ret), None)
    assert not ret.release and b"not numeric" in L._polars_plugin_get_last_error_message()


def _plugin_call(L, name, series, kwargs=None, literals=()):
    """series: [(pyarrow chunks, name)]; literals: trailing one-row literal arrays -> the imported result array"""
    ses, keep = [], []
    for chunks, nm in list(series) + [([lit], "literal") for lit in literals]:
        se, k = _export(chunks, nm)
        ses.append(se); keep.append(k)
    ins = (SeriesExport * len(ses))(*ses)
    kw = pickle.dumps(kwargs) if kwargs else None
    ret = SeriesExport()
    getattr(L, "_polars_plugin_" + name)(ins, len(ses), kw, len(kw) if kw else 0, C.byref(ret), None)
    assert ret.release, (name, L._polars_plugin_get_last_error_message())
    return _import(ret)


def _same_array(a, b):
    assert a.type == b.type and len(a) == len(b) and a.null_count == b.null_count
    na, nb = np.asarray(a.is_null()), np.asarray(b.is_null())
    assert (na == nb).all()
    va, vb = a.to_numpy(zero_copy_only=False)[~na], b.to_numpy(zero_copy_only=False)[~nb]
    if a.type == pa.float64():
        assert (va.view(np.uint64) == vb.view(np.uint64)).all()
    else:
        assert (va == vb).all()


@pytest.mark.gpu
def test_plugin_casts_numeric_columns_like_the_reference(oracle):
    """Every reference function begins with `inputs[k].cast(&DataType::Float64)?` (overlap.rs:120,129; momentum.rs:12; volume.rs:19-31;
    pattern.rs:11-17): an Int64 `volume`, a Float32 OHLC frame or an unsigned column must give, bit for bit, what the same values
    give as Float64 columns -- through the plain and the `_over` symbols, chunked, with nulls where the function accepts them."""
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    L = _lib()
    n = 600
    d = oracle.gen_ohlcv(0x5EED00CA, 1, n, 0)
    f32 = {c: d[c][0].astype(np.float32) for c in ("open", "high", "low", "close")}
    as64 = {c: f32[c].astype(np.float64) for c in f32}                     # what a cast of the f32 column yields
    vol_i = np.round(d["volume"][0]).astype(np.int64)
    vol_f = vol_i.astype(np.float64)
    mask = np.zeros(n, bool); mask[[3, 250, 251]] = True

    def chunks(values, typ, msk=None):
        arr = pa.array(values, type=typ, mask=msk)
        return [arr.slice(0, 170), arr.slice(170, 0), pa.concat_arrays([arr.slice(0, 2), arr.slice(170)]).slice(2)]

    def pair(name, cols_cast, cols_f64, kwargs=None, literals=()):
        got = _plugin_call(L, name, cols_cast, kwargs, literals)
        want = _plugin_call(L, name, cols_f64, kwargs, literals)
        _same_array(got, want)
        return got

    # Int64 volume through OBV / AD / MFI (volume.rs:19-31, :70; momentum.rs:286-295)
    for vt in (pa.int64(), pa.int32(), pa.uint32(), pa.uint64()):
        pair("obv", [(chunks(as64["close"], pa.float64(), mask), "close"), (chunks(vol_i, vt), "volume")],
             [(chunks(as64["close"], pa.float64(), mask), "close"), (chunks(vol_f, pa.float64()), "volume")])
    hlc64 = [(chunks(as64[c], pa.float64()), c) for c in ("high", "low", "close")]
    r = pair("ad", hlc64 + [(chunks(vol_i, pa.int64()), "volume")], hlc64 + [(chunks(vol_f, pa.float64()), "volume")])
    (exp,) = oracle.call("ad", as64["high"], as64["low"], as64["close"], vol_f)
    assert (r.to_numpy(zero_copy_only=False).view(np.uint64) == exp.view(np.uint64)).all()
    lit = pa.array([10], type=pa.int64())
    pair("mfi", hlc64 + [(chunks(vol_i, pa.int64()), "volume")], hlc64 + [(chunks(vol_f, pa.float64()), "volume")], literals=(lit,))
    # Float32 OHLC through ATR (volatility.rs:18-31), one recogniser (pattern.rs:11-17) and a Struct-valued function
    hlc32 = [(chunks(f32[c], pa.float32(), mask if c == "low" else None), c) for c in ("high", "low", "close")]
    hlc64n = [(chunks(as64[c], pa.float64(), mask if c == "low" else None), c) for c in ("high", "low", "close")]
    pair("atr", hlc32, hlc64n, literals=(pa.array([9], type=pa.int64()),))
    ohlc32 = [(chunks(f32[c], pa.float32()), c) for c in ("open", "high", "low", "close")]
    ohlc64 = [(chunks(as64[c], pa.float64()), c) for c in ("open", "high", "low", "close")]
    for cdl in ("cdlengulfing", "cdldoji", "cdlmorningstar"):
        r = pair(cdl, ohlc32, ohlc64)
        assert r.type == pa.int32()
        assert (r.to_numpy() == oracle.pattern(cdl, *[as64[c][None, :] for c in ("open", "high", "low", "close")])[0]).all()
    got = _plugin_call(L, "bbands", [(chunks(f32["close"], pa.float32(), mask), "close")], {"timeperiod": 10})
    want = _plugin_call(L, "bbands", [(chunks(as64["close"], pa.float64(), mask), "close")], {"timeperiod": 10})
    assert got.type == want.type
    for k in range(3):
        _same_array(got.field(k), want.field(k))
    # small integers, unsigned, half precision and Boolean through SMA (overlap.rs:494-500)
    small = (np.arange(n) * 7919 % 251).astype(np.int64)
    for typ, vals in ((pa.uint32(), small), (pa.uint8(), small), (pa.int8(), small - 125), (pa.int16(), small * 100 - 12000),
                      (pa.uint16(), small * 257), (pa.float16(), (small / 8.0).astype(np.float16)), (pa.bool_(), small % 3 == 0)):
        v64 = np.asarray(vals).astype(np.float64)
        r = pair("sma", [(chunks(vals, typ, mask), "x")], [(chunks(v64, pa.float64(), mask), "x")], {"timeperiod": 5})
        xn = v64.copy(); xn[mask] = oracle.NULL
        (exp,) = oracle.call("sma", xn, timeperiod=5)
        en = exp.view(np.uint64) == np.uint64(oracle.NULL_BITS)
        assert (np.asarray(r.is_null()) == en).all()
        assert (r.to_numpy(zero_copy_only=False)[~en].view(np.uint64) == exp[~en].view(np.uint64)).all()
    # 64-bit integers beyond 2^53 round to nearest even, like Rust's `as f64`
    big = np.array([2 ** 53 + 1, 2 ** 53 + 3, -(2 ** 62) - 1, 2 ** 63 - 1, 5], dtype=np.int64)
    r = _plugin_call(L, "medprice", [([pa.array(big)], "high"), ([pa.array(big)], "low")])
    assert (r.to_numpy() == (big.astype(np.float64) + big.astype(np.float64)) / 2.0).all()
    # the _over twin casts as well (ragged groups: 3 groups of unequal length)
    key = np.repeat(np.array([3, 1, 2], dtype=np.int64), [250, 100, 250])
    got = _plugin_call(L, "rsi_over", [([pa.array(f32["close"])], "close"), ([pa.array(key)], "symbol")], literals=(pa.array([14], type=pa.int64()),))
    want = _plugin_call(L, "rsi_over", [([pa.array(as64["close"])], "close"), ([pa.array(key)], "symbol")], literals=(pa.array([14], type=pa.int64()),))
    _same_array(got, want)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(PLUGIN_FUNCS))
def test_every_exported_function_through_its_plugin_symbol(oracle, name):
    """Each exported pair: default period (no kwargs, no literal), pickled kwargs, trailing literal -- against the oracle on
    two-chunk columns; null-bearing where the reference accepts nulls, null-free for the momentum / cycle family, which must
    REFUSE a null like the reference's rechunk().cont_slice()? (momentum.rs:12-13)."""
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from polars_quant_amd._spec import SPEC
    L = _lib()
    cols, params, _outs, fam = SPEC[name]
    has_tp, default = len(params) == 1, (params[0][2] if params else None)
    d = oracle.gen_ohlcv(0x5EED000C, 1, 400, 0)
    data = {c: d["close" if c == "real" else c][0] for c in cols}
    mask = np.zeros(400, bool); mask[[5, 6, 200]] = True
    fn = getattr(L, "_polars_plugin_" + name)
    nullcol = cols[-1] if cols[-1] != "volume" else cols[-2]    # the nulls sit in one column (the last price column)

    def export(msk):
        keep, ses = [], []
        for c in cols:
            arr = pa.array(data[c], mask=msk if c == nullcol else None)
            se, k = _export([arr.slice(0, 150), arr.slice(150)], c)
            ses.append(se); keep.append(k)
        return ses, keep

    if fam == "N-B":
        ses, keep = export(mask)
        ins = (SeriesExport * len(ses))(*ses)
        ret = SeriesExport(); fn(ins, len(ses), None, 0, C.byref(ret), None)
        assert not ret.release and b"nulls" in L._polars_plugin_get_last_error_message()
        mask[:] = False
    ref_in = {c: data[c].copy() for c in cols}
    ref_in[nullcol][mask] = oracle.NULL

    def check(ret, period):
        assert ret.release, L._polars_plugin_get_last_error_message()
        got = _import(ret)
        kw = {"timeperiod": period} if has_tp else {}
        (exp,) = oracle.call(name, *[ref_in[c] for c in cols], **kw)
        en = exp.view(np.uint64) == np.uint64(oracle.NULL_BITS)
        assert (np.asarray(got.is_null()) == en).all(), name
        g = got.to_numpy(zero_copy_only=False)[~en]
        if name in ("ht_dcperiod", "ht_dcphase"):           # transcendental class
            np.testing.assert_allclose(g, exp[~en], rtol=1e-12, atol=1e-12)
        else:
            assert (g.view(np.uint64) == exp[~en].view(np.uint64)).all(), name

    ses, keep = export(mask)
    ins = (SeriesExport * len(ses))(*ses)
    ret = SeriesExport(); fn(ins, len(ses), None, 0, C.byref(ret), None); check(ret, default)
    if has_tp:
        ses, keep = export(mask)
        ins = (SeriesExport * len(ses))(*ses)
        kw = pickle.dumps({"timeperiod": 9})
        ret = SeriesExport(); fn(ins, len(ses), kw, len(kw), C.byref(ret), None); check(ret, 9)
        ses, keep = export(mask)
        se1, keep1 = _export([pa.array([21], type=pa.int64())], "literal")
        ins = (SeriesExport * (len(ses) + 1))(*ses, se1)
        ret = SeriesExport(); fn(ins, len(ses) + 1, None, 0, C.byref(ret), None); check(ret, 21)
    out_field = ArrowSchema()
    getattr(L, "_polars_plugin_field_" + name)(ses[0].field, 1, C.byref(out_field), None, 0)
    assert out_field.format == b"g" and out_field.name == cols[0].encode()


@pytest.mark.gpu
def test_plugin_randomised_chunks_lengths_and_periods(oracle):
    """200 random calls through the plugin symbols: a random function, a series of 0 .. 3000 rows cut into 1 .. 5 chunks (some of them
    slices with a non-zero offset, some empty), nulls where the reference accepts them, the period as default / pickled kwargs /
    trailing literal -- against the oracle on the concatenated column."""
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from polars_quant_amd._spec import SPEC
    L = _lib()
    rng = np.random.default_rng(0x5EED0077)
    names = sorted(PLUGIN_FUNCS)
    for it in range(200):
        name = names[int(rng.integers(0, len(names)))]
        cols, params, _outs, fam = SPEC[name]
        n = int(rng.choice([0, 1, 2, 31, 32, 33, 64, 65, 500, int(rng.integers(0, 3000))]))
        d = oracle.gen_ohlcv(int(rng.integers(1, 1 << 30)), 1, max(n, 1), 0)
        data = {c: d["close" if c == "real" else c][0][:n].copy() for c in cols}
        mask = (rng.random(n) < 0.02) if (fam != "N-B" and rng.random() < 0.5) else np.zeros(n, bool)
        nullcol = cols[-1] if cols[-1] != "volume" else cols[-2]
        cuts = sorted(set([0, n] + [int(x) for x in rng.integers(0, n + 1, size=int(rng.integers(0, 5)))]))
        keep, ses = [], []
        for c in cols:
            arr = pa.array(data[c], mask=mask if c == nullcol else None)
            chunks = []
            for lo, hi in zip(cuts[:-1], cuts[1:]):
                ch = arr.slice(lo, hi - lo)
                if rng.random() < 0.3:   # a slice of a longer array: a non-zero offset into values and validity
                    pad = int(rng.integers(1, 9))
                    ch = pa.concat_arrays([pa.array(np.full(pad, 1e300)), ch]).slice(pad)
                chunks.append(ch)
            if not chunks:
                chunks = [arr]
            se, k = _export(chunks, c)
            ses.append(se); keep.append(k)
        has_tp = len(params) == 1
        mode = int(rng.integers(0, 3)) if has_tp else 0
        period = params[0][2] if has_tp else None
        kwb, extra = None, []
        if mode == 1:
            period = int(rng.choice([0, 1, 2, 5, 14, 30, 100, n, n + 1]))
            kwb = pickle.dumps({"timeperiod": period})
        elif mode == 2:
            period = int(rng.choice([1, 2, 5, 14, 30, 100, max(n, 1)]))
            se1, k1 = _export([pa.array([period], type=pa.int64())], "literal")
            extra, keep = [se1], keep + [k1]
        ins = (SeriesExport * (len(ses) + len(extra)))(*ses, *extra)
        ret = SeriesExport()
        getattr(L, "_polars_plugin_" + name)(ins, len(ses) + len(extra), kwb, len(kwb) if kwb else 0, C.byref(ret), None)
        assert ret.release, (name, n, cuts, mode, period, L._polars_plugin_get_last_error_message())
        got = _import(ret)
        assert len(got) == n, (name, n, len(got))
        if n == 0:
            continue
        ref_in = {c: data[c].copy() for c in cols}
        ref_in[nullcol][mask] = oracle.NULL
        (exp,) = oracle.call(name, *[ref_in[c] for c in cols], **({"timeperiod": period} if has_tp else {}))
        exp = np.asarray(exp).reshape(-1)
        en = exp.view(np.uint64) == np.uint64(oracle.NULL_BITS)
        assert (np.asarray(got.is_null()) == en).all(), (name, n, cuts, mode, period)
        g = got.to_numpy(zero_copy_only=False)[~en]
        if name in ("ht_dcperiod", "ht_dcphase"):
            np.testing.assert_allclose(g, exp[~en], rtol=1e-12, atol=1e-12)
        else:
            ok = (g.view(np.uint64) == exp[~en].view(np.uint64)) | (np.isnan(g) & np.isnan(exp[~en]))
            assert ok.all(), (name, n, cuts, mode, period, int((~ok).sum()))


@pytest.mark.gpu
def test_every_pattern_through_its_plugin_symbol(oracle):
    """The 61 recognisers: (open, high, low, close) in two chunks -> Int32, penetration default 0.3 (pattern.rs:529-532) and as a
    Float64 literal; a null is refused (cont_slice)."""
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from polars_quant_amd._spec import PATTERN_NAMES
    L = _lib()
    d = oracle.gen_ohlcv(0x5EED0005, 1, 600, 1)             # the pattern-rich variant of the generator
    cols = ("open", "high", "low", "close")
    fired = 0
    for name in PATTERN_NAMES:
        fn = getattr(L, "_polars_plugin_" + name)
        for pen in (None, 0.5):
            ses, keep = [], []
            for c in cols:
                arr = pa.array(d[c][0])
                se, k = _export([arr.slice(0, 250), arr.slice(250)], c)
                ses.append(se); keep.append(k)
            if pen is not None:
                se, k = _export([pa.array([pen], type=pa.float64())], "literal")
                ses.append(se); keep.append(k)
            ins = (SeriesExport * len(ses))(*ses)
            ret = SeriesExport(); fn(ins, len(ses), None, 0, C.byref(ret), None)
            assert ret.release, L._polars_plugin_get_last_error_message()
            assert ret.field.contents.format == b"i" and ret.field.contents.name == b"open"
            arr_out = _import(ret)
            assert arr_out.type == pa.int32() and arr_out.null_count == 0 and len(arr_out) == 600
            got = arr_out.to_numpy()
            exp = oracle.pattern(name, d["open"][0], d["high"][0], d["low"][0], d["close"][0], penetration=0.3 if pen is None else pen)
            assert (got == exp.reshape(-1)).all(), (name, pen)
            fired += int((got != 0).any())
    assert fired >= 70          # (two runs per recogniser; a single 600-row series does not trigger the rarest ones)
    # a null anywhere is an error, as in the reference
    ses, keep = [], []
    for c in cols:
        m = np.zeros(600, bool); m[17] = c == "low"
        se, k = _export([pa.array(d[c][0], mask=m)], c); ses.append(se); keep.append(k)
    ins = (SeriesExport * 4)(*ses)
    ret = SeriesExport(); L._polars_plugin_cdldoji(ins, 4, None, 0, C.byref(ret), None)
    assert not ret.release and b"nulls" in L._polars_plugin_get_last_error_message()
    out_field = ArrowSchema()
    L._polars_plugin_field_cdldoji(ses[0].field, 1, C.byref(out_field), None, 0)
    assert out_field.format == b"i"


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(MULTI_FUNCS))
def test_functions_with_several_parameters_through_their_plugin_symbols(oracle, name):
    """ma / t3 / ultosc / adosc / sar / sarext / ht_trendmode: the reference's defaults, every parameter from the pickled kwargs,
    every parameter as a trailing literal, and a mix (first from kwargs, rest literal) -- against the oracle."""
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    L = _lib()
    cols, params = MULTI_FUNCS[name]
    d = oracle.gen_ohlcv(0x5EED000D, 1, 300, 0)
    data = {c: (2 + np.arange(300) % 29).astype(np.float64) if c == "periods" else d["close" if c == "real" else c][0] for c in cols}
    fn = getattr(L, "_polars_plugin_" + name)
    i32 = name == "ht_trendmode"

    def lit(v):
        return _export([pa.array([v], type=pa.float64() if isinstance(v, float) else pa.int64())], "literal")

    def run(kwargs, literals):
        ses, keep = [], []
        for c in cols:
            arr = pa.array(data[c].astype(np.int64)) if c == "periods" else pa.array(data[c])   # the period column as Int64
            se, k = _export([arr.slice(0, 100), arr.slice(100)], c); ses.append(se); keep.append(k)
        for v in literals:
            se, k = lit(v); ses.append(se); keep.append(k)
        ins = (SeriesExport * len(ses))(*ses)
        kw = pickle.dumps(kwargs) if kwargs else None
        ret = SeriesExport(); fn(ins, len(ses), kw, len(kw) if kw else 0, C.byref(ret), None)
        assert ret.release, L._polars_plugin_get_last_error_message()
        return _import(ret)

    def check(got, values):
        (exp,) = oracle.call(name, *[data[c] for c in cols], **{p[0]: v for p, v in zip(params, values)})
        if i32:
            en = exp == np.int32(-2147483648)
            assert got.type == pa.int32() and (np.asarray(got.is_null()) == en).all()
            assert (got.to_numpy(zero_copy_only=False)[~en].astype(np.int32) == exp[~en]).all()
            return
        en = exp.view(np.uint64) == np.uint64(oracle.NULL_BITS)
        assert (np.asarray(got.is_null()) == en).all(), name
        assert (got.to_numpy(zero_copy_only=False)[~en].view(np.uint64) == exp[~en].view(np.uint64)).all(), name

    defaults, tests = [p[1] for p in params], [p[2] for p in params]
    check(run(None, []), defaults)
    if params:
        check(run({p[0]: v for p, v in zip(params, tests)}, []), tests)
        check(run(None, tests), tests)
        check(run({params[0][0]: tests[0]}, [defaults[0]] + tests[1:]), tests)     # kwargs win over the literal of the same parameter


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(STRUCT_FUNCS))
def test_struct_valued_functions_through_their_plugin_symbols(oracle, name):
    """bbands / mama / aroon / macd / ht_phasor / ht_sine: one Struct array with the reference's struct and field names; defaults,
    kwargs and trailing literals -- every field against the oracle."""
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    L = _lib()
    sname, cols, params, fields = STRUCT_FUNCS[name]
    d = oracle.gen_ohlcv(0x5EED000E, 1, 300, 0)
    data = {c: d["close" if c == "real" else c][0] for c in cols}
    fn = getattr(L, "_polars_plugin_" + name)

    def run(kwargs, literals):
        ses, keep = [], []
        for c in cols:
            arr = pa.array(data[c])
            se, k = _export([arr.slice(0, 100), arr.slice(100)], c); ses.append(se); keep.append(k)
        for v in literals:
            se, k = _export([pa.array([v], type=pa.float64() if isinstance(v, float) else pa.int64())], "literal"); ses.append(se); keep.append(k)
        ins = (SeriesExport * len(ses))(*ses)
        kw = pickle.dumps(kwargs) if kwargs else None
        ret = SeriesExport(); fn(ins, len(ses), kw, len(kw) if kw else 0, C.byref(ret), None)
        assert ret.release, L._polars_plugin_get_last_error_message()
        assert ret.field.contents.name == sname.encode() and ret.field.contents.format == b"+s"
        return _import(ret)

    def check(got, values):
        assert pa.types.is_struct(got.type) and [got.type.field(i).name for i in range(got.type.num_fields)] == fields
        exp = oracle.call(name, *[data[c] for c in cols], **{p[0]: v for p, v in zip(params, values)})
        for i, e in enumerate(exp):
            g = got.field(i)
            en = e.view(np.uint64) == np.uint64(oracle.NULL_BITS)
            assert (np.asarray(g.is_null()) == en).all(), (name, fields[i])
            gv = g.to_numpy(zero_copy_only=False)[~en]
            if name in ("ht_phasor", "ht_sine", "mama"):        # transcendental class
                np.testing.assert_allclose(gv, e[~en], rtol=1e-12, atol=1e-12)
            else:
                assert (gv.view(np.uint64) == e[~en].view(np.uint64)).all(), (name, fields[i])

    defaults, tests = [p[1] for p in params], [p[2] for p in params]
    if name != "mama":                       # (fastlimit = slowlimit = 0.0, the reference's default, is degenerate: tested with values)
        check(run(None, []), defaults)
    if params:
        check(run({p[0]: v for p, v in zip(params, tests)}, []), tests)
        check(run(None, tests), tests)
    out_field = ArrowSchema()
    getattr(L, "_polars_plugin_field_" + name)(None, 0, C.byref(out_field), None, 0)
    assert out_field.format == b"+s" and out_field.name == sname.encode() and out_field.n_children == len(fields)
    kids = C.cast(out_field.children, C.POINTER(C.POINTER(ArrowSchema)))
    assert [kids[i].contents.name.decode() for i in range(len(fields))] == fields and all(kids[i].contents.format == b"g" for i in range(len(fields)))


def _groups(oracle, seed=0x5EED0010):
    lens = np.array([37, 1, 120, 64, 5, 300, 33, 2], dtype=np.int64)
    off = np.r_[0, np.cumsum(lens)]
    d = oracle.gen_ohlcv(seed, 1, int(off[-1]), 0)
    d = {k: np.ascontiguousarray(v[0]) for k, v in d.items()}
    d["real"] = d["close"]
    sym = np.repeat(np.arange(len(lens)), lens)
    return d, off, lens, sym


def _key_arrays(sym):
    """the same grouping as different Arrow layouts of the key column"""
    names = np.array([f"SYM{i:04d}{'x' * (i * 3)}" for i in range(int(sym.max()) + 1)])   # some longer than 12 bytes: out-of-line views
    out = {"int64": pa.array(sym.astype(np.int64)), "int32": pa.array(sym.astype(np.int32)), "uint8": pa.array(sym.astype(np.uint8)),
           "float64": pa.array(sym.astype(np.float64)), "utf8": pa.array(names[sym], type=pa.string()),
           "large_utf8": pa.array(names[sym], type=pa.large_string())}
    if hasattr(pa, "string_view"):
        out["string_view"] = pa.array(names[sym], type=pa.string_view())
    out["dictionary"] = pa.array(names[sym]).dictionary_encode()
    return out


@pytest.mark.gpu
def test_batched_over_symbols_equal_one_call_per_group(oracle):
    """`_polars_plugin_<f>_over(columns..., key, literals...)`: the whole sorted frame in one call == the reference's one call per
    group under .over("symbol") (momentum.py:13-16), for every key layout, a chunked key / data, null keys, kwargs and literals."""
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    L = _lib()
    d, off, lens, sym = _groups(oracle)
    n = int(off[-1])

    def per_group(name, cols, **prm):
        outs = None
        for s in range(len(lens)):
            lo, hi = off[s], off[s + 1]
            res = oracle.call(name, *[d[c][lo:hi] for c in cols], **prm)
            outs = [[] for _ in res] if outs is None else outs
            for k, r in enumerate(res):
                outs[k].append(np.asarray(r).reshape(-1))
        return [np.concatenate(o) for o in outs]

    def run(fn, cols, key, literals=(), kwargs=None, split=None):
        ses, keep = [], []
        for c in cols:
            arr = pa.array(d[c])
            se, k = _export([arr] if split is None else [arr.slice(0, split), arr.slice(split)], c); ses.append(se); keep.append(k)
        se, k = _export([key] if split is None else [key.slice(0, split), key.slice(split)], "symbol"); ses.append(se); keep.append(k)
        for v in literals:
            se, k = _export([pa.array([v], type=pa.float64() if isinstance(v, float) else pa.int64())], "literal"); ses.append(se); keep.append(k)
        ins = (SeriesExport * len(ses))(*ses)
        kw = pickle.dumps(kwargs) if kwargs else None
        ret = SeriesExport(); fn(ins, len(ses), kw, len(kw) if kw else 0, C.byref(ret), None)
        assert ret.release, L._polars_plugin_get_last_error_message()
        return _import(ret)

    def same(got, exp, name):
        en = exp.view(np.uint64) == np.uint64(oracle.NULL_BITS)
        assert len(got) == n and (np.asarray(got.is_null()) == en).all(), name
        assert (got.to_numpy(zero_copy_only=False)[~en].view(np.uint64) == exp[~en].view(np.uint64)).all(), name

    keys = _key_arrays(sym)
    (exp,) = per_group("ema", ["real"], timeperiod=9)
    for kind, key in keys.items():
        same(run(L._polars_plugin_ema_over, ["real"], key, literals=[9]), exp, f"ema_over[{kind}]")
    same(run(L._polars_plugin_ema_over, ["real"], keys["utf8"], kwargs={"timeperiod": 9}, split=101), exp, "ema_over chunked + kwargs")
    (exp,) = per_group("ema", ["real"])
    same(run(L._polars_plugin_ema_over, ["real"], keys["int64"]), exp, "ema_over default period (groups shorter than 30 rows are all-null)")
    # a null key is a group of its own
    m = sym == 3
    same(run(L._polars_plugin_ema_over, ["real"], pa.array(sym.astype(np.int64), mask=m), literals=[9]), per_group("ema", ["real"], timeperiod=9)[0], "null key")
    # several columns, several parameters, Int32 output, a Struct, a pattern
    (exp,) = per_group("ultosc", ["high", "low", "close"], timeperiod1=3, timeperiod2=5, timeperiod3=8)
    same(run(L._polars_plugin_ultosc_over, ["high", "low", "close"], keys["int64"], literals=[3, 5, 8]), exp, "ultosc_over")
    (exp,) = per_group("apo", ["real"], fastperiod=3, slowperiod=7, matype=1)
    same(run(L._polars_plugin_apo_over, ["real"], keys["int64"], literals=[3, 7, 1]), exp, "apo_over")
    st = run(L._polars_plugin_macd_over, ["real"], keys["dictionary"], literals=[3, 7, 4])
    for fld, exp in zip(("macd", "macd_signal", "macd_hist"), per_group("macd", ["real"], fastperiod=3, slowperiod=7, signalperiod=4)):
        same(st.field(fld), exp, "macd_over." + fld)
    tm = run(L._polars_plugin_ht_trendmode_over, ["real"], keys["int64"])
    (exp,) = per_group("ht_trendmode", ["real"])
    en = exp == np.int32(-2147483648)
    assert (np.asarray(tm.is_null()) == en).all() and (tm.to_numpy(zero_copy_only=False)[~en].astype(np.int32) == exp[~en]).all()
    rich = oracle.gen_ohlcv(0x5EED0011, 1, n, 1)
    for c in ("open", "high", "low", "close"):
        d[c] = np.ascontiguousarray(rich[c][0])
    got = run(L._polars_plugin_cdlengulfing_over, ["open", "high", "low", "close"], keys["int64"]).to_numpy(zero_copy_only=False)
    exp = np.concatenate([oracle.pattern("cdlengulfing", *[d[c][off[s]:off[s + 1]] for c in ("open", "high", "low", "close")]).reshape(-1)
                          for s in range(len(lens))])
    assert (got == exp).all() and (exp != 0).any()
    # errors: no key column, a key of the wrong length
    ses, keep = [], []
    se, k = _export([pa.array(d["real"])], "real"); ses.append(se); keep.append(k)
    ins = (SeriesExport * 1)(*ses)
    ret = SeriesExport(); L._polars_plugin_ema_over(ins, 1, None, 0, C.byref(ret), None)
    assert not ret.release and b"too few" in L._polars_plugin_get_last_error_message()
    se2, k2 = _export([pa.array(sym[:-3].astype(np.int64))], "symbol")
    ins = (SeriesExport * 2)(ses[0], se2)
    ret = SeriesExport(); L._polars_plugin_ema_over(ins, 2, None, 0, C.byref(ret), None)
    assert not ret.release


@pytest.mark.gpu
@pytest.mark.parametrize("n_groups,glen", [(5, 64), (3, 1100), (1, 300), (7, 33)])
def test_batched_over_balanced_panel(oracle, n_groups, glen):
    """groups of ONE common length are passed on as a regular batch (tiled bodies; the wave-per-symbol forms from 1 024 rows on):
    the same values as one call per group"""
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    L = _lib()
    n = n_groups * glen
    d = oracle.gen_ohlcv(0x5EED0012, 1, n, 0)
    d = {k: np.ascontiguousarray(v[0]) for k, v in d.items()}
    d["real"] = d["close"]
    sym = np.repeat(np.arange(n_groups), glen).astype(np.int64)

    def run(fn, cols, literals=()):
        ses, keep = [], []
        for c in cols:
            se, k = _export([pa.array(d[c])], c); ses.append(se); keep.append(k)
        se, k = _export([pa.array(sym)], "symbol"); ses.append(se); keep.append(k)
        for v in literals:
            se, k = _export([pa.array([v], type=pa.int64())], "literal"); ses.append(se); keep.append(k)
        ins = (SeriesExport * len(ses))(*ses)
        ret = SeriesExport(); fn(ins, len(ses), None, 0, C.byref(ret), None)
        assert ret.release, L._polars_plugin_get_last_error_message()
        return _import(ret)

    for name, cols, lits, prm in (("ema", ["real"], [9], dict(timeperiod=9)), ("sma", ["real"], [5], dict(timeperiod=5)),
                                  ("atr", ["high", "low", "close"], [7], dict(timeperiod=7)), ("rsi", ["real"], [14], dict(timeperiod=14))):
        got = run(getattr(L, f"_polars_plugin_{name}_over"), cols, lits)
        exp = np.concatenate([np.asarray(oracle.call(name, *[d[c][g * glen:(g + 1) * glen] for c in cols], **prm)[0]).reshape(-1) for g in range(n_groups)])
        en = exp.view(np.uint64) == np.uint64(oracle.NULL_BITS)
        assert len(got) == n and (np.asarray(got.is_null()) == en).all(), name
        assert (got.to_numpy(zero_copy_only=False)[~en].view(np.uint64) == exp[~en].view(np.uint64)).all(), name
    # nulls inside a pitched panel (the validity bitmap is applied on the host copy, rows between the groups are padding), Int32 output
    mask = np.zeros(n, bool); mask[::37] = True
    ses, keep = [], []
    se, k = _export([pa.array(d["real"], mask=mask)], "real"); ses.append(se); keep.append(k)
    se, k = _export([pa.array(sym)], "symbol"); ses.append(se); keep.append(k)
    se, k = _export([pa.array([4], type=pa.int64())], "literal"); ses.append(se); keep.append(k)
    ins = (SeriesExport * 3)(*ses)
    ret = SeriesExport(); L._polars_plugin_sma_over(ins, 3, None, 0, C.byref(ret), None)
    got = _import(ret)
    xn = d["real"].copy(); xn[mask] = oracle.NULL
    exp = np.concatenate([np.asarray(oracle.call("sma", xn[g * glen:(g + 1) * glen], timeperiod=4)[0]).reshape(-1) for g in range(n_groups)])
    en = exp.view(np.uint64) == np.uint64(oracle.NULL_BITS)
    assert (np.asarray(got.is_null()) == en).all() and (got.to_numpy(zero_copy_only=False)[~en].view(np.uint64) == exp[~en].view(np.uint64)).all()
    ses, keep = [], []
    for c in ("open", "high", "low", "close"):
        se, k = _export([pa.array(d[c])], c); ses.append(se); keep.append(k)
    se, k = _export([pa.array(sym)], "symbol"); ses.append(se); keep.append(k)
    ins = (SeriesExport * 5)(*ses)
    ret = SeriesExport(); L._polars_plugin_cdlengulfing_over(ins, 5, None, 0, C.byref(ret), None)
    got = _import(ret)
    exp = np.concatenate([oracle.pattern("cdlengulfing", *[d[c][g * glen:(g + 1) * glen] for c in ("open", "high", "low", "close")]).reshape(-1) for g in range(n_groups)])
    assert got.type == pa.int32() and (got.to_numpy() == exp).all()


@pytest.mark.gpu
def test_input_column_cache_never_aliases_a_reused_address(oracle):
    """The plugin keeps the device copy of a one-chunk Float64 input under (buffer address, rows, layout) + a hash of the WHOLE buffer
    (csrc/plugin.hip): the sixty expressions of one `with_columns` upload `close` once.  A freed-and-reused address must not alias: the
    same numpy buffer is handed over with other content -- one element changed in the middle of it -- and the result must follow."""
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    L = _lib()
    L.pq_plugin_cache_stats.argtypes = [C.POINTER(C.c_int64)] * 4
    L.pq_plugin_cache_stats.restype = None
    L.pq_plugin_cache_clear.restype = None

    def stats():
        v = [C.c_int64() for _ in range(4)]
        L.pq_plugin_cache_stats(*[C.byref(x) for x in v])
        return tuple(x.value for x in v)     # hits, misses, bytes, entries
    L.pq_plugin_cache_clear()
    n = 40_000
    buf = np.empty(n)                                             # ONE host buffer for the whole test: the address never changes
    x0 = oracle.gen_ohlcv(0x5EED0C01, 1, n, 0)["close"][0]
    buf[:] = x0
    arr = pa.Array.from_buffers(pa.float64(), n, [None, pa.py_buffer(buf)])     # zero-copy view of buf
    assert arr.buffers()[1].address == buf.ctypes.data

    def call(name, **kw):
        return _plugin_call(L, name, [([arr], "close")], kwargs=kw or None)

    def check(got, name, x, **kw):
        (exp,) = oracle.call(name, x.copy(), **kw)
        en = exp.view(np.uint64) == np.uint64(oracle.NULL_BITS)
        assert (np.asarray(got.is_null()) == en).all()
        assert (got.to_numpy(zero_copy_only=False)[~en].view(np.uint64) == exp[~en].view(np.uint64)).all()
    check(call("ema", timeperiod=20), "ema", x0, timeperiod=20)
    h, m, by, ent = stats()
    assert (h, m, ent) == (0, 1, 1) and by == n * 8
    check(call("sma", timeperiod=10), "sma", x0, timeperiod=10)    # another function, the same column: a hit
    check(call("ema", timeperiod=5), "ema", x0, timeperiod=5)
    assert stats()[:2] == (2, 1)
    # the same address, other content (one value in the middle: a sample of the ends would not see it)
    buf[n // 2 + 17] *= 1.25
    x1 = buf.copy()
    check(call("ema", timeperiod=20), "ema", x1, timeperiod=20)
    h, m, by, ent = stats()
    assert (h, m) == (2, 2) and ent == 2
    check(call("sma", timeperiod=10), "sma", x1, timeperiod=10)
    assert stats()[:2] == (3, 2)
    # back to the first content: its copy is still there
    buf[:] = x0
    check(call("sma", timeperiod=10), "sma", x0, timeperiod=10)
    assert stats()[:2] == (4, 2)
    # a multi-input function, a Struct-valued one and a recogniser share the cache; an `_over` call has its own layout (own entry)
    d = oracle.gen_ohlcv(0x5EED0C02, 1, n, 0)
    cols = {k: pa.array(d[k][0]) for k in ("open", "high", "low", "close")}
    a1 = _plugin_call(L, "atr", [([cols[k]], k) for k in ("high", "low", "close")], kwargs={"timeperiod": 14})
    a2 = _plugin_call(L, "atr", [([cols[k]], k) for k in ("high", "low", "close")], kwargs={"timeperiod": 14})
    _same_array(a1, a2)
    p1 = _plugin_call(L, "cdlengulfing", [([cols[k]], k) for k in ("open", "high", "low", "close")])
    p2 = _plugin_call(L, "cdlengulfing", [([cols[k]], k) for k in ("open", "high", "low", "close")])
    _same_array(p1, p2)
    hits_before = stats()[0]
    _plugin_call(L, "natr", [([cols[k]], k) for k in ("high", "low", "close")], kwargs={"timeperiod": 14})
    assert stats()[0] == hits_before + 3
    # chunked or null-bearing columns are not cached (and still right): the counters do not move
    before = stats()
    whole = pa.array(x0, mask=np.arange(n) == 5)
    _plugin_call(L, "ema", [([whole.slice(0, 100), whole.slice(100)], "close")], kwargs={"timeperiod": 20})
    assert stats() == before
    L.pq_plugin_cache_clear()
    assert stats() == (0, 0, 0, 0)


@pytest.mark.gpu
def test_result_host_pool_never_hands_out_a_block_that_is_still_alive(oracle):
    """The host side of a result column comes from a bounded pool of RELEASED blocks (csrc/plugin.hip HostBuf; blocks of >= 1 MB): a result
    the caller still holds must keep its bytes while later calls run; after its release a later call of about that size may take the
    block over -- and must fill every element of it; PQ_PLUGIN_HOSTPOOL_MB=0 switches the pool off."""
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import os
    L = _lib()
    n = 300_000                                     # 2.4 MB per Float64 column: pooled
    d = oracle.gen_ohlcv(0x5EED0D01, 2, n, 0)
    xa, xb = pa.array(d["close"][0]), pa.array(d["close"][1])

    def bits(arr):
        return arr.to_numpy(zero_copy_only=False).view(np.uint64).copy()

    def expect(x, p):
        (e,) = oracle.call("ema", np.asarray(x), timeperiod=p)
        return e.view(np.uint64)
    ra = _plugin_call(L, "ema", [([xa], "close")], kwargs={"timeperiod": 9})
    keep = bits(ra)
    addr_a = ra.buffers()[1].address
    rb = _plugin_call(L, "ema", [([xb], "close")], kwargs={"timeperiod": 21})      # ra is alive: another block
    assert rb.buffers()[1].address != addr_a
    en = expect(xa, 9) == np.uint64(oracle.NULL_BITS)
    assert (bits(ra) == keep).all() and (keep[~en] == expect(xa, 9)[~en]).all()
    eb = expect(xb, 21)
    assert (bits(rb)[eb != np.uint64(oracle.NULL_BITS)] == eb[eb != np.uint64(oracle.NULL_BITS)]).all()
    del ra                                                                         # released: its block goes to the pool
    rc = _plugin_call(L, "sma", [([xb], "close")], kwargs={"timeperiod": 5})       # same size: may take it over
    (ec,) = oracle.call("sma", np.asarray(xb), timeperiod=5)
    ecb = ec.view(np.uint64)
    got = bits(rc)
    nn = ecb != np.uint64(oracle.NULL_BITS)
    assert (got[nn] == ecb[nn]).all() and (np.asarray(rc.is_null()) == ~nn).all()
    assert (bits(rb)[eb != np.uint64(oracle.NULL_BITS)] == eb[eb != np.uint64(oracle.NULL_BITS)]).all()    # rb untouched by the reuse
    # a three-column Struct result and an Int32 result draw from the same pool
    m1 = _plugin_call(L, "macd", [([xa], "close")])
    del rc, m1
    m2 = _plugin_call(L, "macd", [([xa], "close")])
    em = oracle.call("macd", np.asarray(xa))
    for k, child in enumerate(m2.flatten()):
        e = em[k].view(np.uint64)
        ok = e != np.uint64(oracle.NULL_BITS)
        assert (bits(child)[ok] == e[ok]).all()
    os.environ["PQ_PLUGIN_HOSTPOOL_MB"] = "0"
    try:
        r0 = _plugin_call(L, "ema", [([xa], "close")], kwargs={"timeperiod": 9})
        assert (bits(r0) == keep).all()
    finally:
        del os.environ["PQ_PLUGIN_HOSTPOOL_MB"]
