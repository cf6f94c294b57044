"""-m gpu: the drop-in boundary exercised the way INTEGRATION.md's reference-side stub uses it -- plain C-ABI calls on host
Arrow buffers -- plus the host mirror of the reference's classes (VectorizedBacktester.run()) and thread safety."""
import ctypes as C
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
pa = pytest.importorskip("pyarrow")

NULLB = np.uint64(0x7FF80000504E554C)


def bits(a):
    return np.ascontiguousarray(a).view(np.uint64)


@pytest.fixture(scope="module")
def L():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from polars_quant_amd._lib import lib
    return lib()


def _ck(L, st):
    assert st == 0, L.pq_last_error().decode()


def test_pitched_copies_place_dense_host_columns_at_a_128_byte_pitch(L, oracle):
    """pq_memcpy_h2d_pitched / pq_memcpy_d2h_pitched: a dense [n, len] host column into a device column with a 128-byte row pitch
    (the layout bench.py measures), one C-ABI call on it with pq_batch.stride = the pitch, and back; the padding is untouched."""
    from polars_quant_amd._lib import Batch
    n, T, pitch = 9, 203, 208
    rng = np.random.default_rng(4)
    host = np.ascontiguousarray(rng.normal(size=(n, T)).cumsum(axis=1) + 50.0)
    ctx = C.c_void_p()
    _ck(L, L.pq_ctx_create(0, None, C.byref(ctx)))
    d_in, d_out = C.c_void_p(), C.c_void_p()
    _ck(L, L.pq_malloc(ctx, n * pitch * 8, C.byref(d_in)))
    _ck(L, L.pq_malloc(ctx, n * pitch * 8, C.byref(d_out)))
    poison = np.full((n, pitch), -7.0)
    _ck(L, L.pq_memcpy_h2d(ctx, d_in, poison.ctypes.data_as(C.c_void_p), poison.nbytes))
    _ck(L, L.pq_memcpy_h2d(ctx, d_out, poison.ctypes.data_as(C.c_void_p), poison.nbytes))
    _ck(L, L.pq_memcpy_h2d_pitched(ctx, d_in, pitch * 8, host.ctypes.data_as(C.c_void_p), T * 8, T * 8, n))
    b = Batch(n, T, pitch)
    _ck(L, L.pq_ema(ctx, C.byref(b), d_in, 10, d_out))
    got = np.empty((n, T))
    _ck(L, L.pq_memcpy_d2h_pitched(ctx, got.ctypes.data_as(C.c_void_p), T * 8, d_out, pitch * 8, T * 8, n))
    (exp,) = oracle.call("ema", host, timeperiod=10)
    assert (bits(got) == bits(exp)).all()
    raw = np.empty((n, pitch))
    _ck(L, L.pq_memcpy_d2h(ctx, raw.ctypes.data_as(C.c_void_p), d_out, raw.nbytes))
    assert (raw[:, T:] == -7.0).all()                       # rows between the series are never written
    assert L.pq_memcpy_h2d_pitched(ctx, d_in, 8, host.ctypes.data_as(C.c_void_p), T * 8, T * 8, n) != 0   # width > pitch
    _ck(L, L.pq_free(ctx, d_in)); _ck(L, L.pq_free(ctx, d_out)); _ck(L, L.pq_ctx_destroy(ctx))


def test_arrow_column_through_the_c_abi_only(L, oracle):
    """INTEGRATION.md section 2, step by step, with nothing but C-ABI calls: a host Arrow f64 column with a validity bitmap and
    a NON-ZERO bit offset (a slice) -> pq_host_register -> pq_malloc -> pq_memcpy_h2d -> pq_nulls_from_arrow -> pq_ema ->
    pq_validity_to_arrow -> pq_memcpy_d2h, compared with the oracle on the same nulls."""
    from polars_quant_amd._lib import Batch
    n_series, T = 70, 304
    d = oracle.gen_ohlcv(0x5EED0007, n_series, T, 0)
    flat = d["close"].reshape(-1)
    rng = np.random.default_rng(1)
    mask = rng.random(flat.shape[0] + 13) < 0.02
    big = pa.array(np.concatenate([np.zeros(13), flat]), mask=mask)      # 13 leading rows that the slice drops
    arr = big.slice(13)                                                  # offset 13: bit offset 13 & 7 = 5 into byte 1
    assert arr.offset == 13 and arr.null_count > 0
    validity, values = arr.buffers()
    n = len(arr)
    host_vals = values.address + arr.offset * 8
    ctx = C.c_void_p()
    _ck(L, L.pq_ctx_create(0, None, C.byref(ctx)))
    d_in, d_out, d_bits, d_obits, d_cnt = (C.c_void_p() for _ in range(5))
    nbytes_bits = (arr.offset + n + 7) // 8
    for ptr, sz in ((d_in, n * 8), (d_out, n * 8), (d_bits, nbytes_bits), (d_obits, (n + 7) // 8), (d_cnt, 8)):
        _ck(L, L.pq_malloc(ctx, sz, C.byref(ptr)))
    _ck(L, L.pq_host_register(C.c_void_p(values.address), values.size))                # zero-copy source for the DMA
    _ck(L, L.pq_memcpy_h2d(ctx, d_in, C.c_void_p(host_vals), n * 8))
    _ck(L, L.pq_memcpy_h2d(ctx, d_bits, C.c_void_p(validity.address), nbytes_bits))
    _ck(L, L.pq_nulls_from_arrow(ctx, d_in, d_bits, arr.offset, n))
    b = Batch(n_series, T, T)
    _ck(L, L.pq_ema(ctx, C.byref(b), d_in, C.c_int64(9), d_out))
    _ck(L, L.pq_validity_to_arrow(ctx, d_out, n, d_obits, d_cnt))
    out = np.empty(n)
    obits = np.zeros((n + 7) // 8, np.uint8)
    cnt = np.zeros(1, np.int64)
    _ck(L, L.pq_memcpy_d2h(ctx, out.ctypes.data_as(C.c_void_p), d_out, n * 8))
    _ck(L, L.pq_memcpy_d2h(ctx, obits.ctypes.data_as(C.c_void_p), d_obits, obits.size))
    _ck(L, L.pq_memcpy_d2h(ctx, cnt.ctypes.data_as(C.c_void_p), d_cnt, 8))
    _ck(L, L.pq_host_unregister(C.c_void_p(values.address)))
    for ptr in (d_in, d_out, d_bits, d_obits, d_cnt):
        _ck(L, L.pq_free(ctx, ptr))
    _ck(L, L.pq_ctx_destroy(ctx))
    x = flat.copy()
    x[mask[13:]] = oracle.NULL
    (exp,) = oracle.call("ema", x.reshape(n_series, T), timeperiod=9)
    exp = exp.reshape(-1)
    en = bits(exp) == NULLB
    assert ((bits(out) == bits(exp)) | (np.isnan(out) & np.isnan(exp) & ~en)).all()
    got_valid = np.unpackbits(obits, bitorder="little")[:n].astype(bool)
    assert (got_valid == ~en).all() and cnt[0] == en.sum()
    # the same column through the Python host (api -> the same C ABI): a 1-D Arrow array is ONE series of n rows
    import polars_quant_amd as pq
    got = pq.EMA(arr, 9)
    (exp1,) = oracle.call("ema", x, timeperiod=9)
    en1 = bits(exp1) == NULLB
    assert isinstance(got, pa.Array) and got.null_count == int(en1.sum())
    assert (np.asarray(got.is_null()) == en1).all()
    vals = got.to_numpy(zero_copy_only=False)
    assert (bits(vals[~en1]) == bits(exp1[~en1])).all()


def test_vectorized_backtester_run_shapes(oracle):
    """VectorizedBacktester(price, buy, sell, benchmark).run() -> (positions, capital, summary): vectorized.rs:204-223 and
    polars_quant.pyi:20-49 -- positions {"position"}, capital {"cash", "equity"}, a dict of 8 floats per capital pool."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import polars_quant_amd as pq
    d = oracle.gen_ohlcv(0x5EED0008, 6, 300, 0)
    close = d["close"]
    buy, sell = oracle.macd_cross_signals(close)
    bench = d["open"][0]
    # [T]: one asset, exactly the reference's call
    pos, cap, summ = pq.VectorizedBacktester(close[0], buy[0], sell[0], benchmark=bench, position_size=0.5).run()
    epos, ecash, eeq, es = oracle.backtest(close[0], buy[0], sell[0], benchmark=bench, position_size=0.5)
    assert set(pos) == {"position"} and set(cap) == {"cash", "equity"}
    assert np.asarray(pos["position"]).shape == (300,)
    assert (bits(np.asarray(pos["position"])) == bits(epos)).all()
    assert (bits(np.asarray(cap["cash"])) == bits(ecash)).all() and (bits(np.asarray(cap["equity"])) == bits(eeq)).all()
    assert list(summ) == pq.SUMMARY_KEYS and len(summ) == 8 and all(isinstance(v, float) for v in summ.values())
    np.testing.assert_allclose([summ[k] for k in pq.SUMMARY_KEYS], es, rtol=1e-12, atol=1e-13)
    assert summ["total_trades"] == es[7] and summ["total_trades"] > 0
    # [N, T]: N independent pools; a [T] benchmark is shared by all of them
    pos, cap, summ = pq.VectorizedBacktester(close, buy, sell, benchmark=bench).run()
    epos, ecash, eeq, es = oracle.backtest(close, buy, sell, benchmark=np.tile(bench, (6, 1)))
    assert np.asarray(pos["position"]).shape == (6, 300) and isinstance(summ, list) and len(summ) == 6
    assert (bits(np.asarray(cap["equity"])) == bits(eeq)).all()
    for row, e in zip(summ, es):
        np.testing.assert_allclose([row[k] for k in pq.SUMMARY_KEYS], e, rtol=1e-12, atol=1e-13)
    assert any(abs(row["beta"]) > 0 for row in summ), "the shared benchmark must reach every pool"
    with pytest.raises(pq.PqError, match="buy"):
        pq.VectorizedBacktester(close, buy[0], sell).run()          # [T] signals with [N, T] prices: refused, not read out of bounds
    with pytest.raises(pq.PqError, match="benchmark"):
        pq.VectorizedBacktester(close, buy, sell, benchmark=d["open"][:3]).run()


def test_two_threads_two_contexts(L, oracle):
    """Two host threads, each with its own context and stream, call concurrently; results are right and pq_last_error() is
    thread-local (an error provoked on one thread is not seen by the other)."""
    from polars_quant_amd._lib import Batch
    n_series, T = 130, 312
    d = oracle.gen_ohlcv(0x5EED0009, n_series, T, 0)
    res, errs = {}, {}
    barrier = threading.Barrier(2)

    def worker(tid, name, period):
        try:
            torch.cuda.set_device(0)
            stream = torch.cuda.Stream()
            ctx = C.c_void_p()
            _ck(L, L.pq_ctx_create(0, C.c_void_p(stream.cuda_stream), C.byref(ctx)))
            x = torch.from_numpy(d["close"]).cuda()
            out = torch.empty_like(x)
            b = Batch(n_series, T, T)
            fn = getattr(L, "pq_" + name)
            barrier.wait()
            for _ in range(20):
                _ck(L, fn(ctx, C.byref(b), C.c_void_p(x.data_ptr()), C.c_int64(period), C.c_void_p(out.data_ptr())))
            if tid == 0:    # provoke an argument error on this thread only
                st = fn(ctx, C.byref(b), None, C.c_int64(period), C.c_void_p(out.data_ptr()))
                assert st != 0
            barrier.wait()
            errs[tid] = L.pq_last_error().decode()
            _ck(L, L.pq_ctx_sync(ctx))
            res[tid] = out.cpu().numpy()
            _ck(L, L.pq_ctx_destroy(ctx))
        except Exception as e:  # noqa: BLE001
            res[tid] = e
            try:
                barrier.abort()
            except Exception:  # noqa: BLE001
                pass

    ths = [threading.Thread(target=worker, args=(0, "ema", 12)), threading.Thread(target=worker, args=(1, "sma", 25))]
    for t in ths: t.start()
    for t in ths: t.join()
    for tid, (name, p) in enumerate((("ema", 12), ("sma", 25))):
        assert not isinstance(res[tid], Exception), res[tid]
        (exp,) = oracle.call(name, d["close"], timeperiod=p)
        assert ((bits(res[tid]) == bits(exp))).all(), name
    assert "null pointer" in errs[0] and "null pointer" not in errs[1]


def test_returns_reference_vector_and_parity(oracle):
    """returns(): the one vector the reference itself holds (README.md:66-75: [100, 102, 101, 105] -> [None, 0.02, -0.0098,
    0.0396], printed to 4 d.p.), then parity with the oracle (simple bit-exact; log within 1e-12: device log())."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import polars_quant_amd as pq
    out = pq.returns({"date": ["2024-01-01", "2024-01-02", "2024-01-03", "2024-01-04"], "close": np.array([100.0, 102.0, 101.0, 105.0])},
                     price_col="close", period=1, method="simple")
    r = np.asarray(out["return"])
    assert bits(r)[0] == NULLB and list(np.round(r[1:], 4)) == [0.02, -0.0098, 0.0396]
    d = oracle.gen_ohlcv(0x5EED000A, 70, 304, 0)
    x = d["close"].copy()
    x[3, 10] = oracle.NULL; x[5, :4] = oracle.NULL; x[7, 100] = 0.0
    for period in (1, 5, 20, 0, 400):
        for m, name in ((0, "simple"), (1, "log")):
            (exp,) = oracle.call("returns", x, period=period, method=m)
            got = np.asarray(pq.returns(x, period=period, method=name))
            en = bits(exp) == NULLB
            assert ((bits(got) == NULLB) == en).all(), (period, name)
            if m == 0:
                assert ((bits(got) == bits(exp)) | (np.isnan(got) & np.isnan(exp))).all(), (period, name)
            else:
                ok = ~en & np.isfinite(exp)
                np.testing.assert_allclose(got[ok], exp[ok], rtol=1e-12, atol=1e-15)
    with pytest.raises(ValueError):
        pq.returns(x, method="geometric")


def test_c_abi_communicator_and_summary_gather_single_rank(L):
    """pq_comm_unique_id / pq_comm_init / pq_gather_summaries / pq_comm_destroy through RCCL with a world of one (the only world a
    one-GPU box has): the call sequence a non-Python host performs; ranks > 1 differ only in the peers RCCL connects."""
    import polars_quant_amd as pq
    from polars_quant_amd.distributed import CabiComm
    s = torch.arange(13 * 8, dtype=torch.float64, device="cuda").reshape(13, 8) * 0.5
    comm = CabiComm("cuda:0", 0, 1)
    out = comm.gather_summaries(s, 13)
    torch.cuda.synchronize()
    assert torch.equal(out, s) and out.data_ptr() != s.data_ptr()
    with pytest.raises(pq.PqError):          # a second communicator on the same context is an error, not a leak
        CabiComm("cuda:0", 0, 1)
    comm.close()
    comm = CabiComm("cuda:0", 0, 1)          # and it can be created again after the destroy
    assert torch.equal(comm.gather_summaries(s[:5], 5), s[:5])
    comm.close()


def test_overlapped_gather_equals_the_serial_one_over_consecutive_steps(L, oracle):
    """pq_gather_summaries_begin / _end (the exchange of step k on the communicator's own stream beside the kernels of step k + 1,
    two slots): five consecutive steps with CHANGING inputs; every gathered table equals the table the serial pq_gather_summaries
    returns for that step, and the oracle's summary rows."""
    import polars_quant_amd as pq
    from polars_quant_amd._lib import Batch, BtParams, check
    from polars_quant_amd._spec import BT_DEFAULTS
    from polars_quant_amd.api import ctx
    from polars_quant_amd.distributed import CabiComm, OverlappedGather
    n, T = 96, 640
    data = [oracle.gen_ohlcv(0x5EED0F00 + k, n, T, 0)["close"] for k in range(5)]
    dev = torch.device("cuda:0")
    closes = [torch.from_numpy(c).to(dev) for c in data]
    comm = CabiComm(dev, 0, 1)
    h, b, prm = ctx(0), Batch(n, T, T), BtParams(**BT_DEFAULTS)
    og = OverlappedGather(n, n, dev, comm=comm)
    curves = [torch.empty((n, T), dtype=torch.float64, device=dev) for _ in range(3)]

    def run(k, summ):
        check(L.pq_backtest_macd_cross(h, C.byref(b), C.c_void_p(closes[k].data_ptr()), 12, 26, 9, C.byref(prm),
                                       *[C.c_void_p(t.data_ptr()) for t in curves], C.c_void_p(summ.data_ptr())))
    got = []
    held = {}
    for k in range(5):
        slot = og.acquire()                       # waits for the exchange that used this slot two steps ago
        if slot in held:
            got.append((held[slot], og.all[slot].clone()))
        run(k, og.local[slot])
        og.begin(slot)
        held[slot] = k
        with pytest.raises(pq.PqError):           # a slot with an exchange in flight cannot be begun again
            comm.gather_begin(og.local[slot], n, og.all[slot], slot)
    og.drain()
    for slot, k in held.items():
        got.append((k, og.all[slot].clone()))
    torch.cuda.synchronize()
    assert sorted(k for k, _ in got) == [0, 1, 2, 3, 4]
    for k, table in got:
        serial_local = torch.empty((n, 8), dtype=torch.float64, device=dev)
        run(k, serial_local)
        serial = comm.gather_summaries(serial_local, n)
        torch.cuda.synchronize()
        assert torch.equal(table.view(torch.int64), serial.view(torch.int64)), k
        buy, sell = oracle.macd_cross_signals(data[k])
        _, _, _, exp = oracle.backtest(data[k], buy, sell)
        np.testing.assert_allclose(table.cpu().numpy(), exp.reshape(n, 8), rtol=1e-12, atol=1e-13)
    comm.close()


def test_suite_recorded_twice_fills_either_summary_buffer(L, oracle):
    """Suite.record(summaries=[a, b]): run(slot=k) writes the backtest's summary rows into buffer k and nothing else differs"""
    from polars_quant_amd.suite import Suite
    n, T = 70, 256
    d = oracle.gen_ohlcv(0x5EED0F10, n, T, 0)
    g = {k: torch.from_numpy(v).cuda() for k, v in d.items()}
    st = Suite(n, T, "cuda")
    a, b = torch.zeros((n, 8), dtype=torch.float64, device="cuda"), torch.zeros((n, 8), dtype=torch.float64, device="cuda")
    st.record(g, summaries=[a, b])
    st.run(g, slot=1)
    torch.cuda.synchronize()
    assert float(a.abs().sum()) == 0.0 and float(b.abs().sum()) > 0.0
    st.run(g, slot=0)
    torch.cuda.synchronize()
    assert torch.equal(a.view(torch.int64), b.view(torch.int64))
    buy, sell = oracle.macd_cross_signals(d["close"])
    _, _, _, exp = oracle.backtest(d["close"], buy, sell)
    np.testing.assert_allclose(a.cpu().numpy(), exp.reshape(n, 8), rtol=1e-12, atol=1e-13)
    st.close()


def test_layout_advice_and_no_silent_slow_layout(L, oracle):
    """pq_recommended_stride / pq_layout_check (include/pq_hip.h), and the Python layer's side of it: a Suite re-houses inputs handed
    over at an odd pitch once (then runs the aligned job kernel, not the 8-byte form), host inputs are uploaded pitched, and a device
    tensor at a slow pitch draws a PqLayoutWarning instead of silence."""
    import warnings
    from polars_quant_amd import api
    from polars_quant_amd._lib import Batch
    from polars_quant_amd.suite import Suite
    assert [L.pq_recommended_stride(n) for n in (0, 1, 16, 17, 2520, 2521, 2528)] == [0, 16, 16, 32, 2528, 2528, 2528]
    buf = torch.zeros(4 * 2528 + 2, dtype=torch.float64, device="cuda")
    vp = C.c_void_p
    cols = lambda *ptrs: (vp * len(ptrs))(*ptrs)
    assert L.pq_layout_check(C.byref(Batch(4, 2520, 2528)), cols(buf.data_ptr()), 1) == 0
    assert L.pq_layout_check(C.byref(Batch(4, 2520, 2520)), cols(buf.data_ptr()), 1) == 0             # dense even pitch: the 16-byte forms
    assert L.pq_layout_check(C.byref(Batch(4, 2521, 2521)), cols(buf.data_ptr()), 1) == 100           # PQ_WARN_SLOW_LAYOUT
    assert b"2528" in L.pq_last_error()
    assert L.pq_layout_check(C.byref(Batch(4, 2520, 2528)), cols(buf.data_ptr(), buf.data_ptr() + 8), 2) == 100
    assert L.pq_layout_check(C.byref(Batch(4, 2520, 2528)), cols(buf.data_ptr() + 4), 1) == 100
    assert L.pq_layout_check(None, None, 0) == 1
    # Suite: inputs on a dense ODD pitch -> re-housed once -> the aligned job kernel; results = the oracle's
    n, T = 70, 301
    d = oracle.gen_ohlcv(0x5EED0F20, n, T, 0)
    g = {k: torch.from_numpy(v).cuda() for k, v in d.items()}
    st = Suite(n, T, "cuda")
    assert st.stride == 304
    st.record(g, ["sma", "ema_all", "dm_system_all", "cdl_all", "backtest_macd_cross"])
    assert sorted(st._housed) == ["close", "high", "low", "open", "volume"]
    kernels = {gs["kernel"] for gs in st.grid_stats()}
    assert "seq_jobs_kernel<3>" not in kernels and "seq_jobs_kernel<2>" not in kernels and "seq_jobs_kernel<0>" in kernels, kernels
    st.run(); torch.cuda.synchronize()
    (exp,) = oracle.call("sma", d["close"], timeperiod=30)
    assert (bits(st.out["sma"][0].cpu().numpy()) == bits(exp)).all()
    (exp,) = oracle.call("adx", d["high"], d["low"], d["close"], timeperiod=14)
    assert (bits(st.out["adx"][0].cpu().numpy()) == bits(exp)).all()
    g["close"].mul_(2.0)                                        # the caller changes a re-housed column ...
    st.refresh_inputs(g)                                        # ... and says so
    st.run(); torch.cuda.synchronize()
    (exp,) = oracle.call("sma", d["close"] * 2.0, timeperiod=30)
    assert (bits(st.out["sma"][0].cpu().numpy()) == bits(exp)).all()
    pitched = st.house(g)
    assert all(t.stride(0) == 304 for t in pitched.values())
    with pytest.raises(ValueError):
        st.run_one("sma", g)                                    # a column at another pitch is refused, not misread
    st.close()
    # exact_layout=True keeps the caller's odd pitch (the 8-byte form), as bench.py --exact-layout does
    st = Suite(n, T, "cuda", exact_layout=True)
    st.record({k: torch.from_numpy(v).cuda() for k, v in d.items()}, ["sma", "ema_all"])
    assert "seq_jobs_kernel<3>" in {gs["kernel"] for gs in st.grid_stats()}
    st.close()
    # api.call: host data is uploaded pitched (no warning); a dense odd-pitch DEVICE tensor warns once; both give the oracle's values
    api._warned_layout = False
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        (got,) = api.call("ema", d["close"], timeperiod=10)
    (exp,) = oracle.call("ema", d["close"], timeperiod=10)
    assert got.flags["C_CONTIGUOUS"] and (bits(got) == bits(exp)).all()
    with pytest.warns(api.PqLayoutWarning):
        (got,) = api.call("ema", torch.from_numpy(d["close"]).cuda(), timeperiod=10)
    assert (bits(got.cpu().numpy()) == bits(exp)).all()
    with warnings.catch_warnings():
        warnings.simplefilter("error")                          # once per process
        api.call("ema", torch.from_numpy(d["close"]).cuda(), timeperiod=10)


def test_baseline_config_1_one_symbol_252_days_sma_and_ema_20(L, oracle):
    """BASELINE config 1, literally: 1 symbol x 252 days of f64 close, SMA(20) + EMA(20) -- through the C ABI on a host Arrow buffer (what
    the Rust stub of INTEGRATION.md does) and through the Polars plugin symbols (what `pq.SMA(pl.col("close"), 20)` resolves to), both
    bit for bit the oracle's."""
    import pickle
    from polars_quant_amd._lib import Batch
    close = np.ascontiguousarray(oracle.gen_ohlcv(0x5EED0C01, 1, 252, 0)["close"][0])
    arr = pa.array(close)
    ctx = C.c_void_p()
    _ck(L, L.pq_ctx_create(0, None, C.byref(ctx)))
    d_in, d_out = C.c_void_p(), C.c_void_p()
    _ck(L, L.pq_malloc(ctx, 252 * 8, C.byref(d_in))); _ck(L, L.pq_malloc(ctx, 252 * 8, C.byref(d_out)))
    _ck(L, L.pq_memcpy_h2d(ctx, d_in, C.c_void_p(arr.buffers()[1].address), 252 * 8))
    b = Batch(1, 252, 252)
    for name in ("sma", "ema"):
        _ck(L, getattr(L, "pq_" + name)(ctx, C.byref(b), d_in, 20, d_out))
        got = np.empty(252)
        _ck(L, L.pq_memcpy_d2h(ctx, got.ctypes.data_as(C.c_void_p), d_out, 252 * 8))
        (exp,) = oracle.call(name, close, timeperiod=20)
        assert (bits(got) == bits(exp)).all(), name
        assert (bits(got[:19]) == NULLB).all() and np.isfinite(got[19:]).all()
    _ck(L, L.pq_free(ctx, d_in)); _ck(L, L.pq_free(ctx, d_out)); _ck(L, L.pq_ctx_destroy(ctx))
    import test_polars_plugin as tp
    PL = tp._lib()
    for name in ("sma", "ema"):
        got = tp._plugin_call(PL, name, [([arr], "close")], {"timeperiod": 20})
        (exp,) = oracle.call(name, close, timeperiod=20)
        en = bits(exp) == NULLB
        assert (np.asarray(got.is_null()) == en).all()
        assert (bits(got.to_numpy(zero_copy_only=False)[~en]) == bits(exp[~en])).all()
