"""The benchmark workload as a reusable object: the full indicator suite (every function of SURVEY 8(a)
with the reference's Python-wrapper default parameters) + the fused MACD-cross per-symbol backtest over
one symbol-major [N, T] OHLCV block resident in HBM.  Outputs are allocated once.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from ._lib import Batch, BtParams, check, lib
from ._spec import BT_DEFAULTS, I, PATTERN_NAMES, PATTERN_PEN_DEFAULT, SPEC
from .api import ctx

COLMAP = {"real": "close"}


def algorithmic_bytes_per_row() -> dict:
    """8*(#f64 in + #f64 out) + 4*(#i32 out) per (symbol, day) row, per call."""
    per = {}
    for name, (cols, _p, outs, _f) in SPEC.items():
        per[name] = 8 * len(cols) + sum(8 if dt == "f8" else 4 for _, dt in outs)
    per["cdl_all"] = 8 * 4 + 4 * len(PATTERN_NAMES)          # fused: OHLC read once, 61 int32 written
    per["backtest_macd_cross"] = 8 + 24                       # close in; position, cash, equity out
    return per


class Suite:
    def __init__(self, n_series: int, T: int, device="cuda", stride: int | None = None, exact_layout: bool = False):
        """The suite OWNS its output columns and allocates them at a row pitch that is a multiple of 128 B (pq_recommended_stride:
        every 64 / 128-byte tile piece is then one aligned cache line): `stride` if it is such a multiple, else the smallest one
        >= T.  Inputs handed to record() at another pitch or at a base that is not 16-byte aligned -- a dense odd-T tensor would run
        the 8-byte forms of every kernel, 1.5 x slower -- are re-housed ONCE at that pitch (record() / refresh_inputs()).
        exact_layout=True keeps the caller's layout exactly (stride = `stride` or T, inputs used in place): what the tests of the
        8-byte forms and bench.py --exact-layout measure."""
        self.n, self.T = n_series, T
        self.exact_layout = bool(exact_layout)
        rec = (T + 15) // 16 * 16
        if self.exact_layout:
            self.stride = T if stride is None else int(stride)
        else:
            self.stride = int(stride) if (stride and int(stride) % 16 == 0 and int(stride) >= T) else rec
        assert self.stride >= T
        self._housed = {}
        self.dev = torch.device(device)
        if self.dev.type == "cuda" and self.dev.index is None:     # ("cuda" != "cuda:0" for torch: every input would read as foreign and be re-housed)
            self.dev = torch.device("cuda", torch.cuda.current_device())
        self.batch = Batch(n_series, T, self.stride)
        f64 = lambda: torch.empty((n_series, self.stride), dtype=torch.float64, device=self.dev)[:, :T]
        i32 = lambda: torch.empty((n_series, self.stride), dtype=torch.int32, device=self.dev)[:, :T]
        self.out = {name: [f64() if dt == "f8" else i32() for _, dt in outs] for name, (_c, _p, outs, _f) in SPEC.items()}
        self.pat = {nm: i32() for nm in PATTERN_NAMES}
        self.bt = [f64(), f64(), f64()]
        self.summary = torch.empty((n_series, 8), dtype=torch.float64, device=self.dev)
        self.periods = f64()
        self.periods.copy_(torch.from_numpy(np.tile((2 + np.arange(T) % 29).astype(np.float64), (n_series, 1))))
        self._pens = (C.c_double * 61)(*[PATTERN_PEN_DEFAULT[nm] for nm in PATTERN_NAMES])
        self._pat_ptrs = (C.c_void_p * 61)(*[self.pat[nm].data_ptr() for nm in PATTERN_NAMES])
        self._prm = BtParams(**BT_DEFAULTS)
        self._defaults = {name: [C.c_int64(int(d)) if k == I else C.c_double(float(d)) for _, k, d in p]
                          for name, (_c, p, _o, _f) in SPEC.items()}
        self.bytes_per_row = algorithmic_bytes_per_row()

    def out_bytes(self) -> int:
        return sum(t.numel() * t.element_size() for ts in self.out.values() for t in ts) + \
            sum(t.numel() * 4 for t in self.pat.values()) + sum(t.numel() * 8 for t in self.bt)

    def _col(self, ohlcv, c):
        return self.periods if c == "periods" else ohlcv[COLMAP.get(c, c)]

    def _on_layout(self, ohlcv: dict) -> None:
        """run_one() reads the columns with the suite's batch descriptor: refuse a column at another pitch instead of reading it wrongly
        (record() / run_eager() re-house such columns themselves; direct callers of run_one() pass them through house() first)"""
        for k, t in ohlcv.items():
            if self.n > 1 and t.stride(0) != self.stride:
                raise ValueError(f"input column `{k}` has a row pitch of {t.stride(0)} elements, the suite's is {self.stride}: "
                                 f"pass the columns through Suite.house() (or build the suite with exact_layout=True)")

    def run_one(self, name: str, ohlcv: dict) -> None:
        L, h, b = lib(), ctx(self.dev.index), self.batch
        self._on_layout(ohlcv)
        if name == "cdl_all":
            check(L.pq_cdl_all(h, C.byref(b), *[C.c_void_p(ohlcv[k].data_ptr()) for k in ("open", "high", "low", "close")],
                               self._pens, self._pat_ptrs))
        elif name == "backtest_macd_cross":
            check(L.pq_backtest_macd_cross(h, C.byref(b), C.c_void_p(ohlcv["close"].data_ptr()), 12, 26, 9, C.byref(self._prm),
                                           *[C.c_void_p(t.data_ptr()) for t in self.bt], C.c_void_p(self.summary.data_ptr())))
        elif name == "dmi_all":   # calc_dm evaluated once for its five users (all timeperiod=14 by default)
            o = self.out
            check(L.pq_dmi_all(h, C.byref(b), *[C.c_void_p(ohlcv[k].data_ptr()) for k in ("high", "low", "close")], 14,
                               *[C.c_void_p(o[n][0].data_ptr()) for n in ("dx", "plus_di", "minus_di", "adx", "adxr")]))
        elif name in ("ema_all", "atr_all", "dm_pair", "ad_all", "macd_pair", "apo_ppo", "stoch_all", "sar_pair", "volume_all", "dm_system_all", "cmo_rsi",
                      "sma_ma"):
            # multi-output forms: the listed functions share their inputs and (default) parameters -> one job
            o, P = self.out, lambda k: C.c_void_p(ohlcv[k].data_ptr())
            O = lambda n, i=0: C.c_void_p(o[n][i].data_ptr())
            if name == "ema_all":
                check(L.pq_ema_all(h, C.byref(b), P("close"), 30, O("ema"), O("dema"), O("tema"), O("trix")))
            elif name == "atr_all":
                check(L.pq_atr_all(h, C.byref(b), P("high"), P("low"), P("close"), 14, O("atr"), O("natr")))
            elif name == "dm_pair":
                check(L.pq_dm_pair(h, C.byref(b), P("high"), P("low"), 14, O("plus_dm"), O("minus_dm")))
            elif name == "ad_all":
                check(L.pq_ad_all(h, C.byref(b), P("high"), P("low"), P("close"), P("volume"), 3, 10, O("ad"), O("adosc")))
            elif name == "dm_system_all":
                check(L.pq_dm_system_all(h, C.byref(b), P("high"), P("low"), P("close"), 14, O("dx"), O("plus_di"), O("minus_di"), O("adx"),
                                         O("adxr"), O("atr"), O("natr")))
            elif name == "cmo_rsi":
                check(L.pq_cmo_rsi(h, C.byref(b), P("close"), 14, O("cmo"), O("rsi")))
            elif name == "sma_ma":    # SMA(30) and MA(30, matype 0): the same walk (overlap.rs:857-869), one job, the value written twice
                check(L.pq_sma_ma(h, C.byref(b), P("close"), 30, O("sma"), O("ma")))
            elif name == "volume_all":
                check(L.pq_volume_all(h, C.byref(b), P("high"), P("low"), P("close"), P("volume"), 14, 3, 10, O("mfi"), O("ad"), O("adosc"),
                                      O("obv")))
            elif name == "sar_pair":   # Python-wrapper defaults: every parameter 0.0 (overlap.py:115-157)
                check(L.pq_sar_pair(h, C.byref(b), P("high"), P("low"), *([C.c_double(0.0)] * 10), O("sar"), O("sarext")))
            elif name == "stoch_all":
                check(L.pq_stoch_all(h, C.byref(b), P("high"), P("low"), P("close"), 5, 3, 0, 3, 0, 3, 0, O("stoch", 0), O("stoch", 1),
                                     O("stochf", 0), O("stochf", 1)))
            elif name == "macd_pair":
                check(L.pq_macd_pair(h, C.byref(b), P("close"), 12, 26, 9, 9, O("macd", 0), O("macd", 1), O("macd", 2),
                                     O("macdfix", 0), O("macdfix", 1), O("macdfix", 2)))
            else:
                check(L.pq_apo_ppo(h, C.byref(b), P("close"), 12, 26, 0, O("apo"), O("ppo")))
        elif name == "aroon_all":  # AROON + AROONOSC (both timeperiod=14 by default) from one window scan
            o = self.out
            check(L.pq_aroon_all(h, C.byref(b), C.c_void_p(ohlcv["high"].data_ptr()), C.c_void_p(ohlcv["low"].data_ptr()), 14,
                                 C.c_void_p(o["aroon"][0].data_ptr()), C.c_void_p(o["aroon"][1].data_ptr()),
                                 C.c_void_p(o["aroonosc"][0].data_ptr())))
        elif name == "ht_all":    # the Hilbert pipeline evaluated once for dcperiod / dcphase / phasor / sine
            o = self.out
            outs = [o["ht_dcperiod"][0], o["ht_dcphase"][0], o["ht_phasor"][0], o["ht_phasor"][1], o["ht_sine"][0], o["ht_sine"][1]]
            check(L.pq_ht_all(h, C.byref(b), C.c_void_p(ohlcv["close"].data_ptr()), *[C.c_void_p(t.data_ptr()) for t in outs]))
        else:
            outs = self.out[name]
            cols = SPEC[name][0]
            check(getattr(L, "pq_" + name)(h, C.byref(b), *[C.c_void_p(self._col(ohlcv, c).data_ptr()) for c in cols],
                                           *self._defaults[name], *[C.c_void_p(t.data_ptr()) for t in outs]))

    FUSED = {"dm_system_all": ("dx", "plus_di", "minus_di", "adx", "adxr", "atr", "natr"), "cmo_rsi": ("cmo", "rsi"),
             "ht_all": ("ht_dcperiod", "ht_dcphase", "ht_phasor", "ht_sine"),
             "aroon_all": ("aroon", "aroonosc"), "ema_all": ("ema", "dema", "tema", "trix"), 
             "dm_pair": ("plus_dm", "minus_dm"), "apo_ppo": ("apo", "ppo"),
             "sar_pair": ("sar", "sarext"), "volume_all": ("mfi", "ad", "adosc", "obv"),
             "stoch_all": ("stoch", "stochf"), "macd_pair": ("macd", "macdfix"), "sma_ma": ("sma", "ma")}
    # pq_macd_pair (six output tiles, 27.6 KB) and pq_stoch_all (192 VGPRs since the moving-average cores were slimmed: it now runs in
    # the light job kernel) joined the list in round 3: time-neutral within the noise of a session (4.34 against 4.35 ms per step), two
    # jobs and four column reads fewer.  PQ_SUITE_UNFUSE=name,... keeps the listed ones as separate calls for A/B runs.

    def tasks(self, fused: bool = False):
        """every function of the suite; fused=True replaces the users of a shared core by the multi-output call"""
        names = list(SPEC)
        if fused:
            import os
            skip = set(filter(None, os.environ.get("PQ_SUITE_UNFUSE", "").split(",")))   # A/B runs: keep these as separate calls
            fused_map = {k: v for k, v in self.FUSED.items() if k not in skip}
            covered = {n for v in fused_map.values() for n in v}
            return [n for n in names if n not in covered] + list(fused_map) + ["cdl_all", "backtest_macd_cross"]
        return names + ["cdl_all", "backtest_macd_cross"]

    def run_eager(self, ohlcv: dict) -> None:
        """one step as ~90 separate launches (one per C-ABI call), enqueued on the current stream"""
        ohlcv = self._house(ohlcv)
        with torch.cuda.device(self.dev):
            for name in self.tasks():
                self.run_one(name, ohlcv)

    def _house(self, ohlcv: dict) -> dict:
        """the input columns on the suite's layout: a column already at the suite's pitch (and 16-byte aligned) is used in place,
        any other is copied into a suite-owned pitched buffer -- once; refresh_inputs() repeats the copy after the caller changed it"""
        if self.exact_layout:
            return ohlcv
        out = {}
        for k, t in ohlcv.items():
            ok = (t.dim() == 2 and tuple(t.shape) == (self.n, self.T) and t.dtype == torch.float64 and t.device == self.dev and
                  (self.T <= 1 or t.stride(1) == 1) and (self.n <= 1 or t.stride(0) == self.stride) and t.data_ptr() % 16 == 0)
            if ok:
                out[k] = t
                continue
            buf = self._housed.get(k)
            if buf is None:
                buf = self._housed[k] = torch.zeros((self.n, self.stride), dtype=torch.float64, device=self.dev)
            buf[:, :self.T].copy_(t)
            out[k] = buf[:, :self.T]
        self._stale_warned = False
        return out

    def _needs_housing(self, ohlcv: dict) -> list:
        """the columns of `ohlcv` that record() would copy instead of using in place"""
        if self.exact_layout:
            return []
        return [k for k, t in ohlcv.items()
                if not (t.dim() == 2 and tuple(t.shape) == (self.n, self.T) and t.dtype == torch.float64 and t.device == self.dev and
                        (self.T <= 1 or t.stride(1) == 1) and (self.n <= 1 or t.stride(0) == self.stride) and t.data_ptr() % 16 == 0)]

    house = _house

    def refresh_inputs(self, ohlcv: dict) -> None:
        """after the caller changed input columns that record() had to re-house: copy them again (same buffers, no re-recording).
        run(ohlcv) does this itself; run() without tensors replays on the copies as they are."""
        self._house({k: t for k, t in ohlcv.items() if k in self._housed})

    def record(self, ohlcv: dict, tasks=None, summaries=None) -> None:
        """record the step once (pq_suite_begin/end): the sequential jobs of all functions become one grid per phase.
        summaries = [t0, t1] ([n, 8] f64 each): the step is recorded TWICE, the backtest of recording k writing its summary rows into
        summaries[k] (every other output column is shared) -- run(slot=k) then replays recording k, so that a multi-GPU caller can
        keep the exchange of one table in flight while the next step fills the other (distributed.OverlappedGather)."""
        self.close()
        L, h = lib(), ctx(self.dev.index)
        ohlcv = self._house(ohlcv)
        self._ohlcv = ohlcv  # keep the inputs alive: the suite holds raw device pointers
        handles, keep, ok = [], self.summary, False
        try:
            for summ in (summaries or [self.summary]):
                assert summ.shape == (self.n, 8) and summ.dtype == torch.float64 and summ.is_contiguous()
                self.summary = summ
                with torch.cuda.device(self.dev):
                    check(L.pq_suite_begin(h, C.byref(self.batch)))
                    try:
                        for name in (tasks or self.tasks(fused=True)):
                            self.run_one(name, ohlcv)
                    except Exception:
                        L.pq_suite_abort(h)
                        raise
                    out = C.c_void_p()
                    check(L.pq_suite_end(h, C.byref(out)))
                handles.append(out)
            ok = True
        finally:
            if ok:
                self.summary = summaries[0] if summaries else keep
            else:   # a later recording failed: the earlier handles (job tables, events) must not leak, the object keeps its old table
                self.summary = keep
                for hnd in handles:
                    L.pq_suite_destroy(h, hnd)
        self._summaries = list(summaries) if summaries else [self.summary]
        self._suites = handles
        self._suite = handles[0]

    # ---- staged form: the step split by the input columns a task needs, so that a stage starts as soon as ITS columns have
    # arrived from the host (loader.DeviceFrame.upload records one event per column on the copy stream)
    STAGE_ORDER = ("close", "high", "low", "open", "volume")   # copy order = the order in which the stages become runnable

    def task_inputs(self, name: str) -> set:
        if name == "cdl_all":
            return {"open", "high", "low", "close"}
        if name == "backtest_macd_cross":
            return {"close"}
        members = self.FUSED.get(name) or {"dmi_all": ("dx", "plus_di", "minus_di", "adx", "adxr"), "atr_all": ("atr", "natr"),
                                            "ad_all": ("ad", "adosc"), "macd_pair": ("macd", "macdfix"), "stoch_all": ("stoch", "stochf")}.get(name) or (name,)
        cols = set()
        for m in members:
            cols |= {COLMAP.get(c, c) for c in SPEC[m][0] if c != "periods"}
        return cols

    def record_staged(self, ohlcv: dict) -> list:
        """one recorded suite per prefix of STAGE_ORDER that completes some task's inputs -> [(columns needed, n tasks)]"""
        # The staged step waits for the upload events of the buffers it READS: a column that had to be re-housed would be read from
        # the suite's copy while the events speak of the caller's tensor.  Upload into the suite's own layout instead
        # (DeviceFrame(columns=suite.house(cols), stride=suite.stride), as bench.py --e2e does).
        slow = self._needs_housing(ohlcv)
        if slow:
            raise ValueError(f"record_staged: columns {slow} are not on the suite's row pitch ({self.stride} elements): pass suite.house(columns)")
        self.close()
        L, h = lib(), ctx(self.dev.index)
        self._ohlcv = ohlcv
        stages, have, left = [], set(), list(self.tasks(fused=True))
        for col in self.STAGE_ORDER:
            have.add(col)
            ready = [t for t in left if self.task_inputs(t) <= have]
            left = [t for t in left if t not in ready]
            if ready:
                stages.append((col, ready))
        assert not left, left
        self._stages = []
        with torch.cuda.device(self.dev):
            for col, names in stages:
                check(L.pq_suite_begin(h, C.byref(self.batch)))
                try:
                    for name in names:
                        self.run_one(name, ohlcv)
                except Exception:
                    L.pq_suite_abort(h)
                    raise
                out = C.c_void_p()
                check(L.pq_suite_end(h, C.byref(out)))
                self._stages.append((col, out, len(names)))
        return [(col, n) for col, _, n in self._stages]

    def run_staged(self, events: dict | None = None) -> None:
        """the staged step on the current stream; events[column] (from DeviceFrame.upload): wait for that column's copy first"""
        cur = torch.cuda.current_stream(self.dev)
        with torch.cuda.device(self.dev):
            for col, handle, _n in self._stages:
                if events is not None:
                    for c in self.STAGE_ORDER[: self.STAGE_ORDER.index(col) + 1]:
                        if c in events:
                            cur.wait_event(events[c])
                check(lib().pq_suite_run(ctx(self.dev.index), handle))

    def info(self):
        a, b, c = C.c_int32(), C.c_int32(), C.c_int32()
        check(lib().pq_suite_info(self._suite, C.byref(a), C.byref(b), C.byref(c)))
        return {"phases": a.value, "seq_jobs": b.value, "row_launches": c.value}

    def set_timing(self, on: bool = True) -> None:
        """HIP events around every sequential-job grid (on its launch stream) for the following runs"""
        check(lib().pq_suite_set_timing(self._suite, 1 if on else 0))

    def grid_stats(self):
        """-> list of dicts per SEQ grid: avg_ms over the timed runs, algorithmic bytes per launch, jobs, LDS bytes"""
        out, k = [], 0
        while True:
            ms, by, nj, lds, runs = C.c_double(), C.c_double(), C.c_int32(), C.c_int32(), C.c_int32()
            st = lib().pq_suite_grid_stats(self._suite, k, C.byref(ms), C.byref(by), C.byref(nj), C.byref(lds), C.byref(runs))
            if st != 0:
                break
            var = C.c_int32()
            check(lib().pq_suite_grid_variant(self._suite, k, C.byref(var)))
            out.append({"avg_ms": ms.value, "alg_bytes": by.value, "n_jobs": nj.value, "lds_bytes": lds.value, "runs": runs.value,
                        "kernel": f"seq_jobs_kernel<{var.value}>" if var.value < 3 else ("seq_jobs_kernel<3>" if var.value == 4 else "row_chain")})
            k += 1
        return out

    def span_stats(self, variant: int = 0):
        """-> (mean ms from the earliest start to the latest end of the grids launching seq_jobs_kernel<variant>, their
        algorithmic bytes per step)"""
        ms, by = C.c_double(), C.c_double()
        check(lib().pq_suite_span_stats(self._suite, variant, C.byref(ms), C.byref(by)))
        return ms.value, by.value

    def run(self, ohlcv: dict | None = None, slot: int = 0) -> None:
        """one step: every indicator + all 61 patterns + the MACD-cross backtest, enqueued on the current stream
        (slot: which of the recordings of record(summaries=[...]) to replay)"""
        if getattr(self, "_suite", None) is None:
            self.record(ohlcv)
        elif self._housed:
            # the recording reads suite-owned copies of the columns that were handed over at a slow row pitch: tensors passed here are
            # copied into them again (current stream); without tensors the step replays on the copies as they are -- said once
            if ohlcv is not None:
                self.refresh_inputs(ohlcv)
            elif not getattr(self, "_stale_warned", True):
                import warnings
                from .api import PqLayoutWarning
                self._stale_warned = True
                warnings.warn(f"Suite.run(): columns {sorted(self._housed)} were re-housed at record time; this replay reads those copies, not the "
                              f"caller's tensors -- pass the tensors to run() or call refresh_inputs() after changing them", PqLayoutWarning, stacklevel=2)
        if not Suite._stream_warned and torch.cuda.current_stream(self.dev).cuda_stream != 0:
            # (a replay needs four hardware queues on four compute pipes: the NULL stream + the three side streams of the context are a
            #  process's first four; a created stream shifts them and two chains share a pipe -- DESIGN.md section 6)
            import warnings
            from .api import PqLayoutWarning
            Suite._stream_warned = True
            warnings.warn("Suite.run() on a created stream: on this runtime a recorded suite replays 17-38 % slower there than on the NULL "
                          "(default) stream -- two of its four chains share a compute pipe of the command processor (INTEGRATION.md); "
                          "warned once per process", PqLayoutWarning, stacklevel=2)
        with torch.cuda.device(self.dev):
            check(lib().pq_suite_run(ctx(self.dev.index), self._suites[slot]))

    _stream_warned = False

    def close(self):
        for hnd in getattr(self, "_suites", []) or []:
            check(lib().pq_suite_destroy(ctx(self.dev.index), hnd))
        self._suites = []
        self._suite = None
        for _col, handle, _n in getattr(self, "_stages", []):
            check(lib().pq_suite_destroy(ctx(self.dev.index), handle))
        self._stages = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def suite_bytes_per_row(self) -> int:
        """per-call algorithmic bytes of the whole suite (SURVEY 8d accounting, independent of how calls are fused)"""
        return sum(self.bytes_per_row[t] for t in self.tasks())
