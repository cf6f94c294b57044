"""Market-data files -> pinned host columns -> pitched device columns, with the copies overlapping the compute.

The reference's file side is `prepare_sequential_data` (python/polars_quant/backtest/sequential.py:7-93): every CSV / Parquet file of a
folder, the file stem as the symbol when the file has no symbol column, all symbols aligned on the union of dates, nulls filled
forward (or backward / zero) per symbol and then with a default.  `read_market_folder(..., align=True)` reproduces that contract with
pyarrow (no Polars in the loop) and returns SYMBOL-MAJOR columns -- `[N, T]` f64, the layout every `pq_*` call takes; `align=False`
keeps every symbol's own history: long columns sorted by (symbol, date) + group offsets, i.e. a ragged batch (`pq_batch.offsets`).
Arrow IPC (Feather v2) files are read as well.

`DeviceFrame.upload` pins the host columns (`pq_host_register`: the DMA engine reads them in place, no staging copy) and moves them with
`pq_memcpy_h2d_pitched` on a COPY stream in the order the compute needs them; each column gets an event, so the consumers of `close`
run while `high` is still on the bus (`Suite.record_staged` / `run_staged`).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from pathlib import Path

import numpy as np

NULL_BITS = 0x7FF80000504E554C
_NULL = np.array([NULL_BITS], dtype=np.uint64).view(np.float64)[0]


@dataclass
class HostFrame:
    symbols: list            # group order = row-block order
    dates: np.ndarray        # align=True: the [T] date axis; else the long date column
    columns: dict            # name -> f64 array: [N, T] (aligned) or [rows] (ragged)
    offsets: np.ndarray | None = None   # ragged: n + 1 row indices

    @property
    def n_series(self) -> int:
        return len(self.symbols)


def _read_one(path: Path):
    import pyarrow as pa
    suf = path.suffix.lower()
    if suf in (".parquet", ".pqt"):
        import pyarrow.parquet as pq_
        return pq_.read_table(path)
    if suf in (".arrow", ".ipc", ".feather"):
        try:
            with pa.memory_map(str(path), "r") as src:
                return pa.ipc.open_file(src).read_all()
        except pa.ArrowInvalid:
            with pa.memory_map(str(path), "r") as src:
                return pa.ipc.open_stream(src).read_all()
    if suf == ".csv":
        import pyarrow.csv as pc
        return pc.read_csv(path)
    return None


def read_market_folder(folder, date_col: str = "date", symbol_col: str = "symbol", value_cols=None, align: bool = True,
                       fill_null_strategy: str = "forward", default_fill_value: float = 0.0) -> HostFrame:
    """Every CSV / Parquet / Arrow IPC file of `folder` -> one HostFrame (contract: sequential.py:7-93, see the module docstring).
    value_cols: the numeric columns to keep (default: every column but date / symbol, in first-seen order)."""
    import pyarrow as pa
    import pyarrow.compute as pc
    folder = Path(folder)
    if not folder.exists() or not folder.is_dir():
        raise FileNotFoundError(f"The directory '{folder}' does not exist or is not a directory.")
    tables = []
    for fp in sorted(folder.iterdir()):
        t = _read_one(fp)
        if t is None:
            continue
        if symbol_col not in t.column_names:
            t = t.append_column(symbol_col, pa.array([fp.stem] * t.num_rows, type=pa.string()))
        tables.append(t)
    if not tables:
        raise ValueError(f"No valid CSV, Parquet or Arrow IPC files found in '{folder}'.")
    names = []
    for t in tables:
        for c in t.column_names:
            if c not in (date_col, symbol_col) and c not in names:
                names.append(c)
    if value_cols is not None:
        names = [c for c in value_cols]
    parts = []
    for t in tables:  # diagonal concat: a column a file lacks is null there
        dcol = t[date_col]
        if pa.types.is_timestamp(dcol.type) or pa.types.is_date(dcol.type):   # files disagree on the unit (CSV: s, Parquet: ms / ns)
            dcol = pc.cast(dcol, pa.timestamp("ns"))
        cols = {date_col: dcol, symbol_col: pc.cast(t[symbol_col], pa.string())}
        for c in names:
            cols[c] = pc.cast(t[c], pa.float64()) if c in t.column_names else pa.nulls(t.num_rows, pa.float64())
        parts.append(pa.table(cols))
    tab = pa.concat_tables(parts, promote_options="default").combine_chunks()
    sym = np.asarray(tab[symbol_col].to_pylist(), dtype=object)
    date = tab[date_col].to_numpy(zero_copy_only=False)
    symbols, sym_id = np.unique(sym.astype(str), return_inverse=True)
    vals = {}
    for c in names:
        a = tab[c]
        v = a.to_numpy(zero_copy_only=False).astype(np.float64, copy=True)
        if a.null_count:
            v[np.asarray(a.is_null())] = np.nan
        vals[c] = v
    if not align:
        order = np.lexsort((date, sym_id))
        counts = np.bincount(sym_id, minlength=len(symbols))
        off = np.r_[0, np.cumsum(counts)].astype(np.int64)
        cols = {}
        for c in names:
            v = np.ascontiguousarray(vals[c][order])
            v[np.isnan(v)] = _NULL            # a missing value is a NULL row of the column
            cols[c] = v
        return HostFrame(list(symbols), date[order], cols, off)
    dates, date_id = np.unique(date, return_inverse=True)
    N, T = len(symbols), len(dates)
    cols = {}
    for c in names:
        grid = np.full((N, T), np.nan)
        grid[sym_id, date_id] = vals[c]
        if fill_null_strategy in ("forward", "backward"):
            g = grid if fill_null_strategy == "forward" else grid[:, ::-1]
            idx = np.where(~np.isnan(g), np.arange(T)[None, :], -1)
            np.maximum.accumulate(idx, axis=1, out=idx)
            filled = np.where(idx >= 0, np.take_along_axis(g, np.maximum(idx, 0), axis=1), np.nan)
            grid = filled if fill_null_strategy == "forward" else filled[:, ::-1]
        elif fill_null_strategy != "zero":
            raise ValueError("fill_null_strategy must be 'forward', 'backward' or 'zero'")
        grid = np.where(np.isnan(grid), 0.0 if fill_null_strategy == "zero" else default_fill_value, grid)
        cols[c] = np.ascontiguousarray(grid)
    return HostFrame(list(symbols), dates, cols, None)


@dataclass
class DeviceFrame:
    """Pitched device columns of an aligned HostFrame (+ the events that say when each one has arrived)."""
    columns: dict = field(default_factory=dict)   # name -> [N, T] view of a [N, stride] buffer
    events: dict = field(default_factory=dict)    # name -> torch.cuda.Event recorded on the copy stream behind its H2D
    stride: int = 0
    _host: list = field(default_factory=list)

    @staticmethod
    def allocate(n: int, T: int, names, device="cuda", stride: int | None = None) -> "DeviceFrame":
        import torch
        stride = (T + 15) // 16 * 16 if stride is None else stride   # rows pitched to 128 B (hipMallocPitch-like)
        df = DeviceFrame(stride=stride)
        for c in names:
            df.columns[c] = torch.zeros((n, stride), dtype=torch.float64, device=device)[:, :T]
        return df

    def register(self, host: HostFrame, names=None) -> None:
        """pin the host columns once (an Arrow buffer handed over by the caller is pinned in place)"""
        from ._lib import check, lib
        for c in (names or list(self.columns)):
            a = host.columns[c]
            if not (a.flags["C_CONTIGUOUS"] and a.dtype == np.float64):
                raise ValueError(f"column {c}: the host side must be contiguous f64")
            check(lib().pq_host_register(a.ctypes.data_as(C.c_void_p), a.nbytes))
            self._host.append(a)

    def unregister(self) -> None:
        from ._lib import check, lib
        for a in self._host:
            check(lib().pq_host_unregister(a.ctypes.data_as(C.c_void_p)))
        self._host = []

    def upload(self, host: HostFrame, order=None, copy_stream=None) -> None:
        """H2D of every column on `copy_stream` (default: a side stream of its own), one event per column: a consumer does
        `torch.cuda.current_stream().wait_event(frame.events[name])` (Suite.run_staged does) and starts while later columns copy."""
        import torch

        from ._lib import check, lib
        from .api import ctx
        names = list(order or self.columns)
        any_col = self.columns[names[0]]
        dev = any_col.device
        n, T = any_col.shape
        if copy_stream is None:
            copy_stream = getattr(self, "_copy_stream", None) or torch.cuda.Stream(device=dev)
            self._copy_stream = copy_stream
        copy_stream.wait_stream(torch.cuda.current_stream(dev))   # the previous step's readers are done with the buffers
        with torch.cuda.device(dev), torch.cuda.stream(copy_stream):
            h = ctx(dev.index)
            for c in names:
                a = host.columns[c]
                if a.shape != (n, T):
                    raise ValueError(f"column {c}: host shape {a.shape}, device shape {(n, T)}")
                col = self.columns[c]
                if tuple(col.shape) != (n, T) or (T > 1 and col.stride(1) != 1) or (n > 1 and col.stride(0) != self.stride):
                    # (a pitched copy into a buffer of another pitch writes past its rows)
                    raise ValueError(f"column {c}: device column has a row pitch of {col.stride(0)} elements, the frame's is {self.stride}")
                check(lib().pq_memcpy_h2d_pitched(h, C.c_void_p(self.columns[c].data_ptr()), self.stride * 8, a.ctypes.data_as(C.c_void_p), T * 8,
                                                  T * 8, n))
                ev = self.events.get(c) or torch.cuda.Event()
                ev.record(copy_stream)
                self.events[c] = ev
