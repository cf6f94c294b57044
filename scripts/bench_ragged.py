"""Every sequential function of the suite on a 5 000-group RAGGED batch of the shape `.over("symbol")` produces (groups of 2 400 .. 2 520
rows: a panel with listing gaps) against the same function on the REGULAR 5 000 x 2 520 batch (pitched columns): direct C-ABI calls
with the reference's default parameters on preallocated outputs, ms per call.
  regular        the tiled kernel on the regular batch (what bench.py's suite is made of)
  ragged         the shipped path on the ragged batch: a wave-per-group form where the function has one, else the re-housed tiled path
                 (rg_pack -> the tiled kernel -> rg_unpack, csrc/pq_dev.h launch_seq)
  ragged_gather  PQ_NO_RG_PACK=1 PQ_NO_WT=1: the per-lane gather form every ragged batch ran before round 5
VERDICT r4 item 4: no function slower than 2 x its regular-batch time.  GPU box only; prints one JSON object."""
import ctypes as C
import json
import os
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from polars_quant_amd import api  # noqa: E402
from polars_quant_amd._lib import Batch, check, lib  # noqa: E402
from polars_quant_amd._spec import I, SPEC  # noqa: E402
from polars_quant_amd.synthetic import gen_ohlcv  # noqa: E402

N, T, S = int(os.environ.get("N", 5000)), 2520, 2528
ROW_FUNCS = {"mom", "roc", "rocp", "rocr", "rocr100", "ht_trendline", "ht_trendmode", "trange", "bop", "avgprice", "medprice", "typprice", "wclprice",
             "aroon", "aroonosc", "willr"}
d = gen_ohlcv(0x5EED0002, N, T, 0)
rng = np.random.default_rng(3)
lens = rng.integers(2400, T + 1, size=N).astype(np.int64)
lens[:8] = T
off = np.r_[0, np.cumsum(lens)]
R = int(off[-1])
reg, rag = {}, {}
for k, v in d.items():
    buf = torch.zeros((N, S), dtype=torch.float64, device="cuda")
    buf[:, :T] = torch.from_numpy(v).cuda()
    reg[k] = buf
    rag[k] = torch.from_numpy(np.concatenate([v[s, :lens[s]] for s in range(N)])).cuda()
per = np.tile((2 + np.arange(T) % 29).astype(np.float64), (N, 1))
buf = torch.zeros((N, S), dtype=torch.float64, device="cuda"); buf[:, :T] = torch.from_numpy(per).cuda(); reg["periods"] = buf
rag["periods"] = torch.from_numpy(np.concatenate([per[s, :lens[s]] for s in range(N)])).cuda()
for m in (reg, rag):
    m["real"] = m["close"]
b_reg = Batch(N, T, S)
b_rag, keep = api.ragged_batch(off, torch.device("cuda"))
out_reg = [torch.empty((N, S), dtype=torch.float64, device="cuda") for _ in range(3)]
out_rag = [torch.empty((R,), dtype=torch.float64, device="cuda") for _ in range(3)]
h, L = api.ctx(0), lib()


def timed(fn, reps=10):
    for _ in range(2):
        check(fn())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        check(fn())
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


res = {}
for name, (cols, params, outs, _fam) in SPEC.items():
    if name in ROW_FUNCS:
        continue
    pv = [C.c_int64(int(dv)) if k == I else C.c_double(float(dv)) for _, k, dv in params]
    fn = getattr(L, "pq_" + name)
    mk = lambda b, src, dst: (lambda: fn(h, C.byref(b), *[C.c_void_p(src[c].data_ptr()) for c in cols], *pv, *[C.c_void_p(t.data_ptr()) for t in dst[:len(outs)]]))
    r = {"regular": timed(mk(b_reg, reg, out_reg))}
    api.ragged_rehouse_stats(reset=True); api.wt_stats(reset=True)
    r["ragged"] = timed(mk(b_rag, rag, out_rag))
    r["ragged_path"] = "re-housed tiled" if api.ragged_rehouse_stats() else ("wave per group" if api.wt_stats()[0] else "gather")
    os.environ["PQ_NO_RG_PACK"] = "1"; os.environ["PQ_NO_WT"] = "1"
    r["ragged_gather"] = timed(mk(b_rag, rag, out_rag), reps=4)
    del os.environ["PQ_NO_RG_PACK"], os.environ["PQ_NO_WT"]
    r["ragged_over_regular"] = r["ragged"] / r["regular"]
    res[name] = {k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items()}
# a RECORDED suite on the ragged batch: every sequential function of the list in one recording.  Inside a recording a ragged batch runs
# the per-lane gather bodies of the job kernel (seq_jobs_kernel<2>: one grid, blockIdx.y = job); the same calls made directly take the
# re-housed tiled path / the wave-per-group forms one after the other.
SUITE = [nm for nm in SPEC if nm not in ROW_FUNCS and nm != "mavp"]
outs_suite = {nm: [torch.empty((R,), dtype=torch.float64, device="cuda") for _ in SPEC[nm][2]] for nm in SUITE}
calls = []
for nm in SUITE:
    cols, params, outs, _fam = SPEC[nm]
    pv = [C.c_int64(int(dv)) if k == I else C.c_double(float(dv)) for _, k, dv in params]
    calls.append((getattr(L, "pq_" + nm), [C.c_void_p(rag[c].data_ptr()) for c in cols], pv, [C.c_void_p(t.data_ptr()) for t in outs_suite[nm]]))


def all_calls():
    for fn, ins, pv, os_ in calls:
        check(fn(h, C.byref(b_rag), *ins, *pv, *os_))
    return 0


check(L.pq_suite_begin(h, C.byref(b_rag)))
all_calls()
suite = C.c_void_p()
check(L.pq_suite_end(h, C.byref(suite)))
recorded = {"functions": len(SUITE), "recorded_replay_ms": round(timed(lambda: L.pq_suite_run(h, suite), reps=5), 4),
            "same_calls_direct_ms": round(timed(all_calls, reps=3), 4)}
check(L.pq_suite_destroy(h, suite))
worst = max(res, key=lambda k: res[k]["ragged_over_regular"])
print(json.dumps({"groups": N, "rows": R, "group_len": [int(lens.min()), int(lens.max())], "ms": res,
                  "recorded_ragged_suite": recorded,
                  "worst_ragged_over_regular": {worst: res[worst]["ragged_over_regular"]},
                  "geomean_speedup_over_gather": float(np.exp(np.mean([np.log(v["ragged_gather"] / v["ragged"]) for v in res.values()])))}, indent=1))
