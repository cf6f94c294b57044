"""Direct calls on a REGULAR batch at several batch sizes, ms per call:
  direct    what the library launches for a direct call (Op::DIRECT_LANE_MAX: the per-lane form for compute-bound walks on small batches)
  tiled     PQ_NO_DIRECT_LANE=1: the tiled body (what a recorded suite runs)
  per_lane  the per-lane form reached through a ragged batch of equal groups with PQ_NO_RG_PACK=1 PQ_NO_WT=1 (cross-check)
-> profiles/r05_direct_lane.json"""
import ctypes as C, json, os, sys
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from polars_quant_amd import api
from polars_quant_amd._lib import Batch, check, lib
from polars_quant_amd._spec import I, SPEC
from polars_quant_amd.synthetic import gen_ohlcv
T, S = 2520, 2528
FUNCS = sys.argv[1].split(",") if len(sys.argv) > 1 else ["sar", "stoch", "stochf", "stochrsi", "obv", "mama", "ht_dcperiod", "ht_sine", "kama", "sma", "bbands", "mfi", "ultosc"]
h, L = api.ctx(0), lib()
def timed(fn, reps=6):
    for _ in range(2): check(fn())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): check(fn())
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
out = {}
for N in (1000, 5000, 20000, 50000):
    base = gen_ohlcv(0x5EED0002, 1000, T, 0)
    reg, rag = {}, {}
    for k, v in base.items():
        t = torch.from_numpy(v).cuda().repeat((N + 999) // 1000, 1)[:N].contiguous()
        buf = torch.zeros((N, S), dtype=torch.float64, device="cuda"); buf[:, :T] = t
        reg[k] = buf; rag[k] = t.reshape(-1).contiguous()
    reg["real"], rag["real"] = reg["close"], rag["close"]
    b_reg = Batch(N, T, S)
    b_rag, keep = api.ragged_batch(np.arange(N + 1, dtype=np.int64) * T, torch.device("cuda"))
    o_reg = [torch.empty((N, S), dtype=torch.float64, device="cuda") for _ in range(3)]
    o_rag = [torch.empty((N * T,), dtype=torch.float64, device="cuda") for _ in range(3)]
    for name in FUNCS:
        cols, params, outs, _ = SPEC[name]
        pv = [C.c_int64(int(dv)) if k == I else C.c_double(float(dv)) for _, k, dv in params]
        fn = getattr(L, "pq_" + name)
        mk = lambda b, src, dst: (lambda: fn(h, C.byref(b), *[C.c_void_p(src[c].data_ptr()) for c in cols], *pv, *[C.c_void_p(t.data_ptr()) for t in dst[:len(outs)]]))
        r = {"direct": round(timed(mk(b_reg, reg, o_reg)), 3)}
        os.environ["PQ_NO_DIRECT_LANE"] = "1"
        r["tiled"] = round(timed(mk(b_reg, reg, o_reg)), 3)
        del os.environ["PQ_NO_DIRECT_LANE"]
        os.environ["PQ_NO_RG_PACK"] = "1"; os.environ["PQ_NO_WT"] = "1"
        r["per_lane"] = round(timed(mk(b_rag, rag, o_rag)), 3)
        del os.environ["PQ_NO_RG_PACK"], os.environ["PQ_NO_WT"]
        out.setdefault(name, {})[N] = r
    del reg, rag, o_reg, o_rag
    torch.cuda.empty_cache()
print(json.dumps(out, indent=1))
