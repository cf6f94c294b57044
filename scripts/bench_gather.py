"""World-of-one measurement of the per-step exchange (the only world a one-GPU box has): the MACD-cross backtest on a 625 / 5000-symbol
shard x 2520 days with (a) no exchange, (b) pq_gather_summaries in series on the step's stream (round 4's form), (c) the exchange
double-buffered on the communicator's own stream (pq_gather_summaries_begin / _end), and the bare latency of one pq_gather_summaries.
With one rank RCCL's all-gather is a device-to-device copy kernel: its LAUNCH + completion latency is what (b) adds per step, a lower
bound of what 8 ranks add (their kernel also waits for the peers).  Prints one JSON object."""
import ctypes as C
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from polars_quant_amd._lib import Batch, BtParams, check, lib
from polars_quant_amd._spec import BT_DEFAULTS
from polars_quant_amd.api import ctx
from polars_quant_amd.distributed import CabiComm, OverlappedGather
from polars_quant_amd.synthetic import gen_ohlcv

T, PITCH = 2520, 2528
dev = torch.device("cuda:0")
full = gen_ohlcv(0x5EED0002, 5000, T, 0)["close"]
comm = CabiComm(dev, 0, 1)
L, h, prm = lib(), ctx(0), BtParams(**BT_DEFAULTS)
out = {"days": T, "world": 1, "shards": {}}


def timed(fn, steps=200, warm=20):
    for k in range(warm):
        fn(k)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for k in range(steps):
        fn(k)
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / steps * 1e3      # microseconds per step


for n in (625, 5000):
    buf = torch.zeros((n, PITCH), dtype=torch.float64, device=dev)
    buf[:, :T] = torch.from_numpy(full[:n].copy()).to(dev)
    b = Batch(n, T, PITCH)
    curves = [torch.empty((n, PITCH), dtype=torch.float64, device=dev) for _ in range(3)]
    og = OverlappedGather(n, n, dev, comm=comm)

    def run(summ):
        check(L.pq_backtest_macd_cross(h, C.byref(b), C.c_void_p(buf.data_ptr()), 12, 26, 9, C.byref(prm),
                                       *[C.c_void_p(t.data_ptr()) for t in curves], C.c_void_p(summ.data_ptr())))

    def kernel_only(k):
        run(og.local[k & 1])

    def serial(k):
        run(og.local[0])
        check(L.pq_gather_summaries(h, C.c_void_p(og.local[0].data_ptr()), n, C.c_void_p(og.all[0].data_ptr())))

    def overlapped(k):
        slot = og.acquire()
        run(og.local[slot])
        og.begin(slot)

    def gather_alone(k):
        check(L.pq_gather_summaries(h, C.c_void_p(og.local[0].data_ptr()), n, C.c_void_p(og.all[0].data_ptr())))

    r = {"kernel_only_us": timed(kernel_only), "serial_us": timed(serial)}
    r["overlapped_us"] = timed(overlapped)
    og.drain(); torch.cuda.synchronize()
    r["gather_alone_us"] = timed(gather_alone)
    r["same_table"] = bool(torch.equal(og.all[0].view(torch.int64), og.local[0].view(torch.int64)))
    out["shards"][n] = r
k, s = out["shards"][625], out["shards"][5000]
out["projected_8gpu_factor"] = {"kernel_only": s["kernel_only_us"] / k["kernel_only_us"], "serial_gather": s["kernel_only_us"] / k["serial_us"],
                                "overlapped_gather": s["kernel_only_us"] / k["overlapped_us"]}
comm.close()
print(json.dumps(out, indent=1))
