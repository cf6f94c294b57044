"""ms per recorded suite step on a shard of n symbols (the bench's suite, nothing else): python scripts/step_time.py 625 [1250 ...]
Environment switches of the library / of Suite apply (PQ_SUITE_UNFUSE, PQ_WT_SUITE, PQ_LIB_PATH, PQ_SMALL_SHARD ...).  One line per size."""
import sys

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch

from polars_quant_amd.suite import Suite
from polars_quant_amd.synthetic import gen_ohlcv

T = 2520
PITCH = (T + 15) // 16 * 16
sizes = [int(a) for a in sys.argv[1:]] or [625]
_mode = __import__("os").environ.get("PQ_STEP_STREAM")   # A/B: the step on a stream of its own instead of the null stream
if _mode in ("ext-nb", "ext-b"):   # a stream created with the HIP API itself (non-blocking / blocking), wrapped for torch
    import ctypes
    torch.cuda.init()
    path = next(l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l)
    hip = ctypes.CDLL(path)
    h = ctypes.c_void_p()
    assert hip.hipStreamCreateWithFlags(ctypes.byref(h), 1 if _mode == "ext-nb" else 0) == 0
    torch.cuda.set_stream(torch.cuda.ExternalStream(h.value))
elif _mode:
    torch.cuda.set_stream(torch.cuda.Stream(priority=int(__import__("os").environ.get("PQ_STEP_STREAM_PRIO", "0"))))
full = gen_ohlcv(0x5EED0002, max(sizes), T, 0)
for n in sizes:
    g = {}
    for k, v in full.items():
        buf = torch.zeros((n, PITCH), dtype=torch.float64, device="cuda")
        buf[:, :T] = torch.from_numpy(v[:n].copy()).cuda()
        g[k] = buf[:, :T]
    st = Suite(n, T, "cuda", stride=PITCH)
    st.record(g)
    for _ in range(5):
        st.run()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            st.run()
        e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20)
    import time
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        st.run()
    host = (time.perf_counter() - t0) / 20 * 1e3      # the host's share: 20 steps enqueued, nothing waited for
    torch.cuda.synchronize()
    print(f"n={n} ms_per_step={best:.4f} host_enqueue_ms_per_step={host:.4f} info={st.info()}", flush=True)
    st.close()
