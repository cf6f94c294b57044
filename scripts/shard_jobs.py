"""Per-job schedule of one suite step on a shard of n symbols (PQ_SUITE_DEBUG=1 prints '[pq suite] job ...' lines on stderr):
which job bounds the step when the chip is not full.   python scripts/shard_jobs.py 625"""
import os
import sys

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
os.environ.setdefault("PQ_SUITE_DEBUG", "1")
import torch

from polars_quant_amd.suite import Suite
from polars_quant_amd.synthetic import gen_ohlcv

n = int(sys.argv[1]) if len(sys.argv) > 1 else 625
T = 2520
PITCH = (T + 15) // 16 * 16
full = gen_ohlcv(0x5EED0002, n, T, 0)
g = {}
for k, v in full.items():
    buf = torch.zeros((n, PITCH), dtype=torch.float64, device="cuda")
    buf[:, :T] = torch.from_numpy(v).cuda()
    g[k] = buf[:, :T]
st = Suite(n, T, "cuda", stride=PITCH)
st.record(g)
for _ in range(3):
    st.run()
torch.cuda.synchronize()
print("---- last step ----", file=sys.stderr, flush=True)
st.run()
torch.cuda.synchronize()
st.close()
