import sys; sys.path.insert(0, ".")
import torch
from polars_quant_amd.suite import Suite
from oracle import pq_oracle as oracle
N, T = 5000, 2520
d = oracle.gen_ohlcv(0x5EED0002, N, T, 0)
g = {k: torch.from_numpy(v).cuda() for k, v in d.items()}
st = Suite(N, T, "cuda")
def timeit(tasks, label):
    st.record(g, tasks)
    for _ in range(2): st.run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): st.run()
    e1.record(); e1.synchronize()
    print(f"{label:30s} {e0.elapsed_time(e1)/5:8.3f} ms")
for t in sys.argv[1:]:
    timeit(t.split(","), t)
