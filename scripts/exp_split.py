"""Which side sets the step: SEQ grids, the ROW chain, or the backtest?  (run on the GPU box)"""
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
    print(f"{label:34s} {e0.elapsed_time(e1)/5:8.3f} ms  {st.info()}", flush=True)
allt = st.tasks(fused=True)
rows, seqs = [], []
for t in allt:
    st.record(g, [t]); i = st.info()
    (rows if i["seq_jobs"] == 0 else seqs).append(t)
print("ROW-only:", rows)
timeit(allt, "full")
timeit(seqs, "SEQ tasks only")
timeit(rows, "ROW tasks only")
timeit([t for t in allt if t != "backtest_macd_cross"], "full minus backtest")
timeit([t for t in seqs if t != "backtest_macd_cross"], "SEQ minus backtest")
timeit([t for t in allt if t != "cdl_all"], "full minus cdl_all")
