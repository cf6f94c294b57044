"""Solo replay time of every SEQ task of the suite (run on the GPU box) -> sorted table + sum."""
import sys; sys.path.insert(0, ".")
import torch
from polars_quant_amd.suite import Suite
from oracle import pq_oracle as oracle
N, T = 5000, 2520
d = oracle.gen_ohlcv(0x5EED0002, N, T, 0)
g = {k: torch.from_numpy(v).cuda() for k, v in d.items()}
st = Suite(N, T, "cuda")
res = []
for t in st.tasks(fused=True):
    st.record(g, [t]); info = st.info()
    if info["seq_jobs"] == 0: continue
    for _ in range(2): st.run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): st.run()
    e1.record(); e1.synchronize()
    res.append((e0.elapsed_time(e1) / 5, t, info["seq_jobs"], info["phases"]))
res.sort(reverse=True)
for ms, t, nj, ph in res: print(f"{t:22s} {ms:7.3f} ms  jobs={nj} phases={ph}")
print("sum", sum(r[0] for r in res))
