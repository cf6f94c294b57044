"""Replay time of chosen subsets of the suite's tasks (run on the GPU box): what slows a job down when the rest runs beside it?
usage: python scripts/exp_subset.py [name=task,task,... ...]   (default: a fixed list of subsets)"""
import sys; sys.path.insert(0, ".")
import torch
from polars_quant_amd.suite import Suite
from oracle import pq_oracle as oracle
N, T = 5000, 2520
d = oracle.gen_ohlcv(0x5EED0002, N, T, 0)
g = {k: torch.from_numpy(v).cuda() for k, v in d.items()}
st = Suite(N, T, "cuda")
def timeit(tasks, label, reps=5):
    st.record(g, tasks)
    for _ in range(2): st.run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): st.run()
    e1.record(); e1.synchronize()
    print(f"{label:40s} {e0.elapsed_time(e1)/reps:8.3f} ms  {st.info()}", flush=True)
allt = st.tasks(fused=True)
heavy = ["ht_all", "mama", "stoch"]
sets = {"ht_all": ["ht_all"], "heavy": heavy, "heavy+cdl": heavy + ["cdl_all"],
        "all-heavy": [t for t in allt if t not in heavy], "all-cdl": [t for t in allt if t != "cdl_all"], "all": allt,
        "ht_all+light": [t for t in allt if t not in ("mama", "stoch", "cdl_all")]}
for a in sys.argv[1:]:
    k, v = a.split("="); sets = {k: v.split(",")} if a == sys.argv[1] else {**sets, k: v.split(",")}
for k, v in sets.items(): timeit(v, k)
if len(sys.argv) == 1:
    light = []
    for t in allt:
        if t in heavy or t == "cdl_all": continue
        st.record(g, [t])
        if st.info()["seq_jobs"]: light.append(t)
    print("light SEQ tasks:", light)
    timeit(light, "light SEQ only")
    timeit(light[::2], "light SEQ even half")
    timeit(light[1::2], "light SEQ odd half")
    timeit(light[::4], "light SEQ quarter")
    timeit(light + heavy, "all SEQ")
