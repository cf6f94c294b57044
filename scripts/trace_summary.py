"""Summarise a rocprofv3 kernel_trace.csv: timeline of one steady-state step of the suite replay."""
import csv, sys, glob
from collections import defaultdict
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = n.replace("void ", "").replace("row_kernel<", "row<").replace("seq_kernel<", "seq<")
    return n.split("(")[0][:46]
# cluster launches into steps: a new cluster starts when a kernel starts after everything before it has ended
clusters, cur, end = [], [], 0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if cur and s > end + 5000:
        clusters.append(cur); cur = []
    cur.append(r); end = max(end, e)
clusters.append(cur)
steps = [c for c in clusters if sum("seq_jobs_kernel" in r["Kernel_Name"] for r in c) >= 2 and len(c) >= 4]
print(f"{len(rows)} kernel launches, {len(steps)} suite steps")
c = steps[len(steps) // 2]
t0 = int(c[0]["Start_Timestamp"])
tot = defaultdict(float)
for r in c:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    nm = short(r["Kernel_Name"])
    q = f" q={r['Queue_Id']}" if "Queue_Id" in r else ""
    print(f"{s/1e3:9.1f} us  +{(e-s)/1e3:8.1f} us ={e/1e3:8.1f}{q}  grid=({r['Grid_Size_X']},{r['Grid_Size_Y']}) vgpr={r['VGPR_Count']} scratch={r['Scratch_Size']}  {nm}")
    tot[nm] += (e - s) / 1e3
span = (max(int(r["End_Timestamp"]) for r in c) - t0) / 1e3
print(f"step span {span:.1f} us (launches of one step overlap on 4 streams)")
for k, v in sorted(tot.items(), key=lambda kv: -kv[1])[:8]:
    print(f"   {v:9.1f} us  {k}")
# gaps between consecutive steps (last kernel end -> next step's first kernel start)
gaps = []
for a, b in zip(steps[:-1], steps[1:]):
    gaps.append((int(b[0]["Start_Timestamp"]) - max(int(r["End_Timestamp"]) for r in a)) / 1e3)
if gaps:
    spans = [(max(int(r["End_Timestamp"]) for r in c) - int(c[0]["Start_Timestamp"])) / 1e3 for c in steps]
    print(f"step spans: median {sorted(spans)[len(spans)//2]:.1f} us; inter-step gaps: median {sorted(gaps)[len(gaps)//2]:.1f} us, max {max(gaps):.1f} us")
