"""Summarise a rocprofv3 kernel_trace.csv: per-step timeline of the suite replay (last full step)."""
import csv, sys, glob
from collections import defaultdict
path = sys.argv[1]
f = glob.glob(path + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = n.replace("void ", "")
    for a, b in (("row_kernel<", "row<"), ("seq_kernel<", "seq<")):
        n = n.replace(a, b)
    return n.split("(")[0][:46]
# find the suite replays: sequences starting at seq_jobs_kernel with the largest grid.y
idx = [i for i, r in enumerate(rows) if "seq_jobs_kernel" in r["Kernel_Name"]]
# take the window covering the last complete step before the per-task probes: search for repeating pattern
starts = [i for i in idx if (i == 0 or "seq_jobs_kernel" not in rows[i-1]["Kernel_Name"]) ]
print("n kernels", len(rows), "suite-ish starts", len(starts))
# print timeline of the 3rd replay
import itertools
seqs = []
cur = []
for r in rows:
    cur.append(r)
# identify step boundaries by the first seq_jobs launch with max grid y
maxy = max(int(r["Grid_Size_Y"]) for r in rows if "seq_jobs_kernel" in r["Kernel_Name"])
bounds = [i for i, r in enumerate(rows) if "seq_jobs_kernel" in r["Kernel_Name"] and int(r["Grid_Size_Y"]) == maxy]
print("steps found", len(bounds))
a, b = bounds[-2], bounds[-1]
t0 = int(rows[a]["Start_Timestamp"])
tot = defaultdict(float)
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    nm = short(r["Kernel_Name"])
    print(f"{s/1e3:9.1f} us  +{(e-s)/1e3:8.1f} us  grid=({r['Grid_Size_X']},{r['Grid_Size_Y']}) lds={r['LDS_Block_Size']} vgpr={r['VGPR_Count']} scr={r['Scratch_Size']}  {nm}")
    tot[nm] += (e - s) / 1e3
print("step span us:", (int(rows[b]["Start_Timestamp"]) - t0) / 1e3)
for k, v in sorted(tot.items(), key=lambda kv: -kv[1])[:12]:
    print(f"   {v:9.1f} us  {k}")
