"""Aggregate rocprofv3 --pmc SQ counter passes per kernel (sum over dispatch dims, mean over launches).
usage: python scripts/pmc_sq.py dir1 [dir2 ...]"""
import csv, glob, sys
from collections import defaultdict
agg = defaultdict(lambda: defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        per = defaultdict(float)
        for r in csv.DictReader(open(f)):
            key = (r["Kernel_Name"].split("(")[0].replace("void ", "")[:40], r["Grid_Size_Y"] if "Grid_Size_Y" in r else "", r["Dispatch_Id"], r["Counter_Name"])
            per[key] += float(r["Counter_Value"])
        for (k, gy, did, c), v in per.items():
            agg[(k, gy)][c].append(v)
for (k, gy), cs in sorted(agg.items()):
    if not any(x in k for x in ("seq_jobs", "cdl", "row_jobs", "bt_wave")): continue
    print(f"{k} y={gy}: " + "  ".join(f"{c}={sum(v)/len(v):.4g}" for c, v in sorted(cs.items())))
