"""Mean per launch of every counter in a rocprofv3 --pmc counter_collection.csv, per kernel.  usage: pmc_generic.py <dir>"""
import csv, glob, sys
from collections import defaultdict
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
agg = defaultdict(lambda: defaultdict(list))
for r in csv.DictReader(open(f)):
    agg[r["Kernel_Name"].split("(")[0].replace("void ", "")[:44]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in sorted(agg.items()):
    if "at::" in k or "rocclr" in k: continue
    print(f"{k:44s} " + "  ".join(f"{c}={sum(v)/len(v):.4g}" for c, v in sorted(cs.items())) + f"  launches={len(next(iter(cs.values())))}")
