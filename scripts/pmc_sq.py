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
N_SIMD = 1024   # 256 CUs x 4
tot_valu = 0.0
for (k, gy), cs in sorted(agg.items()):
    if not any(x in k for x in ("seq_jobs", "seq_mj", "cdl", "row_jobs", "bt_wave")): continue
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    print(f"{k} y={gy}: " + "  ".join(f"{c}={v:.4g}" for c, v in sorted(m.items())))
    # VALU-busy per SIMD: SQ_ACTIVE_INST_VALU counts quad-cycles (one per 64-wide VALU pass of 4 clocks), summed over the chip; the
    # kernel's own clocks are GRBM_GUI_ACTIVE / 8 (that counter is the sum over the 8 XCDs)
    if m.get("SQ_ACTIVE_INST_VALU") and m.get("GRBM_GUI_ACTIVE"):
        cyc = m["GRBM_GUI_ACTIVE"] / 8.0
        busy = 4.0 * m["SQ_ACTIVE_INST_VALU"] / (N_SIMD * cyc)
        conf = m.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(m.get("SQ_LDS_IDX_ACTIVE", 0.0), 1.0)
        print(f"    -> {cyc / 2.4e6:.3f} ms of clocks at 2.4 GHz (this kernel ALONE under the profiler); VALU busy per SIMD {busy:.3f}; "
              f"LDS bank conflicts {conf:.3f} of the LDS index cycles")
        tot_valu += 4.0 * m["SQ_ACTIVE_INST_VALU"]
if tot_valu:
    print(f"step: VALU clocks summed over the launches of one step (every launch is one line above) = {tot_valu:.4g} SIMD-clocks; a 3.9 ms step is "
          f"{N_SIMD * 3.9e-3 * 2.4e9:.4g} SIMD-clocks on {N_SIMD} SIMDs at 2.4 GHz: the vector ALUs are busy {tot_valu / (N_SIMD * 3.9e-3 * 2.4e9):.3f} of it")
