"""Aggregate rocprofv3 --pmc SQ counter passes per kernel: sum over the dispatches of ONE step (a kernel launched twice per step --
seq_jobs_kernel<0> -- counts twice), mean over the profiled steps (= the dispatch count of cdl_all_kernel, which runs once per step).
usage: python scripts/pmc_sq.py dir1 [dir2 ...]"""
import csv, glob, sys
from collections import defaultdict
tot = defaultdict(lambda: defaultdict(float))      # kernel -> counter -> sum over every dispatch
disp = defaultdict(lambda: defaultdict(set))       # kernel -> counter -> dispatch ids
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:40]
            tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[k][r["Counter_Name"]].add(r["Dispatch_Id"])
N_SIMD = 1024   # 256 CUs x 4
steps = {c: len(v) for c, v in disp.get("cdl_all_kernel<true, true>", {}).items()}
tot_valu = 0.0
for k in sorted(tot):
    if not any(x in k for x in ("seq_jobs", "cdl", "row_jobs", "bt_wave")): continue
    m = {c: v / max(steps.get(c, 1), 1) for c, v in tot[k].items()}            # per step
    per_step = {c: len(disp[k][c]) / max(steps.get(c, 1), 1) for c in tot[k]}
    n_launch = max(per_step.values()) if per_step else 1
    print(f"{k} ({n_launch:g} launch(es) per step; counters summed over them): " + "  ".join(f"{c}={v:.4g}" for c, v in sorted(m.items())))
    # VALU-busy per SIMD: SQ_ACTIVE_INST_VALU counts quad-cycles (one per 64-wide VALU pass of 4 clocks), summed over the chip; the
    # kernel's own clocks are GRBM_GUI_ACTIVE / 8 (that counter is the sum over the 8 XCDs)
    if m.get("SQ_ACTIVE_INST_VALU") and m.get("GRBM_GUI_ACTIVE"):
        cyc = m["GRBM_GUI_ACTIVE"] / 8.0
        busy = 4.0 * m["SQ_ACTIVE_INST_VALU"] / (N_SIMD * cyc)
        conf = m.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(m.get("SQ_LDS_IDX_ACTIVE", 0.0), 1.0)
        print(f"    -> {cyc / 2.4e6:.3f} ms of clocks at 2.4 GHz (its launches ONE AFTER THE OTHER under the profiler); VALU busy per SIMD while it runs "
              f"{busy:.3f}; LDS bank conflicts {conf:.3f} of the LDS index cycles")
        tot_valu += 4.0 * m["SQ_ACTIVE_INST_VALU"]
if tot_valu:
    print(f"step: VALU clocks of every launch of one step = {tot_valu:.4g} SIMD-clocks; a 3.9 ms step is {N_SIMD * 3.9e-3 * 2.4e9:.4g} SIMD-clocks "
          f"on {N_SIMD} SIMDs at 2.4 GHz: the vector ALUs are busy {tot_valu / (N_SIMD * 3.9e-3 * 2.4e9):.3f} of it")
