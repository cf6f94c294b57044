"""Turn the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs of the same bench.py command) into
profiles/<tag>_pmc_traffic.json: mean HBM bytes per launch for every kernel.

Units / corrections (MI355X_MICROARCH.md, HBM section): both counters are in KiB; on gfx950 FETCH_SIZE reports exactly half
of the bytes of a wide coalesced (16 B/lane) read stream, so it is doubled; WRITE_SIZE is exact for 16 B/lane stores.
Calibration inside this very profile: cdl_all_kernel writes 61 int32 columns = 3.07 GB algorithmic and WRITE_SIZE reads
3.08 GB; it reads 4 f64 columns = 403 MB and 2 x FETCH_SIZE reads 430 MB.
usage: python scripts/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r04 [steps incl. warm-up of the profiled command]
The file carries the source hash of the build it was collected on (bench.source_hash): bench.py quotes it only for that build.
"""
import csv, glob, json, sys
from collections import defaultdict
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from bench import source_hash  # noqa: E402

def per_kernel(path, counter):
    f = glob.glob(path + "/**/*counter_collection.csv", recursive=True)[0]
    agg = defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            agg[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in agg.items()}

fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
write = per_kernel(sys.argv[2], "WRITE_SIZE")
out = {}
for k in sorted(set(fetch) | set(write)):
    fk, nf = fetch.get(k, (0.0, 0))
    wk, nw = write.get(k, (0.0, 0))
    out[k] = {"launches": max(nf, nw), "fetch_bytes_per_launch": 2.0 * fk * 1024.0, "write_bytes_per_launch": wk * 1024.0,
              "hbm_bytes_per_launch": 2.0 * fk * 1024.0 + wk * 1024.0,
              "raw_FETCH_SIZE_KiB": fk, "raw_WRITE_SIZE_KiB": wk}
nsteps = int(sys.argv[4]) if len(sys.argv) > 4 else 23
json.dump({"note": "mean per launch; FETCH_SIZE doubled (gfx950 correction), KiB -> bytes", "source_hash": source_hash(),
           "launches_per_step": {k: round(v["launches"] / nsteps) for k, v in out.items() if round(v["launches"] / nsteps) >= 1},
           "kernels": out},
          open(sys.argv[3] + "_pmc_traffic.json", "w"), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"])[:6]:
    print(f"{k[:40]:40s} {v['hbm_bytes_per_launch']/1e9:7.3f} GB/launch  (read {v['fetch_bytes_per_launch']/1e9:.3f} write {v['write_bytes_per_launch']/1e9:.3f})")
