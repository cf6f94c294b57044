"""How fast does the Hilbert pipeline (cycle.rs:27-63) forget its start?  CPU only (the oracle): a walk started at row s against the full
walk -- (a) rows until the two agree bit for bit for good (they mostly do not within a useful distance: median ~570 rows, worst
~1 700), (b) worst relative difference W rows after the late start (a few ulp from W = 640 on).  This is what the time-split Hilbert job
of small shards rests on (csrc/fused.hip ht_all_time_split: W = 640, hand-over check at 1e-13).   python scripts/ht_warmup_error.py [n_series]"""
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np

from oracle import pq_oracle as o

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
T = 2520
x = o.gen_ohlcv(0x5EED0002, N, T, 0)["close"]
out = {"series": N, "days": T, "bitwise_merge_distance_rows": {}, "worst_relative_difference": {}}
for name, idx in (("ht_dcperiod", [0]), ("ht_phasor", [0, 1])):
    full = o.call(name, x)
    for s in (630, 1260):
        part = o.call(name, np.ascontiguousarray(x[:, s:]))
        md = np.zeros(N, dtype=np.int64)
        for k in idx:
            neq = full[k][:, s:].view(np.uint64) != part[k].view(np.uint64)
            md = np.maximum(md, np.where(neq.any(axis=1), (T - s) - np.argmax(neq[:, ::-1], axis=1), 0))
        out["bitwise_merge_distance_rows"][f"{name}@{s}"] = dict(zip(("p50", "p90", "p99", "max"), np.percentile(md, [50, 90, 99, 100]).tolist()))
        for W in (256, 384, 512, 640, 768):
            worst = 0.0
            for k in idx:
                a, b = full[k][:, s + W:], part[k][:, W:]
                scale = np.maximum(np.abs(a), np.abs(x[:, s + W:]) if name == "ht_phasor" else 0.0)   # phasor components: of the price level
                worst = max(worst, float(np.nanmax(np.abs(a - b) / np.where(scale > 0, scale, 1.0))))
            out["worst_relative_difference"][f"{name}@{s}+{W}"] = worst
print(json.dumps(out, indent=1))
