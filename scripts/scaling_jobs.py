"""How does the SEQ job grid scale with the number of identical jobs? (throughput ceiling of the LDS-tile body)"""
import sys
sys.path.insert(0, ".")
import ctypes as C
import torch
from polars_quant_amd._lib import Batch, check, lib
from polars_quant_amd.api import ctx
from oracle import pq_oracle as oracle
N, T = 5000, 2520
d = oracle.gen_ohlcv(0x5EED0002, N, T, 0)
g = {k: torch.from_numpy(v).cuda() for k, v in d.items()}
L, h, b = lib(), ctx(0), Batch(N, T, T)
outs = [torch.empty((N, T), dtype=torch.float64, device="cuda") for _ in range(64 * 3)]
vp = lambda t: C.c_void_p(t.data_ptr())
def run(label, nj, add, bytes_per_row):
    check(L.pq_suite_begin(h, C.byref(b)))
    for j in range(nj): add(j)
    s = C.c_void_p(); check(L.pq_suite_end(h, C.byref(s)))
    for _ in range(2): check(L.pq_suite_run(h, s))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): check(L.pq_suite_run(h, s))
    e1.record(); e1.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f"{label:14s} jobs={nj:3d} {ms:8.3f} ms  {bytes_per_row * nj * N * T / ms / 1e6:8.1f} GB/s")
    check(L.pq_suite_destroy(h, s))
for nj in (1, 4, 8, 16, 32, 64):
    run("ema(30)", nj, lambda j: check(L.pq_ema(h, C.byref(b), vp(g["close"]), 30, vp(outs[j]))), 16)
for nj in (8, 32, 64):
    run("sma(30)", nj, lambda j: check(L.pq_sma(h, C.byref(b), vp(g["close"]), 30, vp(outs[j]))), 16)
for nj in (8, 32, 64):
    run("macd", nj, lambda j: check(L.pq_macd(h, C.byref(b), vp(g["close"]), 12, 26, 9, vp(outs[3*j]), vp(outs[3*j+1]), vp(outs[3*j+2]))), 32)
for nj in (8, 32):
    run("atr(14)", nj, lambda j: check(L.pq_atr(h, C.byref(b), vp(g["high"]), vp(g["low"]), vp(g["close"]), 14, vp(outs[j]))), 32)
