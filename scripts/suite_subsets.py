"""Time suite replays for subsets of tasks (which jobs set the critical path of the SEQ grid?)."""
import sys, time
sys.path.insert(0, ".")
import torch
import polars_quant_amd as pq
from polars_quant_amd.suite import Suite
from oracle import pq_oracle as oracle
N, T = 5000, 2520
d = oracle.gen_ohlcv(0x5EED0002, N, T, 0)
g = {k: torch.from_numpy(v).cuda() for k, v in d.items()}
st = Suite(N, T, "cuda")
SEQ0 = ["bbands","dema","ema","kama","ma","mama","midpoint","midprice","sar","sarext","sma","t3","tema","wma",
        "adx","dx","plus_di","minus_di","plus_dm","minus_dm","cmo","macd","macdfix","mfi","rsi","trix","ultosc",
        "atr","natr","ad","adosc","obv","ht_dcperiod","ht_dcphase","ht_phasor","ht_sine","backtest_macd_cross"]
def timeit(tasks, label):
    st.record(g, tasks)
    for _ in range(2): st.run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): st.run()
    e1.record(); e1.synchronize()
    print(f"{label:40s} {e0.elapsed_time(e1)/5:8.3f} ms  {st.info()}")
timeit(SEQ0, "all phase-0 SEQ")
HT = ["ht_dcperiod","ht_dcphase","ht_phasor","ht_sine","mama"]
timeit([t for t in SEQ0 if t not in HT], "no HT/mama")
timeit([t for t in SEQ0 if t not in HT and t != "backtest_macd_cross"], "no HT, no backtest")
timeit([t for t in SEQ0 if t not in HT + ["backtest_macd_cross","mfi","ultosc","kama"]], "no HT/bt/mfi/ultosc/kama")
timeit(HT, "HT only")
timeit(["ema"]*1, "ema x1")
timeit(["ema","dema","tema","t3","rsi","macd","trix","obv","ad","atr"], "10 light jobs")
timeit(["sma","wma","bbands","midpoint","midprice","cmo"], "6 ring jobs")
timeit(["mfi","ultosc","kama"], "3 large ring jobs")
timeit(["adx","dx","plus_di","minus_di","plus_dm","minus_dm"], "DM family")
timeit(["backtest_macd_cross"], "backtest only")
