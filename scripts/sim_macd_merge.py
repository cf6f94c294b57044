"""How many rows until a MACD state machine restarted in the middle of a series has merged BITWISE with the one that ran from row 0
(the warm-up length of the speculative chunks in csrc/ops_backtest_wave.h)?  CPU only: the oracle's macd on suffixes of the SURVEY 8(d)
series against the same rows of the full run.  Prints the distribution and the failure fraction per warm-up length."""
import sys

import numpy as np

sys.path.insert(0, ".")
from oracle import pq_oracle as o

N, T, L = 400, 2520, 1200
x = o.gen_ohlcv(0x5EED0002, N, T, 0)["close"]
bits = lambda a: np.ascontiguousarray(a).view(np.uint64)
full = o.call("macd", x)
ms = []
for r0 in range(100, 1300, 100):
    sub = o.call("macd", np.ascontiguousarray(x[:, r0:r0 + L]))
    neq = np.zeros((N, L), bool)
    for k in range(2):          # macd and signal line (the histogram is their difference)
        neq |= bits(sub[k]) != bits(full[k][:, r0:r0 + L])
    ms.append(np.where(neq.any(axis=1), L - np.argmax(neq[:, ::-1], axis=1), 0))
ms = np.concatenate(ms)
print("rows until merged: median %d, 90 %% %d, 99 %% %d, 99.9 %% %d, max %d" % tuple(np.percentile(ms, q) for q in (50, 90, 99, 99.9, 100)))
for W in (480, 520, 560, 600, 640):
    print(f"warm-up {W} rows: {100 * (ms > W).mean():.3f} % of the chunks not merged")
