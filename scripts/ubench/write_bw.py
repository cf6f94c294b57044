"""Streaming write / copy bandwidth of the box with plain torch kernels (fill_, copy_) -- the ceiling for fully coalesced traffic."""
import torch
def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps
for gb in (1, 4, 12):
    n = gb * (1 << 30) // 8
    x = torch.empty(n, dtype=torch.float64, device="cuda"); y = torch.empty_like(x)
    t = timed(lambda: x.fill_(1.0)); print(f"fill  {gb:3d} GB: {gb*1.0737/t:6.2f} TB/s written")
    t = timed(lambda: y.copy_(x)); print(f"copy  {gb:3d} GB: {gb*1.0737/t:6.2f} TB/s read + the same written")
    t = timed(lambda: x.sum()); print(f"sum   {gb:3d} GB: {gb*1.0737/t:6.2f} TB/s read")
    del x, y
