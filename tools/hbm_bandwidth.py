"""Developer aid: what this box's HBM delivers to plain torch kernels on 4 GiB of float64 -- the practical ceiling next to
the 8 TB/s peak the roofline is priced on (profiles/r06_hbm_bandwidth.txt).  Usage (GPU box): python tools/hbm_bandwidth.py"""
import time

import torch

x = torch.empty(2**29, dtype=torch.float64, device="cuda")  # 4 GiB
y = torch.empty_like(x)
for name, fn in (("copy (read + write)", lambda: y.copy_(x)), ("fill (write)", lambda: y.fill_(1.0)), ("sum (read)", lambda: x.sum())):
  for _ in range(3):
    fn()
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(10):
    fn()
  torch.cuda.synchronize()
  dt = (time.perf_counter() - t0) / 10
  nbytes = x.numel() * 8 * (2 if "copy" in name else 1)
  print(f"{name}: {nbytes / dt / 1e12:.2f} TB/s")
