"""GPU box: practical HBM roofline at the byte counts of the cfg3 kernels: torch elementwise copy (read N, write N)
for several N, timed with HIP events over back-to-back launches."""
import torch
for mb in (8, 33.5, 67, 134, 268, 1024):
    n = int(mb * 1e6 / 2)
    x = torch.empty(n, dtype=torch.bfloat16, device="cuda").normal_()
    y = torch.empty_like(x)
    for _ in range(5): y.copy_(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 50
    e0.record()
    for _ in range(reps): y.copy_(x)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    print(f"copy {mb:7.1f} MB read + {mb:7.1f} MB write: {us:8.1f} us  {2 * mb / us * 1e-6 * 1e6 / 1e6:6.2f} TB/s" if False else
          f"copy {mb:7.1f} MB -> {us:8.1f} us   {2 * mb * 1e6 / (us * 1e-6) / 1e12:5.2f} TB/s (read+write)")
    z = torch.empty(n, dtype=torch.bfloat16, device="cuda")
    e0.record()
    for _ in range(reps): z.fill_(1.0)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    print(f"fill {mb:7.1f} MB -> {us:8.1f} us   {mb * 1e6 / (us * 1e-6) / 1e12:5.2f} TB/s (write only)")
    e0.record()
    for _ in range(reps): s = x.sum()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    print(f"sum  {mb:7.1f} MB -> {us:8.1f} us   {mb * 1e6 / (us * 1e-6) / 1e12:5.2f} TB/s (read only)")
