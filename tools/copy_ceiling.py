#!/usr/bin/env python
"""
What a plain device-to-device copy of the matrix reaches on this GPU (read + write stream), as the
yardstick for the one-read-one-write kernels (linearize, posterior pass): torch's copy kernel and
hipMemcpyDtoD on 10^6 x 5408 fp64 (43.26 GB read + 43.26 GB written).
"""
import torch

rows, cols = 1000000, 5408
a = torch.empty((rows, cols), dtype=torch.float64, device="cuda").uniform_(-50.0, 0.0)
b = torch.empty_like(a)
gb = 2 * a.numel() * 8 / 1e9
for name, fn in (("torch copy_", lambda: b.copy_(a)), ("torch add (x + 1.0 -> out)", lambda: torch.add(a, 1.0, out=b))):
    ts = []
    for i in range(8):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        torch.cuda.synchronize()
        if i >= 2:
            ts.append(s.elapsed_time(e))
    ts.sort()
    print("%-28s median %.3f ms = %.2f TB/s (read + write)" % (name, ts[len(ts) // 2], gb / ts[len(ts) // 2]))
