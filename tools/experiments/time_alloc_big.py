#!/usr/bin/env python
"""What does a process's first LARGE allocation cost?  (the record buffer of 10^7 rows is 69 GB)"""
import sys
import time

import torch

torch.zeros(1, device="cuda"); torch.cuda.synchronize()
sizes = [int(float(a) * (1 << 30)) for a in sys.argv[1:]] or [int(g * (1 << 30)) for g in (8, 32, 60, 63.9, 64.1, 80)]
for n in sizes:
    t0 = time.perf_counter(); x = torch.empty(n, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize(); t1 = time.perf_counter()
    x.zero_(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("%.1f GiB (%.1f GB): torch.empty %.1f ms, first fill %.1f ms" % (n / (1 << 30), n / 1e9, (t1 - t0) * 1e3, (t2 - t1) * 1e3), flush=True)
    del x; torch.cuda.empty_cache()
    t0 = time.perf_counter(); x = torch.empty(n, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize(); t1 = time.perf_counter()
    print("   again after empty_cache: torch.empty %.1f ms" % ((t1 - t0) * 1e3), flush=True)
    del x; torch.cuda.empty_cache()
a = torch.empty(35 << 30, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
t0 = time.perf_counter(); b = torch.empty(35 << 30, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
print("a second 35 GiB beside a first: %.1f ms" % ((time.perf_counter() - t0) * 1e3))
