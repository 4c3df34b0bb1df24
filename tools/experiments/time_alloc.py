#!/usr/bin/env python
"""What a fresh device allocation costs (round-5 diagnostic for the records stage): torch.empty of N GB, first touch, second
touch; run once plainly and once with PYTORCH_HIP_ALLOC_CONF=expandable_segments:True."""
import os, sys, time
import torch
torch.zeros(1, device="cuda"); torch.cuda.synchronize()
print("allocator conf:", os.environ.get("PYTORCH_HIP_ALLOC_CONF", "(default)"))
for gb in (1, 7, 7, 20, 70):
    n = int(gb * 1e9)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    buf = torch.empty(n, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize(); t1 = time.perf_counter()
    buf.fill_(1)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    buf.fill_(2)
    torch.cuda.synchronize(); t3 = time.perf_counter()
    print("%3d GB: empty %.1f ms (%.0f GB/s), first fill %.1f ms, second fill %.1f ms" % (gb, (t1 - t0) * 1e3, gb / (t1 - t0), (t2 - t1) * 1e3, (t3 - t2) * 1e3))
    del buf
    torch.cuda.empty_cache()
