#!/usr/bin/env python
"""
Practical HBM read ceiling of the device at hand: a bare streaming-read kernel over a 43 GB
buffer (the size of the 1M x 5408 fp64 matrix), for several grid sizes.  The streaming EM
kernel's roofline fraction is quoted against the 8 TB/s spec; this number says how much of the
gap is the kernel's and how much is the memory system's.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mixemt_amd import _lib
from mixemt_amd._dev import current_stream

lib = _lib.load()
nbytes = int(float(sys.argv[1])) if len(sys.argv) > 1 and sys.argv[1][0] != "-" else 43264000000
buf = torch.empty(nbytes // 8, dtype=torch.float64, device="cuda").uniform_()
sink = torch.zeros(4, dtype=torch.int32, device="cuda")
NAMES = {0: "grid-stride", 1: "blocked, nt", 2: "dealt 43 KB rows, nt, ring 3, 512 thr",
         3: "records: 5408 code bytes at 4 B/lane + 27-entry table at 8 B/lane, nt, 256 thr"}
patterns = (3,) if "--records" in sys.argv else (0, 1, 2, 3)
for blocked in patterns:
    if blocked == 3:
        nbytes = nbytes // 5624 * 5624
    for wg in ((1, 2) if blocked == 2 else ((2, 3, 4) if blocked == 3 else (1, 2, 4, 8, 16))):
        times = []
        for rep in range(6):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            _lib.check(lib.mxm_diag_stream_read(buf.data_ptr(), nbytes, wg, blocked, sink.data_ptr(),
                                                current_stream()), "diag")
            b.record()
            torch.cuda.synchronize()
            times.append(a.elapsed_time(b))
        best, med = min(times[1:]), sorted(times[1:])[len(times[1:]) // 2]
        print("bare read (%s), %2d WG/CU: median %.3f ms = %.2f TB/s (best %.3f ms = %.2f TB/s)"
              % (NAMES[blocked], wg, med, nbytes / med / 1e9, best,
                 nbytes / best / 1e9))
