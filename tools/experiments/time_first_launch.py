#!/usr/bin/env python
"""How long the FIRST kernel launch of libmixemt_hip.so takes in a process (the code object is loaded lazily): round-5 diagnostic."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
t0 = time.perf_counter(); torch.zeros(1, device="cuda"); torch.cuda.synchronize(); print("context: %.1f ms" % ((time.perf_counter() - t0) * 1e3))
from mixemt_amd import _lib
from mixemt_amd._dev import current_stream
t0 = time.perf_counter(); lib = _lib.load(); print("dlopen + bind: %.1f ms" % ((time.perf_counter() - t0) * 1e3))
a = torch.zeros(8, dtype=torch.float64, device="cuda"); out = torch.zeros(1, dtype=torch.float64, device="cuda"); torch.cuda.synchronize()
for i in range(3):
    t0 = time.perf_counter()
    lib.mxm_l1_exp_diff(a.data_ptr(), a.data_ptr(), 8, out.data_ptr(), current_stream()); torch.cuda.synchronize()
    print("library kernel launch %d: %.2f ms" % (i, (time.perf_counter() - t0) * 1e3))
print("library size: %.1f MB" % (os.path.getsize(_lib.LIB_PATH) / 1e6))
