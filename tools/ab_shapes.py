#!/usr/bin/env python
"""
In-process A/B of the two single-restart shapes of the streaming kernel (interleaved rounds, one
process, one device -- cdna_hip_programming.md methodology rule 24): HIP-event time of the kernel
alone, ROUNDS x ITERS launches per shape, median / min per shape.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy
import torch
from mixemt_amd import _lib, em

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
n_haps = 5408
lib = _lib.load()
dev = torch.device("cuda")
mat = torch.empty((rows, n_haps), dtype=torch.float64, device=dev).uniform_(-50.0, 0.0)
plan = em.EmPlan(mat, torch.ones(rows, dtype=torch.float64, device=dev))
init = numpy.random.default_rng(1).dirichlet([1.0] * n_haps)[None, :]
ln0, p0 = em.log_inits(init)
props, lnp = torch.from_numpy(p0).to(dev), torch.from_numpy(ln0).to(dev)
colsum = torch.zeros_like(props)
ROUNDS, ITERS = 8, 12
times = {0: [], 1: []}
for rnd in range(ROUNDS):
    for shape in (0, 1):
        lib.mxm_set_v1_shape(shape)
        for it in range(ITERS):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); b.record()
            lib.mxm_set_timing_events(a.cuda_event, b.cuda_event)
            plan.em_iter(props, lnp, None, colsum)
            lib.mxm_set_timing_events(None, None)
            torch.cuda.synchronize()
            if it >= 2:
                times[shape].append(a.elapsed_time(b))
lib.mxm_set_v1_shape(1)
for shape, name in ((0, "256 thr x 2 WG/CU, ring 2"), (1, "512 thr x 1 WG/CU, ring 3")):
    t = numpy.array(times[shape])
    gb = rows * n_haps * 8 / 1e9
    print("shape %d (%s): median %.3f ms (%.2f TB/s)  min %.3f  max %.3f  n=%d"
          % (shape, name, numpy.median(t), gb / numpy.median(t), t.min(), t.max(), len(t)))
