#!/usr/bin/env python
"""
In-process A/B of two builds of libmixemt_hip.so (same ABI, e.g. different -D tuning macros or
two revisions): interleaved rounds on one device and the SAME buffers, wall time per call by HIP
events on the stream (cdna_hip_programming.md methodology rule 24 -- run-to-run differences
between processes are ~5 % here, larger than most of what is being compared).

    python tools/ab_libs.py A.so B.so [rows] [op ...]
    ops: iter1 iter2 iter3 (streaming step, 1/2/3 restarts)  iter_f32  linearize  posterior
"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy
import torch
from mixemt_amd import _lib, em
from mixemt_amd._dev import current_stream


def bind(path):
    lib = ctypes.CDLL(os.path.abspath(path))
    for name, (restype, argtypes) in _lib.SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = restype, argtypes
    return lib


paths = sys.argv[1:3]
rows = int(sys.argv[3]) if len(sys.argv) > 3 else 1000000
ops = sys.argv[4:] or ["iter1", "iter3", "linearize", "posterior"]
n_haps = 5408
_lib.load()
libs = [bind(p) for p in paths]
dev = torch.device("cuda")
mat = torch.empty((rows, n_haps), dtype=torch.float64, device=dev).uniform_(-50.0, 0.0)
wts = torch.ones(rows, dtype=torch.float64, device=dev)
plan = em.EmPlan(mat, wts, n_runs=4)
plan32 = em.EmPlan(mat, wts, n_runs=1, storage="f32") if "iter_f32" in ops else None
init = numpy.random.default_rng(1).dirichlet([1.0] * n_haps, size=4)
ln0, p0 = em.log_inits(init)
props, lnp = torch.from_numpy(p0).to(dev), torch.from_numpy(ln0).to(dev)
colsum = torch.zeros_like(props)
post = torch.empty((rows, n_haps), dtype=torch.float64, device=dev) if "posterior" in ops else None


def run(op, lib):
    if op.startswith("iter") and op != "iter_f32":
        b = int(op[4:])
        plan.lib = lib
        plan.em_iter(props[:b], lnp[:b], None, colsum[:b])
    elif op == "iter_f32":
        plan32.lib = lib
        plan32.em_iter(props[:1], lnp[:1], None, colsum[:1])
    elif op == "linearize":
        _lib.check(lib.mxm_linearize(mat.data_ptr(), mat.stride(0), rows, n_haps, plan.lin.data_ptr(),
                                     plan.lin.stride(0), plan.rowmax.data_ptr(), current_stream()), op)
    elif op == "posterior":
        _lib.check(lib.mxm_em_step(mat.data_ptr(), mat.stride(0), wts.data_ptr(), lnp.data_ptr(), rows, n_haps,
                                   post.data_ptr(), post.stride(0), 0, None, plan.ws.data_ptr(), plan.ws_bytes,
                                   current_stream()), op)
    else:
        raise SystemExit("unknown op " + op)


ROUNDS, ITERS = 6, 8
for op in ops:
    times = [[], []]
    for rnd in range(ROUNDS):
        for which in (0, 1):
            for it in range(ITERS):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                run(op, libs[which])
                b.record()
                torch.cuda.synchronize()
                if it >= 2:
                    times[which].append(a.elapsed_time(b))
    med = [float(numpy.median(t)) for t in times]
    print("%-10s rows %8d   A %.3f ms (min %.3f)   B %.3f ms (min %.3f)   B/A %.3f"
          % (op, rows, med[0], min(times[0]), med[1], min(times[1]), med[1] / med[0]))
