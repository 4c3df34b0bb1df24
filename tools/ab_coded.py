#!/usr/bin/env python
"""
In-process A/B of two builds of libmixemt_hip.so on the records iteration (mxm_em_iter_coded: row pass + column reduce):
interleaved rounds on one device and the SAME records, HIP events per round (tools/ab_libs.py is the dense matrix's).

    python -m mixemt_amd.build -DMXM_CODED_CHECK=0 --out build_ab/libmxm_nocheck.so
    python tools/ab_coded.py mixemt_amd/lib/libmixemt_hip.so build_ab/libmxm_nocheck.so [rows]
"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy
import torch
from mixemt_amd import _lib, em, phylotree, preprocess, synth

em.QUADS = False        # this tool measures the records' own pass (em_iter_coded_kernel): no quad dictionary beside them


def bind(path):
    lib = ctypes.CDLL(os.path.abspath(path))
    for name, (restype, argtypes) in _lib.SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = restype, argtypes
    return lib


paths = sys.argv[1:3]
rows = int(sys.argv[3]) if len(sys.argv) > 3 else 1000000
_lib.load()
libs = [bind(p) for p in paths]
refseq = phylotree.load_rsrs(); phy = phylotree.load_build17(refseq); haps = sorted(phy.hap_var)
tables = preprocess.HapVarTables.build(refseq, phy, haps)
row_ptr, site, obs, _ = synth.synth_rows(tables, len(refseq), 0, rows, seed=1)
cm = preprocess.build_em_records_device(tables, row_ptr, site, obs)
wts = torch.ones(rows, dtype=torch.float64, device="cuda")
plan = em.EmPlan(None, wts, records=cm)
props = torch.from_numpy(numpy.random.default_rng(1).dirichlet([1.0] * len(haps))[None, :]).cuda()
lnp = props.log()
out = [torch.zeros_like(props) for _ in libs]
state = em.new_state(1, props.device)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
times = [[] for _ in libs]
for rnd in range(12):
    for i, lib in enumerate(libs):
        plan.lib = lib
        plan.em_iter(props, lnp, state, out[i])
        torch.cuda.synchronize()
        ev[0].record()
        for _ in range(20):
            plan.em_iter(props, lnp, state, out[i])
        ev[1].record()
        torch.cuda.synchronize()
        times[i].append(ev[0].elapsed_time(ev[1]) / 20)
rel = float(((out[0] - out[1]).abs() / out[0].abs().clamp_min(1e-300)).max())
mass = [float((props * o).sum()) for o in out]
print("%d rows as records; mxm_em_iter_coded, 12 interleaved rounds of 20 launches; same sums: %s (max relative difference %.2e; "
      "sum_h p_h T_h = %.9f / %.9f)" % (rows, torch.equal(out[0], out[1]), rel, mass[0], mass[1]))
for p, t in zip(paths, times):
    t = sorted(t)
    print("%-40s median %.4f ms  min %.4f  max %.4f" % (os.path.basename(p), t[len(t) // 2], t[0], t[-1]))
