#!/usr/bin/env python
"""
Round 6, VERDICT r5 #3: the quad dictionary's codes ranked by OCCURRENCES (then value) against ranked by value alone --
do the hot 32-byte table entries of a row land in different LDS bank groups, and what does the pass gain?

    python tools/ab_quad_ranking.py [rows] [--pairs] lib_a.so [lib_b.so ...]

Every library builds ITS OWN quad dictionary over the same records (mxm_build_quads) and runs the one-restart step
(mxm_em_iter_coded: em_iter_quad_coded_kernel + column reduce) on it: interleaved rounds, HIP events, the same
proportions; column sums against the first library's.  One library alone = a profiling target for a counter pass
(tools/sq_summary.py: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE).
"""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy
import torch
from mixemt_amd import _lib, em, phylotree, preprocess, synth

args = sys.argv[1:]
pairs = "--pairs" in args
if pairs:
    args.remove("--pairs")
rows = int(args.pop(0)) if args and args[0].isdigit() else 1000000
paths = args or [_lib.LIB_PATH]


def bind(path):
    lib = ctypes.CDLL(os.path.abspath(path))
    for name, (restype, argtypes) in _lib.SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = restype, argtypes
    return lib


_lib.load()
refseq = phylotree.load_rsrs()
phy = phylotree.load_build17(refseq)
haps = sorted(phy.hap_var)
tables = preprocess.HapVarTables.build(refseq, phy, haps)
H = len(haps)
row_ptr, site, obs, _ = synth.synth_rows(tables, len(refseq), 0, rows, seed=1, pairs=pairs)
cm = preprocess.build_em_records_device(tables, row_ptr, site, obs)
dev = cm.rec.device
wts = torch.ones(rows, dtype=torch.float64, device=dev)
numpy.random.seed(7)
props = torch.from_numpy(em.init_props(H, 1.0)[None, :]).to(dev)
# proportions as they are late in a run (concentrated on a few haplogroups): what most iterations look like
late = numpy.full(H, 1e-7)
late[[10, 2000, 4000]] = [0.6, 0.3, 0.1]
late = torch.from_numpy((late / late.sum())[None, :]).to(dev)
em.QUADS = False
plans = []
for path in paths:
    lib = bind(path)
    plan = em.EmPlan(None, wts, records=cm)
    plan.lib = lib
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    assert plan.attach_quads(True)
    torch.cuda.synchronize()
    plans.append((os.path.basename(path), plan, (time.perf_counter() - t0) * 1e3))
print("one MI355X; %d %s x %d haplogroups; %d rows with quads" % (rows, "synth-pe-v1 fragments" if pairs else "synth-v1 reads", H, plans[0][1].coded.n_quad_rows))
want = None
for label, p in (("first-iteration proportions", props), ("late proportions", late)):
    times = [[] for _ in plans]
    sums = []
    for rnd in range(5):
        for i, (name, plan, build_ms) in enumerate(plans):
            colsum = torch.zeros_like(p)
            state = em.new_state(1, dev)
            for it in range(8):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                plan.em_iter(p, p.log(), state, colsum)
                b.record()
                torch.cuda.synchronize()
                if it >= 2:
                    times[i].append(a.elapsed_time(b))
            if rnd == 0:
                sums.append(colsum.cpu().numpy())
    for i, (name, plan, build_ms) in enumerate(plans):
        rel = float((numpy.abs(sums[i] - sums[0]) / numpy.abs(sums[0]).max()).max())
        print("%-28s %-28s step %.4f ms (median of %d; min %.4f)   dictionary built in %.1f ms   sums within %.1e of the first"
              % (name, label, float(numpy.median(times[i])), len(times[i]), min(times[i]), build_ms, rel))
