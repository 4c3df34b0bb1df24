#!/usr/bin/env python
"""
Round 6: several restarts over records of FEWER rows than the 3*10^5 from which one restart takes the quad dictionary --
the one-launch loop over the records (restarts one after another) against the per-iteration kernels beside a quad
dictionary (full tiles of three restarts share a pass).  Where is the crossover?

    python tools/time_multi_restart_routes.py [rows ...]
Wall time of em.em_loop over 64 iterations of every restart (tolerance 0), the dictionary's build time apart.
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy
import torch
from mixemt_amd import _lib, em, phylotree, preprocess, synth

sizes = [int(a) for a in sys.argv[1:]] or [20000, 40000, 80000, 150000, 300000]
refseq = phylotree.load_rsrs()
phy = phylotree.load_build17(refseq)
haps = sorted(phy.hap_var)
tables = preprocess.HapVarTables.build(refseq, phy, haps)
H = len(haps)
lib = _lib.load()
ITERS = 64
print("one MI355X; %d haplogroups; em.em_loop over %d iterations of each restart (tolerance 0), wall time incl. launches and read-backs" % (H, ITERS))
for rows in sizes:
    row_ptr, site, obs, _ = synth.synth_rows(tables, len(refseq), 0, rows, seed=1)
    cm = preprocess.build_em_records_device(tables, row_ptr, site, obs)
    wts = torch.ones(rows, dtype=torch.float64, device="cuda")
    for n_runs in (3, 4, 6, 10):
        inits = numpy.random.RandomState(7).dirichlet([1.0] * H, size=n_runs)
        out = []
        for label, quads, fused in (("records, one launch per restart", False, -1), ("records + quads, tiles of three", True, 0),
                                    ("records + quads, one per pass", True, 0)):
            em.QUADS = False
            plan = em.EmPlan(None, wts, n_runs=n_runs, records=cm)
            t_build = 0.0
            if quads:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                assert plan.attach_quads(True)
                torch.cuda.synchronize()
                t_build = (time.perf_counter() - t0) * 1e3
            lib.mxm_set_loop_fused(fused, 0)
            lib.mxm_set_coded_batch_tile(1 if label.endswith("one per pass") else 3)
            try:
                em.em_loop(plan, inits, 0.0, 3)
                torch.cuda.synchronize()
                best = None
                for rep in range(2):
                    t0 = time.perf_counter()
                    _, _, states = em.em_loop(plan, inits, 0.0, ITERS)
                    torch.cuda.synchronize()
                    dt = time.perf_counter() - t0
                    best = dt if best is None else min(best, dt)
            finally:
                lib.mxm_reset_tuning()
            done = sum(s[1] for s in states)
            out.append((label, best * 1e6 / done, t_build))
            del plan
        em.QUADS = "auto"
        print("%7d rows, %2d restarts: " % (rows, n_runs) + "   ".join("%s %.1f us per restart-iteration%s" % (l, us, (" (dictionary %.1f ms)" % b) if b else "") for l, us, b in out))
    del cm
    torch.cuda.empty_cache()
