#!/usr/bin/env python
"""
Wall time of a full multi-restart run_em (BASELINE config 3 shape: synth-v1 reads x 5408
haplogroups, n_multi sequential Dirichlet inits, defaults tol 1e-4 / max_iter 10000) with the
restart schedules of mxm_em_loop (mxm_set_compact_restarts: 0 all together,
1 running ones packed, 2 one full tile at a time with slot refill).

    python tools/time_restarts.py [--reads N] [--multi M]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy
import torch
from mixemt_amd import _lib, em, phylotree, preprocess, synth

ap = argparse.ArgumentParser()
ap.add_argument("--reads", type=int, default=1000000)
ap.add_argument("--multi", type=int, default=10)
opts = ap.parse_args()
args = argparse.Namespace(init_alpha=1.0, tolerance=1e-4, max_iter=10000, n_multi=opts.multi, verbose=False)
refseq = phylotree.load_rsrs()
phy = phylotree.load_build17(refseq)
haps = sorted(phy.hap_var)
tables = preprocess.HapVarTables.build(refseq, phy, haps)
row_ptr, site, obs, _ = synth.synth_reads(tables, len(refseq), opts.reads, seed=1)
mat = preprocess.build_em_matrix_device(tables, row_ptr, site, obs)
wts = torch.ones(opts.reads, dtype=torch.float64, device="cuda")
lib = _lib.load()
numpy.random.seed(7)
inits = numpy.stack([em.init_props(len(haps), 1.0) for _ in range(opts.multi)])
plan = em.EmPlan(mat, wts, n_runs=opts.multi)
res = {}
for on in (2, 1, 0, 2, 1):
    lib.mxm_set_compact_restarts(on)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ln_cur, ln_new, states = em.em_loop(plan, inits, args.tolerance, args.max_iter)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    iters = [s[1] for s in states]
    res[on] = (ln_new.cpu().numpy(), iters)
    print("schedule %d: EM loop %.3f s, iterations per restart %s (sum %d) -> %.1f restart-iterations/s"
          % (on, dt, iters, sum(iters), sum(iters) / dt))
lib.mxm_set_compact_restarts(2)
# one restart per pass (no sharing of matrix reads at all): the batched kernels' summation order differs,
# the stopping iterations must not
lib.mxm_set_batch_tile(1)
torch.cuda.synchronize(); t0 = time.perf_counter()
ln_cur, ln_new, states = em.em_loop(plan, inits, args.tolerance, args.max_iter)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
lib.mxm_set_batch_tile(4)
one = (ln_new.cpu().numpy(), [s[1] for s in states])
print("one restart per pass: EM loop %.3f s, iterations per restart %s -> %.1f restart-iterations/s"
      % (dt, one[1], sum(one[1]) / dt))
print("tile 4 vs one per pass: same iteration counts: %s; max |delta props| %.3e"
      % (one[1] == res[2][1], float(numpy.abs(numpy.exp(one[0]) - numpy.exp(res[2][0])).max())))
for a in (1, 2):
    print("schedule %d vs 0: same iteration counts: %s; max |delta ln p| over finite entries: %.3e"
          % (a, res[0][1] == res[a][1],
             float(numpy.nanmax(numpy.abs(numpy.where(numpy.isfinite(res[0][0]), res[0][0] - res[a][0], 0.0))))))
