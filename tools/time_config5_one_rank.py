#!/usr/bin/env python
"""
BASELINE config 5 on ONE rank (round 6): 64 restarts on 10^6 x 5408 held as records + quad dictionary, run to convergence
through dist.run_em_restart_parallel(records=...) -- with full tiles of three restarts sharing a pass
(em_iter_quad_batched_kernel) and with one restart per pass (round 5's schedule), same inits, same process.

    python tools/time_config5_one_rank.py [rows] [restarts]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy
import torch
from mixemt_amd import _lib, dist as mdist, em, phylotree, preprocess, synth

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
n_multi = int(sys.argv[2]) if len(sys.argv) > 2 else 64
refseq = phylotree.load_rsrs()
phy = phylotree.load_build17(refseq)
haps = sorted(phy.hap_var)
tables = preprocess.HapVarTables.build(refseq, phy, haps)
row_ptr, site, obs, _ = synth.synth_rows(tables, len(refseq), 0, rows, seed=1)
cm = preprocess.build_em_records_device(tables, row_ptr, site, obs)
wts = torch.ones(rows, dtype=torch.float64, device="cuda")
args = argparse.Namespace(init_alpha=1.0, tolerance=1e-4, max_iter=10000, n_multi=n_multi, verbose=False)
lib = _lib.load()
print("one MI355X, one rank; %d restarts on %d synth-v1 reads x %d haplogroups as records (+ quad dictionary), to convergence" % (n_multi, rows, len(haps)))
out = {}
for tile in (3, 1):
    lib.mxm_set_coded_batch_tile(tile)
    try:
        numpy.random.seed(7)
        timing = {}
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = mdist.run_em_restart_parallel(None, wts, args, want_read_mix=False, timing=timing, records=cm)
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
    finally:
        lib.mxm_reset_tuning()
    iters = [int(x) for x in res["iters"]]
    out[tile] = (timing["loop_s"], iters, res["run_props"])
    print("%d restart(s) per pass: loop %.3f s (wall %.3f s), %d restart-iterations (%d .. %d per restart) = %.1f restart-iterations/s, %.3f ms each"
          % (tile, timing["loop_s"], wall, sum(iters), min(iters), max(iters), sum(iters) / timing["loop_s"], timing["loop_s"] * 1e3 / sum(iters)))
a, b = out[3], out[1]
print("iteration counts equal: %s; max |d props| over all restarts %.2e; loop time %.3f -> %.3f s (%.1f %% less)"
      % (a[1] == b[1], float(numpy.abs(numpy.asarray(a[2]) - numpy.asarray(b[2])).max()), b[0], a[0], 100.0 * (1.0 - a[0] / b[0])))
