#!/usr/bin/env python
"""
Wall time per EM iteration of whole run_em calls on cache-resident matrices (the regime real
mixemt inputs live in: de-duplicated signatures, preprocess.py:163-174): the one-launch loop
("one launch": em_fused_cols_kernel up to 1536 rows -- columns split over the workgroups, matrix in
registers -- and em_fused_loop_kernel above; "rows split" forces the latter) against the per-iteration
kernels, with and without the hipGraph replay.

    python tools/time_small_runs.py [--rows 600,2400,10000,...]
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
ap.add_argument("--rows", default="600,2400,4600,10000,30000,100000")
ap.add_argument("--max-iter", type=int, default=200)
ap.add_argument("--restarts", type=int, default=1, help="restarts per run (the one-launch loop takes them one after "
                "another, the per-iteration kernels four per pass)")
ap.add_argument("--coded-grids", default="", help="also time the one-launch loop over records with these many workgroups")
ap.add_argument("--coded-only", action="store_true", help="skip the dense matrix's loops")
ap.add_argument("--stamps", action="store_true",
                help="with a -DFUSED_STAMPS build of the library (MXM_LIB=...): per-phase time shares of the one-launch loop")
opts = ap.parse_args()
refseq = phylotree.load_rsrs(); phy = phylotree.load_build17(refseq); haps = sorted(phy.hap_var)
tables = preprocess.HapVarTables.build(refseq, phy, haps)
args = argparse.Namespace(init_alpha=1.0, tolerance=1e-4, max_iter=opts.max_iter, n_multi=1, verbose=False)
lib = _lib.load()
print("one MI355X; %d haplogroups; fixed %d iterations (tolerance never met that early); wall time of "
      "em.em_loop incl. launch and final state read-back" % (len(haps), opts.max_iter))
for n_rows in [int(x) for x in opts.rows.split(",")]:
    row_ptr, site, obs, _ = synth.synth_rows(tables, len(refseq), 0, n_rows, seed=1)
    mat = preprocess.build_em_matrix_device(tables, row_ptr, site, obs)
    wts = torch.ones(n_rows, dtype=torch.float64, device="cuda")
    plan = em.EmPlan(mat, wts, n_runs=opts.restarts)
    numpy.random.seed(7)
    init = numpy.stack([em.init_props(len(haps), 1.0) for _ in range(opts.restarts)])
    out = {}
    for label, fused, graph in (() if opts.coded_only else
                                (("one launch", 1, 0), ("one launch, rows split", 2, 0), ("kernels", 0, 0),
                                 ("kernels+graph", 0, 1), ("one launch", 1, 0), ("one launch, rows split", 2, 0),
                                 ("kernels", 0, 0))):
        lib.mxm_set_loop_fused(fused, 0)
        lib.mxm_set_loop_graph(graph)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ln_cur, ln_new, states = em.em_loop(plan, init, 0.0, opts.max_iter)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        out[label] = ln_new.cpu().numpy()
        if opts.stamps and fused:
            import ctypes
            st = (ctypes.c_ulonglong * 8)()
            _lib.check(lib.mxm_diag_fused_stamps(plan.ws.data_ptr(), st), "mxm_diag_fused_stamps")
            if st[5]:
                names = ("row pass", "barrier 1", "slice reduce", "barrier 2", "normalise+test")
                print("        workgroup 0, us per iteration (stamped build): "
                      + ", ".join("%s %.2f" % (n, st[i] * 0.01 / st[5]) for i, n in enumerate(names))
                      + "  (sum %.2f)" % (sum(st[:5]) * 0.01 / st[5]))
        n_it = sum(st[1] for st in states)
        print("%7d rows (%6.1f MB)  %-23s %4d restart-iterations  %8.2f ms  %7.1f us per restart-iteration"
              % (n_rows, n_rows * len(haps) * 8 / 1e6, label, n_it, dt * 1e3, dt * 1e6 / n_it))
    lib.mxm_set_loop_fused(-1, 0)
    cplan = em.EmPlan(mat, wts, n_runs=opts.restarts, storage="coded")
    variants = [(-1, 0, "records, one launch", 0), (0, 0, "records, kernels", 0), (0, 1, "records, kernels+graph", 0),
                (-1, 0, "records, one launch", 0), (0, 0, "records, kernels", 0)]
    variants += [(-1, 0, "records, one launch/%d" % int(g), int(g)) for g in opts.coded_grids.split(",") if g]
    for fused, graph, label, grid in variants:
        if cplan.coded is None:
            break
        lib.mxm_set_fused_coded_grid(grid)
        lib.mxm_set_loop_fused(fused, 0)
        lib.mxm_set_loop_graph(graph)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ln_cur, ln_new, states = em.em_loop(cplan, init, 0.0, opts.max_iter)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        n_it = sum(st[1] for st in states)
        print("%7d rows (%6.1f MB)  %-23s %4d restart-iterations  %8.2f ms  %7.1f us per restart-iteration"
              % (n_rows, cplan.coded_bytes / 1e6, label, n_it, dt * 1e3, dt * 1e6 / n_it))
        if opts.stamps and fused:
            import ctypes
            st = (ctypes.c_ulonglong * 8)()
            _lib.check(lib.mxm_diag_fused_stamps(cplan.ws.data_ptr(), st), "mxm_diag_fused_stamps")
            if st[5]:
                names = ("row pass", "barrier 1", "slice reduce", "barrier 2", "normalise+test")
                print("        workgroup 0, us per iteration (stamped build): "
                      + ", ".join("%s %.2f" % (n, st[i] * 0.01 / st[5]) for i, n in enumerate(names))
                      + "  (sum %.2f)" % (sum(st[:5]) * 0.01 / st[5]))
    lib.mxm_set_fused_coded_grid(0)
    if not opts.coded_only:
      print("        max |delta ln p| one launch vs kernels over finite entries: %.2e"
          % float(numpy.nanmax(numpy.abs(numpy.where(numpy.isfinite(out["kernels"]), out["one launch"] - out["kernels"], 0.0)))))
lib.mxm_set_loop_fused(-1, 0)
lib.mxm_set_loop_graph(-1)
