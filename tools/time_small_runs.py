#!/usr/bin/env python
"""Wall time of whole run_em calls on small matrices with / without the hipGraph loop."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import argparse
import numpy
import torch
from mixemt_amd import _lib, em, phylotree, preprocess

g = numpy.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "g4_run_em.npz"))
refseq = phylotree.load_rsrs(); phy = phylotree.load_build17(refseq); haps = sorted(phy.hap_var)
tables = preprocess.HapVarTables.build(refseq, phy, haps)
mat = preprocess.build_em_matrix_device(tables, g["row_ptr"], g["site"], g["obs"])
wts = torch.from_numpy(g["wts"]).cuda()
args = argparse.Namespace(init_alpha=1.0, tolerance=1e-4, max_iter=10000, n_multi=1, verbose=False)
lib = _lib.load()
sub = mat[:, torch.tensor([10, 11, 2000, 3000, 4000], device="cuda")].contiguous()
for name, m in (("600 x 5408", mat), ("600 x 5", sub)):
    for mode in (0, 1, 0, 1):
        lib.mxm_set_loop_graph(mode)
        numpy.random.seed(7)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        res = em.run_em_ex(m, wts, args, want_read_mix=False)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print("%-10s graph=%d  %4d iterations  %.2f ms  (%.1f us/iteration)" % (name, mode, res["iters"][0], dt * 1e3, dt * 1e6 / res["iters"][0]))
lib.mxm_set_loop_graph(-1)
