#!/usr/bin/env python
"""Build the synth-v1 matrix once per kernel (profiling target): python tools/run_build_only.py [rows] [kernel...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy
import torch
from mixemt_amd import phylotree, preprocess, synth

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
kernels = sys.argv[2:] or ["packed", "bytes"]
refseq = phylotree.load_rsrs()
phy = phylotree.load_build17(refseq)
haps = sorted(phy.hap_var)
tables = preprocess.HapVarTables.build(refseq, phy, haps)
row_ptr, site, obs, _ = synth.synth_reads(tables, len(refseq), rows, seed=1)
dev = torch.device("cuda")
rp = torch.from_numpy(row_ptr).to(dev)
si = torch.from_numpy(site.view(numpy.int16)).to(dev)
ob = torch.from_numpy(obs).to(dev)
out = torch.empty((rows, len(haps)), dtype=torch.float64, device=dev)
tables.device(); tables.packed_device()
for kern in kernels:
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        preprocess.build_em_matrix_device(tables, rp, si, ob, out=out, kernel=kern)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print("%s rep %d: %.2f ms  (%.3g cells/s, %.0f GB/s written)" % (kern, rep, dt * 1e3, rows * len(haps) / dt, rows * len(haps) * 8 / dt / 1e9))
