#!/usr/bin/env python
"""
Build the synth-v1 matrix once per kernel variant (timing / profiling target):
    python tools/run_build_only.py [rows] [variant ...]
variants: bytes, sparse (marker kernel), records (marker kernel -> row-dictionary records, no dense matrix),
          lut (rows in given order), lut+sort (position-sorted row order); "linearize" times mxm_linearize alone.
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy
import torch
from mixemt_amd import _lib, phylotree, preprocess, synth

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
variants = sys.argv[2:] or ["sparse", "records", "bytes", "lut", "lut+sort", "linearize"]
refseq = phylotree.load_rsrs()
phy = phylotree.load_build17(refseq)
haps = sorted(phy.hap_var)
tables = preprocess.HapVarTables.build(refseq, phy, haps)
row_ptr, site, obs, _ = synth.synth_rows(tables, len(refseq), 0, rows, seed=1)
dev = torch.device("cuda")
rp = torch.from_numpy(row_ptr).to(dev)
si = torch.from_numpy(site.view(numpy.int16)).to(dev)
ob = torch.from_numpy(obs).to(dev)
out = torch.empty((rows, len(haps)), dtype=torch.float64, device=dev)
lin = torch.empty((rows, len(haps)), dtype=torch.float64, device=dev)
rowmax = torch.empty(rows, dtype=torch.float64, device=dev)
tables.device(); tables.lut_device(); tables.sparse_device()
lib = _lib.load()
cells = rows * len(haps)
print("one MI355X; %d synth-v1 reads x %d haplogroups (%.1f observed sites per read); wall time per call, "
      "inputs resident" % (rows, len(haps), row_ptr[-1] / float(rows)))
for var in variants:
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        if var == "linearize":
            _lib.check(lib.mxm_linearize(out.data_ptr(), out.stride(0), rows, len(haps), lin.data_ptr(), lin.stride(0),
                                         rowmax.data_ptr(), torch.cuda.current_stream().cuda_stream), "mxm_linearize")
        else:
            parts = var.split("+")
            if parts[0] == "records":
                cm = preprocess.build_em_records_device(tables, rp, si, ob)
                del cm
            else:
                preprocess.build_em_matrix_device(tables, rp, si, ob, out=out, kernel=parts[0],
                                                  sort_rows=("sort" in parts) if parts[0] == "lut" else "auto")
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        nbytes = cells * 8 * (2 if var == "linearize" else 1)
        print("%-11s rep %d: %7.2f ms  (%.3g cells/s, %.0f GB/s %s)"
              % (var, rep, dt * 1e3, cells / dt, nbytes / dt / 1e9,
                 "read + written" if var == "linearize" else "written"))
