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

argv = sys.argv[1:]
pairs = "--pairs" in argv                 # synth-pe-v1 rows (2 x 150 merged mates) instead of synth-v1
if pairs:
    argv.remove("--pairs")
read_len = 150
if "--read-len" in argv:
    at = argv.index("--read-len")
    read_len = int(argv[at + 1])
    del argv[at:at + 2]
max_entries = None
if "--max-entries" in argv:
    at = argv.index("--max-entries")
    max_entries = int(argv[at + 1])
    del argv[at:at + 2]
long_off = "--no-long" in argv            # round 5's routing: rows beyond 64 observations to the fallback list
if long_off:
    argv.remove("--no-long")
rows = int(argv[0]) if argv else 1000000
variants = argv[1:] or ["sparse", "records", "bytes", "lut", "lut+sort", "linearize"]
refseq = phylotree.load_rsrs()
phy = phylotree.load_build17(refseq)
haps = sorted(phy.hap_var)
tables = preprocess.HapVarTables.build(refseq, phy, haps)
row_ptr, site, obs, _ = (synth.synth_rows(tables, len(refseq), 0, rows, seed=1, pairs=True) if pairs else
                         synth.synth_rows(tables, len(refseq), 0, rows, seed=1, read_len=read_len))
dev = torch.device("cuda")
rp = torch.from_numpy(row_ptr).to(dev)
si = torch.from_numpy(site.view(numpy.int16)).to(dev)
ob = torch.from_numpy(obs).to(dev)
out = torch.empty((rows, len(haps)), dtype=torch.float64, device=dev)
lin = torch.empty((rows, len(haps)), dtype=torch.float64, device=dev)
rowmax = torch.empty(rows, dtype=torch.float64, device=dev)
tables.device(); tables.lut_device(); tables.sparse_device()
lib = _lib.load()
lib.mxm_set_sparse_long_rows(0 if long_off else 1)
if max_entries is not None:
    lib.mxm_set_sparse_long_entries(max_entries)
    print("(rows of the long launch with more than %d marker entries -> fallback list)" % max_entries)
cells = rows * len(haps)
lens = numpy.diff(row_ptr)
print("one MI355X; %d %s x %d haplogroups (%.1f observed sites per row, %.1f %% above 64, %.2f %% above 128)%s; wall time per call, "
      "inputs resident" % (rows, "synth-pe-v1 fragments" if pairs else "synth-v1 reads of %d bp" % read_len, len(haps), lens.mean(),
                           100.0 * (lens > 64).mean(), 100.0 * (lens > 128).mean(),
                           "; rows beyond 64 observations to the fallback list" if long_off else ""))
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
        print("%-11s rep %d: %7.2f ms  (%.3g cells/s, %.0f GB/s %s)%s"
              % (var, rep, dt * 1e3, cells / dt, nbytes / dt / 1e9,
                 "read + written" if var == "linearize" else "written",
                 "  rows left to the fallback kernel: %d" % preprocess.build_em_matrix_device.last_fallback if var.split("+")[0] in ("sparse", "records") else ""))
