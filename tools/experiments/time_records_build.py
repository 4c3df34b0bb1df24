#!/usr/bin/env python
"""
Where does build_em_records_device spend its time on rows that come from ALIGNMENTS (mates: long rows, many of them past
the marker kernel's 64 observations)?  python tools/experiments/time_records_build.py [fragments]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy
import torch
from mixemt_amd import _lib, alignments, phylotree, preprocess, synth

n_frag = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
refseq = phylotree.load_rsrs(); phy = phylotree.load_build17(refseq); haps = sorted(phy.hap_var)
tables = preprocess.HapVarTables.build(refseq, phy, haps)
cols = synth.synth_alignments(tables, refseq, n_frag, seed=1)
enc = alignments.encode_alignments(cols, tables.sites, len(refseq), 30, 30)
nobs = numpy.diff(enc.row_ptr)
print("%d rows from %d alignments: observations per row mean %.1f, median %d, max %d; rows with more than 64: %d (%.1f %%)"
      % (enc.n_rows, len(cols), nobs.mean(), numpy.median(nobs), nobs.max(), int((nobs > 64).sum()), 100.0 * (nobs > 64).mean()))
torch.zeros(1, device="cuda"); _lib.load(); torch.cuda.synchronize()
for rep in range(4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    cm = preprocess.build_em_records_device(tables, enc.row_ptr, enc.site, enc.obs)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    nd = cm.ndist_host()
    print("call %d: %.1f ms; %d rows handed back by the marker kernel, %d wide (16-bit codes), %d dense beside the records; stages %s"
          % (rep, (t1 - t0) * 1e3, preprocess.build_em_matrix_device.last_fallback, int((nd > 256).sum()), cm.rest_rows.numel(),
             getattr(preprocess.build_em_records_device, "last_timing", None)))
    del cm
