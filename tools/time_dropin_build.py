#!/usr/bin/env python
"""
Wall time of the DROP-IN entry point preprocess.build_em_matrix(refseq, phylo, reads, haplogroups,
args) -- signature strings in, numpy matrix out, like the reference's (preprocess.py:177-198) --
split into its host and device parts.

    python tools/time_dropin_build.py [reads]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy
import torch
from mixemt_amd import phylotree, preprocess, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
refseq = phylotree.load_rsrs()
phy = phylotree.load_build17(refseq)
haps = sorted(phy.hap_var)
tables = preprocess.HapVarTables.build(refseq, phy, haps)
row_ptr, site, obs, _ = synth.synth_reads(tables, len(refseq), n, seed=1)
reads = synth.signatures(tables, row_ptr, site, obs)
preprocess.build_em_matrix(refseq, phy, reads[:64], haps, None)            # warm up (tables, library)
t0 = time.perf_counter()
csr = preprocess.encode_signatures(reads, tables)
t1 = time.perf_counter()
csr_py = preprocess._encode_signatures_py(reads[:20000], tables)
t2 = time.perf_counter()
dev = preprocess.build_em_matrix_device(tables, *csr)
torch.cuda.synchronize()
t3 = time.perf_counter()
host = dev.cpu().numpy()
t4 = time.perf_counter()
# the alternative that was tried for the drop-in return path: one page-locked staging buffer
staged = torch.empty(dev.shape, dtype=dev.dtype, pin_memory=True)
staged.copy_(dev, non_blocking=True)
torch.cuda.synchronize()
pinned = staged.numpy()
t5 = time.perf_counter()
assert numpy.array_equal(pinned, host)
whole0 = time.perf_counter()
mat = preprocess.build_em_matrix(refseq, phy, reads, haps, None)
whole = time.perf_counter() - whole0
assert numpy.array_equal(mat, host)
print("%d reads x %d haplogroups" % (n, len(haps)))
print("  signatures -> CSR, library host parser : %.3f s" % (t1 - t0))
print("  signatures -> CSR, item-by-item Python : %.3f s (extrapolated from 20000 reads)" % ((t2 - t1) * n / 20000.0))
print("  H2D + build kernel                     : %.3f s" % (t3 - t2))
print("  matrix D2H (%.1f GB, pageable .cpu())   : %.3f s (%.1f GB/s)" % (host.nbytes / 1e9, t4 - t3, host.nbytes / 1e9 / (t4 - t3)))
print("  matrix D2H into a fresh page-locked buffer (alloc + copy): %.3f s (%.1f GB/s) -- not used: slower"
      % (t5 - t4, host.nbytes / 1e9 / (t5 - t4)))
print("  build_em_matrix() as called by the reference's host code: %.3f s" % whole)
