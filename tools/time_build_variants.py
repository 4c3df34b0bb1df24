#!/usr/bin/env python
"""
The marker build kernel alone (mxm_build_em_matrix_sparse: no leftover-row pass, no host sync), timed by HIP
events, for one or several builds of the library (-D tuning macros: SPB_SLOTS, SPB_WAVES, SPB_PASSES):
    python tools/time_build_variants.py [rows] lib1.so [lib2.so ...]
Prints ms per launch (median of 7 after 2 warm-ups), GB/s written, rows handed to the fallback list, and checks
256 sampled rows of every variant against the C oracle.
"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy
import torch
from mixemt_amd import _lib, phylotree, preprocess, synth
from oracle import c_oracle

args = sys.argv[1:]
pairs = "--pairs" in args                                 # synth-pe-v1 rows (2 x 150 merged mates) instead of synth-v1
if pairs:
    args.remove("--pairs")
read_len = 150
if "--read-len" in args:
    at = args.index("--read-len")
    read_len = int(args[at + 1])
    del args[at:at + 2]
max_entries = None
if "--max-entries" in args:
    at = args.index("--max-entries")
    max_entries = int(args[at + 1])
    del args[at:at + 2]
long_off = "--no-long" in args                            # round 5's routing: rows beyond 64 observations to the fallback list
if long_off:
    args.remove("--no-long")
rows = int(args.pop(0)) if args and args[0].isdigit() else 1000000
paths = args or [_lib.LIB_PATH]
_lib.load()
refseq = phylotree.load_rsrs()
phy = phylotree.load_build17(refseq)
haps = sorted(phy.hap_var)
tables = preprocess.HapVarTables.build(refseq, phy, haps)
row_ptr, site, obs, _ = (synth.synth_rows(tables, len(refseq), 0, rows, seed=1, pairs=True) if pairs else
                         synth.synth_rows(tables, len(refseq), 0, rows, seed=1, read_len=read_len))
lens = numpy.diff(row_ptr)
dev = torch.device("cuda")
rp = torch.from_numpy(row_ptr).to(dev)
si = torch.from_numpy(site.view(numpy.int16)).to(dev)
ob = torch.from_numpy(obs).to(dev)
out = torch.empty((rows, len(haps)), dtype=torch.float64, device=dev)
enc = tables.sparse_device()
fallback = torch.empty(rows, dtype=torch.int64, device=dev)
n_fb = torch.zeros(1, dtype=torch.int64, device=dev)
stream = torch.cuda.current_stream().cuda_stream
pick = numpy.sort(numpy.random.default_rng(5).choice(rows, size=min(256, rows), replace=False))
sub_ptr = numpy.zeros(len(pick) + 1, dtype=numpy.int64)
sub_ptr[1:] = numpy.cumsum(row_ptr[pick + 1] - row_ptr[pick])
sub_site = numpy.concatenate([site[row_ptr[r]:row_ptr[r + 1]] for r in pick])
sub_obs = numpy.concatenate([obs[row_ptr[r]:row_ptr[r + 1]] for r in pick])
want = c_oracle.build_em_matrix(tables.expected, tables.lhit, tables.lmiss, sub_ptr, sub_site, sub_obs, len(haps))
print("one MI355X; %d %s x %d haplogroups (%.1f sites per row, %.1f %% above 64, %.2f %% above 128); the marker kernel%s alone, HIP events"
      % (rows, "synth-pe-v1 fragments" if pairs else "synth-v1 reads of %d bp" % read_len, len(haps), lens.mean(),
         100.0 * (lens > 64).mean(), 100.0 * (lens > 128).mean(), " (rows beyond 64 observations to the fallback list)" if long_off else "s"))
for path in paths:
    lib = ctypes.CDLL(os.path.abspath(path))
    fn = lib.mxm_build_em_matrix_sparse
    fn.restype, fn.argtypes = _lib.SIGNATURES["mxm_build_em_matrix_sparse"]
    if hasattr(lib, "mxm_set_sparse_long_rows"):
        lib.mxm_set_sparse_long_rows(0 if long_off else 1)
        if max_entries is not None and hasattr(lib, "mxm_set_sparse_long_entries"):
            lib.mxm_set_sparse_long_entries(max_entries)
    out.fill_(7.0)
    times = []
    for rep in range(9):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        rc = fn(enc["maj"].data_ptr(), enc["lhit"].data_ptr(), enc["lmiss"].data_ptr(), enc["mk_ptr"].data_ptr(),
                enc["mk_hap"].data_ptr(), enc["mk_base"].data_ptr(), rp.data_ptr(), si.data_ptr(), ob.data_ptr(), 0, rows,
                len(haps), len(tables.sites), out.data_ptr(), out.stride(0), fallback.data_ptr(), n_fb.data_ptr(), stream)
        b.record()
        torch.cuda.synchronize()
        assert rc == 0, rc
        if rep >= 2:
            times.append(a.elapsed_time(b))
    left = set(fallback[:int(n_fb.item())].cpu().numpy().tolist())
    got = out[torch.from_numpy(pick).to(dev)].cpu().numpy()
    ok = all(numpy.array_equal(got[i], want[i]) for i, r in enumerate(pick) if int(r) not in left)
    ms = float(numpy.median(times))
    print("%-34s %7.2f ms  (min %.2f)  %.0f GB/s written  fallback rows %d  sampled rows %s"
          % (os.path.basename(path), ms, min(times), rows * len(haps) * 8 / ms / 1e6, int(n_fb.item()),
             "bit-exact" if ok else "DIFFER"))
