#!/usr/bin/env python
"""
The row-dictionary iteration (mxm_em_iter_coded: em_iter_coded_kernel + column reduce; the records' coded rows only,
no dense rest) timed by events for one or several builds of the library (-D macros of coded_kernels.hpp):
    python tools/time_coded_variants.py [rows] lib1.so [lib2.so ...]
Records are built once by the installed library (straight from synth-v1 observations, no dense matrix).  Prints ms
per call (median of 15 after 3 warm-ups) and the largest relative difference of the column sums against (a) the
installed library's and (b) an fp64 torch evaluation of the same sums from decoded rows (first 4096 rows).
"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy
import torch
from mixemt_amd import _lib, phylotree, preprocess, synth

args = sys.argv[1:]
rows = int(args.pop(0)) if args and args[0].isdigit() else 1000000
paths = [_lib.LIB_PATH] + args
lib0 = _lib.load()
refseq = phylotree.load_rsrs()
phy = phylotree.load_build17(refseq)
haps = sorted(phy.hap_var)
tables = preprocess.HapVarTables.build(refseq, phy, haps)
row_ptr, site, obs, _ = synth.synth_rows(tables, len(refseq), 0, rows, seed=1)
dev = torch.device("cuda")
cm = preprocess.build_em_records_device(tables, row_ptr, site, obs)
H = cm.n_haps
coded = cm.struct()
n_coded = int((cm.ndist > 0).sum().item())
rng = numpy.random.default_rng(3)
props_h = rng.dirichlet(numpy.full(H, 0.05))
props = torch.from_numpy(props_h).to(dev)
w = torch.from_numpy(rng.integers(1, 4, size=rows).astype(numpy.float64)).to(dev)
nbytes = lib0.mxm_workspace_bytes(rows, H, 1)
ws = torch.empty(nbytes // 8 + 1, dtype=torch.float64, device=dev)
stream = torch.cuda.current_stream().cuda_stream
print("one MI355X; %d synth-v1 reads x %d haplogroups as records (%d coded rows, %.2f GB); mxm_em_iter_coded, events"
      % (rows, H, n_coded, cm.used / 1e9))


def small_reference():
    """column sums of the first 4096 rows from decoded rows, fp64 torch"""
    n = min(rows, 4096)
    wide_n = cm.wide_rows()
    wide_n = wide_n[wide_n < n].contiguous()
    small_reference.keep = wide_n
    sub = _lib.Coded(cm.rec.data_ptr(), cm.rec_off.data_ptr(), cm.ndist.data_ptr(), n, None, 0, None, 0,
                     wide_n.data_ptr() if wide_n.numel() else None, int(wide_n.numel()))
    P = torch.zeros((n, H), dtype=torch.float64, device=dev)
    _lib.check(lib0.mxm_decode_rows(ctypes.byref(sub), H, P.data_ptr(), P.stride(0), stream), "decode")
    live = (cm.ndist[:n] > 0).to(torch.float64)
    z = P @ props
    cf = torch.where(z > 0, w[:n] * live / z, torch.zeros_like(z))
    return sub, (P * cf[:, None]).sum(0)


sub, want_small = small_reference()
base = None
for path in paths:
    lib = ctypes.CDLL(os.path.abspath(path))
    fn = lib.mxm_em_iter_coded
    fn.restype, fn.argtypes = _lib.SIGNATURES["mxm_em_iter_coded"]
    col = torch.zeros(H, dtype=torch.float64, device=dev)
    rc = fn(ctypes.byref(sub), w.data_ptr(), props.data_ptr(), H, 1, None, col.data_ptr(), ws.data_ptr(), nbytes, stream)
    assert rc == 0, rc
    small = float(((col - want_small).abs() / want_small.abs().clamp_min(1e-300)).max().item())
    times = []
    for rep in range(18):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        rc = fn(ctypes.byref(coded), w.data_ptr(), props.data_ptr(), H, 1, None, col.data_ptr(), ws.data_ptr(), nbytes,
                stream)
        b.record()
        torch.cuda.synchronize()
        assert rc == 0, rc
        if rep >= 3:
            times.append(a.elapsed_time(b))
    if base is None:
        base = col.clone()
    rel = float(((col - base).abs() / base.abs().clamp_min(1e-300)).max().item())
    print("%-28s %.3f ms (min %.3f)   vs installed %.2e   vs torch on 4096 rows %.2e"
          % (os.path.basename(path), float(numpy.median(times)), min(times), rel, small), flush=True)
