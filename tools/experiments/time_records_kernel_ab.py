#!/usr/bin/env python
"""mxm_build_em_records alone (the marker kernel writing records), the product against a timing build without the shared bump
pointer (-DRECORDS_FIXED_SLOTS: a fixed slot per row): what does one atomic add per row on one address cost it?
    python -m mixemt_amd.build -DRECORDS_FIXED_SLOTS --out build_ab/libmxm_recfixed.so
    python tools/experiments/time_records_kernel_ab.py [rows]"""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy
import torch
from mixemt_amd import _lib, phylotree, preprocess, synth
from mixemt_amd._dev import as_device, current_stream

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
refseq = phylotree.load_rsrs(); phy = phylotree.load_build17(refseq); haps = sorted(phy.hap_var)
tables = preprocess.HapVarTables.build(refseq, phy, haps)
H, S = len(haps), len(tables.sites)
row_ptr, site, obs, _ = synth.synth_rows(tables, len(refseq), 0, rows, seed=1)
enc = tables.sparse_device()
dev = enc["maj"].device
rp, si, ob = as_device(row_ptr, torch.int64, dev), as_device(site, torch.uint16, dev), as_device(obs, torch.uint8, dev)
ldc = (H + 7) // 8 * 8
cap = rows * (ldc + 16 * 400)
rec = torch.empty(cap, dtype=torch.uint8, device=dev)
rec_off = torch.empty(rows, dtype=torch.int64, device=dev); ndist = torch.empty(rows, dtype=torch.int32, device=dev)
rowmax = torch.empty(rows, dtype=torch.float64, device=dev)
fallback = torch.empty(rows, dtype=torch.int64, device=dev)
for path in ("mixemt_amd/lib/libmixemt_hip.so", "build_ab/libmxm_recfixed.so"):
    lib = ctypes.CDLL(os.path.join(ROOT, path))
    for name, (restype, argtypes) in _lib.SIGNATURES.items():
        fn = getattr(lib, name); fn.restype, fn.argtypes = restype, argtypes
    for rep in range(3):
        counters = torch.zeros(5, dtype=torch.int64, device=dev)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        rc = lib.mxm_build_em_records(enc["maj"].data_ptr(), enc["lhit"].data_ptr(), enc["lmiss"].data_ptr(), enc["mk_ptr"].data_ptr(),
                                      enc["mk_hap"].data_ptr(), enc["mk_base"].data_ptr(), rp.data_ptr(), si.data_ptr(), ob.data_ptr(),
                                      0, rows, H, S, 0, 0, rec.data_ptr(), cap, rec_off.data_ptr(), ndist.data_ptr(), rowmax.data_ptr(),
                                      counters[0:2].data_ptr(), fallback.data_ptr(), counters[2:3].data_ptr(), current_stream())
        torch.cuda.synchronize()
        print("%-24s rc %d: %.2f ms (%d rows handed back)" % (os.path.basename(path), rc, (time.perf_counter() - t0) * 1e3, int(counters[2])), flush=True)
