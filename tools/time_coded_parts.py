#!/usr/bin/env python
"""
Where a pass of em_iter_coded_kernel goes: the same records timed (HIP events around mxm_em_iter_coded, median of 15)
  full        all rows: the main loop (wide rows passed over as empty steps) + the wide rows' loop
  no wide     n_wide = 0 (the wide rows' loop skipped: its share by difference)
    python tools/time_coded_parts.py [rows ...]
"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy
import torch
from mixemt_amd import _lib, phylotree, preprocess, synth

lib = _lib.load()
refseq = phylotree.load_rsrs()
phy = phylotree.load_build17(refseq)
haps = sorted(phy.hap_var)
tables = preprocess.HapVarTables.build(refseq, phy, haps)
dev = torch.device("cuda")
stream = torch.cuda.current_stream().cuda_stream
for rows in [int(x) for x in sys.argv[1:]] or [10000, 125000, 1000000]:
    row_ptr, site, obs, _ = synth.synth_rows(tables, len(refseq), 0, rows, seed=1)
    cm = preprocess.build_em_records_device(tables, row_ptr, site, obs)
    H = cm.n_haps
    wide = cm.wide_rows()
    props = torch.from_numpy(numpy.random.default_rng(3).dirichlet(numpy.full(H, 0.05))).to(dev)
    nbytes = lib.mxm_workspace_bytes(rows, H, 1)
    ws = torch.empty(nbytes // 8 + 1, dtype=torch.float64, device=dev)
    col = torch.zeros(H, dtype=torch.float64, device=dev)

    def coded(with_wide):
        return _lib.Coded(cm.rec.data_ptr(), cm.rec_off.data_ptr(), cm.ndist.data_ptr(), rows, None, 0, None, 0,
                          wide.data_ptr() if with_wide and wide.numel() else None, int(wide.numel()) if with_wide else 0)
    out = []
    for label, c in (("full", coded(True)), ("no wide", coded(False)), ("full", coded(True)), ("no wide", coded(False))):
        times = []
        for rep in range(18):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            rc = lib.mxm_em_iter_coded(ctypes.byref(c), None, props.data_ptr(), H, 1, None, col.data_ptr(), ws.data_ptr(), nbytes, stream)
            b.record()
            torch.cuda.synchronize()
            assert rc == 0
            if rep >= 3:
                times.append(a.elapsed_time(b) * 1e3)
        out.append("%s %.1f us" % (label, float(numpy.median(times))))
    print("%8d rows (%d wide, %d without a record): kernel + column reduce  " % (rows, wide.numel(), cm.rest_rows.numel()) + "   ".join(out), flush=True)
