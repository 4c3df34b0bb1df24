#!/usr/bin/env python
"""
Counter calibration for the row-dictionary iteration.  FETCH_SIZE on gfx950 tallies the L2's memory-side
read requests at 64 B each; a 128-B request of a wide coalesced stream therefore shows as half its bytes
(MI355X_MICROARCH.md), and for em_iter_coded_kernel's mix of 4-B-per-lane code words and 8-B-per-lane table reads over
records that are not line aligned the factor is something else again.  Instead of guessing it, this script runs, on the
SAME records,
    diag_stream_coded_kernel   exactly the EM kernel's loads and nothing else (mxm_diag_stream_coded)
    em_iter_coded_kernel       the EM iteration (mxm_em_iter_coded)
under `rocprofv3 --pmc FETCH_SIZE`; tools/pmc_summary.py then takes
    factor  = bytes the records hold (printed here as one JSON line) / FETCH_SIZE of the bare reader
    traffic = FETCH_SIZE of em_iter_coded_kernel x factor
so the EM kernel's traffic is stated against a reader of exactly its bytes through exactly its access pattern.
    python tools/pmc_calibrate_coded.py [rows] > calibration.json
"""
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy
import torch
from mixemt_amd import _lib, em, phylotree, preprocess, synth

em.QUADS = False        # this tool measures the records' own pass (em_iter_coded_kernel): no quad dictionary beside them

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
lib = _lib.load()
refseq = phylotree.load_rsrs()
phy = phylotree.load_build17(refseq)
haps = sorted(phy.hap_var)
tables = preprocess.HapVarTables.build(refseq, phy, haps)
row_ptr, site, obs, _ = synth.synth_rows(tables, len(refseq), 0, rows, seed=1)
mat = preprocess.build_em_matrix_device(tables, row_ptr, site, obs)          # the same matrix bench.py --storage coded encodes
wts = torch.ones(rows, dtype=torch.float64, device="cuda")
plan = em.EmPlan(mat, wts, storage="coded")
H = plan.n_haps
nd = plan.coded_ndist.to(torch.int64)
ldc = (H + 7) // 8 * 8
read_bytes = int(((nd > 0) & (nd <= 256)).sum().item()) * ldc + int((nd > 256).sum().item()) * 2 * ldc + 8 * int(nd.sum().item())
props = torch.from_numpy(numpy.random.default_rng(3).dirichlet(numpy.full(H, 0.05))).cuda()
col = torch.zeros(H, dtype=torch.float64, device="cuda")
sink = torch.zeros(4, dtype=torch.int32, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
times = {"diag_stream_coded_kernel": [], "em_iter_coded_kernel": []}
for rep in range(6):
    for name in times:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        if name.startswith("diag"):
            _lib.check(lib.mxm_diag_stream_coded(ctypes.byref(plan.coded), H, 2, sink.data_ptr(), stream), "mxm_diag_stream_coded")
        else:
            _lib.check(lib.mxm_em_iter_coded(ctypes.byref(plan.coded), wts.data_ptr(), props.data_ptr(), H, 1, None,
                                             col.data_ptr(), plan.ws.data_ptr(), plan.ws_bytes, stream), "mxm_em_iter_coded")
        b.record()
        torch.cuda.synchronize()
        if rep:
            times[name].append(a.elapsed_time(b))
print(json.dumps({"rows": rows, "haps": H, "storage": "coded", "record_bytes_read_per_pass": read_bytes,
                  "rows_byte_coded": int(((nd > 0) & (nd <= 256)).sum().item()), "rows_16bit": int((nd > 256).sum().item()),
                  "rows_dense": int((nd == 0).sum().item()),
                  "bare_reader_ms": float(numpy.median(times["diag_stream_coded_kernel"])),
                  "bare_reader_TBps": read_bytes / float(numpy.median(times["diag_stream_coded_kernel"])) / 1e9,
                  "em_iter_coded_ms_with_column_reduce": float(numpy.median(times["em_iter_coded_kernel"]))}))
