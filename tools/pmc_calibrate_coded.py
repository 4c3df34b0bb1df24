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

With --quads the same for the quad dictionary beside the records (em_iter_quad_coded_kernel, the kernel a plan of
3*10^5 rows or more runs): diag_stream_quads_kernel reads the quad records of the rows that have them through the quad
pass's own loads (8 B/lane of codes, 2 x 16 B/lane of table); the rows left to the records' pass (a few per cent) are
not in the bare reader, so its factor is applied to the whole launch and the JSON line says how many bytes that covers.
    python tools/pmc_calibrate_coded.py [rows] --quads > calibration_quads.json
--records: over the records as the build leaves them (bench.py --records) instead of the encoder's (--storage coded); the
two differ by a few per cent in bytes, and bench.py quotes a file only for the kind of records it ran on.
"""
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy
import torch
from mixemt_amd import _lib, em, phylotree, preprocess, synth

quads = "--quads" in sys.argv[1:]
from_build = "--records" in sys.argv[1:]        # the records as the build leaves them (bench.py --records), not as the encoder does
args = [a for a in sys.argv[1:] if a not in ("--quads", "--records")]
em.QUADS = False        # (attached below when asked for: the records' own pass otherwise -- em_iter_coded_kernel)

rows = int(args[0]) if args else 1000000
lib = _lib.load()
refseq = phylotree.load_rsrs()
phy = phylotree.load_build17(refseq)
haps = sorted(phy.hap_var)
tables = preprocess.HapVarTables.build(refseq, phy, haps)
row_ptr, site, obs, _ = synth.synth_rows(tables, len(refseq), 0, rows, seed=1)
wts = torch.ones(rows, dtype=torch.float64, device="cuda")
if from_build:                                  # bench.py --records: a signature's values stay apart where the encoder merges equal ones
    plan = em.EmPlan(None, wts, records=preprocess.build_em_records_device(tables, row_ptr, site, obs))
else:
    mat = preprocess.build_em_matrix_device(tables, row_ptr, site, obs)      # the same matrix bench.py --storage coded encodes
    plan = em.EmPlan(mat, wts, storage="coded")
H = plan.n_haps
nd = plan.coded_ndist.to(torch.int64)
ldc = (H + 7) // 8 * 8
read_bytes = int(((nd > 0) & (nd <= 256)).sum().item()) * ldc + int((nd > 256).sum().item()) * 2 * ldc + 8 * int(nd.sum().item())
bare, kernel = "diag_stream_coded_kernel", "em_iter_coded_kernel"
extra = {}
if quads:
    if not plan.attach_quads(True):
        sys.exit("no quad dictionary for these rows")
    bare, kernel = "diag_stream_quads_kernel", "em_iter_quad_coded_kernel"
    read_bytes = int(plan.quad_bytes)                                         # what the bare reader reads
    extra = {"quad_rows": int(plan.quad_rows_n), "kernel_bytes_per_pass": int(plan.coded_record_bytes),
             "bare_reader_covers": plan.quad_bytes / float(plan.coded_record_bytes)}
props = torch.from_numpy(numpy.random.default_rng(3).dirichlet(numpy.full(H, 0.05))).cuda()
col = torch.zeros(H, dtype=torch.float64, device="cuda")
sink = torch.zeros(4, dtype=torch.int32, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
times = {bare: [], kernel: []}
for rep in range(6):
    for name in times:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        if name == "diag_stream_quads_kernel":
            _lib.check(lib.mxm_diag_stream_quads(ctypes.byref(plan.coded), H, 2, sink.data_ptr(), stream), "mxm_diag_stream_quads")
        elif name.startswith("diag"):
            _lib.check(lib.mxm_diag_stream_coded(ctypes.byref(plan.coded), H, 2, sink.data_ptr(), stream), "mxm_diag_stream_coded")
        else:
            _lib.check(lib.mxm_em_iter_coded(ctypes.byref(plan.coded), wts.data_ptr(), props.data_ptr(), H, 1, None,
                                             col.data_ptr(), plan.ws.data_ptr(), plan.ws_bytes, stream), "mxm_em_iter_coded")
        b.record()
        torch.cuda.synchronize()
        if rep:
            times[name].append(a.elapsed_time(b))
sweep = {}
if quads:                                       # what the quad pass's loads reach alone, by workgroups per CU
    for wg in (1, 2, 3, 4, 6, 8):
        best = []
        for rep in range(4):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            _lib.check(lib.mxm_diag_stream_quads(ctypes.byref(plan.coded), H, wg, sink.data_ptr(), stream), "mxm_diag_stream_quads")
            b.record()
            torch.cuda.synchronize()
            if rep:
                best.append(a.elapsed_time(b))
        sweep[str(wg)] = round(read_bytes / float(numpy.median(best)) / 1e9, 3)
    extra["bare_reader_TBps_by_workgroups_per_cu"] = sweep
print(json.dumps({"rows": rows, "haps": H, "storage": "coded", "matrix": "records" if from_build else "encoded", "bare_reader": bare, "kernel": kernel,
                  "record_bytes_read_per_pass": read_bytes, **extra,
                  "rows_byte_coded": int(((nd > 0) & (nd <= 256)).sum().item()), "rows_16bit": int((nd > 256).sum().item()),
                  "rows_dense": int((nd == 0).sum().item()),
                  "bare_reader_ms": float(numpy.median(times[bare])),
                  "bare_reader_TBps": read_bytes / float(numpy.median(times[bare])) / 1e9,
                  "em_iter_coded_ms_with_column_reduce": float(numpy.median(times[kernel]))}))
