#!/usr/bin/env python
"""What does the FIRST quad build of a process cost, and which part?  python tools/experiments/time_quad_build.py [rows]"""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy
import torch
from mixemt_amd import _lib, em, phylotree, preprocess, synth
from mixemt_amd._dev import current_stream

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
refseq = phylotree.load_rsrs(); phy = phylotree.load_build17(refseq); haps = sorted(phy.hap_var)
tables = preprocess.HapVarTables.build(refseq, phy, haps)
H = len(haps)
row_ptr, site, obs, _ = synth.synth_rows(tables, len(refseq), 0, rows, seed=1)
cm = preprocess.build_em_records_device(tables, row_ptr, site, obs)
lib = _lib.load()
torch.cuda.synchronize()


def lap(t0, what):
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    print("  %-40s %8.2f ms" % (what, (t1 - t0) * 1e3), flush=True)
    return time.perf_counter()


coded = cm.struct()
for rep in range(3):
    print("build %d" % rep)
    t = time.perf_counter()
    cap = rows * (2048 + 32 * 112)
    qrec = torch.empty(cap, dtype=torch.uint8, device="cuda")
    t = lap(t, "torch.empty of %.1f GB" % (cap / 1e9))
    qoff = torch.empty(rows, dtype=torch.int64, device="cuda")
    nquad = torch.empty(rows, dtype=torch.int32, device="cuda")
    stats = torch.empty(2, dtype=torch.int64, device="cuda")
    t = lap(t, "small tensors")
    _lib.check(lib.mxm_build_quads(ctypes.byref(coded), H, qrec.data_ptr(), cap, qoff.data_ptr(), nquad.data_ptr(), stats.data_ptr(),
                                   current_stream()), "mxm_build_quads")
    t = lap(t, "mxm_build_quads (kernel)")
    used = stats.cpu()
    t = lap(t, "stats to the host")
    nq = nquad.cpu().numpy()
    t = lap(t, "nquad to the host")
    nd = cm.ndist_host()
    quad_rows = numpy.flatnonzero(nq > 0)
    byte_rows = numpy.flatnonzero((nd > 0) & (nd <= 256) & (nq == 0))
    t = lap(t, "lists (numpy)")
    a = torch.from_numpy(quad_rows).to("cuda"); b = torch.from_numpy(byte_rows).to("cuda")
    t = lap(t, "lists to the device")
    del qrec, qoff, nquad
print("EmPlan.attach_quads(True), three plans:")
os.environ["MXM_PIPELINE_TIMING"] = "1"
wts = torch.ones(rows, dtype=torch.float64, device="cuda")
for rep in range(3):
    t0 = time.perf_counter()
    plan = em.EmPlan(None, wts, records=cm)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    plan.attach_quads(True)
    torch.cuda.synchronize()
    print("  plan %.1f ms, attach %.1f ms: %s" % ((t1 - t0) * 1e3, (time.perf_counter() - t1) * 1e3, plan.quad_laps), flush=True)
    del plan
