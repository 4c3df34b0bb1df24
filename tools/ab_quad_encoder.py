#!/usr/bin/env python
"""
The quad dictionary's two encoders on the same records (mxm_set_quad_encoder: 0 = a workgroup per row, 1 = a wave per row):
ms per mxm_build_quads call by HIP events (median of 5 after one warm-up), rows with quads, and every row's record bytes
compared between the two.
    python tools/ab_quad_encoder.py [--reads N] [--pairs]
"""
import argparse
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy
import torch
from mixemt_amd import _lib, phylotree, preprocess, synth
from mixemt_amd._dev import current_stream

ap = argparse.ArgumentParser()
ap.add_argument("--reads", type=int, default=1000000)
ap.add_argument("--pairs", action="store_true")
ap.add_argument("--compare-rows", type=int, default=20000)
opt = ap.parse_args()
lib = _lib.load()
refseq = phylotree.load_rsrs()
phy = phylotree.load_build17(refseq)
haps = sorted(phy.hap_var)
tables = preprocess.HapVarTables.build(refseq, phy, haps)
row_ptr, site, obs, _ = synth.synth_rows(tables, len(refseq), 0, opt.reads, seed=1, pairs=opt.pairs)
cm = preprocess.build_em_records_device(tables, row_ptr, site, obs)
dev = cm.rec.device
coded = cm.struct()
R, H = cm.n_rows, cm.n_haps
nd = cm.ndist_host()
n_byte = int(((nd > 0) & (nd <= 256)).sum())
cap = n_byte * (2048 + 32 * 224) + 5120 * 65536 + (1 << 20)
print("one MI355X; %d %s x %d haplogroups, %d byte-coded rows; mxm_build_quads alone, HIP events"
      % (R, "synth-pe-v1 fragments" if opt.pairs else "synth-v1 reads", H, n_byte))
kept = {}
for kind in (0, 1, 0, 1):
    _lib.check(lib.mxm_set_quad_encoder(kind), "mxm_set_quad_encoder")
    qrec = torch.empty(cap, dtype=torch.uint8, device=dev)
    qoff = torch.empty(R, dtype=torch.int64, device=dev)
    nquad = torch.empty(R, dtype=torch.int32, device=dev)
    stats = torch.empty(2, dtype=torch.int64, device=dev)
    times = []
    for rep in range(6):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        _lib.check(lib.mxm_build_quads(ctypes.byref(coded), H, qrec.data_ptr(), qrec.numel(), qoff.data_ptr(), nquad.data_ptr(),
                                       stats.data_ptr(), current_stream()), "mxm_build_quads")
        b.record()
        torch.cuda.synchronize()
        if rep:
            times.append(a.elapsed_time(b))
    used, left = (int(v) for v in stats.cpu())
    assert used <= cap, (used, cap)
    nq = nquad.cpu().numpy()
    print("encoder %d (%s per row): %6.2f ms (min %.2f)  rows with quads %d, left without %d, %.2f GB reserved"
          % (kind, "a wave" if kind else "a workgroup", float(numpy.median(times)), min(times), int((nq > 0).sum()), left, used / 1e9))
    if kind not in kept:
        rows = numpy.flatnonzero(nq > 0)[:opt.compare_rows]
        off = qoff.cpu().numpy()
        q = qrec.cpu().numpy()
        kept[kind] = (nq, {int(r): q[int(off[r]):int(off[r]) + 2048 + 32 * int(nq[r])].copy() for r in rows})
    del qrec
assert numpy.array_equal(kept[0][0], kept[1][0]), "nquad differs"
bad = sum(0 if numpy.array_equal(kept[0][1][r], kept[1][1][r]) else 1 for r in kept[0][1])
print("nquad equal for all %d rows; record bytes of the first %d rows with quads: %d differ" % (R, len(kept[0][1]), bad))
assert bad == 0
