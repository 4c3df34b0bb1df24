#!/usr/bin/env python
"""
Round 6: mxm_em_iter_coded over records + quad dictionary with three restarts, full tiles of three through
em_iter_quad_batched_kernel against one restart per pass -- in-process, the same buffers, HIP events on the stream.

    python tools/time_quads_batched.py [rows] [lib.so ...]

Extra libraries (builds with other -D macros: `python -m mixemt_amd.build --out X.so -D NAME=VALUE`) are bound beside the
in-tree one and timed on the same plan.
"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy
import torch
from mixemt_amd import _lib, em, phylotree, preprocess, synth
from mixemt_amd._dev import current_stream

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
extra = sys.argv[2:]


def bind(path):
    lib = ctypes.CDLL(os.path.abspath(path))
    for name, (restype, argtypes) in _lib.SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = restype, argtypes
    return lib


refseq = phylotree.load_rsrs()
phy = phylotree.load_build17(refseq)
haps = sorted(phy.hap_var)
tables = preprocess.HapVarTables.build(refseq, phy, haps)
H = len(haps)
row_ptr, site, obs, _ = synth.synth_rows(tables, len(refseq), 0, rows, seed=1)
cm = preprocess.build_em_records_device(tables, row_ptr, site, obs)
dev = cm.rec.device
plan = em.EmPlan(None, torch.ones(rows, dtype=torch.float64, device=dev), n_runs=3, records=cm)
assert plan.attach_quads(True)
print("one MI355X; %d rows x %d haplogroups: %d with quads, %d byte-coded without, %d wide"
      % (rows, H, plan.coded.n_quad_rows, plan.coded.n_byte_rows, plan.coded.n_wide))
numpy.random.seed(7)
B = 3
props = torch.from_numpy(numpy.stack([em.init_props(H, 1.0) for _ in range(B)])).to(dev)
libs = [("in-tree", _lib.load())] + [(os.path.basename(p), bind(p)) for p in extra]


def step(lib, tile, colsum, state):
    lib.mxm_set_coded_batch_tile(tile)
    _lib.check(lib.mxm_em_iter_coded(ctypes.byref(plan.coded), plan.wts.data_ptr(), props.data_ptr(), H, B, state.data_ptr(),
                                     colsum.data_ptr(), plan.ws.data_ptr(), plan.ws_bytes, current_stream()), "mxm_em_iter_coded")


want = None
for name, lib in libs:
    for tile in (1, 3):
        colsum = torch.zeros((B, H), dtype=torch.float64, device=dev)
        state = em.new_state(B, dev)
        for _ in range(3):
            step(lib, tile, colsum, state)
        torch.cuda.synchronize()
        best = []
        for rep in range(3):
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
            for _ in range(10):
                step(lib, tile, colsum, state)
            ev1.record()
            torch.cuda.synchronize()
            best.append(ev0.elapsed_time(ev1) / 10)
        got = colsum.cpu().numpy()
        if want is None:
            want = got
        err = float((numpy.abs(got - want) / numpy.abs(want).max(axis=1, keepdims=True)).max())
        print("%-28s %d restart(s) per pass   %7.3f ms per iteration of 3 restarts (min of 3 x 10; all: %s) = %.3f per restart-iteration   sums within %.1e"
              % (name, tile, min(best), " ".join("%.3f" % b for b in best), min(best) / B, err))
        lib.mxm_reset_tuning()
