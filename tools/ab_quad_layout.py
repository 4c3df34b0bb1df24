#!/usr/bin/env python
"""
Does WHERE the encoder puts the quad records matter to the pass that reads them?  The workgroup-per-row encoder
scatters consecutive rows over 2048 open chunks, the wave-per-row encoder writes a run of ~200 consecutive rows one after
the other.  One process, the same records: a dictionary from each encoder, mxm_em_iter_coded over each in turn
(one restart, HIP events, 3 x 20 steps, alternating).
    python tools/ab_quad_layout.py [rows]
"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy
import torch
from mixemt_amd import _lib, em, phylotree, preprocess, synth
from mixemt_amd._dev import current_stream

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
lib = _lib.load()
refseq = phylotree.load_rsrs()
phy = phylotree.load_build17(refseq)
haps = sorted(phy.hap_var)
tables = preprocess.HapVarTables.build(refseq, phy, haps)
H = len(haps)
row_ptr, site, obs, _ = synth.synth_rows(tables, len(refseq), 0, rows, seed=1)
cm = preprocess.build_em_records_device(tables, row_ptr, site, obs)
dev = cm.rec.device
plans = {}
for kind in (0, 1):
    _lib.check(lib.mxm_set_quad_encoder(kind), "mxm_set_quad_encoder")
    plans[kind] = em.EmPlan(None, torch.ones(rows, dtype=torch.float64, device=dev), n_runs=1, records=cm)
    assert plans[kind].attach_quads(True)
lib.mxm_set_quad_encoder(1)
numpy.random.seed(7)
props = torch.from_numpy(em.init_props(H, 1.0)[None, :].copy()).to(dev)
print("one MI355X; %d rows x %d haplogroups, %d rows with quads; one restart per pass" % (rows, H, plans[0].coded.n_quad_rows))


def step(plan, colsum, state):
    _lib.check(lib.mxm_em_iter_coded(ctypes.byref(plan.coded), plan.wts.data_ptr(), props.data_ptr(), H, 1, state.data_ptr(),
                                     colsum.data_ptr(), plan.ws.data_ptr(), plan.ws_bytes, current_stream()), "mxm_em_iter_coded")


sums = {}
for rnd in range(3):
    for kind in (0, 1):
        plan = plans[kind]
        colsum = torch.zeros((1, H), dtype=torch.float64, device=dev)
        state = em.new_state(1, dev)
        for _ in range(3):
            step(plan, colsum, state)
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(20):
            step(plan, colsum, state)
        ev1.record()
        torch.cuda.synchronize()
        sums[kind] = colsum.cpu().numpy()
        print("round %d: dictionary laid out by encoder %d (%s per row): %.4f ms per step"
              % (rnd, kind, "a wave" if kind else "a workgroup", ev0.elapsed_time(ev1) / 20))
print("column sums equal bit for bit: %s" % numpy.array_equal(sums[0], sums[1]))
