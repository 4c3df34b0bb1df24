#!/usr/bin/env python
"""Where the contributors stage of tools/run_pipeline.py spends its time right after run_em (round-5 diagnostic)."""
import argparse, ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy, torch
from mixemt_amd import _lib, assign, em, phylotree, preprocess, synth
from mixemt_amd._dev import current_stream
refseq = phylotree.load_rsrs(); phy = phylotree.load_build17(refseq); haps = sorted(phy.hap_var)
tables = preprocess.HapVarTables.build(refseq, phy, haps)
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
row_ptr, site, obs, _ = synth.synth_rows(tables, len(refseq), 0, rows, seed=1)
cm = preprocess.build_em_records_device(tables, row_ptr, site, obs)
wts = torch.ones(rows, dtype=torch.float64, device="cuda")
args = argparse.Namespace(init_alpha=1.0, tolerance=1e-4, max_iter=30, n_multi=1, verbose=False)
numpy.random.seed(7)
res = em.run_em_ex(None, wts, args, want_read_mix=False, records=cm)
torch.cuda.synchronize()
def lap(label, t0):
    torch.cuda.synchronize(); t1 = time.perf_counter(); print("%-40s %8.2f ms" % (label, (t1 - t0) * 1e3)); return t1
for rep in range(2):
    print("--- pass %d" % rep)
    t = time.perf_counter()
    lib = _lib.load(); dev = cm.rec.device
    lnp = torch.from_numpy(res["ln_theta_k"]).to(dev).contiguous(); props = torch.exp(lnp); t = lap("ln_theta upload + exp", t)
    best = torch.zeros(cm.n_rows, dtype=torch.int32, device=dev); votes = torch.zeros(cm.n_haps, dtype=torch.float64, device=dev); t = lap("best / votes buffers", t)
    nbytes = lib.mxm_workspace_bytes(cm.n_rows, cm.n_haps, 1); ws = torch.empty(nbytes // 8 + 1, dtype=torch.float64, device=dev); t = lap("workspace (%d MB)" % (nbytes >> 20), t)
    coded = cm.struct(); t = lap("descriptor (wide rows list)", t)
    _lib.check(lib.mxm_row_argmax_votes_coded(ctypes.byref(coded), cm.n_haps, 1, lnp.data_ptr(), props.data_ptr(), cm.rowmax.data_ptr(), 0, 0, 0, 0, 0,
                                              best.data_ptr(), votes.data_ptr(), ws.data_ptr(), nbytes, current_stream()), "votes"); t = lap("mxm_row_argmax_votes_coded", t)
    out = assign._contributors_on_device(best, votes, cm.n_haps, 10); t = lap("first seen + host order", t)
    del ws, best, votes
