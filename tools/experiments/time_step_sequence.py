#!/usr/bin/env python
"""Kernel time of the records' row pass step by step (HIP events around the launch): does it depend on the iteration?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy, torch
from mixemt_amd import _lib, em, phylotree, preprocess, synth
rows = 1000000
refseq = phylotree.load_rsrs(); phy = phylotree.load_build17(refseq); haps = sorted(phy.hap_var)
tables = preprocess.HapVarTables.build(refseq, phy, haps)
H = len(haps)
row_ptr, site, obs, _ = synth.synth_rows(tables, len(refseq), 0, rows, seed=1)
cm = preprocess.build_em_records_device(tables, row_ptr, site, obs)
wts = torch.ones(rows, dtype=torch.float64, device="cuda")
lib = _lib.load()
for mode in (True, False):
    em.QUADS = mode
    plan = em.EmPlan(None, wts, records=cm)
    numpy.random.seed(7)
    init = em.init_props(H, 1.0)[None, :]
    props = torch.from_numpy(init).cuda(); ln_a = props.log(); ln_b = ln_a.clone()
    colsum = torch.zeros_like(props); state = em.new_state(1, "cuda")
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(40)]
    for a, b in evs:                                   # (a torch event has no handle before its first record)
        a.record(); b.record()
    for variant in ("em_iter + finalize (props move)", "em_iter only (props fixed)"):
        for i in range(40):
            lib.mxm_set_timing_events(evs[i][0].cuda_event, evs[i][1].cuda_event)
            plan.em_iter(props, ln_a, state, colsum)
            lib.mxm_set_timing_events(None, None)
            if variant.startswith("em_iter + finalize"):
                plan.finalize(colsum, ln_a, ln_b, props, state, 0.0, 1 << 30)
        torch.cuda.synchronize()
        ks = [a.elapsed_time(b) for a, b in evs]
        print("quads=%s %s: %s" % (mode, variant, " ".join("%.2f" % k for k in ks)), flush=True)
