#!/usr/bin/env python
"""The FIRST EmPlan with a quad dictionary in a fresh process, stage by stage (the pipeline's order: tables -> records -> plan)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["MXM_PIPELINE_TIMING"] = "1"
import numpy
import torch
from mixemt_amd import _lib, em, phylotree, preprocess, synth

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
refseq = phylotree.load_rsrs(); phy = phylotree.load_build17(refseq); haps = sorted(phy.hap_var)
tables = preprocess.HapVarTables.build(refseq, phy, haps)
row_ptr, site, obs, _ = synth.synth_reads(tables, len(refseq), rows, seed=1)
torch.zeros(1, device="cuda"); _lib.load(); torch.cuda.synchronize()
cm = preprocess.build_em_records_device(tables, row_ptr, site, obs)
torch.cuda.synchronize()
wts = numpy.ones(rows)
for rep in range(3):
    t0 = time.perf_counter()
    em.QUADS = False
    plan = em.EmPlan(None, wts, records=cm)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    ok = plan.attach_quads(True)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("plan %.1f ms; attach %.1f ms: %s" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, plan.quad_laps), flush=True)
    del plan
