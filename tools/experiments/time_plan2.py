"""Round 6: where a cold process spends the plan stage (EmPlan over records + quad dictionary) -- host time stamps of
its parts without synchronising between them (profiles/r06/experiments.md; DESIGN section 6: a 25 ms stall at the plan's
synchronize on some boxes' first process, absent under rocprofv3).  python tools/experiments/time_plan2.py
"""
import os, sys, time, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy, torch
from mixemt_amd import _lib, em, phylotree, preprocess, synth
refseq = phylotree.load_rsrs(); phy = phylotree.load_build17(refseq); haps = sorted(phy.hap_var)
tables = preprocess.HapVarTables.build(refseq, phy, haps)
rows = 1000000
row_ptr, site, obs, _ = synth.synth_rows(tables, len(refseq), 0, rows, seed=1, pairs="--pairs" in sys.argv)
torch.zeros(1, device="cuda"); _lib.load(); torch.cuda.synchronize()
tables.sparse_device(); tables.lut_device(); torch.cuda.synchronize()
T0 = time.perf_counter()
def stamp(msg):
    sys.stderr.write("%8.1f ms  %s\n" % ((time.perf_counter() - T0) * 1e3, msg))
def wrap(obj, name):
    fn = getattr(obj, name)
    def inner(*a, **k):
        stamp("-> " + name)
        r = fn(*a, **k)
        stamp("<- " + name)
        return r
    setattr(obj, name, inner)
wrap(em, "_workspace"); wrap(em.EmPlan, "attach_quads"); wrap(preprocess.CodedMatrix, "ndist_host"); wrap(preprocess.CodedMatrix, "wide_rows")
wrap(em, "device_empty"); wrap(em.EmPlan, "encode") if hasattr(em.EmPlan, "encode") else None
wts = torch.ones(rows, dtype=torch.float64, device="cuda")
stamp("build records")
cm = preprocess.build_em_records_device(tables, row_ptr, site, obs)
stamp("records returned"); torch.cuda.synchronize(); stamp("records drained")
plan = em.EmPlan(None, wts, n_runs=1, records=cm)
stamp("plan returned"); torch.cuda.synchronize(); stamp("plan drained")
