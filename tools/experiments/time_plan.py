import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy, torch
from mixemt_amd import _lib, em, phylotree, preprocess, synth
refseq = phylotree.load_rsrs(); phy = phylotree.load_build17(refseq); haps = sorted(phy.hap_var)
tables = preprocess.HapVarTables.build(refseq, phy, haps)
rows = 1000000
row_ptr, site, obs, _ = synth.synth_rows(tables, len(refseq), 0, rows, seed=1)
lib = _lib.load()
wts = torch.ones(rows, dtype=torch.float64, device="cuda")
for long_rows in (1, 0, 1, 0):
    lib.mxm_set_sparse_long_rows(long_rows)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    cm = preprocess.build_em_records_device(tables, row_ptr, site, obs)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    plan = em.EmPlan(None, wts, n_runs=1, records=cm)
    t3 = time.perf_counter(); torch.cuda.synchronize(); t4 = time.perf_counter()
    nd = cm.ndist_host()
    print("long rows %d: records %.1f ms (+%.1f to drain), plan %.1f ms (+%.1f to drain); wide %d, byte %d, rest %d; quads %d"
          % (long_rows, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, (nd > 256).sum(), ((nd > 0) & (nd <= 256)).sum(), cm.rest_rows.numel(), plan.coded.n_quad_rows))
    del plan, cm
    torch.cuda.empty_cache()
