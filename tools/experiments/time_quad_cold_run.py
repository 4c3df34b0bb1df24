import os, sys, time, argparse
sys.path.insert(0, "/root/repo" if os.path.isdir("/root/repo/mixemt_amd") else ".")
import numpy, torch
from mixemt_amd import _lib, em, phylotree, preprocess, synth
rows = 1000000
refseq = phylotree.load_rsrs(); phy = phylotree.load_build17(refseq); haps = sorted(phy.hap_var)
tables = preprocess.HapVarTables.build(refseq, phy, haps)
row_ptr, site, obs, _ = synth.synth_reads(tables, len(refseq), rows, seed=1)
torch.zeros(1, device="cuda"); _lib.load(); torch.cuda.synchronize()
cm = preprocess.build_em_records_device(tables, row_ptr, site, obs)
torch.cuda.synchronize()
real = em.EmPlan.attach_quads
def timed(self, mode=None, cap=None, min_rows=None):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ok = real(self, mode, cap, min_rows)
    torch.cuda.synchronize(); print("  attach_quads(%r) -> %s in %.1f ms" % (mode, ok, (time.perf_counter() - t0) * 1e3), flush=True)
    return ok
em.EmPlan.attach_quads = timed
real_loop = em.em_loop
def loop(*a, **k):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = real_loop(*a, **k)
    torch.cuda.synchronize(); print("  em_loop in %.1f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
    return out
em.em_loop = loop
real_collect = em.collect_result
def collect(*a, **k):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = real_collect(*a, **k)
    torch.cuda.synchronize(); print("  collect_result in %.1f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
    return out
em.collect_result = collect
args = argparse.Namespace(init_alpha=1.0, tolerance=1e-4, max_iter=10000, n_multi=1, verbose=False)
wts = numpy.ones(rows)
for q in (True, True, False):
    em.QUADS = q
    numpy.random.seed(7)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = em.run_em_ex(None, wts, args, want_read_mix=False, records=cm)
    torch.cuda.synchronize()
    print("QUADS=%s: run_em_ex %.1f ms (plan_s %.1f, loop_s %.1f)" % (q, (time.perf_counter() - t0) * 1e3, r["plan_s"] * 1e3, r["loop_s"] * 1e3), flush=True)
