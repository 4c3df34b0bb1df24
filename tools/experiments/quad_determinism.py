#!/usr/bin/env python
"""Are two quad dictionaries built from the same records interchangeable bit for bit?  python quad_determinism.py [rows]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy
import torch
from mixemt_amd import _lib, em, phylotree, preprocess, synth

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 400000
refseq = phylotree.load_rsrs(); phy = phylotree.load_build17(refseq); haps = sorted(phy.hap_var)
tables = preprocess.HapVarTables.build(refseq, phy, haps)
H = len(haps)
row_ptr, site, obs, _ = synth.synth_rows(tables, len(refseq), 0, rows, seed=1)
cm = preprocess.build_em_records_device(tables, row_ptr, site, obs)
wts = torch.ones(rows, dtype=torch.float64, device="cuda")
props = torch.from_numpy(numpy.random.default_rng(1).dirichlet([1.0] * H)[None, :]).cuda()


def sums(plan, reps=3):
    out = []
    for _ in range(reps):
        colsum = torch.zeros((1, H), dtype=torch.float64, device="cuda")
        plan.em_iter(props, props.log(), em.new_state(1, "cuda"), colsum)
        torch.cuda.synchronize()
        out.append(colsum[0].cpu().numpy())
    return out


plans = []
for i in range(2):
    plan = em.EmPlan(None, wts, records=cm)
    plan.attach_quads(True)
    plans.append(plan)
a, b = sums(plans[0]), sums(plans[1])
print("rows %d: quad rows %d / %d, byte rows %d / %d" % (rows, plans[0].coded.n_quad_rows, plans[1].coded.n_quad_rows,
                                                         plans[0].coded.n_byte_rows, plans[1].coded.n_byte_rows))
print("same plan, reruns equal:", all(numpy.array_equal(a[0], x) for x in a), all(numpy.array_equal(b[0], x) for x in b))
print("two plans equal:", numpy.array_equal(a[0], b[0]), "max rel diff %.3e" % (numpy.abs(a[0] - b[0]).max() / numpy.abs(a[0]).max()))
nq0, nq1 = plans[0]._quad_keep[2].cpu().numpy(), plans[1]._quad_keep[2].cpu().numpy()
print("nquad equal:", numpy.array_equal(nq0, nq1), "differing rows:", int((nq0 != nq1).sum()))
q0, o0 = plans[0]._quad_keep[0], plans[0]._quad_keep[1].cpu().numpy()
q1, o1 = plans[1]._quad_keep[0], plans[1]._quad_keep[1].cpu().numpy()
bad = 0
rng = numpy.random.default_rng(3)
for r in rng.choice(numpy.flatnonzero(nq0 > 0), size=min(3000, int((nq0 > 0).sum())), replace=False):
    n = 2048 + 32 * int(nq0[r])
    x = q0[int(o0[r]):int(o0[r]) + n].cpu().numpy()
    y = q1[int(o1[r]):int(o1[r]) + n].cpu().numpy()
    bad += int(not numpy.array_equal(x, y))
print("records differing among 3000 sampled rows:", bad)
plain = em.EmPlan(None, wts, records=cm)
c = sums(plain)
print("no quads: reruns equal:", all(numpy.array_equal(c[0], x) for x in c), "vs quads max rel diff %.3e" % (numpy.abs(a[0] - c[0]).max() / numpy.abs(c[0]).max()))
