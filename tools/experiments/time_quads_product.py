#!/usr/bin/env python
"""
The product's records iteration with and without the quad dictionary, same process, same records:
    python tools/experiments/time_quads_product.py [rows]
per-iteration step (mxm_em_iter_coded: row passes + column reduce) by events, interleaved rounds; then run_em to
convergence both ways (loop seconds, plan seconds, iteration count, proportions).
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy
import torch
from mixemt_amd import _lib, em, phylotree, preprocess, synth

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
refseq = phylotree.load_rsrs(); phy = phylotree.load_build17(refseq); haps = sorted(phy.hap_var)
tables = preprocess.HapVarTables.build(refseq, phy, haps)
H = len(haps)
row_ptr, site, obs, _ = synth.synth_rows(tables, len(refseq), 0, rows, seed=1)
cm = preprocess.build_em_records_device(tables, row_ptr, site, obs)
wts = torch.ones(rows, dtype=torch.float64, device="cuda")
props = torch.from_numpy(numpy.random.default_rng(1).dirichlet([1.0] * H)[None, :]).cuda()
lnp = props.log()
plans = {}
for label, mode in (("records", False), ("records + quads", True)):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    em.QUADS = mode
    plan = em.EmPlan(None, wts, records=cm)
    torch.cuda.synchronize()
    plans[label] = plan
    print("%-16s plan %.1f ms; quad rows %d, byte rows left %d, wide %d; bytes per pass %.3f GB %s"
          % (label, (time.perf_counter() - t0) * 1e3, plan.coded.n_quad_rows, plan.coded.n_byte_rows, plan.coded_wide,
             plan.coded_bytes / 1e9, getattr(plan, "quad_laps", None) or ""), flush=True)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
times = {k: [] for k in plans}
out = {k: torch.zeros((1, H), dtype=torch.float64, device="cuda") for k in plans}
state = em.new_state(1, "cuda")
for rnd in range(8):
    for k, plan in plans.items():
        plan.em_iter(props, lnp, state, out[k])
        torch.cuda.synchronize()
        ev[0].record()
        for _ in range(20):
            plan.em_iter(props, lnp, state, out[k])
        ev[1].record()
        torch.cuda.synchronize()
        times[k].append(ev[0].elapsed_time(ev[1]) / 20)
lib = _lib.load()
plan = plans["records + quads"]
for left in (8, 16, 22, 32, 48, 64, 96, 128, 192):
    lib.mxm_set_quad_left_grid(left)
    plan.em_iter(props, lnp, state, out["records + quads"])
    torch.cuda.synchronize()
    ev[0].record()
    for _ in range(20):
        plan.em_iter(props, lnp, state, out["records + quads"])
    ev[1].record()
    torch.cuda.synchronize()
    print("leftover pass on %3d workgroups: %.4f ms per iteration" % (left, ev[0].elapsed_time(ev[1]) / 20), flush=True)
lib.mxm_set_quad_left_grid(0)
a, b = (out[k][0].cpu().numpy() for k in plans)
print("column sums: max relative difference %.2e" % (numpy.abs(a - b).max() / numpy.abs(a).max()))
for k, t in times.items():
    t = sorted(t)
    print("%-16s mxm_em_iter_coded: median %.4f ms  min %.4f  max %.4f   (%.2f TB/s of what it reads)"
          % (k, t[len(t) // 2], t[0], t[-1], plans[k].coded_bytes / t[len(t) // 2] / 1e9))
del plans
args = argparse.Namespace(init_alpha=1.0, tolerance=1e-4, max_iter=10000, n_multi=1, verbose=False)
res = {}
for label, mode in (("records", False), ("records + quads", True), ("records", False), ("records + quads", True)):
    em.QUADS = mode
    numpy.random.seed(7)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = em.run_em_ex(None, wts, args, want_read_mix=False, records=cm)
    torch.cuda.synchronize()
    res[label] = r
    print("%-16s run_em %.1f ms: loop %.1f ms, %d iterations = %.4f ms per iteration"
          % (label, (time.perf_counter() - t0) * 1e3, r["loop_s"] * 1e3, r["iters"][0], r["loop_s"] * 1e3 / r["iters"][0]), flush=True)
print("same stop: %s; proportions: max difference %.2e" % (res["records"]["iters"] == res["records + quads"]["iters"],
                                                          numpy.abs(res["records"]["props"] - res["records + quads"]["props"]).max()))
