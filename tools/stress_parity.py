#!/usr/bin/env python
"""
Randomised parity stress on the GPU box (not part of the suite: minutes of oracle time):
run_em end to end -- one-launch loop, per-iteration kernels, chunked launches -- against the oracle on
random shapes, weights, -inf patterns and restart counts; build kernels against the C oracle on random
row sets.  Every case prints one line; the script exits non-zero on the first mismatch.

    python tools/stress_parity.py [--cases N] [--seed S] [--budget SECONDS]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy
import torch
from mixemt_amd import _lib, em, phylotree, preprocess, synth
from oracle import c_oracle, em_oracle

ap = argparse.ArgumentParser()
ap.add_argument("--cases", type=int, default=60)
ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--budget", type=float, default=600.0)
opts = ap.parse_args()
lib = _lib.load()
rng = numpy.random.default_rng(opts.seed)
t_start = time.time()
fails = 0


def ns(**kw):
    a = argparse.Namespace(init_alpha=1.0, tolerance=1e-4, max_iter=400, n_multi=1, verbose=False)
    for k, v in kw.items():
        setattr(a, k, v)
    return a


widths = [1, 1, 2, 3, 4, 5, 8, 9, 16, 17, 31, 32, 33, 63, 64,        # the narrow (log-space) kernels: the refinement EM's shapes
          65, 66, 67, 127, 128, 129, 255, 256, 511, 512, 513, 1000, 1023, 1024, 1025, 2047, 2048, 2049,
          3000, 4095, 4096, 4097, 5407, 5408, 5409, 6143, 6144, 6145, 7000, 8191, 8192,
          8193, 9001, 9600, 9601, 10241, 12000]                    # round 5: beyond the streaming / log-space / finalize kernels' tables
for case in range(opts.cases):
    if time.time() - t_start > opts.budget:
        print("budget used up after %d cases" % case)
        break
    n_haps = int(rng.choice(widths))
    n_rows = int(rng.choice([1, 2, 3, 7, 64, 255, 256, 257, 300, 511, 513, 768, 769, 1000, 1500, 2500]))
    n_rows = min(n_rows, max(1, int(4.0e6 // n_haps)))           # keep the oracle affordable
    n_multi = int(rng.choice([1, 1, 1, 2, 3, 5]))
    mat = rng.normal(size=(n_rows, n_haps)) * rng.choice([1.0, 3.0, 8.0]) - 10.0
    hot = rng.integers(0, min(n_haps, 9), size=n_rows)
    mat[numpy.arange(n_rows), hot] += rng.choice([5.0, 12.0, 30.0])
    if rng.random() < 0.5:
        mat[rng.random(mat.shape) < rng.choice([0.001, 0.01, 0.2])] = -numpy.inf
        mat[numpy.arange(n_rows), hot] = numpy.where(numpy.isfinite(mat[numpy.arange(n_rows), hot]),
                                                     mat[numpy.arange(n_rows), hot], -3.0)
    wts = rng.integers(0, 6, size=n_rows).astype(numpy.float64)
    wts[rng.integers(0, n_rows)] = 3.0
    if rng.random() < 0.3:
        wts = wts * rng.random(n_rows)                           # fractional weights
        wts[0] = 1.5
    args = ns(n_multi=n_multi, max_iter=int(rng.choice([5, 60, 400])))
    seed = int(rng.integers(1, 1 << 30))
    trace = []
    numpy.random.seed(seed)
    with numpy.errstate(all="ignore"):
        want_props, want_mix = em_oracle.run_em(mat, wts, args, trace=trace)
    want_iters = [t["iters"] for t in trace]
    line = "case %3d: %5d x %5d, n_multi %d, max_iter %3d, iterations %s" % (case, n_rows, n_haps, n_multi,
                                                                             args.max_iter, want_iters)
    seen = {}
    for label, mode, chunk in (("one-launch", 1, 0), ("rows-split", 2, 0), ("kernels", 0, 0),
                               ("chunks", 1, int(rng.integers(1, 9))), ("rows-chunks", 2, int(rng.integers(1, 9)))):
        lib.mxm_set_loop_fused(mode, chunk)
        numpy.random.seed(seed)
        res = em.run_em_ex(mat, wts, args)
        got_mix = res["read_mix"].cpu().numpy()
        ok = res["iters"] == want_iters
        ok = ok and float(numpy.nanmax(numpy.abs(res["props"] - want_props))) < 1e-9
        ok = ok and numpy.array_equal(numpy.isfinite(got_mix), numpy.isfinite(want_mix))
        with numpy.errstate(all="ignore"):
            ok = ok and float(numpy.nanmax(numpy.abs(numpy.exp(got_mix) - numpy.exp(want_mix)))) < 1e-9
        seen[label] = res
        # chunked launches of one form must reproduce its single launch bit for bit, the reported L1 included
        twin = {"chunks": "one-launch", "rows-chunks": "rows-split"}.get(label)
        if twin is not None:
            same = numpy.array_equal(res["run_props"], seen[twin]["run_props"], equal_nan=True)
            same = same and all((a == b) or (a != a and b != b) for a, b in zip(res["l1"], seen[twin]["l1"]))
            ok = ok and same
        if not ok:
            fails += 1
            line += "  %s MISMATCH (iters %s, dprops %.2e)" % (label, res["iters"],
                                                              float(numpy.nanmax(numpy.abs(res["props"] - want_props))))
    # the same run over row-dictionary records where the shape allows them (even H in 66..8192): a random row holds H
    # distinct values -- byte codes up to 256 columns, 16-bit codes up to 1024, dense beyond -- through the one-launch
    # loop over records and through the per-iteration kernels
    if n_haps % 2 == 0 and 66 <= n_haps <= 8192:
        for label, mode, chunk in (("records-one-launch", 1, 0), ("records-chunks", 1, int(rng.integers(1, 9))),
                                   ("records-kernels", 0, 0)):
            lib.mxm_set_loop_fused(mode, chunk)
            numpy.random.seed(seed)
            res = em.run_em_ex(mat, wts, args, storage="coded")
            got_mix = res["read_mix"].cpu().numpy()
            ok = res["iters"] == want_iters and res["storage"] == "coded"
            ok = ok and float(numpy.nanmax(numpy.abs(res["props"] - want_props))) < 1e-9
            ok = ok and numpy.array_equal(numpy.isfinite(got_mix), numpy.isfinite(want_mix))
            with numpy.errstate(all="ignore"):
                ok = ok and float(numpy.nanmax(numpy.abs(numpy.exp(got_mix) - numpy.exp(want_mix)))) < 1e-9
            if not ok:
                fails += 1
                line += "  %s MISMATCH (iters %s, dprops %.2e)" % (label, res["iters"],
                                                                  float(numpy.nanmax(numpy.abs(res["props"] - want_props))))
        line += "  [+ records x3]"
    print(line + ("" if "MISMATCH" in line else "  ok"))
    sys.stdout.flush()
lib.mxm_set_loop_fused(-1, 0)

# ---- build kernels on random row sets of Build 17 ------------------------------------------------------
refseq = phylotree.load_rsrs()
phy = phylotree.load_build17(refseq)
haps = sorted(phy.hap_var)
for case in range(20):
    if time.time() - t_start > opts.budget * 1.3:
        break
    n_cols = int(rng.choice([1, 3, 4, 5, 63, 64, 65, 66, 130, 1023, 1024, 1025, 2050, 3000, 5408]))
    first = int(rng.integers(0, len(haps) - n_cols + 1))
    sub = haps[first:first + n_cols]
    tables = preprocess.HapVarTables.build(refseq, phy, sub)
    n_rows = int(rng.choice([1, 9, 130, 1000, 5000]))
    read_len = int(rng.choice([30, 150, 150, 600, 3000]))
    if rng.random() < 0.3:                                    # round 6: merged mates (two thirds of the rows above 64 sites)
        read_len = 0
        row_ptr, site, obs, _ = synth.synth_pairs(tables, len(refseq), n_rows, seed=int(rng.integers(1, 1 << 30)),
                                                  contrib=(0, n_cols // 2, n_cols - 1))
    else:
        row_ptr, site, obs, _ = synth.synth_reads(tables, len(refseq), n_rows, seed=int(rng.integers(1, 1 << 30)),
                                                  read_len=read_len, contrib=(0, n_cols // 2, n_cols - 1))
    obs = obs.copy()
    obs[rng.random(obs.size) < 0.02] = ord("N")
    want = c_oracle.build_em_matrix(tables.expected, tables.lhit, tables.lmiss, row_ptr, site, obs, n_cols)
    line = "build %2d: %5d rows x %4d columns, read length %4d (up to %d sites per row):" % (
        case, n_rows, n_cols, read_len, int(numpy.diff(row_ptr).max()))
    for kernel in ("lut", "bytes", "sparse"):
        if kernel == "sparse" and tables.lut() is None:
            continue
        for sort_rows in ((False, True) if kernel == "lut" else (False,)):
            got = preprocess.build_em_matrix_device(tables, row_ptr, site, obs, kernel=kernel,
                                                    sort_rows=sort_rows).cpu().numpy()
            if not numpy.array_equal(got, want):
                fails += 1
                line += "  %s%s MISMATCH" % (kernel, "+sort" if sort_rows else "")
    # row-dictionary forms on the same rows: EM from the encoded dense matrix and from records built without it,
    # against the oracle's run_em on the reference-exact matrix (even widths in the streaming kernel's range)
    if n_cols % 2 == 0 and 66 <= n_cols and n_rows * n_cols <= 6.0e6 and tables.lut() is not None:
        wts = rng.integers(1, 5, size=n_rows).astype(numpy.float64)
        args = ns(max_iter=int(rng.choice([5, 40])), n_multi=int(rng.choice([1, 1, 3, 4])))     # (3, 4: a shared pass of three + one)
        seed = int(rng.integers(1, 1 << 30))
        trace = []
        numpy.random.seed(seed)
        want_props, want_mix = em_oracle.run_em(want, wts, args, trace=trace)
        cm = preprocess.build_em_records_device(tables, row_ptr, site, obs)
        for label, kw in (("coded", dict(read_hap_mat=want, storage="coded")), ("records", dict(read_hap_mat=None, records=cm)),
                          ("records+quads", dict(read_hap_mat=None, records=cm))):
            # round 5: the same from records with a quad dictionary beside them, through the per-iteration kernels that
            # read it (forced: at these sizes "auto" builds none and the one-launch loop would run)
            quads = label.endswith("quads")
            em.QUADS = True if quads else False
            lib.mxm_set_loop_fused(0 if quads else -1, 0)
            numpy.random.seed(seed)
            res = em.run_em_ex(kw.pop("read_hap_mat"), wts, args, **kw)
            em.QUADS = "auto"
            lib.mxm_reset_tuning()
            got_mix = res["read_mix"].cpu().numpy()
            ok = res["iters"] == [t["iters"] for t in trace] and res["storage"] == "coded"
            ok = ok and float(numpy.abs(res["props"] - want_props).max()) < 1e-9
            ok = ok and float(numpy.abs(numpy.exp(got_mix) - numpy.exp(want_mix)).max()) < 1e-9
            ok = ok and numpy.array_equal(got_mix.argmax(axis=1), want_mix.argmax(axis=1))
            if not ok:
                fails += 1
                line += "  %s MISMATCH" % label
        line += "  [coded + records (+ quads) EM, %d rows dense]" % int(cm.rest_rows.numel())
    print(line + ("" if "MISMATCH" in line else "  bit-exact"))
    sys.stdout.flush()
print("%d mismatches" % fails)
sys.exit(1 if fails else 0)
