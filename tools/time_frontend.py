#!/usr/bin/env python
"""
Front-end timing: N synthetic fragments (synth-aln-v1) as columns through
the library's batched encoder (mxm_aln_encode) with 1 / 4 / 8 / 16 host threads, the adapter that reads pysam-like
OBJECTS into columns, and -- on a sample -- the object-by-object path the reference takes (preprocess.py:99-174).
CPU only (the encoder is a host function): runs in the build container and on the GPU box alike.

    python tools/time_frontend.py [--fragments 1000000] [--sample 20000] [--threads 1,4,8,16]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))

import numpy

from mixemt_amd import alignments, phylotree, preprocess, synth


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fragments", type=int, default=1000000)
    ap.add_argument("--sample", type=int, default=20000)
    ap.add_argument("--threads", default="1,4,8,16")
    opts = ap.parse_args()
    refseq = phylotree.load_rsrs()
    phy = phylotree.load_build17(refseq)
    tables = preprocess.HapVarTables.build(refseq, phy, sorted(phy.hap_var))
    var_pos = phy.get_variant_pos()
    print("host: %d CPUs visible" % (os.cpu_count() or 0))
    t0 = time.perf_counter()
    cols = synth.synth_alignments(tables, refseq, opts.fragments, seed=1)
    print("synth-aln-v1: %d fragments -> %d alignments, %d bases (generated in %.1f s)"
          % (cols.n_frag, len(cols), len(cols.seq), time.perf_counter() - t0))
    for nt in [int(x) for x in opts.threads.split(",")]:
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            enc = alignments.encode_alignments(cols, var_pos, len(refseq), 30, 30, n_threads=nt)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        print("batched encoder, %2d thread(s): %.3f s = %.2f s per 10^6 alignments  (%d fragments with a site -> %d rows, "
              "%d observations, max weight %d)" % (nt, best, best / len(cols) * 1e6, enc.n_fragments, enc.n_rows,
                                                  len(enc.site), int(enc.weights.max())))
    from _fake_aln import from_columns
    small = synth.synth_alignments(tables, refseq, opts.sample, seed=2)
    objs = from_columns(small)
    t0 = time.perf_counter()
    back = alignments.AlignmentColumns.from_alignments(objs)
    t_adapter = time.perf_counter() - t0
    t0 = time.perf_counter()
    enc = alignments.encode_alignments(back, var_pos, len(refseq), 30, 30)
    t_enc = time.perf_counter() - t0
    t0 = time.perf_counter()
    obs = preprocess.process_reads(objs, var_pos, 30, 30)
    t_proc = time.perf_counter() - t0
    t0 = time.perf_counter()
    sigs = preprocess.reduce_reads(obs)
    rows = sorted(s for s in sigs if s)
    t_red = time.perf_counter() - t0
    same = enc.signatures() == rows and enc.read_ids == [sigs[r] for r in rows]
    n = len(objs)
    print("sample of %d alignment OBJECTS: adapter (objects -> columns) %.3f s = %.1f s per 10^6; encoder %.3f s; "
          "object-by-object path: process_reads %.2f s + reduce_reads / sort %.2f s = %.1f s per 10^6 alignments; "
          "same rows, weights and id lists: %s" % (n, t_adapter, t_adapter / n * 1e6, t_enc, t_proc, t_red,
                                                   (t_proc + t_red) / n * 1e6, same))
    return 0 if same else 1


if __name__ == "__main__":
    sys.exit(main())
