#!/usr/bin/env python3
"""
Time the library's BAM reader (alignments.read_bam -> mxm_bam_read) on synthetic alignments written as BAM by
tests/_bam_writer.py, and the whole host front end from the file: read_bam + encode_alignments.
    python tools/time_bam_reader.py --fragments 1000000 --threads 1,4,16
Host only (no GPU work).
"""
import argparse
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fragments", type=int, default=200000)
    ap.add_argument("--threads", default="1,4,16")
    ap.add_argument("--level", type=int, default=6)
    ap.add_argument("--seed", type=int, default=1)
    opts = ap.parse_args()
    import _bam_writer as bw
    from mixemt_amd import alignments, phylotree, preprocess, synth
    refseq = phylotree.load_rsrs()
    phy = phylotree.load_build17(refseq)
    haps = sorted(phy.hap_var)
    tables = preprocess.HapVarTables.build(refseq, phy, haps)
    t0 = time.perf_counter()
    cols = synth.synth_alignments(tables, refseq, opts.fragments, seed=opts.seed)
    t1 = time.perf_counter()
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "synth.bam")
        stream = bw.write_bam(path, cols, level=opts.level)
        t2 = time.perf_counter()
        print("%d alignments of %d fragments: generated in %.1f s, written in %.1f s: %.1f MB of BAM (%.1f MB inflated)"
              % (len(cols), opts.fragments, t1 - t0, t2 - t1, os.path.getsize(path) / 1e6, len(stream) / 1e6), flush=True)
        del stream
        for th in [int(x) for x in opts.threads.split(",")]:
            best = None
            for _ in range(3):
                a = time.perf_counter()
                got = alignments.read_bam(path, n_threads=th)
                b = time.perf_counter()
                best = b - a if best is None else min(best, b - a)
            a = time.perf_counter()
            enc = alignments.encode_alignments(got, tables.sites, len(refseq), 30, 30, n_threads=th)
            b = time.perf_counter()
            print("threads %2d: read_bam %.3f s (%.2f s per 10^6 alignments, %.0f MB/s of file); encode %.3f s; %d rows"
                  % (th, best, best * 1e6 / max(len(got), 1), os.path.getsize(path) / 1e6 / best, b - a, enc.n_rows), flush=True)
        assert len(got) == len(cols)


if __name__ == "__main__":
    main()
