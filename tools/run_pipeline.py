#!/usr/bin/env python
"""
The in-scope part of bin/mixemt's process_and_report (bin/mixemt:248-342) on
synthetic reads, device-resident end to end:

    CSR observations -> build_em_matrix_device -> run_em -> contributors from read votes
    -> refinement EM on the contributor columns -> read assignment -> contributor table

    python tools/run_pipeline.py [--reads N] [--seed S] [--multi M] [--dense [--storage f64|f32|coded|auto]] [--alignments]
                                 [--pairs | --read-len L]

--alignments (round 5): start one step earlier, from ALIGNMENTS -- N synthetic fragments (synth-aln-v1: mates, indels,
clips, low qualities, duplicates) as columns -> the library's batched front end (alignments.encode_alignments =
process_reads + reduce_reads + the row order of build_em_input, preprocess.py:99-174, :218-225) -> CSR + weights + id
groups, and the stage is timed beside the others.

Default (round 3): the build leaves the matrix as row-dictionary records -- no dense matrix, no posterior matrix; the
contributors, the vote table and the reduced matrix for the refinement come from the records.  --dense takes the
reference's own data flow (dense matrix -> run_em -> posterior matrix -> votes) on the device.
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy
import torch

from mixemt_amd import _lib, assign, em, phylotree, preprocess, synth


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=20000)
    ap.add_argument("--seed", type=int, default=7)
    ap.add_argument("--multi", type=int, default=1)
    ap.add_argument("--dense", action="store_true",
                    help="build the dense matrix and the posterior matrix (the reference's data flow); default: records only")
    ap.add_argument("--records", action="store_true", help="(the default since round 3; kept for old command lines)")
    ap.add_argument("--alignments", action="store_true",
                    help="input = synthetic alignments as columns through the batched front end (instead of ready-made rows)")
    ap.add_argument("--bam", action="store_true",
                    help="with --alignments: write them as a BAM file first (tests/_bam_writer.py, untimed) and start from "
                         "the FILE: the library's reader (alignments.read_bam) in front of the encoder")
    ap.add_argument("--pairs", action="store_true",
                    help="rows = paired-end fragments (synth-pe-v1: 2 x 150 bp, insert 350-500, mates merged as preprocess.py:118-138 "
                         "merges them): two thirds of them observe more than 64 sites")
    ap.add_argument("--read-len", type=int, default=150, help="length of the single-end reads (synth-v1; 250: 38 %% of the rows above 64 sites)")
    ap.add_argument("--threads", type=int, default=0, help="with --alignments: host threads of the encoder (0 = its default)")
    ap.add_argument("--storage", default="auto", choices=["f64", "f32", "coded", "auto"],
                    help="with --dense: form of the matrix the EM loop streams (EmPlan): coded = lossless row dictionaries, "
                         "auto = coded from 1.5e7 cells on")
    opts = ap.parse_args()
    opts.records = not opts.dense
    args = argparse.Namespace(init_alpha=1.0, tolerance=1e-4, max_iter=10000, n_multi=opts.multi,
                              verbose=True, min_reads=10, min_fold=2.0, storage=opts.storage)
    numpy.random.seed(opts.seed)                       # bin/mixemt:507-508

    t0 = time.perf_counter()
    refseq = phylotree.load_rsrs()
    phy = phylotree.load_build17(refseq)
    haps = sorted(phy.hap_var)
    tables = preprocess.HapVarTables.build(refseq, phy, haps)
    weights_host = None
    if opts.alignments:
        from mixemt_amd import alignments
        cols = synth.synth_alignments(tables, refseq, opts.reads, seed=1)
        sys.stderr.write("%d synthetic alignments of %d fragments as columns (%.1f s of generation, not part of the pipeline)\n"
                         % (len(cols), cols.n_frag, time.perf_counter() - t0))
        if opts.bam:
            import tempfile
            sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
            import _bam_writer
            t0 = time.perf_counter()
            tmp = tempfile.TemporaryDirectory()
            bam_path = os.path.join(tmp.name, "synth.bam")
            _bam_writer.write_bam(bam_path, cols)
            n_written = len(cols)
            sys.stderr.write("written as %s (%.1f MB) in %.1f s (not part of the pipeline)\n"
                             % (bam_path, os.path.getsize(bam_path) / 1e6, time.perf_counter() - t0))
            alignments.read_bam(bam_path, n_threads=opts.threads)                 # (page cache warm, as a file just written is)
            t0 = time.perf_counter()
            cols = alignments.read_bam(bam_path, n_threads=opts.threads)
            t_bam = time.perf_counter() - t0
            assert len(cols) == n_written
            sys.stderr.write("BAM reader (BGZF inflate -> records -> columns, names -> fragment indices): %.1f ms for %d alignments "
                             "(%.2f s per 10^6, %.0f MB/s of file)\n"
                             % (t_bam * 1e3, len(cols), t_bam / max(1, len(cols)) * 1e6, os.path.getsize(bam_path) / 1e6 / t_bam))
            tmp.cleanup()
        t0 = time.perf_counter()
        enc = alignments.encode_alignments(cols, tables.sites, len(refseq), 30, 30, n_threads=opts.threads)
        t_enc = time.perf_counter() - t0
        row_ptr, site, obs = enc.row_ptr, enc.site, enc.obs
        reads = enc.read_ids
        weights_host = enc.weights
        sys.stderr.write("front end (alignments -> fragments -> signatures -> de-dup -> sorted rows): %.1f ms for %d alignments "
                         "(%.2f s per 10^6): %d fragments with a site, %d distinct signatures, %d observations, %d dropped\n"
                         % (t_enc * 1e3, len(cols), t_enc / max(1, len(cols)) * 1e6, enc.n_fragments, enc.n_rows, len(site),
                            len(enc.dropped)))
        opts.reads = enc.n_rows
    else:
        if opts.pairs:                                    # synth-pe-v1: 2 x 150 fragments, mates merged into one row
            row_ptr, site, obs, who = synth.synth_pairs(tables, len(refseq), opts.reads, seed=1)
        else:
            row_ptr, site, obs, who = synth.synth_reads(tables, len(refseq), opts.reads, seed=1, read_len=opts.read_len)
        lens = numpy.diff(row_ptr)
        sys.stderr.write("[pipeline] rows: %s, %.1f sites per row, %.1f %% above 64, %.2f %% above 128\n"
                         % ("synth-pe-v1 (2 x 150, insert 350-500, mates merged)" if opts.pairs else "synth-v1, %d bp" % opts.read_len,
                            lens.mean(), 100.0 * (lens > 64).mean(), 100.0 * (lens > 128).mean()))
        reads = [[str(i)] for i in range(opts.reads)]     # the read ids behind each row (synthetic input, like the fragments)
        sys.stderr.write("Using %d variant sites from %d haplogroups; %d synthetic fragments (%.1f s)\n"
                         % (len(tables.sites), len(haps), opts.reads, time.perf_counter() - t0))

    # the process's first HIP calls (context, the library's code object) are the runtime's, not the pipeline's: timed apart
    t0 = time.perf_counter()
    torch.zeros(1, device="cuda")
    _lib.load()
    torch.cuda.synchronize()
    sys.stderr.write("device context + library load: %.1f ms\n" % ((time.perf_counter() - t0) * 1e3))
    t_all = time.perf_counter()

    # one-time table set-up (the reference's HapVarBaseMatrix.__init__, preprocess.py:39-67): marker lists and lookup
    # tables encoded on the host and uploaded
    t0 = time.perf_counter()
    tables.sparse_device()
    tables.lut_device()
    torch.cuda.synchronize()
    sys.stderr.write("haplogroup tables encoded and uploaded: %.1f ms\n" % ((time.perf_counter() - t0) * 1e3))

    t0 = time.perf_counter()
    wts = (torch.ones(opts.reads, dtype=torch.float64, device="cuda") if weights_host is None
           else torch.from_numpy(weights_host).to(device="cuda", dtype=torch.float64))
    if opts.records:
        em_mat = None
        # the CSR observations go to the device first (round 6: timed apart -- 111 MB for 10^6 150-bp reads, 217 MB for 10^6
        # merged mates; the stage time below is the upload + the build, as in earlier rounds)
        t_up = time.perf_counter()
        row_ptr_d = torch.from_numpy(numpy.ascontiguousarray(row_ptr)).cuda()
        site_d = torch.from_numpy(numpy.ascontiguousarray(site).view(numpy.int16)).cuda()
        obs_d = torch.from_numpy(numpy.ascontiguousarray(obs)).cuda()
        torch.cuda.synchronize()
        t_up = time.perf_counter() - t_up
        cm = preprocess.build_em_records_device(tables, row_ptr_d, site_d, obs_d)
        torch.cuda.synchronize()
        sys.stderr.write("EM input %d x %d built on the device as records in %.1f ms (of which %.1f ms the upload of the %d MB of "
                         "observations): %.2f GB, %d rows dense beside them (a dense matrix would be %.1f GB)\n"
                         % (cm.n_rows, cm.n_haps, (time.perf_counter() - t0) * 1e3, t_up * 1e3,
                            (row_ptr_d.numel() * 8 + site_d.numel() * 2 + obs_d.numel()) >> 20, cm.used / 1e9,
                            cm.rest_rows.numel(), cm.n_rows * cm.n_haps * 8 / 1e9))
    else:
        cm = None
        em_mat = preprocess.build_em_matrix_device(tables, row_ptr, site, obs)
        torch.cuda.synchronize()
        sys.stderr.write("EM input matrix %d x %d built on the device in %.1f ms\n"
                         % (em_mat.shape[0], em_mat.shape[1], (time.perf_counter() - t0) * 1e3))

    t0 = time.perf_counter()
    res = em.run_em_ex(em_mat, wts, args, want_read_mix=not opts.records, records=cm)
    props, read_mix = res["props"], res["read_mix"]
    torch.cuda.synchronize()
    sys.stderr.write("run_em: %.1f ms, of which the EM loop %.1f ms (%d iterations, %s matrix) and the plan "
                     "(allocation + linearise / encode) %.1f ms\n"
                     % ((time.perf_counter() - t0) * 1e3, res["loop_s"] * 1e3, sum(res["iters"]), res["storage"],
                        res["plan_s"] * 1e3))

    order = numpy.argsort(props)[::-1]
    sys.stderr.write("\nTop 10 haplogroups by proportion...\n")
    for i in range(10):
        sys.stderr.write("%d\t%0.6f\t%s\n" % (i + 1, props[order[i]], haps[order[i]]))
    t0 = time.perf_counter()
    if opts.records:
        # votes and contributors from the records' log tables under theta_k (no posterior matrix exists)
        seen, votes = assign.vote_table_from_records(cm, res["ln_theta_k"], None)     # every run of a --multi N
        sys.stderr.write("\nTop 10 haplogroups by read probabilities...\n")
        for hap_i in sorted(seen, key=lambda h: -votes[h])[:10]:
            sys.stderr.write("%s\t%d\n" % (haps[hap_i], int(votes[hap_i])))
        sys.stderr.write("\n")
        cons = [int(h) for h in seen if votes[h] >= args.min_reads]
    else:
        assign.report_read_votes(haps, read_mix, 10)
        cons = assign.find_contribs_from_reads(read_mix, wts, args)
    torch.cuda.synchronize()
    sys.stderr.write("contributors from read votes: %.1f ms\n" % ((time.perf_counter() - t0) * 1e3))
    contribs = sorted(([haps[c], props[c]] for c in cons), key=lambda c: c[1], reverse=True)
    fmt = "hap%%0%dd" % len(str(len(contribs) + 1))
    contribs = [[fmt % (i + 1)] + c for i, c in enumerate(contribs)]
    if not contribs:
        sys.stderr.write("\n0 contributors passed filtering steps.\n")
        return 1

    sys.stderr.write("Refining contribution estimates...\n")
    t0 = time.perf_counter()
    if opts.records:
        sub, sub_haps = preprocess.reduce_em_records(cm, haps, contribs)
    else:
        sub, sub_haps = preprocess.reduce_em_matrix(em_mat, haps, contribs)
    results = em.run_em(sub, wts, args)
    torch.cuda.synchronize()
    sys.stderr.write("refinement run_em on %d x %d: %.1f ms\n" % (sub.shape[0], sub.shape[1], (time.perf_counter() - t0) * 1e3))
    contribs = assign.update_contribs(contribs, results, sub_haps)
    t0 = time.perf_counter()
    table = assign.assign_read_indexes(contribs, results, sub_haps, reads, args.min_fold)
    sys.stderr.write("read assignment (device kernel + host table of %d ids): %.1f ms\n"
                     % (opts.reads, (time.perf_counter() - t0) * 1e3))

    sys.stderr.write("tables -> read assignment, all stages: %.1f ms\n" % ((time.perf_counter() - t_all) * 1e3))
    print("hap#   Haplogroup      Contribution   Reads")
    print("-------------------------------------------")
    for hap_id, group, prop in contribs:
        print("%s %s %s %s" % (hap_id.ljust(6), group.ljust(15), ("%.4f" % prop).rjust(12),
                               ("%d" % table.count(hap_id)).rjust(7)))
    print("unassigned %d" % table.count("unassigned"))
    return 0


if __name__ == "__main__":
    sys.exit(main())
