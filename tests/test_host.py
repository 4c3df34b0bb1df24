"""
Host logic of the product package against the reference-derived fixtures:
Phylotree parsing, table encoding, signature encoding, the synthetic generator.
CPU only (no kernels are called).
"""
import hashlib
import os

import numpy
import pytest

from conftest import golden
from mixemt_amd import phylotree, preprocess, synth


def test_toy_tree_matches_reference_docstring(toy):
    """phylotree.py:491-499 lists the toy tree's hap_var."""
    _, phy, _ = toy
    want = {'A': ['A1G', 'A2T', 'A4T'], 'B': ['A1G', 'A3T', 'A5T', 'A6T', 'A8T'],
            'C': ['A1G', 'A3T', 'T5A', 'A6T'], 'D': ['A1G', 'A3T', 'A5T', 'A7T', 'A9T'],
            'E': ['A1G', 'A3T', 'A4T', 'A5T', 'A7T'], 'F': ['A1G', 'A3T', 'A5T', 'A6T'],
            'G': ['A1G', 'A3T', 'A5T', 'A7T'], 'H': ['A1G', 'A3T', 'A5T'], 'I': ['A1G']}
    assert phy.hap_var == want
    assert phy.get_variant_pos() == list(range(9))
    assert phy.variants[4] == {'T': 1, 'A': 1} and phy.variants[3] == {'T': 2}


def test_toy_markers_match_reference_test(toy):
    """preprocess_test.py:34-50: marker table of the toy tree."""
    ref, phy, haps = toy
    tables = preprocess.HapVarTables.build(ref, phy, haps)
    markers = {'A': {1: 'T', 3: 'T', 0: 'G'}, 'B': {0: 'G', 2: 'T', 4: 'T', 5: 'T', 7: 'T'},
               'C': {0: 'G', 2: 'T', 5: 'T'}, 'D': {0: 'G', 2: 'T', 4: 'T', 6: 'T', 8: 'T'},
               'E': {0: 'G', 2: 'T', 3: 'T', 4: 'T', 6: 'T'}, 'F': {0: 'G', 2: 'T', 4: 'T', 5: 'T'},
               'G': {0: 'G', 2: 'T', 4: 'T', 6: 'T'}, 'H': {0: 'G', 2: 'T', 4: 'T'}, 'I': {0: 'G'}}
    for j, hap in enumerate(haps):
        for pos in range(9):
            want = markers[hap].get(pos, 'A')
            assert chr(tables.expected[pos, j]) == want
    # mut_prob scaling (preprocess_test.py:64-74 uses other weights; defaults here)
    assert numpy.allclose(numpy.exp(tables.lhit), 1 - numpy.minimum(0.5, 0.01 * numpy.array(
        [sum(phy.variants[p].values()) for p in range(9)])))


def test_line_parser_anonymous_nodes():
    rows = ["R, A1G ,,", ",, A2T ,,", ",,, A3T ,,", ",N, A4T ,,", ",,, A5T ,,"]
    phy = phylotree.Phylotree(rows, anon_haps=True)
    assert phy.names == ["R", "R[1]", "R[1][1]", "N", "N[1]"]
    assert phy.parent == [-1, 0, 1, 0, 3]
    assert phy.hap_var["R[1][1]"] == ["A1G", "A2T", "A3T"]
    phy2 = phylotree.Phylotree(rows, anon_haps=False)
    assert sorted(phy2.hap_var) == ["N", "R"]


def test_annotations_and_filters():
    rows = ["R, (A1G) A2T! C3d 4.1T a5c ,,", ",K, G1A T2A ,,"]
    phy = phylotree.Phylotree(rows)
    assert phy.hap_var["R"] == ["A1G", "A2T", "A5C"]        # indels dropped, annotations stripped
    assert phy.hap_var["K"] == ["G1A", "T2A", "A5C"]        # newer mutation masks the older
    assert phylotree.Phylotree(rows, rm_unstable=True).get_variant_pos() == [1, 4]
    assert phylotree.Phylotree(rows, rm_unstable=True, rm_backmut=True).get_variant_pos() == [4]
    assert phylotree.pos_from_var("(A95c)") == 94 and phylotree.der_allele("(A95c)") == "C"
    assert phylotree.der_allele("A263G!") == "G" and phylotree.anc_allele("(A95c)") == "A"
    phy.add_custom_hap("mine", ["A1T", "G9A"])
    with pytest.raises(ValueError):
        phy.add_custom_hap("mine", ["A1T"])
    phy.ignore_sites("1,4-5")
    assert phy.get_variant_pos() == [1]
    assert phy.variants[1]["T"] == 2                         # counters accumulate (reference quirk)


def test_build17_tables_match_reference(b17):
    refseq, phy, haps, tables = b17
    g = golden("g0_tables_b17")
    assert len(refseq) == 16569
    assert haps == str(g["hap_names"]).split("\n")
    assert int(g["n_haps"]) == 5408 and int(g["n_sites"]) == 4070
    assert numpy.array_equal(tables.sites, g["sites"])
    dense = numpy.empty((len(tables.sites), len(haps)), dtype=numpy.uint8)
    dense[:, :] = g["ref_codes"][:, None]
    dense[g["marker_site"], g["marker_hap"]] = g["marker_base"]
    assert len(g["marker_hap"]) == 263826
    assert numpy.array_equal(tables.expected[:, :len(haps)], dense)
    assert hashlib.sha256(dense.tobytes()).hexdigest() == str(g["dense_sha256"])
    assert not tables.expected[:, len(haps):].any()
    assert numpy.array_equal(tables.lhit, numpy.log(1.0 - g["mut_prob"]))
    assert numpy.array_equal(tables.lmiss, numpy.log(g["mut_prob"] / 3.0))


def test_encode_signatures_roundtrip_and_errors(b17):
    refseq, phy, haps, tables = b17
    row_ptr, site, obs, who = synth.synth_reads(tables, len(refseq), 50, seed=9)
    sigs = synth.signatures(tables, row_ptr, site, obs)
    rp, si, ob = preprocess.encode_signatures(sigs, tables)
    assert numpy.array_equal(rp, row_ptr) and numpy.array_equal(si, site) and numpy.array_equal(ob, obs)
    assert rp.dtype == numpy.int64 and si.dtype == numpy.uint16 and ob.dtype == numpy.uint8
    with pytest.raises(ValueError):
        preprocess.encode_signatures(["%d:A" % tables.sites[0], ""], tables)          # reference: int('') ValueError
    with pytest.raises(KeyError):
        preprocess.encode_signatures(["0:A"], tables)              # position 0 is not a variant site
    rp, si, ob = preprocess.encode_signatures(["%d:N,%d:ac" % (tables.sites[0], tables.sites[1])], tables)
    assert list(ob) == [ord("N"), 0]


def test_native_and_itemwise_signature_parsers_agree(b17):
    """
    encode_signatures goes through the library's host parser (mxm_encode_signatures) and keeps
    the item-by-item Python expressions of the reference for anything unusual: both must give the
    same CSR, and the unusual inputs must raise what the reference raises.
    """
    from mixemt_amd import build
    build.build()                                     # host function of the in-tree library
    refseq, phy, haps, tables = b17
    rng = numpy.random.default_rng(4)
    sites = numpy.asarray(tables.sites)
    sigs = []
    for _ in range(400):
        picked = numpy.sort(rng.choice(sites, size=int(rng.integers(1, 60)), replace=False))
        sigs.append(",".join("%0*d:%s" % (int(rng.integers(1, 7)), p, rng.choice(["A", "C", "G", "T", "N", "a", "-", "AC", ""]))
                             for p in picked))
    fast = preprocess._encode_signatures_native(sigs, tables)
    slow = preprocess._encode_signatures_py(sigs, tables)
    assert fast is not None
    for a, b in zip(fast, slow):
        assert a.dtype == b.dtype and numpy.array_equal(a, b)
    s0 = int(sites[0])
    # inputs the C parser hands back; the Python expressions then decide, like the reference
    for sig, exc in (("", ValueError), ("%d:A," % s0, ValueError), (",%d:A" % s0, ValueError),
                     ("%d:A:C" % s0, ValueError), ("x:A", ValueError), ("%d" % s0, ValueError),
                     ("0:A", KeyError), ("-5:A", KeyError), ("99999999999:A", KeyError)):
        assert preprocess._encode_signatures_native(["%d:A" % s0, sig], tables) is None
        with pytest.raises(exc):
            preprocess.encode_signatures(["%d:A" % s0, sig], tables)
    # int() accepts more than digits: same answer through the slow path
    rp, si, ob = preprocess.encode_signatures([" %d :G" % s0, "%d_0:A" % (int(sites[1]) // 10)] if int(sites[1]) % 10 == 0
                                              else [" %d :G" % s0], tables)
    assert si[0] == 0 and ob[0] == ord("G")
    assert preprocess._encode_signatures_native([], tables) is None
    assert preprocess.encode_signatures([], tables)[0].tolist() == [0]


def test_synth_v1_is_pinned(b17):
    """The generator's stream is part of the bench definition: same seed, same bytes."""
    refseq, phy, haps, tables = b17
    g = golden("g2_build_b17")
    row_ptr, site, obs, who = synth.synth_reads(tables, len(refseq), 1032, seed=2)
    assert numpy.array_equal(row_ptr, g["row_ptr"])
    assert numpy.array_equal(site, g["site"]) and numpy.array_equal(obs, g["obs"])
    counts = numpy.diff(row_ptr)
    assert counts.min() >= 1 and 30 < counts.mean() < 45
    assert set(numpy.unique(obs)) <= set(b"ACGTN")


def test_reduce_em_matrix_numpy():
    """preprocess.py:230-251."""
    mat = numpy.arange(12.0).reshape(3, 4)
    sub, names = preprocess.reduce_em_matrix(mat, list("WXYZ"), [["hap1", "Y", 0.7], ["hap2", "W", 0.3]])
    assert names == ["W", "Y"] and numpy.array_equal(sub, mat[:, [0, 2]])


def test_mutation_weight_scaling_reference_cases(toy):
    """preprocess_test.py:52-74: hit = 1 - mu, miss = mu / 3 with mu = min(mut_max, mut_wt * count)."""
    import math
    ref, phy, haps = toy
    flat = preprocess.HapVarTables.build(ref, phy, haps, mut_wt=0.10, mut_max=0.10)
    assert numpy.array_equal(flat.lhit, numpy.array([math.log(1.0 - 0.1)] * 9))
    assert numpy.array_equal(flat.lmiss, numpy.array([math.log(0.1 / 3.0)] * 9))
    scaled = preprocess.HapVarTables.build(ref, phy, haps, mut_wt=0.10, mut_max=0.50)
    # sites 3 and 4 (0-based) carry two mutation events in the toy tree
    for pos, mu in ((0, 0.1), (3, 0.2), (4, 0.2), (2, 0.1)):
        assert scaled.lhit[pos] == math.log(1.0 - mu) and scaled.lmiss[pos] == math.log(mu / 3.0)


def test_other_trees_and_flags_match_reference_digests():
    """Build 16, named-only, stable-only and strict variants of the tree parse to exactly what the
    reference's parser produces (digests in g0b), and both shipped reference sequences are intact."""
    import hashlib
    g = golden("g0b_trees")
    rsrs, rcrs = phylotree.load_rsrs(), phylotree.load_rcrs()
    assert hashlib.sha256(rsrs.encode()).hexdigest() == str(g["rsrs_sha256"])
    assert hashlib.sha256(rcrs.encode()).hexdigest() == str(g["rcrs_sha256"])
    cases = (("b16", phylotree.load_build16(rsrs)),
             ("b17_named", phylotree.load_build17(rsrs, anon_haps=False)),
             ("b17_stable", phylotree.load_build17(rsrs, rm_unstable=True)),
             ("b17_strict", phylotree.load_build17(rsrs, rm_unstable=True, rm_backmut=True)))
    for tag, tree in cases:
        names = sorted(tree.hap_var)
        text = "\n".join("%s\t%s" % (n, ",".join(tree.hap_var[n])) for n in names)
        counts = "\n".join("%d\t%s" % (p, ",".join("%s=%d" % kv for kv in sorted(tree.variants[p].items())))
                           for p in sorted(tree.variants))
        assert len(names) == int(g[tag + "_n_haps"]) and len(tree.variants) == int(g[tag + "_n_sites"])
        assert hashlib.sha256(text.encode()).hexdigest() == str(g[tag + "_hap_var_sha256"]), tag
        assert hashlib.sha256(counts.encode()).hexdigest() == str(g[tag + "_variants_sha256"]), tag


def test_bench_without_a_gpu_fails_loudly_instead_of_downgrading():
    """`python bench.py --gpus 2` where no GPU exists: the ranks it starts report the missing GPU and the
    parent exits non-zero with no JSON line (never a silent n_gpus = 1 result)."""
    import subprocess
    import sys
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    proc = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--total-rows", "1000"],
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert proc.returncode != 0
    assert not [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
    assert "no ROCm GPU" in proc.stderr


def test_synth_rows_is_one_global_read_set_whatever_the_slice(b17):
    """bench.py's strong scaling: every rank generates ITS rows of one block-defined read set; any
    slicing of the row range gives the same observations."""
    refseq, phy, haps, tables = b17
    whole = synth.synth_rows(tables, len(refseq), 0, 1000, seed=3, block=300)
    lo, hi = 250, 910
    part = synth.synth_rows(tables, len(refseq), lo, hi, seed=3, block=300)
    rp = whole[0]
    assert numpy.array_equal(part[0], rp[lo:hi + 1] - rp[lo])
    assert numpy.array_equal(part[1], whole[1][rp[lo]:rp[hi]])
    assert numpy.array_equal(part[2], whole[2][rp[lo]:rp[hi]])
    assert numpy.array_equal(part[3], whole[3][lo:hi])
    again = synth.synth_rows(tables, len(refseq), lo, hi, seed=3, block=300)
    assert all(numpy.array_equal(a, b) for a, b in zip(part, again))
    other = synth.synth_rows(tables, len(refseq), lo, hi, seed=4, block=300)
    assert not numpy.array_equal(part[2], other[2][:len(part[2])])
    empty = synth.synth_rows(tables, len(refseq), 5, 5)
    assert empty[0].tolist() == [0] and empty[1].size == 0


def test_sparse_tables_are_the_dense_table(b17):
    """maj + markers reproduce expected[S][H] exactly."""
    refseq, phy, haps, tables = b17
    sp = tables.sparse()
    dense = numpy.repeat(sp["maj"][:, None], len(haps), axis=1)
    site_of = numpy.repeat(numpy.arange(len(sp["maj"])), numpy.diff(sp["mk_ptr"]))
    dense[site_of, sp["mk_hap"]] = sp["mk_base"]
    assert numpy.array_equal(dense, tables.expected[:, :len(haps)])
    assert len(sp["mk_hap"]) < 0.01 * dense.size


def test_gather_csr_of_a_row_subset():
    """preprocess._gather_csr (the records build's dense leftover rows): CSR of a row subset, any order."""
    import torch
    rng = numpy.random.default_rng(5)
    lens = rng.integers(0, 9, size=40)
    row_ptr = numpy.concatenate([[0], numpy.cumsum(lens)]).astype(numpy.int64)
    site = rng.integers(0, 4000, size=int(row_ptr[-1])).astype(numpy.int16)
    obs = rng.integers(65, 90, size=int(row_ptr[-1])).astype(numpy.uint8)
    rows = numpy.array([3, 39, 0, 17, 18, 5], dtype=numpy.int64)
    new_ptr, sub_site, sub_obs = preprocess._gather_csr(torch.from_numpy(row_ptr), torch.from_numpy(site),
                                                        torch.from_numpy(obs), torch.from_numpy(rows))
    assert numpy.array_equal(numpy.diff(new_ptr.numpy()), lens[rows])
    want_site = numpy.concatenate([site[row_ptr[r]:row_ptr[r + 1]] for r in rows])
    want_obs = numpy.concatenate([obs[row_ptr[r]:row_ptr[r + 1]] for r in rows])
    assert numpy.array_equal(sub_site.numpy(), want_site) and numpy.array_equal(sub_obs.numpy(), want_obs)


def test_bench_quotes_counter_traffic_of_the_kernel_it_ran(tmp_path):
    """bench.pmc_traffic: the committed counter file of the SAME workload, storage and kernel instance;
    for the headline (10^6 x 5408, dense fp64, one restart) that is 43.4 GB +- 2 % per launch -- never the
    0.97 GB of the dense leftover rows that a row-dictionary run sends through the same kernel template."""
    import json
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    algo = 1e6 * 5408 * 8
    name = "em_iter_wide_kernel<512, 6, 1, 3, 1>"
    got = bench.pmc_traffic(10 ** 6, 5408, "f64", name, algo)
    assert got is not None and abs(got[0] - 43.4e9) < 0.02 * 43.4e9, got
    assert "coded" not in got[1]
    # the leftover-rows instance of a coded run is never taken for the dense matrix's kernel
    assert bench.pmc_traffic(10 ** 6, 5408, "f64", "em_iter_wide_kernel<256, 11, 1, 2, 1>", algo) is None
    # another workload has no figure
    assert bench.pmc_traffic(123456, 5408, "f64", name, 123456 * 5408 * 8.0) is None
    # selection rules on a scratch tree: storage must match, the later round wins, out-of-range figures are refused
    for rnd, fname, storage, val in (("r01", "pmc_traffic_a.json", "f64", 1.01 * algo), ("r07", "pmc_traffic_a.json", "f64", 1.02 * algo),
                                     ("r09", "pmc_traffic_coded_a.json", "coded", 1.0 * algo), ("r10", "pmc_traffic_b.json", "f64", 0.02 * algo)):
        os.makedirs(tmp_path / "profiles" / rnd, exist_ok=True)
        with open(tmp_path / "profiles" / rnd / fname, "w") as fout:
            json.dump({name: {"hbm_bytes_per_launch": val},
                       "_workload": {"rows_per_gpu": 10 ** 6, "haps": 5408, "storage": storage}}, fout)
    val, path = bench.pmc_traffic(10 ** 6, 5408, "f64", name, algo, root=str(tmp_path))
    assert val == 1.02 * algo and path == os.path.join("profiles", "r07", "pmc_traffic_a.json")


def test_bench_quotes_calibrated_counter_traffic_for_the_records_kernel():
    """`bench.py --storage coded` quotes roofline.traffic for em_iter_coded_kernel from a counter file whose
    factor was CALIBRATED (a bare reader of exactly the same records in the same FETCH_SIZE pass), under the dense line's
    [0.9, 1.5] x rule: within 10 % of the 5.8 GB of codes + P tables the kernel has to read at 10^6 x 5408."""
    import json
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cal = json.load(open(os.path.join(root, "profiles", "r04", "pmc_calibration_coded.json")))
    algo = float(cal["record_bytes_read_per_pass"])
    got = bench.pmc_traffic(10 ** 6, 5408, "coded", "em_iter_coded_kernel*", algo)
    assert got is not None and abs(got[0] - algo) < 0.10 * algo, got
    data = json.load(open(os.path.join(root, got[1])))
    rec = next(v for k, v in data.items() if k.startswith("em_iter_coded_kernel"))
    assert "diag_stream_coded_kernel" in rec["fetch_correction_source"] and 1.0 < rec["fetch_correction"] <= 2.0
    # the dense matrix's figure is never taken for it, nor the other way round
    assert bench.pmc_traffic(10 ** 6, 5408, "f64", "em_iter_coded_kernel*", algo) is None
    # the quad dictionary's kernel has a calibration of its own (a bare reader of the quad records: --quads), and neither
    # kernel's figure answers for the other
    calq = json.load(open(os.path.join(root, "profiles", "r05", "pmc_calibration_quads.json")))
    algo_q = float(calq["kernel_bytes_per_pass"])
    got_q = bench.pmc_traffic(10 ** 6, 5408, "coded", "em_iter_quad_coded_kernel*", algo_q)
    assert got_q is not None and abs(got_q[0] - algo_q) < 0.05 * algo_q and got_q[1] != got[1], got_q
    rec_q = next(v for k, v in json.load(open(os.path.join(root, got_q[1]))).items() if k.startswith("em_iter_quad_coded_kernel"))
    assert "diag_stream_quads_kernel" in rec_q["fetch_correction_source"] and 1.0 < rec_q["fetch_correction"] <= 2.0
    assert calq["bare_reader_covers"] > 0.9


def test_committed_profiles_name_the_kernel_instances_this_source_builds():
    """profile files must come from the binary that ships.  The latest round's rocprofv3 kernel statistics
    and counter summary name exactly the template instances this source tree launches for the benchmark's workload
    (streaming kernel shape from the library itself, the build kernel's column ranges and the lookup-table / coded
    shapes from the source's defaults)."""
    import ctypes
    import glob
    import json
    import re
    from mixemt_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rounds = sorted(glob.glob(os.path.join(root, "profiles", "r[0-9]*")), key=lambda p: int(re.search(r"r(\d+)$", p).group(1)))
    # (a round's directory fills up as the round goes; its kernel statistics are regenerated from the final binary)
    latest = [p for p in rounds if os.path.exists(os.path.join(p, "bench_1m_kernel_stats.csv"))][-1]
    stats = open(os.path.join(latest, "bench_1m_kernel_stats.csv")).read()
    traffic = json.load(open(os.path.join(latest, "pmc_traffic_1m.json")))
    lib = _lib.load()
    buf = ctypes.create_string_buffer(96)
    assert lib.mxm_describe_stream_kernel(5408, 1, buf, len(buf)) == 0
    stream_kernel = buf.value.decode()
    src = open(os.path.join(root, "mixemt_amd", "csrc", "mixemt_hip.hip")).read()
    passes = int(re.search(r"#define SPB_PASSES (\d+)", src).group(1))
    lut = open(os.path.join(root, "mixemt_amd", "csrc", "build_lut_kernels.hpp")).read()
    cpl = int(re.search(r"#define LUT_CPL (\d+)", lut).group(1))
    expected = [stream_kernel, "build_sparse_kernel<11, %d, false, 1>" % passes, "build_lut_kernel<6, %d>" % cpl,
                "em_iter_coded_kernel<256, 6, 4, 2>", "em_fused_coded_kernel<6, 4, false>", "encode_wide_rows_kernel<6>"]
    for name in expected:
        assert name in stats, "%s: not in %s/bench_1m_kernel_stats.csv -- regenerate the profiles from this binary" % (name, latest)
    assert stream_kernel in traffic and traffic["_workload"]["storage"] == "f64"
    assert abs(traffic[stream_kernel]["hbm_bytes_per_launch"] / (1e6 * 5408 * 8) - 1.0) < 0.02


def test_synth_pe_v1_is_pinned(b17):
    """synth-pe-v1 (round 6; 2 x 150 paired-end fragments, mates merged into one row as preprocess.py:118-138 merges
    them): the draw order is the definition -- same seed, same bytes; two thirds of the rows observe more than 64 sites."""
    refseq, phy, haps, tables = b17
    row_ptr, site, obs, who = synth.synth_pairs(tables, len(refseq), 4000, seed=5)
    lens = numpy.diff(row_ptr)
    assert lens.min() >= 1 and 0.6 < (lens > 64).mean() < 0.75 and (lens > 128).mean() < 0.04
    # ascending sites inside every row (no overlap between the mates at inserts >= 2 x 150)
    step = numpy.diff(site.astype(numpy.int64))
    inner = numpy.ones(len(step), dtype=bool)
    inner[row_ptr[1:-1] - 1] = False
    assert (step[inner] > 0).all()
    digest = hashlib.sha256(row_ptr.tobytes() + site.tobytes() + obs.tobytes() + who.tobytes()).hexdigest()
    again = synth.synth_pairs(tables, len(refseq), 4000, seed=5)
    assert hashlib.sha256(b"".join(a.tobytes() for a in again)).hexdigest() == digest
    assert digest[:16] == SYNTH_PE_V1_DIGEST, digest[:16]
    # blocks of synth_rows: one global fragment set whatever the slice
    a = synth.synth_rows(tables, len(refseq), 100, 700, seed=3, block=256, pairs=True)
    b = synth.synth_rows(tables, len(refseq), 0, 1024, seed=3, block=256, pairs=True)
    assert numpy.array_equal(a[1], b[1][b[0][100]:b[0][700]]) and numpy.array_equal(a[3], b[3][100:700])
    with pytest.raises(ValueError):
        synth.synth_pairs(tables, len(refseq), 10, insert=(200, 300))


SYNTH_PE_V1_DIGEST = "54e74634cdf719fa"


def test_bench_rank_census_rules():
    """bench.rank_census: a multi-GPU line is only sane when the all-reduce of ones saw --gpus ranks and, under
    RCCL, no two ranks sit on one PCI address."""
    import importlib.util
    import types
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)

    class Ones(object):
        def __init__(self, total):
            self.total = total

        def item(self):
            return self.total

    def fake(world_seen, devices):
        props = types.SimpleNamespace(name="AMD Instinct MI355X", pci_domain_id=0, pci_bus_id=5, pci_device_id=0, uuid="u0")
        torch = types.SimpleNamespace(cuda=types.SimpleNamespace(get_device_properties=lambda dev: props),
                                      ones=lambda n, dtype=None, device=None: Ones(float(world_seen)), float64=None)

        def gather(out, mine):
            for i, d in enumerate(devices):
                out[i] = d

        dist = types.SimpleNamespace(all_reduce=lambda t, op=None: None, ReduceOp=types.SimpleNamespace(SUM=0), all_gather_object=gather)
        return torch, dist

    dev = types.SimpleNamespace(index=0)
    eight = [{"rank": r, "name": "x", "pci": "0000:%02x:00" % (5 + r), "uuid": None} for r in range(8)]
    torch, dist = fake(8, eight)
    ok = bench.rank_census(torch, dist, dev, 0, 8, True, "nccl", 8)
    assert ok["ok"] and ok["ranks_seen"] == 8 and len(ok["devices"]) == 8 and ok["backend"] == "nccl"
    torch, dist = fake(7, eight)
    assert not bench.rank_census(torch, dist, dev, 0, 8, True, "nccl", 8)["ok"]
    shared = [dict(d, pci="0000:05:00") if d["rank"] in (2, 3) else d for d in eight]
    torch, dist = fake(8, shared)
    bad = bench.rank_census(torch, dist, dev, 0, 8, True, "nccl", 8)
    assert not bad["ok"] and "share a GPU" in bad["note"]
    assert bench.rank_census(torch, dist, dev, 0, 8, True, "gloo", 8)["ok"]       # (test set-ups share the one GPU over gloo)
    one = bench.rank_census(torch, dist, dev, 0, 1, False, "nccl", 1)
    assert one["ok"] and one["ranks_seen"] == 1 and one["backend"] is None
