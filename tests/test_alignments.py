"""
The batched alignment front end (mixemt_amd/alignments.py -> mxm_aln_encode, csrc/aln_encode.hpp) against the
reference's own front end: run on golden g11 (preprocess.py:99-139, :163-174, :218-225 executed by the reference,
tools/gen_golden.py) and, on inputs the fixture does not hold, against the object-by-object restatement
preprocess.process_reads / reduce_reads (itself pinned by g11 and by the reference's unit cases, test_frontend.py).
CPU only: the encoder is a host function of the library.
"""
import json

import numpy
import pytest

from _fake_aln import FakeAln, FakeBam, from_columns
from mixemt_amd import alignments, preprocess


def _python_path(alns, var_pos, min_mq, min_bq):
    obs = preprocess.process_reads(alns, var_pos, min_mq, min_bq)
    sigs = preprocess.reduce_reads(obs)
    dropped = sigs.pop("", [])
    rows = sorted(sigs)
    return obs, rows, [len(sigs[r]) for r in rows], [sigs[r] for r in rows], dropped


def _same_as_python(alns, var_pos, ref_len, min_mq=30, min_bq=30, cols=None, n_threads=0):
    cols = cols if cols is not None else alignments.AlignmentColumns.from_alignments(alns)
    enc = alignments.encode_alignments(cols, var_pos, ref_len, min_mq, min_bq, n_threads=n_threads)
    obs, rows, weights, ids, dropped = _python_path(alns, var_pos, min_mq, min_bq)
    got = enc.read_obs(numpy.asarray(var_pos))
    assert got == obs and list(got) == list(obs)                     # the dict AND its insertion order
    assert enc.signatures() == rows
    assert enc.weights.tolist() == weights and enc.weights.dtype == numpy.int64
    assert enc.read_ids == ids and len(enc.read_ids) == len(rows)
    assert enc.dropped == dropped
    assert enc.n_fragments == len(obs)
    # the CSR is the signatures' own content
    sites = numpy.asarray(var_pos)
    for r in (0, len(rows) // 2, len(rows) - 1) if rows else ():
        a, b = int(enc.row_ptr[r]), int(enc.row_ptr[r + 1])
        assert ",".join("%d:%s" % (sites[s], chr(o)) for s, o in zip(enc.site[a:b], enc.obs[a:b])) == rows[r]
    return enc


def test_g11_the_reference_run(b17):
    """The reference's process_reads / reduce_reads / build_em_input on 454 alignments, bit for bit."""
    from conftest import golden
    refseq, phy, haps, tables = b17
    g = golden("g11_frontend")
    alns = [FakeAln(*rec) for rec in json.loads(str(g["alns"]))]
    var_pos = phy.get_variant_pos()
    enc = _same_as_python(alns, var_pos, len(refseq), int(g["min_mq"]), int(g["min_bq"]))
    want = {name: {int(p): b for p, b in obs.items()} for name, obs in json.loads(str(g["read_obs"])).items()}
    assert enc.read_obs(numpy.asarray(var_pos)) == want
    assert enc.signatures() == str(g["signatures"]).split("\n")
    assert numpy.array_equal(enc.weights, g["weights"])
    assert enc.read_ids == json.loads(str(g["read_ids"]))
    assert enc.dropped == [str(g["empty_name"])]
    sigs = json.loads(str(g["read_sigs"]))
    assert {s: ids for s, ids in zip(enc.signatures(), enc.read_ids)} == {s: ids for s, ids in sigs.items() if s}


def test_reference_unit_cases():
    """preprocess_test.py:126-196 through the encoder (the cases test_frontend.py runs through the python path)."""
    aln1 = FakeAln("read1", 10, 30, "AAAAATAAAATAAAAT", [30] * 16, "16M")
    qq = [33] * 12
    qq[3] = 20
    aln2 = FakeAln("read2", 12, 20, "AAAGAAGAAAAG", qq, "5M2D7M")
    base = [aln1, aln2, FakeAln("read3", 0, 0)]
    for min_mq, min_bq in ((20, 10), (25, 10), (20, 30)):
        _same_as_python(base, [15, 20, 25], 64, min_mq, min_bq)
    _same_as_python(base + [FakeAln("read1", 30, 30, "AAAAACAAAACAAAAT", [30] * 16, "16M")], [15, 20, 25, 35, 40], 64, 20, 10)
    _same_as_python(base + [FakeAln("read1", 20, 20, "AAAAATAAAACAAAAT", [30] * 16, "16M")], [15, 20, 25, 35], 64, 20, 10)
    qq = [30] * 16
    qq[0] = 5
    _same_as_python(base + [FakeAln("read1", 20, 20, "AAAAATAAAACAAAAC", qq, "16M")], [15, 20, 25, 35], 64, 20, 10)
    _same_as_python([FakeAln("r", 10, 60, "ACGTACGT", None, "8M")], [11, 12], 64, 20, 30)


def test_conflicts_ns_case_and_cigar_operations():
    sites = [3, 5, 8, 12, 20, 21, 30]
    alns = [
        FakeAln("n_only", 0, 60, "acgNacgt", [40] * 8, "8M"),                    # site 3 is 'N': dropped; 5 -> 'C'
        FakeAln("abc", 0, 60, "AAAAAAAAAAAAA", [40] * 13, "13M"),
        FakeAln("abc", 4, 60, "ACAAGAAAA", [40] * 9, "9M"),                      # site 5: A vs C -> N; 8: A vs G -> N
        FakeAln("abc", 5, 60, "AAAAAAAA", [40] * 8, "8M"),                       # 'N' stays 'N' whatever comes next
        FakeAln("ops", 0, 60, "TTTTGGGGCCCCAAAATTTT", [40] * 20, "2S2M1I3M4D2=3X2N3M2H"),
        FakeAln("lowq", 0, 60, "ACGTACGTACGTA", [40, 40, 40, 3] + [40] * 9, "13M"),
        FakeAln("filtered", 0, 10, "ACGTACGTACGTA", [40] * 13, "13M"),           # mapping quality: never seen
        FakeAln("nothing", 13, 60, "ACGTAC", [40] * 6, "6M"),                    # covers no site: not a fragment
        FakeAln("dup1", 18, 60, "ACGT", None, "4M"),
        FakeAln("dup2", 18, 60, "ACGTT", None, "5M"),
        FakeAln("all_n", 28, 60, "AANAA", [40] * 5, "5M"),                       # its only site is 'N': empty signature
        FakeAln("edge", 27, 60, "ACGTACGT", [40] * 8, "8M"),                     # runs past the reference's end (32)
        FakeAln("neg", -2, 60, "ACGTACGT", [40] * 8, "8M"),
    ]
    enc = _same_as_python(alns, sites, 32, 30, 30)
    assert enc.dropped == ["all_n"] and "filtered" not in enc.read_ids.names[:0]
    obs = enc.read_obs(numpy.asarray(sites))
    assert obs["n_only"] == {5: "C"} and obs["abc"] == {3: "A", 12: "A"} and obs["all_n"] == {}
    assert "nothing" not in obs and "filtered" not in obs
    assert any(len(ids) == 2 for ids in enc.read_ids)                            # dup1 / dup2 share a row


def test_row_order_is_pythons_string_order():
    """sorted() over 'pos:base,...' strings: '730:A' < '73:A' (a digit sorts before ':'), ',' before any digit, a
    signature before every longer one it is a prefix of."""
    sites = [7, 73, 730, 731, 7300, 7301]
    seq = {7: "A", 73: "C", 730: "G", 731: "T", 7300: "A", 7301: "C"}
    alns = []
    rng = numpy.random.default_rng(3)
    for i in range(300):
        a = int(rng.choice([0, 60, 700, 7290]))
        n = int(rng.integers(5, 60))
        text = "".join(seq.get(p, "A") if rng.random() > 0.2 else "ACGT"[int(rng.integers(0, 4))] for p in range(a, a + n))
        alns.append(FakeAln("q%d" % i, a, 60, text, None, "%dM" % n))
    alns += [FakeAln("two_a", 5, 60, "AAA", None, "3M"), FakeAln("two_b", 70, 60, "AAAAAAAAAA", None, "10M")]
    alns.append(FakeAln("two_a", 70, 60, "AAAAAAAAAA", None, "10M"))             # 7:A,73:A  vs  73:A
    enc = _same_as_python(alns, sites, 7400, 30, 30)
    rows = enc.signatures()
    assert rows == sorted(rows) and len(rows) > 20
    assert any(r.startswith("730:") for r in rows) and any(r.startswith("73:") for r in rows)


@pytest.mark.parametrize("n_frag,seed,threads", [(3000, 5, 1), (20000, 6, 4)])
def test_synthetic_alignments_equal_the_python_path(b17, n_frag, seed, threads):
    """synth-aln-v1 (mates, indels, clips, N, lower case, low / missing qualities, duplicates), shuffled."""
    from mixemt_amd import synth
    refseq, phy, haps, tables = b17
    cols = synth.synth_alignments(tables, refseq, n_frag, seed=seed)
    alns = from_columns(cols)
    # the adapter reads the objects back into the very same columns
    back = alignments.AlignmentColumns.from_alignments(alns)
    for key in ("ref_start", "mapq", "cig_ptr", "cigar", "seq_ptr", "seq", "has_qual"):
        assert numpy.array_equal(getattr(back, key), getattr(cols, key)), key
    assert [back.names[f] for f in back.frag] == [cols.names[f] for f in cols.frag]
    enc = _same_as_python(alns, phy.get_variant_pos(), len(refseq), 30, 30, cols=cols, n_threads=threads)
    assert enc.weights.max() > 10 and len(cols) > n_frag                         # duplicates collapse; mates exist
    # conflicting overlaps really occur: a site two high-quality reads of one fragment disagree on is gone from it
    var_pos = phy.get_variant_pos()
    merged = enc.read_obs(tables.sites)
    lost = 0
    for i, aln in enumerate(alns[:4000]):
        alone = preprocess.process_reads([aln], var_pos, 30, 30).get(aln.query_name, {})
        lost += sum(1 for pos, base in alone.items() if base != "N" and pos not in merged[aln.query_name])
    assert lost > 0


def test_what_the_encoder_hands_back(b17):
    """An unknown CIGAR operation or a CIGAR that runs past its sequence: NeedsSlowPath; build_em_input's "auto" then
    takes the object-by-object path (which raises what the reference raises), "batched" passes the refusal on."""
    refseq, phy, haps, tables = b17
    var_pos = phy.get_variant_pos()
    good = FakeAln("ok", 100, 60, "ACGT" * 30, None, "120M")
    for bad in (FakeAln("b", 100, 60, "ACGT" * 5, None, "10M5B10M"), FakeAln("c", 100, 60, "ACGT" * 5, None, "30M")):
        cols = alignments.AlignmentColumns.from_alignments([good, bad])
        with pytest.raises(alignments.NeedsSlowPath):
            alignments.encode_alignments(cols, var_pos, len(refseq), 30, 30)
    with pytest.raises(alignments.NeedsSlowPath):
        alignments.AlignmentColumns.from_alignments([FakeAln("u", 1, 60, "ACéT", None, "4M")])
    with pytest.raises(alignments.NeedsSlowPath):
        alignments.AlignmentColumns.from_alignments([FakeAln("q", 1, 60, "ACGT", [30, 30], "4M")])
    # columns that do not hang together are refused before the library sees them
    with pytest.raises(ValueError):
        alignments.AlignmentColumns([0], [60], [0], [0, 1], [16 << 4], [0, 9], numpy.zeros(4, dtype=numpy.uint8), None, None, ["x"])
    bad_frag = alignments.AlignmentColumns([0], [60], [3], [0, 1], [4 << 4], [0, 4], numpy.frombuffer(b"ACGT", dtype=numpy.uint8),
                                           None, None, ["x"])
    with pytest.raises(ValueError):
        alignments.encode_alignments(bad_frag, var_pos, len(refseq), 30, 30)


def test_empty_input_and_read_id_groups(b17):
    refseq, phy, haps, tables = b17
    enc = alignments.encode_alignments(alignments.AlignmentColumns.from_alignments([]), phy.get_variant_pos(), len(refseq), 30, 30)
    assert enc.n_rows == 0 and enc.signatures() == [] and len(enc.read_ids) == 0 and enc.dropped == []
    groups = alignments.ReadIdGroups(numpy.array([0, 2, 3]), numpy.array([1, 0, 2]), ["a", "b", "c"])
    assert groups == [["b", "a"], ["c"]] and groups[1] == ["c"] and groups[-2] == ["b", "a"] and groups[0:2] == [["b", "a"], ["c"]]
    assert list(groups) == groups.tolist() and groups.counts().tolist() == [2, 1] and not (groups == [["b"], ["c"]])
    with pytest.raises(IndexError):
        groups[2]


@pytest.mark.slow
def test_a_million_fragments_equal_the_python_path(b17):
    """VERDICT r4 #2's size: 10^6 fragments (1.4 * 10^6 alignments) -- minutes of interpreter time for the python path."""
    from mixemt_amd import synth
    refseq, phy, haps, tables = b17
    cols = synth.synth_alignments(tables, refseq, 1000000, seed=5)
    _same_as_python(from_columns(cols), phy.get_variant_pos(), len(refseq), 30, 30, cols=cols)
