"""
Host front end (row f-4): alignments -> observations -> signatures, and the
save/load formats.  The reference's own expectations (preprocess_test.py:126-241)
restated on a pysam stand-in.  CPU only.
"""
import numpy
import pytest

from _fake_aln import FakeAln, FakeBam
from mixemt_amd import io as mio
from mixemt_amd import preprocess


def _alns():
    aln1 = FakeAln("read1", 10, 30, "AAAAATAAAATAAAAT", [30] * 16, "16M")
    qq = [33] * 12
    qq[3] = 20
    aln2 = FakeAln("read2", 12, 20, "AAAGAAGAAAAG", qq, "5M2D7M")
    aln3 = FakeAln("read3", 0, 0)
    return [aln1, aln2, aln3]


def test_process_reads_reference_cases():
    """preprocess_test.py:126-141."""
    alns = _alns()
    assert preprocess.process_reads(alns, [15, 20, 25], 20, 10) == {
        "read1": {15: "T", 20: "T", 25: "T"}, "read2": {15: "G", 20: "G", 25: "G"}}
    assert preprocess.process_reads(alns, [15, 20, 25], 25, 10) == {
        "read1": {15: "T", 20: "T", 25: "T"}}
    assert preprocess.process_reads(alns, [15, 20, 25], 20, 30) == {
        "read1": {15: "T", 20: "T", 25: "T"}, "read2": {20: "G", 25: "G"}}


def test_process_reads_paired_end_cases():
    """preprocess_test.py:143-196: mates merge; a conflicting overlap is dropped; a low-quality
    base does not conflict."""
    alns = _alns() + [FakeAln("read1", 30, 30, "AAAAACAAAACAAAAT", [30] * 16, "16M")]
    assert preprocess.process_reads(alns, [15, 20, 25, 35, 40], 20, 10) == {
        "read1": {15: "T", 20: "T", 25: "T", 35: "C", 40: "C"}, "read2": {15: "G", 20: "G", 25: "G"}}
    alns = _alns() + [FakeAln("read1", 20, 20, "AAAAATAAAACAAAAT", [30] * 16, "16M")]
    assert preprocess.process_reads(alns, [15, 20, 25, 35], 20, 10) == {
        "read1": {15: "T", 25: "T", 35: "T"}, "read2": {15: "G", 20: "G", 25: "G"}}
    qq = [30] * 16
    qq[0] = 5
    alns = _alns() + [FakeAln("read1", 20, 20, "AAAAATAAAACAAAAC", qq, "16M")]
    assert preprocess.process_reads(alns, [15, 20, 25, 35], 20, 10) == {
        "read1": {15: "T", 20: "T", 25: "T", 35: "C"}, "read2": {15: "G", 20: "G", 25: "G"}}


def test_missing_qualities_are_accepted():
    aln = FakeAln("r", 10, 60, "ACGTACGT", None, "8M")
    assert preprocess.process_reads([aln], [11, 12], 20, 30) == {"r": {11: "C", 12: "G"}}


def test_signatures_and_reduction():
    """preprocess_test.py:199-241."""
    obs = {1: "A", 2: "C", 3: "G", 4: "T"}
    sig = preprocess.read_signature(obs)
    assert sig == "1:A,2:C,3:G,4:T"
    assert dict(preprocess.pos_obs_from_sig(sig)) == obs
    with pytest.raises(TypeError):
        preprocess.read_signature({"A": "A", 2: "C"})
    with pytest.raises(TypeError):
        preprocess.read_signature("1:A,2:C")
    reads = {"read1": {1: "A", 2: "C"}, "read2": {3: "G", 4: "T"}, "read3": {2: "C", 1: "A"}}
    assert preprocess.reduce_reads(reads) == {"1:A,2:C": ["read1", "read3"], "3:G,4:T": ["read2"]}
    reads["read3"] = {2: "C", 1: "T"}
    assert preprocess.reduce_reads(reads) == {"1:A,2:C": ["read1"], "3:G,4:T": ["read2"],
                                              "1:T,2:C": ["read3"]}


def test_dump_and_load_roundtrip(tmp_path):
    """bin/mixemt:168-245 formats."""
    prefix = str(tmp_path / "run")
    haps = ["A", "B'c", "D/E"]
    reads = [["r1", "r2"], ["r3"]]
    em_mat = numpy.arange(6.0).reshape(2, 3)
    mix = -em_mat
    props = numpy.array([0.5, 0.25, 0.25])
    mio.dump_all(prefix, haps, reads, em_mat, (props, mix))
    assert open(prefix + ".reads").read() == "0\tr1\tr2\n1\tr3\n"
    h2, r2, wts, init, (p2, m2) = mio.load_prev(prefix)
    assert h2 == haps and r2 == reads and list(wts) == [2, 1]
    assert numpy.array_equal(init, em_mat) and numpy.array_equal(m2, mix) and numpy.array_equal(p2, props)
    # a missing .em.npy only disables refinement
    import os
    os.remove(prefix + ".em.npy")
    assert mio.load_prev(prefix)[3] is None
    os.remove(prefix + ".prop.npy")
    with pytest.raises(ValueError):
        mio.load_prev(prefix)
