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


# ---- g11: the reference's own front end RUN on synthetic alignments (tools/gen_golden.py) -------------------
def _g11():
    import json
    from conftest import golden
    g = golden("g11_frontend")
    alns = [FakeAln(*rec) for rec in json.loads(str(g["alns"]))]
    return g, alns


def test_process_and_reduce_reads_equal_the_reference_run(b17):
    """preprocess.py:99-139, :163-174 on 454 alignments: mates, conflicting overlaps ('N' -> dropped), low mapping /
    base qualities, missing quality arrays, indels, soft clips, lower-case bases, duplicates, and one fragment
    left with no site at all (empty signature)."""
    import json
    refseq, phy, haps, tables = b17
    g, alns = _g11()
    got = preprocess.process_reads(alns, phy.get_variant_pos(), int(g["min_mq"]), int(g["min_bq"]))
    want = {name: {int(p): b for p, b in obs.items()} for name, obs in json.loads(str(g["read_obs"])).items()}
    assert got == want
    assert got[str(g["empty_name"])] == {}
    sigs = preprocess.reduce_reads(got)
    assert {k: list(v) for k, v in sigs.items()} == json.loads(str(g["read_sigs"]))
    assert any(len(v) > 1 for v in sigs.values())                     # duplicates really collapse
    # the fixture exercises what it claims to
    assert any(a.query_qualities is None for a in alns) and any("I" in a.cigarstring for a in alns)
    assert any("D" in a.cigarstring for a in alns) and any("S" in a.cigarstring for a in alns)
    assert any(a.query_sequence.islower() for a in alns) and any(a.mapping_quality < 30 for a in alns)


@pytest.mark.gpu
@pytest.mark.parametrize("frontend", ["batched", "python"])
@pytest.mark.parametrize("as_records", [False, True])
def test_build_em_input_equals_the_reference_run(b17, as_records, frontend, capsys):
    """preprocess.py:201-227 end to end: rows = sorted distinct signatures, weights, read-id lists, haplogroup order
    and the matrix itself (sha256 of the reference's) -- dense and as row-dictionary records; the fragment the
    reference dies on (recorded in the fixture) is skipped with a warning."""
    import argparse
    import hashlib
    import json
    import torch
    refseq, phy, haps, tables = b17
    g, alns = _g11()
    assert str(g["ref_died_with"]).startswith("ValueError")
    args = argparse.Namespace(min_mq=int(g["min_mq"]), min_bq=int(g["min_bq"]), verbose=False)
    mat, wts, hap_order, read_ids = preprocess.build_em_input(FakeBam(alns), refseq, phy, args, as_records=as_records,
                                                              frontend=frontend)
    assert preprocess.build_em_input.last_frontend == frontend
    assert preprocess.build_em_input.last_dropped == [str(g["empty_name"])]
    assert "skipped 1 fragment" in capsys.readouterr().err
    assert hashlib.sha256("\n".join(hap_order).encode()).hexdigest() == str(g["hap_sha256"])
    assert numpy.array_equal(wts, g["weights"])
    assert read_ids == json.loads(str(g["read_ids"]))
    rows = str(g["signatures"]).split("\n")
    if as_records:
        # decode the records: P = exp(M - rowmax) bit for bit what mxm_linearize makes of the reference's matrix;
        # the log tables give the matrix itself
        cm = mat
        assert cm.n_rows == len(rows) and cm.n_haps == len(hap_order)
        cols = torch.arange(cm.n_haps, dtype=torch.int32, device="cuda")
        dense, _ = preprocess.reduce_em_records(cm, hap_order, [[None, h, 0.0] for h in hap_order])
        host = dense.cpu().numpy()
    else:
        host = mat
    assert host.shape == (len(rows), len(hap_order))
    assert hashlib.sha256(numpy.ascontiguousarray(host).tobytes()).hexdigest() == str(g["mat_sha256"])
    assert numpy.array_equal(host[:3], g["mat_rows"])


@pytest.mark.gpu
def test_build_em_input_from_a_bam_file_equals_the_reference_run(b17, tmp_path, capsys):
    """The same run started from a BAM FILE (golden g11's alignments written by tests/_bam_writer.py): the library's reader
    in front of the encoder gives the reference's matrix, weights and read-id lists."""
    import argparse
    import hashlib
    import json
    import _bam_writer
    from mixemt_amd import alignments
    refseq, phy, haps, tables = b17
    g, alns = _g11()
    path = str(tmp_path / "g11.bam")
    _bam_writer.write_bam(path, alignments.AlignmentColumns.from_alignments(alns), block_bytes=5000)
    args = argparse.Namespace(min_mq=int(g["min_mq"]), min_bq=int(g["min_bq"]), verbose=False)
    mat, wts, hap_order, read_ids = preprocess.build_em_input(path, refseq, phy, args)
    assert preprocess.build_em_input.last_frontend == "batched"
    assert preprocess.build_em_input.last_dropped == [str(g["empty_name"])]
    assert "skipped 1 fragment" in capsys.readouterr().err
    assert numpy.array_equal(wts, g["weights"])
    assert read_ids == json.loads(str(g["read_ids"]))
    assert hashlib.sha256(numpy.ascontiguousarray(mat).tobytes()).hexdigest() == str(g["mat_sha256"])
    with pytest.raises(ValueError, match="batched front end"):
        preprocess.build_em_input(path, refseq, phy, args, frontend="python")


@pytest.mark.gpu
def test_an_open_handle_that_names_its_file_is_read_by_the_library(b17, tmp_path, capsys):
    """bin/mixemt:139-147 hands build_em_input an open pysam.AlignmentFile; it carries `.filename`, and the batched front end
    reads that file itself instead of iterating fetch() -- same matrix, weights and ids; a handle whose file the reader does
    not take (here: SAM text) still goes through its objects."""
    import argparse
    import hashlib
    import json
    import _bam_writer
    from mixemt_amd import alignments
    refseq, phy, haps, tables = b17
    g, alns = _g11()
    path = str(tmp_path / "g11.bam")
    _bam_writer.write_bam(path, alignments.AlignmentColumns.from_alignments(alns))

    class Handle(FakeBam):
        def __init__(self, alns, filename):
            FakeBam.__init__(self, alns)
            self.filename, self.fetched = filename, 0

        def fetch(self):
            self.fetched += 1
            return FakeBam.fetch(self)

    args = argparse.Namespace(min_mq=int(g["min_mq"]), min_bq=int(g["min_bq"]), verbose=False)
    for name, want_source, want_fetch in ((path.encode(), "file of the handle", 0), (str(tmp_path / "reads.sam"), "objects", 1)):
        if want_fetch:
            open(name, "w").write("@HD\tVN:1.6\n")
        handle = Handle(alns, name)
        mat, wts, hap_order, read_ids = preprocess.build_em_input(handle, refseq, phy, args)
        assert preprocess.build_em_input.last_source == want_source and handle.fetched == want_fetch
        assert numpy.array_equal(wts, g["weights"]) and read_ids == json.loads(str(g["read_ids"]))
        assert hashlib.sha256(numpy.ascontiguousarray(mat).tobytes()).hexdigest() == str(g["mat_sha256"])
    capsys.readouterr()


# ---- g12: the -s / -l files as the reference's own dump_all writes them and its load_prev reads them ----------------
def _g12():
    import json
    from conftest import golden
    g = golden("g12_io_formats")
    return g, str(g["haps"]).split("\n"), json.loads(str(g["reads"]))


def _files(prefix):
    out = {}
    for ext in ("haps", "reads", "em.npy", "mat.npy", "prop.npy"):
        with open("%s.%s" % (prefix, ext), "rb") as fin:
            out["file_" + ext.replace(".", "_")] = numpy.frombuffer(fin.read(), dtype=numpy.uint8)
    return out


def test_dump_all_writes_the_reference_bytes_and_load_prev_reads_them(tmp_path):
    """bin/mixemt:168-245 run by the reference itself (tools/gen_golden.py, g12): our dump_all writes the same five
    files byte for byte (numpy's .npy header included), our load_prev returns what the reference's returns from the
    reference's files -- either tool can resume the other's run."""
    g, haps, reads = _g12()
    prefix = str(tmp_path / "ours")
    mio.dump_all(prefix, haps, reads, g["em"], (g["props"], g["mix"]))
    for key, mine in _files(prefix).items():
        assert numpy.array_equal(mine, g[key]), key
    ref_prefix = str(tmp_path / "theirs")
    for key in ("haps", "reads", "em.npy", "mat.npy", "prop.npy"):
        with open("%s.%s" % (ref_prefix, key), "wb") as fout:
            fout.write(g["file_" + key.replace(".", "_")].tobytes())
    import json
    h2, r2, wts, init, (props, mat) = mio.load_prev(ref_prefix)
    assert h2 == str(g["loaded_haps"]).split("\n") and r2 == json.loads(str(g["loaded_reads"]))
    assert numpy.array_equal(wts, g["loaded_wts"]) and numpy.array_equal(init, g["loaded_init"])
    assert numpy.array_equal(props, g["loaded_props"]) and numpy.array_equal(mat, g["loaded_mat"])


@pytest.mark.gpu
def test_dump_all_from_device_tensors_writes_the_reference_bytes(tmp_path):
    """The same files streamed from matrices that live on the GPU (row slabs through a memory map): same bytes."""
    import torch
    g, haps, reads = _g12()
    prefix = str(tmp_path / "dev")
    old = mio.SLAB_BYTES
    mio.SLAB_BYTES = 64                       # two rows per slab: the slab loop really loops
    try:
        mio.dump_all(prefix, haps, reads, torch.from_numpy(g["em"]).cuda(),
                     (torch.from_numpy(g["props"]).cuda(), torch.from_numpy(g["mix"]).cuda()))
    finally:
        mio.SLAB_BYTES = old
    for key, mine in _files(prefix).items():
        assert numpy.array_equal(mine, g[key]), key
