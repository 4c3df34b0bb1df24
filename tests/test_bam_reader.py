"""
The library's BAM reader (mxm_bam_read, csrc/bam_reader.hpp -> alignments.read_bam): the columns it hands the batched
front end are what pysam hands the reference's (bin/mixemt:139-147, preprocess.py:209, :118-132).  pysam / samtools are
not installed here, so the files come from tests/_bam_writer.py (written from the SAM/BAM specification, and checked
below to be plain multi-member gzip that Python's own gzip module inflates to the same stream).  CPU only.
"""
import gzip
import json
import os

import numpy
import pytest

import _bam_writer as bw
from _fake_aln import FakeAln
from mixemt_amd import alignments, synth


def _cols(alns):
    return alignments.AlignmentColumns.from_alignments(alns)


def _same_columns(got, want, upper=True):
    assert len(got) == len(want)
    for key in ("ref_start", "mapq", "cig_ptr", "cigar", "seq_ptr"):
        assert numpy.array_equal(getattr(got, key), getattr(want, key)), key
    seq = want.seq
    if upper:
        seq = numpy.frombuffer(want.seq.tobytes().upper(), dtype=numpy.uint8)
    assert numpy.array_equal(got.seq, seq)
    want_has = want.has_qual if want.has_qual is not None else numpy.ones(len(want), dtype=numpy.uint8)
    # an alignment without bases has no quality bytes to carry the 0xFF marker: pysam reports None for it as well
    empty = numpy.diff(want.seq_ptr) == 0
    assert numpy.array_equal(got.has_qual, numpy.where(empty, 0, want_has))
    if want.qual is not None:
        own = numpy.repeat(got.has_qual != 0, numpy.diff(got.seq_ptr))
        assert numpy.array_equal(got.qual[own], want.qual[own])
    # fragments: the same partition of the alignments, the same names, numbered by first appearance
    assert [got.names[int(f)] for f in got.frag] == [want.names[int(f)] for f in want.frag]
    first = {}
    for f in got.frag:
        first.setdefault(int(f), len(first))
    assert all(k == v for k, v in first.items())


SMALL = [
    FakeAln("read1", 10, 30, "AAAAATAAAATAAAAT", [30] * 16, "16M"),
    FakeAln("read2", 12, 20, "AAAGAAGAAAAG", [33, 33, 33, 20] + [33] * 8, "5M2D7M"),
    FakeAln("read3", 0, 0, "", None, ""),                                    # no bases, no CIGAR ('*' / '*')
    FakeAln("read1", 30, 60, "ACGTNACGTRYKMA", None, "2S3M1I4M2N2=1X1H"),    # mate: same name; no qualities; odd ops
    FakeAln("r5", 16000, 255, "ACGTACG", [0, 1, 2, 93, 40, 41, 254], "7M"),  # odd length, quality extremes
]


def test_round_trip_small(tmp_path):
    cols = _cols(SMALL)
    path = str(tmp_path / "small.bam")
    stream = bw.write_bam(path, cols)
    with gzip.open(path, "rb") as fin:                      # BGZF is multi-member gzip: an independent inflater agrees
        assert fin.read() == stream
    got = alignments.read_bam(path)
    _same_columns(got, cols)
    assert list(got.names) == ["read1", "read2", "read3", "r5"]
    assert got.bam_counts == (5, 0)
    assert got.seq.tobytes().decode() == "AAAAATAAAATAAAAT" "AAAGAAGAAAAG" "ACGTNACGTRYKMA" "ACGTACG"
    assert got.ref_id.tolist() == [0] * 5 and got.flag.tolist() == [0] * 5


@pytest.mark.parametrize("block_bytes,threads", [(37, 1), (100, 3), (4096, 4), (65280, 2)])
def test_records_straddling_bgzf_blocks(tmp_path, b17, block_bytes, threads):
    refseq, phy, haps, tables = b17
    cols = synth.synth_alignments(tables, refseq, 700, seed=11)
    path = str(tmp_path / "synth.bam")
    bw.write_bam(path, cols, block_bytes=block_bytes, level=1)
    got = alignments.read_bam(path, n_threads=threads)
    _same_columns(got, cols)


def test_unplaced_records_references_and_flags(tmp_path):
    cols = _cols(SMALL)
    unplaced = bw.record("lost", -1, -1, 0, 4, [], b"ACGT", b"\x1e" * 4)
    placed_unmapped = bw.record("mate_of_r5", 1, 99, 0, 4 | 8, [], b"GGCC", None)      # placed at its mate's position
    path = str(tmp_path / "mixed.bam")
    bw.write_bam(path, cols, refs=(("chrM", 16569), ("other", 5000)), ref_id=[0, 0, 0, 1, 1], flag=[99, 147, 4, 0x900, 0x400],
                 extra_records=[(2, unplaced), (4, placed_unmapped), (99, unplaced)])
    got = alignments.read_bam(path)
    assert got.bam_counts == (8, 2)
    assert len(got) == 6
    assert list(got.names) == ["read1", "read2", "read3", "mate_of_r5", "r5"]
    assert got.ref_id.tolist() == [0, 0, 0, 1, 1, 1]
    assert got.flag.tolist() == [99, 147, 4, 0x900, 12, 0x400]
    assert got.ref_start.tolist() == [10, 12, 0, 30, 99, 16000]
    assert numpy.diff(got.cig_ptr).tolist() == [1, 3, 0, 8, 0, 1]
    assert got.has_qual.tolist() == [1, 1, 0, 0, 0, 1]


def test_empty_file_and_header_only(tmp_path):
    path = str(tmp_path / "empty.bam")
    bw.write_bam(path, _cols([]))
    got = alignments.read_bam(path)
    assert len(got) == 0 and got.n_frag == 0 and got.bam_counts == (0, 0)
    path2 = str(tmp_path / "no_eof.bam")
    bw.write_stream(path2, bw.header([("chrM", 16569)]), eof=False)
    assert len(alignments.read_bam(path2)) == 0


def test_what_the_reader_refuses(tmp_path):
    with pytest.raises(OSError):
        alignments.read_bam(str(tmp_path / "missing.bam"))
    plain = tmp_path / "plain.gz"
    with gzip.open(str(plain), "wb") as fout:
        fout.write(b"BAM\1" + b"\0" * 8)
    with pytest.raises(ValueError, match="not a BGZF file"):
        alignments.read_bam(str(plain))
    sam = tmp_path / "reads.sam"
    sam.write_text("@HD\tVN:1.6\n")
    with pytest.raises(ValueError, match="not a BGZF file"):
        alignments.read_bam(str(sam))
    notbam = tmp_path / "notbam.bgz"
    bw.write_stream(str(notbam), b"hello world, this is not a BAM stream")
    with pytest.raises(ValueError, match="no BAM magic"):
        alignments.read_bam(str(notbam))
    cols = _cols(SMALL)
    good = tmp_path / "good.bam"
    stream = bw.write_bam(str(good), cols)
    cut = tmp_path / "cut.bam"
    bw.write_stream(str(cut), stream[:-9])                                   # the last record loses its tail
    with pytest.raises(ValueError, match="truncated alignment record"):
        alignments.read_bam(str(cut))
    raw = good.read_bytes()
    (tmp_path / "short.bam").write_bytes(raw[:len(raw) // 2])                # a BGZF block cut in the middle
    with pytest.raises(ValueError, match="BGZF"):
        alignments.read_bam(str(tmp_path / "short.bam"))
    damaged = bytearray(raw)
    damaged[40] ^= 0x5a                                                      # inside the first block's deflate stream
    (tmp_path / "damaged.bam").write_bytes(bytes(damaged))
    with pytest.raises(ValueError, match="BGZF"):
        alignments.read_bam(str(tmp_path / "damaged.bam"))
    long_cigar = bw.record("huge", 0, 5, 60, 0, [(8 << 4) | 4, (70000 << 4) | 3], b"ACGTACGT", None)
    bw.write_stream(str(tmp_path / "cg.bam"), bw.header([("chrM", 16569)]) + long_cigar)
    with pytest.raises(ValueError, match="CG tag"):
        alignments.read_bam(str(tmp_path / "cg.bam"))


def test_g11_through_a_bam_file(tmp_path, b17):
    """Golden g11's 454 alignments written as BAM, read back by the library and encoded: the reference's own rows."""
    from conftest import golden
    refseq, phy, haps, tables = b17
    g = golden("g11_frontend")
    alns = [FakeAln(*rec) for rec in json.loads(str(g["alns"]))]
    path = str(tmp_path / "g11.bam")
    bw.write_bam(path, _cols(alns), block_bytes=3000)
    enc = alignments.encode_alignments(alignments.read_bam(path), phy.get_variant_pos(), len(refseq), int(g["min_mq"]),
                                       int(g["min_bq"]))
    assert enc.signatures() == str(g["signatures"]).split("\n")
    assert numpy.array_equal(enc.weights, g["weights"])
    assert enc.read_ids == json.loads(str(g["read_ids"]))
    assert enc.dropped == [str(g["empty_name"])]
    want = {name: {int(p): b for p, b in obs.items()} for name, obs in json.loads(str(g["read_obs"])).items()}
    assert enc.read_obs(numpy.asarray(phy.get_variant_pos())) == want


def test_encoded_rows_equal_those_of_the_columns(tmp_path, b17):
    refseq, phy, haps, tables = b17
    cols = synth.synth_alignments(tables, refseq, 5000, seed=12)
    path = str(tmp_path / "s.bam")
    bw.write_bam(path, cols)
    assert os.path.getsize(path) > 100000
    a = alignments.encode_alignments(alignments.read_bam(path), tables.sites, len(refseq), 30, 30)
    b = alignments.encode_alignments(cols, tables.sites, len(refseq), 30, 30)
    for key in ("row_ptr", "site", "obs", "weights"):
        assert numpy.array_equal(getattr(a, key), getattr(b, key)), key
    assert a.read_ids == b.read_ids and a.dropped == b.dropped and a.signatures() == b.signatures()


def test_views_outlive_the_columns_object(tmp_path, b17):
    """read_bam's arrays are views of the library's own; each keeps the handle alive."""
    import gc
    refseq, phy, haps, tables = b17
    cols = synth.synth_alignments(tables, refseq, 300, seed=13)
    path = str(tmp_path / "v.bam")
    bw.write_bam(path, cols)
    got = alignments.read_bam(path)
    seq, cigar = got.seq, got.cigar
    assert not seq.flags.writeable
    del got
    gc.collect()
    junk = [numpy.ones(1 << 20, dtype=numpy.uint8) for _ in range(8)]              # (reuse freed memory, if any was)
    assert numpy.array_equal(seq, numpy.frombuffer(cols.seq.tobytes().upper(), dtype=numpy.uint8))
    assert numpy.array_equal(cigar, cols.cigar)
    del junk


def test_reader_under_address_sanitizer(tmp_path, b17):
    """The reader parses files from outside: built host-only with ASan + UBSan and run over good, cut and damaged files
    (every BAM field that sizes something set to extremes, random byte flips in the record stream)."""
    import shutil
    import struct
    import subprocess
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "harness")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
           "-I" + os.path.join(root, "include"), "-I" + os.path.join(root, "mixemt_amd", "csrc"),
           os.path.join(root, "tests", "native", "bam_reader_harness.cpp"), "-o", exe, "-lz", "-lpthread"]
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert proc.returncode == 0, proc.stdout
    refseq, phy, haps, tables = b17
    cols = synth.synth_alignments(tables, refseq, 120, seed=14)
    good = str(tmp_path / "good.bam")
    stream = bw.write_bam(good, cols, block_bytes=1500)
    files = [good]
    big = str(tmp_path / "big.bam")                                # enough records for the reader's threaded stages
    big_cols = synth.synth_alignments(tables, refseq, 20000, seed=16)   # (enough for every threaded stage of reader and encoder)
    assert len(big_cols) >= 4096
    bw.write_bam(big, big_cols, level=1)
    raw = open(good, "rb").read()
    rng = numpy.random.default_rng(15)
    head = len(bw.header([("chrM", 16569)]))
    for k in range(60):                                           # damaged record streams, valid BGZF around them
        bad = bytearray(stream)
        for _ in range(int(rng.integers(1, 4))):
            at = int(rng.integers(4, len(bad)))
            bad[at] = int(rng.integers(0, 256))
        name = str(tmp_path / ("flip%d.bam" % k))
        bw.write_stream(name, bytes(bad), block_bytes=int(rng.integers(50, 4000)))
        files.append(name)
    for k, (off, fmt, val) in enumerate([(head, "<i", 0x7fffffff), (head, "<i", -1), (head, "<i", 31), (head + 4 + 8, "<B", 0),
                                         (head + 4 + 8, "<B", 255), (head + 4 + 12, "<H", 65535), (head + 4 + 16, "<i", 0x7fffffff),
                                         (head + 4 + 16, "<i", -5), (head + 4 + 16, "<I", 0xaaaaaaab), (head + 4 + 16, "<I", 0xaaaaaaaa + 60), (4, "<i", 0x7ffffff0), (4, "<i", -1),
                                         (head - 4 - 5 - 4, "<i", 0x7fffffff), (head - 4 - 4 - 5 - 4 - 4, "<i", 1 << 30)]):
        bad = bytearray(stream)
        bad[off:off + struct.calcsize(fmt)] = struct.pack(fmt, val)
        name = str(tmp_path / ("field%d.bam" % k))
        bw.write_stream(name, bytes(bad), block_bytes=700)
        files.append(name)
    for k in range(30):                                           # damaged files (headers, deflate streams, trailers)
        bad = bytearray(raw)
        if k % 3 == 0:
            bad = bad[:int(rng.integers(0, len(bad)))]
        else:
            for _ in range(int(rng.integers(1, 5))):
                bad[int(rng.integers(0, len(bad)))] = int(rng.integers(0, 256))
        name = str(tmp_path / ("file%d.bam" % k))
        open(name, "wb").write(bytes(bad))
        files.append(name)
    open(str(tmp_path / "zero.bam"), "wb").close()
    files.append(str(tmp_path / "zero.bam"))
    proc = subprocess.run([exe] + files + [big], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert proc.returncode == 0, proc.stdout[-4000:]
    lines = proc.stdout.strip().split("\n")
    assert len(lines) == 2 * (len(files) + 1)
    assert " rc=0 n_aln=%d n_frag=%d " % (len(big_cols), big_cols.n_frag) in lines[-1]          # three threads
    assert lines[-1].split("check=")[1].strip() == lines[-2].split("check=")[1].strip()         # ... as one
    assert " rc=0 n_aln=%d " % len(cols) in lines[0] and lines[0].split("check=")[1] == lines[1].split("check=")[1]
    assert sum(" rc=-4 " in ln for ln in lines) > 20              # most damage is noticed; none of it crashes
    # the threaded stages of the reader AND of the batched encoder (the harness runs both) under ThreadSanitizer
    exe_t = str(tmp_path / "harness_tsan")
    cmd_t = [c if c != "-fsanitize=address,undefined" else "-fsanitize=thread" for c in cmd[:-3]] + [exe_t, "-lz", "-lpthread"]
    cmd_t = [c for c in cmd_t if c != "-fno-sanitize-recover=all"]
    proc = subprocess.run(cmd_t, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if proc.returncode == 0:
        run = subprocess.run([exe_t, big], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
        if "unexpected memory mapping" not in run.stdout:         # (a kernel setting some hosts have: not the code's)
            assert run.returncode == 0 and "ThreadSanitizer" not in run.stdout, run.stdout[-3000:]
            out = run.stdout.strip().split("\n")
            assert out[-1].split("check=")[1].strip() == out[-2].split("check=")[1].strip()
