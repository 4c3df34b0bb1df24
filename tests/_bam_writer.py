"""
A minimal BAM writer for the reader's tests (SAM/BAM specification, sections 4.1 BGZF and 4.2 the alignment records;
pysam / samtools are not installed here).  Test infrastructure only: it writes what `samtools view -b` would for the
fields the front end reads, and fixed values for the rest (no mate, no tags).
"""
import struct
import zlib

import numpy

SEQ_CODES = "=ACMGRSVTWYHKDBN"
_CODE_OF = numpy.full(256, 15, dtype=numpy.uint8)
for _i, _c in enumerate(SEQ_CODES):
    _CODE_OF[ord(_c)] = _i
    _CODE_OF[ord(_c.lower())] = _i

BGZF_EOF = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")


def bgzf_block(payload, level=6):
    comp = zlib.compressobj(level, zlib.DEFLATED, -15)
    cdata = comp.compress(payload) + comp.flush()
    bsize = 12 + 6 + len(cdata) + 8
    assert bsize <= 65536 and len(payload) <= 65536
    return (b"\x1f\x8b\x08\x04" + struct.pack("<IBBH", 0, 0, 0xff, 6) + b"BC" + struct.pack("<HH", 2, bsize - 1) + cdata
            + struct.pack("<II", zlib.crc32(payload) & 0xffffffff, len(payload)))


def reg2bin(beg, end):
    end -= 1
    for shift, base in ((14, 4681), (17, 585), (20, 73), (23, 9), (26, 1)):
        if beg >> shift == end >> shift:
            return base + (beg >> shift)
    return 0


def record(name, ref_id, pos, mapq, flag, cigar, seq, qual, tags=b""):
    """One alignment record (block_size included).  cigar: uint32 values (len << 4 | op); seq: bytes; qual: bytes, or None
    for 'absent' (0xFF throughout, section 4.2.3)."""
    l_seq = len(seq)
    codes = _CODE_OF[numpy.frombuffer(seq, dtype=numpy.uint8)] if l_seq else numpy.zeros(0, dtype=numpy.uint8)
    if l_seq & 1:
        codes = numpy.append(codes, numpy.uint8(0))
    packed = ((codes[0::2] << 4) | codes[1::2]).astype(numpy.uint8).tobytes()
    ref_span = sum(int(c) >> 4 for c in cigar if (int(c) & 15) in (0, 2, 3, 7, 8))
    qbytes = b"\xff" * l_seq if qual is None else bytes(bytearray(qual))
    assert len(qbytes) == l_seq
    rname = name.encode("ascii") + b"\0"
    body = (struct.pack("<iiBBHHHiiii", ref_id, pos, len(rname), mapq, reg2bin(max(pos, 0), max(pos, 0) + max(ref_span, 1)),
                        len(cigar), flag, l_seq, -1, -1, 0)
            + rname + b"".join(struct.pack("<I", int(c)) for c in cigar) + packed + qbytes + tags)
    return struct.pack("<i", len(body)) + body


def header(refs, text="@HD\tVN:1.6\tSO:coordinate\n"):
    text = text + "".join("@SQ\tSN:%s\tLN:%d\n" % r for r in refs)
    out = b"BAM\1" + struct.pack("<i", len(text)) + text.encode("ascii") + struct.pack("<i", len(refs))
    for name, length in refs:
        out += struct.pack("<i", len(name) + 1) + name.encode("ascii") + b"\0" + struct.pack("<i", length)
    return out


def write_stream(path, stream, block_bytes=65280, level=6, eof=True):
    """The uncompressed BAM stream cut into BGZF blocks of `block_bytes` (records may straddle blocks, as htslib's do
    for long records; small values exercise that)."""
    with open(path, "wb") as fout:
        for a in range(0, len(stream), block_bytes):
            fout.write(bgzf_block(stream[a:a + block_bytes], level))
        if eof:
            fout.write(BGZF_EOF)


def write_bam(path, cols, refs=(("chrM", 16569),), ref_id=None, flag=None, extra_records=(), block_bytes=65280, level=6):
    """alignments.AlignmentColumns -> a BAM file.  extra_records: (index, record bytes) inserted BEFORE alignment `index`
    (or at the end for index >= len)."""
    parts = [header(list(refs))]
    extra = sorted(extra_records, key=lambda e: e[0])
    e = 0
    seq = cols.seq.tobytes()
    for i in range(len(cols)):
        while e < len(extra) and extra[e][0] <= i:
            parts.append(extra[e][1])
            e += 1
        a, b = int(cols.seq_ptr[i]), int(cols.seq_ptr[i + 1])
        qual = None
        if cols.qual is not None and (cols.has_qual is None or cols.has_qual[i]):
            qual = cols.qual[a:b].tobytes()
        cig = cols.cigar[int(cols.cig_ptr[i]):int(cols.cig_ptr[i + 1])]
        parts.append(record(cols.names[int(cols.frag[i])], 0 if ref_id is None else int(ref_id[i]), int(cols.ref_start[i]),
                            int(cols.mapq[i]), 0 if flag is None else int(flag[i]), cig, seq[a:b], qual))
    parts.extend(rec for _, rec in extra[e:])
    stream = b"".join(parts)
    write_stream(path, stream, block_bytes, level)
    return stream
