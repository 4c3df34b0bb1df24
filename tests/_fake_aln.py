"""
Minimal stand-in for pysam.AlignedSegment / AlignmentFile for the front-end
tests (the reference's own tests build pysam objects in memory,
preprocess_test.py:102-124; pysam is not installed here).  Supports M/=/X, I,
D/N, S cigar operations -- what get_aligned_pairs(matches_only=True) needs.
"""
import re


class FakeAln(object):
    def __init__(self, name, start=0, mq=0, seq=None, quals=None, cigar=None):
        self.query_name = name
        self.reference_start = start
        self.mapping_quality = mq
        self.query_sequence = seq
        self.query_qualities = quals
        self.cigarstring = cigar

    def get_aligned_pairs(self, matches_only=False):
        assert matches_only
        pairs, q, r = [], 0, self.reference_start
        for n, op in re.findall(r"(\d+)([MIDNS=X])", self.cigarstring or ""):
            n = int(n)
            if op in "M=X":
                pairs.extend((q + i, r + i) for i in range(n))
                q += n
                r += n
            elif op in "IS":
                q += n
            else:
                r += n
        return pairs


class FakeBam(object):
    def __init__(self, alns):
        self.alns = list(alns)

    def fetch(self):
        return iter(self.alns)


def from_columns(cols):
    """alignments.AlignmentColumns -> list of FakeAln (the same alignments, one object each)."""
    ops = "MIDNSHP=XB"
    out = []
    seq = cols.seq.tobytes().decode("ascii")
    for i in range(len(cols)):
        a, b = int(cols.seq_ptr[i]), int(cols.seq_ptr[i + 1])
        quals = None
        if cols.qual is not None and (cols.has_qual is None or cols.has_qual[i]):
            quals = cols.qual[a:b].tolist()
        cig = "".join("%d%s" % (int(c) >> 4, ops[int(c) & 15])
                      for c in cols.cigar[int(cols.cig_ptr[i]):int(cols.cig_ptr[i + 1])])
        out.append(FakeAln(cols.names[int(cols.frag[i])], int(cols.ref_start[i]), int(cols.mapq[i]), seq[a:b], quals, cig))
    return out
