"""
synth-v1: the synthetic read generator behind bench.py and the size-scaled
parity tests (SURVEY.md section 8 d).  The build's own code; the reference has
no generator (its tests use hand-written signatures, em_test.py:89-92).

A mixture of three contributors -- columns 10, 2000, 4000 of sorted(hap_var)
(= A12a, H63, M9a'b for Build 17 + RSRS) at 0.6 / 0.3 / 0.1 -- sheds reads of
L = 150 bp at uniform starts; a read observes every variant site it covers and
shows its contributor's expected base, or with probability e = 0.005 one of
the other three.  One matrix row per read, weight 1, generation order.

Everything is vectorised numpy on `numpy.random.default_rng(seed)` (PCG64); the
draw order below IS the definition of synth-v1:
    1. contributor index per read      rng.choice(len(props), R, p=props)
    2. start per read                  rng.integers(0, len(refseq) - L, R)
       (reads covering no site are redrawn, all of them at once, until none)
    3. error flag per observation      rng.random(nnz) < e
    4. substitution offset per obs.    rng.integers(1, 4, nnz)
"""

import numpy

ALPHABET = numpy.frombuffer(b"ACGT", dtype=numpy.uint8)
DEFAULT_CONTRIB = (10, 2000, 4000)
DEFAULT_PROPS = (0.6, 0.3, 0.1)
READ_LEN = 150
ERR_RATE = 0.005


def synth_reads(tables, ref_len, n_reads, seed=1, contrib=DEFAULT_CONTRIB,
                props=DEFAULT_PROPS, read_len=READ_LEN, err=ERR_RATE):
    """
    -> (row_ptr[R+1] int64, site[nnz] uint16, obs[nnz] uint8, who[R] int64)
    CSR observations in the encoding of preprocess.encode_signatures, plus the
    contributor each read was drawn from.
    """
    rng = numpy.random.default_rng(seed)
    sites = tables.sites
    who = rng.choice(len(props), size=n_reads, p=numpy.asarray(props, dtype=float))
    start = rng.integers(0, ref_len - read_len, size=n_reads)
    lo = numpy.searchsorted(sites, start, side="left")
    hi = numpy.searchsorted(sites, start + read_len, side="left")
    empty = numpy.flatnonzero(hi == lo)
    while empty.size:
        start[empty] = rng.integers(0, ref_len - read_len, size=empty.size)
        lo[empty] = numpy.searchsorted(sites, start[empty], side="left")
        hi[empty] = numpy.searchsorted(sites, start[empty] + read_len, side="left")
        empty = empty[hi[empty] == lo[empty]]
    counts = (hi - lo).astype(numpy.int64)
    row_ptr = numpy.zeros(n_reads + 1, dtype=numpy.int64)
    numpy.cumsum(counts, out=row_ptr[1:])
    nnz = int(row_ptr[-1])
    # site index of every observation: lo[row] + (position within the row)
    row_of = numpy.repeat(numpy.arange(n_reads, dtype=numpy.int64), counts)
    site = (lo[row_of] + (numpy.arange(nnz, dtype=numpy.int64) - row_ptr[row_of])).astype(numpy.int64)
    hap_col = numpy.asarray(contrib, dtype=numpy.int64)[who][row_of]
    truth = tables.expected[site, hap_col]
    flip = rng.random(nnz) < err
    shift = rng.integers(1, 4, size=nnz)
    # index of the true base in ACGT (0 if it is something else, e.g. N)
    code = numpy.zeros(nnz, dtype=numpy.int64)
    for i, base in enumerate(ALPHABET):
        code[truth == base] = i
    wrong = ALPHABET[(code + shift) % 4]
    obs = numpy.where(flip, wrong, truth).astype(numpy.uint8)
    return row_ptr, site.astype(numpy.uint16), obs, who


def synth_pairs(tables, ref_len, n_frag, seed=1, contrib=DEFAULT_CONTRIB, props=DEFAULT_PROPS, read_len=READ_LEN,
                insert=(350, 500), err=ERR_RATE):
    """
    synth-pe-v1 (round 6): paired-end FRAGMENTS as matrix rows -- two reads of `read_len` bp at the ends of an insert of
    insert[0] .. insert[1] bp (outer distance, uniform), merged into ONE row as the reference merges mates
    (preprocess.py:118-138: one observation dict per query name): the row observes every variant site under either
    read, in ascending position.  With 2 x 150 bp and inserts of 350-500 the mates do not overlap, so no site is seen
    twice; a row holds ~74 sites on Build 17, two thirds of the rows more than 64.
    Draw order (the definition, rng = numpy.random.default_rng(seed)):
        1. contributor per fragment        rng.choice(len(props), R, p=props)
        2. start per fragment              rng.integers(0, ref_len - insert[1], R)
        3. insert per fragment             rng.integers(insert[0], insert[1] + 1, R)
           (fragments covering no site: start and insert redrawn, all of them at once, until none)
        4. error flag per observation      rng.random(nnz) < err
        5. substitution offset per obs.    rng.integers(1, 4, nnz)
    -> (row_ptr[R+1] int64, site[nnz] uint16, obs[nnz] uint8, who[R] int64) like synth_reads.
    """
    if insert[0] < 2 * read_len:
        raise ValueError("synth_pairs: overlapping mates are not modelled (insert[0] >= 2 * read_len)")
    rng = numpy.random.default_rng(seed)
    sites = tables.sites
    who = rng.choice(len(props), size=n_frag, p=numpy.asarray(props, dtype=float))
    start = rng.integers(0, ref_len - insert[1], size=n_frag)
    size = rng.integers(insert[0], insert[1] + 1, size=n_frag)

    def windows(st, sz):
        lo1 = numpy.searchsorted(sites, st, side="left")
        hi1 = numpy.searchsorted(sites, st + read_len, side="left")
        lo2 = numpy.searchsorted(sites, st + sz - read_len, side="left")
        hi2 = numpy.searchsorted(sites, st + sz, side="left")
        return lo1, hi1, lo2, hi2

    lo1, hi1, lo2, hi2 = windows(start, size)
    empty = numpy.flatnonzero((hi1 - lo1) + (hi2 - lo2) == 0)
    while empty.size:
        start[empty] = rng.integers(0, ref_len - insert[1], size=empty.size)
        size[empty] = rng.integers(insert[0], insert[1] + 1, size=empty.size)
        a, b, c, d = windows(start[empty], size[empty])
        lo1[empty], hi1[empty], lo2[empty], hi2[empty] = a, b, c, d
        empty = empty[(b - a) + (d - c) == 0]
    n1 = (hi1 - lo1).astype(numpy.int64)
    counts = n1 + (hi2 - lo2).astype(numpy.int64)
    row_ptr = numpy.zeros(n_frag + 1, dtype=numpy.int64)
    numpy.cumsum(counts, out=row_ptr[1:])
    nnz = int(row_ptr[-1])
    row_of = numpy.repeat(numpy.arange(n_frag, dtype=numpy.int64), counts)
    within = numpy.arange(nnz, dtype=numpy.int64) - row_ptr[row_of]
    first = within < n1[row_of]
    site = numpy.where(first, lo1[row_of] + within, lo2[row_of] + (within - n1[row_of])).astype(numpy.int64)
    hap_col = numpy.asarray(contrib, dtype=numpy.int64)[who][row_of]
    truth = tables.expected[site, hap_col]
    flip = rng.random(nnz) < err
    shift = rng.integers(1, 4, size=nnz)
    code = numpy.zeros(nnz, dtype=numpy.int64)
    for i, base in enumerate(ALPHABET):
        code[truth == base] = i
    wrong = ALPHABET[(code + shift) % 4]
    obs = numpy.where(flip, wrong, truth).astype(numpy.uint8)
    return row_ptr, site.astype(numpy.uint16), obs, who


GEN_BLOCK = 125000          # rows per independently seeded block of synth_rows


def synth_rows(tables, ref_len, lo, hi, seed=1, block=GEN_BLOCK, pairs=False, **kw):
    """
    Rows [lo, hi) of a synth-v1 read set of any size, without generating the rest:
    the set is defined block by block -- block k (rows k*block ... (k+1)*block - 1) is
    synth_reads(..., n_reads=block, seed=[seed, k]) -- so that every rank of a row-sharded
    run builds its own shard of ONE global problem, whatever the number of ranks
    (bench.py strong scaling; SURVEY.md section 8 d/e).  pairs=True: blocks of synth_pairs (synth-pe-v1) instead.
    -> (row_ptr[hi-lo+1], site, obs, who) like synth_reads.
    """
    lo, hi = int(lo), int(hi)
    if hi <= lo:
        return (numpy.zeros(1, dtype=numpy.int64), numpy.zeros(0, dtype=numpy.uint16),
                numpy.zeros(0, dtype=numpy.uint8), numpy.zeros(0, dtype=numpy.int64))
    ptrs, sites, obss, whos = [], [], [], []
    base = 0
    for k in range(lo // block, (hi - 1) // block + 1):
        rp, st, ob, who = (synth_pairs if pairs else synth_reads)(tables, ref_len, block, seed=[int(seed), k], **kw)
        a = max(lo - k * block, 0)
        b = min(hi - k * block, block)
        ptrs.append(rp[a:b] - rp[a] + base)
        base += int(rp[b] - rp[a])
        sites.append(st[rp[a]:rp[b]])
        obss.append(ob[rp[a]:rp[b]])
        whos.append(who[a:b])
    row_ptr = numpy.concatenate(ptrs + [numpy.array([base], dtype=numpy.int64)])
    return row_ptr, numpy.concatenate(sites), numpy.concatenate(obss), numpy.concatenate(whos)


def signatures(tables, row_ptr, site, obs):
    """CSR observations -> the reference's signature strings (preprocess.py:142-148)."""
    out = []
    pos = tables.sites
    for i in range(len(row_ptr) - 1):
        beg, end = int(row_ptr[i]), int(row_ptr[i + 1])
        out.append(",".join("%d:%s" % (pos[site[j]], chr(obs[j])) for j in range(beg, end)))
    return out


# --------------------------------------------------------------------------------------------------
# synth-aln-v1: synthetic ALIGNMENTS as columns (alignments.AlignmentColumns) -- input of the front end
# (preprocess.process_reads -> reduce_reads, preprocess.py:99-174) at any size, generated without a Python
# object per read.  The build's own code (the reference's tests build a handful of pysam objects by hand,
# preprocess_test.py:102-124).  tests/_fake_aln.py turns the columns into pysam-like objects for the
# object-by-object path, so both front ends can be run on the same input.
# --------------------------------------------------------------------------------------------------
def _ragged(lengths):
    """(owner index, offset within owner) of every element of segments with the given lengths."""
    lengths = numpy.asarray(lengths, dtype=numpy.int64)
    ptr = numpy.zeros(len(lengths) + 1, dtype=numpy.int64)
    numpy.cumsum(lengths, out=ptr[1:])
    owner = numpy.repeat(numpy.arange(len(lengths), dtype=numpy.int64), lengths)
    return ptr, owner, numpy.arange(int(ptr[-1]), dtype=numpy.int64) - ptr[owner]


def synth_alignments(tables, refseq, n_frag, seed=1, contrib=DEFAULT_CONTRIB, props=DEFAULT_PROPS, mate_share=0.4,
                     err=0.004, shuffle=True):
    """
    -> alignments.AlignmentColumns of n_frag fragments (plus a mate for `mate_share` of them) shed by the three
    contributors' sequences (the reference sequence carrying the contributor's expected base at every variant site):
    lengths 80-150; a sixth each with an insertion, a deletion, a soft clip; sequencing errors and a few 'N' bases;
    a tenth in lower case; mapping qualities from {60 x4, 42, 30, 29, 10, 0}; base qualities 31-40, a fifth of the
    alignments with a tenth of their bases below 30, a fifth with NO quality array; a third of the fragments at one of
    five fixed starts with length 120 (equal signatures: weights > 1); mates overlap their fragment's first read by
    20-60 bases, and half of them get one substituted base inside the overlap (half of those at low quality) -- where
    that base sits on a variant site the fragment sees two different bases there.  Alignment order is shuffled.
    """
    from .alignments import AlignmentColumns
    rng = numpy.random.default_rng([int(seed), 0xA11])
    ref_len = len(refseq)
    sites = numpy.asarray(tables.sites)
    srcs = []
    for col in contrib:
        seq = numpy.frombuffer(refseq.encode("ascii"), dtype=numpy.uint8).copy()
        seq[sites] = tables.expected[:, col]
        srcs.append(seq)
    src = numpy.stack(srcs)                                              # [3][ref_len]
    who = rng.choice(len(props), size=n_frag, p=numpy.asarray(props, dtype=float))
    length = rng.integers(80, 151, size=n_frag)
    start = rng.integers(0, ref_len - 400, size=n_frag)
    fixed = rng.random(n_frag) < 0.33
    start[fixed] = rng.choice(numpy.array([310, 2700, 7020, 11710, 16120]), size=int(fixed.sum()))
    length[fixed] = 120
    has_mate = rng.random(n_frag) < mate_share
    m_idx = numpy.flatnonzero(has_mate)
    overlap = rng.integers(20, 61, size=len(m_idx))
    # alignment table: fragments' first reads, then the mates
    a_frag = numpy.concatenate([numpy.arange(n_frag, dtype=numpy.int64), m_idx])
    a_who = numpy.concatenate([who, who[m_idx]])
    a_len = numpy.concatenate([length, rng.integers(80, 151, size=len(m_idx))])
    a_start = numpy.concatenate([start, start[m_idx] + length[m_idx] - overlap])
    n_aln = len(a_frag)
    is_mate = numpy.arange(n_aln) >= n_frag
    kind = rng.choice(6, size=n_aln)                                     # 0-2 plain, 3 ins, 4 del, 5 clip
    kind[is_mate] = 0
    at = rng.integers(10, 70, size=n_aln)                                # where the indel sits (all lengths are >= 80)
    k = numpy.where(kind == 3, rng.integers(1, 4, size=n_aln), numpy.where(kind == 4, rng.integers(1, 6, size=n_aln),
                                                                            rng.integers(2, 9, size=n_aln)))
    mapq = rng.choice(numpy.array([60, 60, 60, 60, 42, 30, 29, 10, 0]), size=n_aln).astype(numpy.int32)
    qkind = rng.choice(5, size=n_aln)                                    # 0-2 ok, 3 low, 4 none
    lower = rng.random(n_aln) < 0.1
    # bases: source position of every query offset (-1 = a random base: inserted / clipped)
    seq_ptr, own, off = _ragged(a_len)
    kk, aa, kd = k[own], at[own], kind[own]
    shift = numpy.zeros(len(own), dtype=numpy.int64)
    rand = numpy.zeros(len(own), dtype=bool)
    ins = kd == 3
    rand |= ins & (off >= aa) & (off < aa + kk)
    shift[ins & (off >= aa + kk)] = -kk[ins & (off >= aa + kk)]
    dele = kd == 4
    shift[dele & (off >= aa)] = kk[dele & (off >= aa)]
    clip = kd == 5
    rand |= clip & (off < kk)
    shift[clip & (off >= kk)] = -kk[clip & (off >= kk)]
    pos = a_start[own] + off + shift
    seq = src[a_who[own], numpy.clip(pos, 0, ref_len - 1)]
    seq[rand] = ALPHABET[rng.integers(0, 4, size=int(rand.sum()))]
    flip = rng.random(len(seq)) < err
    seq[flip] = ALPHABET[rng.integers(0, 4, size=int(flip.sum()))]
    seq[rng.random(len(seq)) < 0.0005] = ord("N")
    qual = rng.integers(31, 41, size=len(seq)).astype(numpy.uint8)
    low = (qkind[own] == 3) & (rng.random(len(seq)) < 0.1)
    qual[low] = rng.integers(2, 30, size=int(low.sum())).astype(numpy.uint8)
    # mates: one substituted base inside the overlap for half of them, half of those at low quality
    hit = numpy.flatnonzero(is_mate & (rng.random(n_aln) < 0.5))
    where = seq_ptr[hit] + (rng.random(len(hit)) * overlap[hit - n_frag]).astype(numpy.int64)
    code = numpy.zeros(len(where), dtype=numpy.int64)
    for i, base in enumerate(ALPHABET):
        code[seq[where] == base] = i
    seq[where] = ALPHABET[(code + rng.integers(1, 4, size=len(where))) % 4]
    qual[where] = numpy.where(rng.random(len(where)) < 0.5, 5, 38).astype(numpy.uint8)
    seq[lower[own]] |= 0x20                                               # lower case (letters only: A-Z, N)
    has_qual = (qkind != 4).astype(numpy.uint8)
    # CIGARs (BAM encoding: length << 4 | op): plain LM; aM kI (L-a-k)M; aM kD (L-a)M; kS (L-k)M
    n_ops = numpy.where(kind <= 2, 1, numpy.where(kind == 5, 2, 3))
    cig_ptr, c_own, c_off = _ragged(n_ops)
    ck, ca, cl, ckd = k[c_own], at[c_own], a_len[c_own], kind[c_own]
    op = numpy.zeros(len(c_own), dtype=numpy.int64)
    ln = cl.copy()
    sel = (ckd == 3)
    ln[sel & (c_off == 0)] = ca[sel & (c_off == 0)]
    op[sel & (c_off == 1)] = 1
    ln[sel & (c_off == 1)] = ck[sel & (c_off == 1)]
    ln[sel & (c_off == 2)] = (cl - ca - ck)[sel & (c_off == 2)]
    sel = (ckd == 4)
    ln[sel & (c_off == 0)] = ca[sel & (c_off == 0)]
    op[sel & (c_off == 1)] = 2
    ln[sel & (c_off == 1)] = ck[sel & (c_off == 1)]
    ln[sel & (c_off == 2)] = (cl - ca)[sel & (c_off == 2)]
    sel = (ckd == 5)
    op[sel & (c_off == 0)] = 4
    ln[sel & (c_off == 0)] = ck[sel & (c_off == 0)]
    ln[sel & (c_off == 1)] = (cl - ck)[sel & (c_off == 1)]
    cigar = ((ln << 4) | op).astype(numpy.uint32)
    names = ["f%07d" % i for i in range(n_frag)]
    if shuffle:                                                           # file order is not fragment order
        perm = rng.permutation(n_aln)
        new_sptr, s_own, s_off = _ragged(a_len[perm])
        gather = seq_ptr[perm][s_own] + s_off
        seq, qual = seq[gather], qual[gather]
        new_cptr, cc_own, cc_off = _ragged(n_ops[perm])
        cigar = cigar[cig_ptr[perm][cc_own] + cc_off]
        seq_ptr, cig_ptr = new_sptr, new_cptr
        a_start, mapq, a_frag, has_qual = a_start[perm], mapq[perm], a_frag[perm], has_qual[perm]
        # fragment ids in order of first appearance in the file, as a reader's name table would number them
        first = numpy.full(n_frag, n_aln, dtype=numpy.int64)
        numpy.minimum.at(first, a_frag, numpy.arange(n_aln, dtype=numpy.int64))
        rank = numpy.empty(n_frag, dtype=numpy.int64)
        rank[numpy.argsort(first, kind="stable")] = numpy.arange(n_frag, dtype=numpy.int64)
        a_frag = rank[a_frag]
        inv = numpy.argsort(rank)
        names = [names[i] for i in inv]
    return AlignmentColumns(a_start, mapq, a_frag, cig_ptr, cigar, seq_ptr, seq, qual, has_qual, names)
