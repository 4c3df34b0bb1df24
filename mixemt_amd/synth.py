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


GEN_BLOCK = 125000          # rows per independently seeded block of synth_rows


def synth_rows(tables, ref_len, lo, hi, seed=1, block=GEN_BLOCK, **kw):
    """
    Rows [lo, hi) of a synth-v1 read set of any size, without generating the rest:
    the set is defined block by block -- block k (rows k*block ... (k+1)*block - 1) is
    synth_reads(..., n_reads=block, seed=[seed, k]) -- so that every rank of a row-sharded
    run builds its own shard of ONE global problem, whatever the number of ranks
    (bench.py strong scaling; SURVEY.md section 8 d/e).
    -> (row_ptr[hi-lo+1], site, obs, who) like synth_reads.
    """
    lo, hi = int(lo), int(hi)
    if hi <= lo:
        return (numpy.zeros(1, dtype=numpy.int64), numpy.zeros(0, dtype=numpy.uint16),
                numpy.zeros(0, dtype=numpy.uint8), numpy.zeros(0, dtype=numpy.int64))
    ptrs, sites, obss, whos = [], [], [], []
    base = 0
    for k in range(lo // block, (hi - 1) // block + 1):
        rp, st, ob, who = synth_reads(tables, ref_len, block, seed=[int(seed), k], **kw)
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
