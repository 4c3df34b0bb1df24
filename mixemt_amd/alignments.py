"""
The alignment front end, batched: alignments held as COLUMNS go through ONE call of
the library's host encoder (mxm_aln_encode, csrc/aln_encode.hpp) instead of one interpreter step per aligned base:

    process_reads   /root/reference/mixemt/preprocess.py:99-139    alignments -> {fragment: {site: base}}
    read_signature  :142-148, reduce_reads :163-174                -> {signature: [fragment ids]}
    build_em_input  :218-220, :225                                 rows = sorted(signatures), weights, id lists

    cols = AlignmentColumns.from_alignments(bam.fetch())       # pysam.AlignedSegment-like objects, read in bulk
    cols = read_bam("sample.bam")                              # or: the file itself, through the library's BAM reader
    enc  = encode_alignments(cols, tables.sites, ref_len, min_mq, min_bq)
    enc.row_ptr, enc.site, enc.obs      the CSR observations build_em_matrix_device / build_em_records_device take
    enc.weights                         fragments per signature (int64)
    enc.read_ids                        ReadIdGroups: row -> fragment names, materialised only when asked for
    enc.dropped                         fragments left with no site (the reference dies on them, :156-160)
    enc.signatures()                    the signature strings (tests, `-v`)

preprocess.process_reads / reduce_reads (object by object, per base) stay as the slow path: anything the encoder
hands back (a CIGAR that runs past its sequence, non-ASCII sequence text) goes through them, so that the exception
raised is the reference's.
"""

import re

import numpy

from . import _lib

_CIGAR_RE = re.compile(r"(\d+)([MIDNSHP=XB])")
_CIGAR_OPS = {c: i for i, c in enumerate("MIDNSHP=XB")}


class NeedsSlowPath(ValueError):
    """The batched encoder handed the input back (message says why): take preprocess.process_reads."""


class AlignmentColumns(object):
    """
    Alignments as columns (include/mixemt_hip.h, mxm_aln_columns):
        ref_start[n] int64, mapq[n] int32, frag[n] int64 (index into `names`), cig_ptr[n+1] / cigar[] uint32
        (BAM encoding: length << 4 | op), seq_ptr[n+1] / seq[] uint8 (ASCII) / qual[] uint8, has_qual[n] uint8
    names: the fragments' query_names (list, or any indexable), mates sharing one.
    """

    def __init__(self, ref_start, mapq, frag, cig_ptr, cigar, seq_ptr, seq, qual, has_qual, names):
        self.ref_start = numpy.ascontiguousarray(ref_start, dtype=numpy.int64)
        self.mapq = numpy.ascontiguousarray(mapq, dtype=numpy.int32)
        self.frag = numpy.ascontiguousarray(frag, dtype=numpy.int64)
        self.cig_ptr = numpy.ascontiguousarray(cig_ptr, dtype=numpy.int64)
        self.cigar = numpy.ascontiguousarray(cigar, dtype=numpy.uint32)
        self.seq_ptr = numpy.ascontiguousarray(seq_ptr, dtype=numpy.int64)
        self.seq = numpy.ascontiguousarray(seq, dtype=numpy.uint8)
        self.qual = None if qual is None else numpy.ascontiguousarray(qual, dtype=numpy.uint8)
        self.has_qual = None if has_qual is None else numpy.ascontiguousarray(has_qual, dtype=numpy.uint8)
        self.names = names
        n = len(self.ref_start)
        if not (len(self.mapq) == len(self.frag) == n and len(self.cig_ptr) == len(self.seq_ptr) == n + 1):
            raise ValueError("alignment columns of different lengths")
        if n and (int(self.cig_ptr[-1]) > len(self.cigar) or int(self.seq_ptr[-1]) > len(self.seq)
                  or (self.qual is not None and len(self.qual) < len(self.seq))
                  or (self.has_qual is not None and len(self.has_qual) != n)):
            raise ValueError("alignment columns: offsets run past their arrays")
        #  the C ABI carries no array lengths: an offset array that starts elsewhere than 0 or steps backwards
        # would make the encoder's walk read cigar[] / seq[] / qual[] outside their arrays
        for name, ptr in (("cig_ptr", self.cig_ptr), ("seq_ptr", self.seq_ptr)):
            if int(ptr[0]) != 0 or (n and int(numpy.diff(ptr).min()) < 0):
                raise ValueError("alignment columns: %s must start at 0 and never step backwards" % name)

    def __len__(self):
        return len(self.ref_start)

    @property
    def n_frag(self):
        return len(self.names)

    @classmethod
    def from_alignments(cls, alns):
        """
        pysam.AlignedSegment-like objects -> columns: one pass, a handful of attribute reads per ALIGNMENT (none per
        base).  Uses `cigartuples` where the object has it (pysam), else parses `cigarstring`.  Raises NeedsSlowPath
        for what cannot be held as bytes (non-ASCII sequence text, a quality array of another length).
        """
        starts, mapqs, frags, cig_len, seq_len, has_q = [], [], [], [], [], []
        cig, seqs, quals = [], [], []
        ids = {}
        names = []
        any_q = False
        for aln in alns:
            name = aln.query_name
            f = ids.get(name)
            if f is None:
                f = ids[name] = len(names)
                names.append(name)
            frags.append(f)
            starts.append(aln.reference_start if aln.reference_start is not None else -1)
            mapqs.append(aln.mapping_quality)
            tuples = getattr(aln, "cigartuples", None)
            if tuples is None:
                text = getattr(aln, "cigarstring", None) or ""
                tuples = [(_CIGAR_OPS[op], int(n)) for n, op in _CIGAR_RE.findall(text)]
            cig_len.append(len(tuples))
            for op, n in tuples:
                cig.append((int(n) << 4) | int(op))
            seq = aln.query_sequence or ""
            try:
                raw = seq.encode("ascii") if isinstance(seq, str) else bytes(seq)
            except UnicodeEncodeError:
                raise NeedsSlowPath("non-ASCII sequence text in %r" % (name,))
            seqs.append(raw)
            seq_len.append(len(raw))
            q = aln.query_qualities
            if q is None:
                has_q.append(0)
                quals.append(None)
            else:
                if len(q) != len(raw):
                    raise NeedsSlowPath("quality array of %r does not match its sequence" % (name,))
                has_q.append(1)
                any_q = True
                quals.append(q)
        n = len(starts)
        cig_ptr = numpy.zeros(n + 1, dtype=numpy.int64)
        numpy.cumsum(numpy.asarray(cig_len, dtype=numpy.int64), out=cig_ptr[1:])
        seq_ptr = numpy.zeros(n + 1, dtype=numpy.int64)
        numpy.cumsum(numpy.asarray(seq_len, dtype=numpy.int64), out=seq_ptr[1:])
        seq = numpy.frombuffer(b"".join(seqs), dtype=numpy.uint8)
        qual = None
        if any_q:
            qual = numpy.zeros(len(seq), dtype=numpy.uint8)
            for i, q in enumerate(quals):
                if q is not None and len(q):
                    arr = numpy.asarray(q)
                    if arr.min() < 0 or arr.max() > 255:
                        raise NeedsSlowPath("base quality outside 0..255")
                    qual[seq_ptr[i]:seq_ptr[i + 1]] = arr
        return cls(starts, mapqs, frags, cig_ptr, numpy.asarray(cig, dtype=numpy.uint32), seq_ptr, seq, qual,
                   numpy.asarray(has_q, dtype=numpy.uint8), names)

    def struct(self):
        """(mxm_aln_columns, the arrays it points into)."""
        keep = (self.ref_start, self.mapq, self.frag, self.cig_ptr, self.cigar, self.seq_ptr, self.seq, self.qual, self.has_qual)

        def p(arr):
            return None if arr is None or arr.size == 0 else arr.ctypes.data

        cols = _lib.AlnColumns(len(self), self.n_frag, p(self.ref_start), p(self.mapq), p(self.frag), self.cig_ptr.ctypes.data,
                               p(self.cigar), self.seq_ptr.ctypes.data, p(self.seq), p(self.qual), p(self.has_qual))
        return cols, keep


class FragmentNames(object):
    """The fragments' read names as one byte string + offsets (read_bam): a sequence of str, decoded when looked at."""

    def __init__(self, data, off):
        self.data, self.off = data, off

    def __len__(self):
        return len(self.off) - 1

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[k] for k in range(*i.indices(len(self)))]
        i = int(i)
        if i < 0:
            i += len(self)
        if not 0 <= i < len(self):
            raise IndexError("fragment index out of range")
        return self.data[self.off[i]:self.off[i + 1]].tobytes().decode("ascii", "replace")

    def __iter__(self):
        return (self[i] for i in range(len(self)))


def read_bam(path, n_threads=0):
    """
    A BAM file -> AlignmentColumns through the library's reader (mxm_bam_read, csrc/bam_reader.hpp): what the reference
    gets from `pysam.AlignmentFile(path).fetch()` (bin/mixemt:139-147, preprocess.py:209) -- every record placed on a
    reference, in file order -- without a Python object per alignment.  The columns carry `.ref_id`, `.flag` (per
    alignment, not used by the encoder) and `.bam_counts` = (records in the file, skipped because unplaced).
    Raises OSError when the file cannot be read, ValueError when it is not a BAM file the reader takes.
    """
    import ctypes
    import os
    lib = _lib.load()
    handle = ctypes.c_void_p()
    rc = lib.mxm_bam_read(os.fsencode(path), int(n_threads), ctypes.byref(handle))
    if rc == -6:
        raise OSError(lib.mxm_last_error().decode("utf-8", "replace"))
    if rc == -4:
        raise ValueError(lib.mxm_last_error().decode("utf-8", "replace"))
    _lib.check(rc, "mxm_bam_read")
    owner = _BamHandle(lib, handle)
    sz = _lib.BamSizes()
    _lib.check(lib.mxm_bam_sizes_of(handle, ctypes.byref(sz)), "mxm_bam_sizes_of")
    st = _lib.AlnColumns()
    _lib.check(lib.mxm_bam_columns(handle, ctypes.byref(st)), "mxm_bam_columns")

    def view(ptr, n, dtype):
        # the library's own arrays, not copies (the two big ones are a byte per base each); every view keeps the
        # handle alive through the ctypes block it is made from
        n = int(n)
        if not n or not ptr:
            return numpy.empty(0, dtype=dtype)
        block = (ctypes.c_uint8 * (n * numpy.dtype(dtype).itemsize)).from_address(ptr)
        block._owner = owner
        out = numpy.frombuffer(block, dtype=dtype)
        out.flags.writeable = False
        return out

    n = int(sz.n_aln)
    names = numpy.empty(int(sz.names_bytes), dtype=numpy.uint8)
    name_off = numpy.empty(int(sz.n_frag) + 1, dtype=numpy.int64)
    ref_id = numpy.empty(n, dtype=numpy.int32)
    flag = numpy.empty(n, dtype=numpy.uint16)
    _lib.check(lib.mxm_bam_fetch_names(handle, names.ctypes.data, name_off.ctypes.data, ref_id.ctypes.data,
                                       flag.ctypes.data), "mxm_bam_fetch_names")
    cols = AlignmentColumns(view(st.ref_start, n, numpy.int64), view(st.mapq, n, numpy.int32), view(st.frag, n, numpy.int64),
                            view(st.cig_ptr, n + 1, numpy.int64), view(st.cigar, sz.n_cigar, numpy.uint32),
                            view(st.seq_ptr, n + 1, numpy.int64), view(st.seq, sz.n_bases, numpy.uint8),
                            view(st.qual, sz.n_bases, numpy.uint8) if st.qual else None,
                            view(st.has_qual, n, numpy.uint8), FragmentNames(names, name_off))
    cols.ref_id, cols.flag = ref_id, flag
    cols.bam_counts = (int(sz.n_records_total), int(sz.n_skipped_unplaced))
    return cols


class _BamHandle(object):
    """Owns an mxm_bam: freed when the last array viewing it goes."""

    def __init__(self, lib, handle):
        self._free, self._handle = lib.mxm_bam_free, handle

    def __del__(self):
        handle, self._handle = self._handle, None
        if handle and self._free is not None:
            self._free(handle)


class ReadIdGroups(object):
    """
    The read-id lists of build_em_input's fourth result (preprocess.py:225) without a Python list per row: row i's
    fragments are names[frag[ptr[i]:ptr[i+1]]].  Behaves like the reference's list of lists (len, indexing, iteration,
    equality with one) and materialises a row only when someone looks at it -- a `.reads` writer, an assembler.
    """

    def __init__(self, ptr, frag, names):
        self.ptr, self.frag, self.names = ptr, frag, names

    def __len__(self):
        return len(self.ptr) - 1

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[k] for k in range(*i.indices(len(self)))]
        if i < 0:
            i += len(self)
        if not 0 <= i < len(self):
            raise IndexError("row index out of range")
        names = self.names
        return [names[f] for f in self.frag[self.ptr[i]:self.ptr[i + 1]]]

    def __iter__(self):
        names, frag, ptr = self.names, self.frag, self.ptr
        for i in range(len(self)):
            yield [names[f] for f in frag[ptr[i]:ptr[i + 1]]]

    def __eq__(self, other):
        try:
            return len(other) == len(self) and all(a == b for a, b in zip(self, other))
        except TypeError:
            return NotImplemented

    def counts(self):
        """Fragments per row (the weights, preprocess.py:220)."""
        return numpy.diff(self.ptr)

    def tolist(self):
        return list(self)


class EncodedReads(object):
    """Result of encode_alignments (see the module docstring)."""

    def __init__(self, row_ptr, site, obs, weights, read_ids, dropped, n_fragments, text, text_off, handle, frag_nnz):
        self.row_ptr, self.site, self.obs, self.weights = row_ptr, site, obs, weights
        self.read_ids, self.dropped, self.n_fragments = read_ids, dropped, n_fragments
        self._text, self._text_off = text, text_off
        self._handle, self._frag_nnz, self._frag = handle, frag_nnz, None     # the per-fragment lists stay in the library

    def __del__(self):
        self.release()

    def release(self):
        """Give the library's copy of the per-fragment observations back (read_obs() needs it)."""
        handle, self._handle = getattr(self, "_handle", None), None
        if handle and _lib is not None and _lib.load is not None:          # (module teardown at interpreter exit)
            lib = _lib.load()
            if lib is not None:
                lib.mxm_aln_free(handle)

    @property
    def n_rows(self):
        return len(self.weights)

    def signatures(self):
        """The rows' signature strings 'pos:base,...' in row order (= sorted()); fetched from the library on demand."""
        if not self.n_rows:
            return []
        if self._text is None:
            if not self._handle:
                raise ValueError("signatures: the encoder's result has been released")
            text = numpy.empty(int(self._text_off[-1]), dtype=numpy.uint8)
            _lib.check(_lib.load().mxm_aln_fetch(self._handle, None, None, None, None, None, None, None, text.ctypes.data, None),
                       "mxm_aln_fetch")
            self._text = text
        return self._text.tobytes().decode("ascii").split("\n")[:-1]

    def read_obs(self, sites):
        """process_reads' own result {fragment name: {0-based site: base}} in its dict order (tests; small inputs)."""
        if self._frag is None:
            if not self._handle:
                raise ValueError("read_obs: the encoder's result has been released")
            frag_id = numpy.empty(self.n_fragments, dtype=numpy.int64)
            frag_ptr = numpy.empty(self.n_fragments + 1, dtype=numpy.int64)
            f_site = numpy.empty(self._frag_nnz, dtype=numpy.uint16)
            f_obs = numpy.empty(self._frag_nnz, dtype=numpy.uint8)
            _lib.check(_lib.load().mxm_aln_fetch_fragments(self._handle, frag_id.ctypes.data, frag_ptr.ctypes.data,
                                                           f_site.ctypes.data, f_obs.ctypes.data), "mxm_aln_fetch_fragments")
            self._frag = (frag_id, frag_ptr, f_site, f_obs)
        frag_id, frag_ptr, f_site, f_obs = self._frag
        names = self.read_ids.names
        out = {}
        for k, f in enumerate(frag_id):
            a, b = int(frag_ptr[k]), int(frag_ptr[k + 1])
            out[names[f]] = {int(sites[s]): chr(o) for s, o in zip(f_site[a:b], f_obs[a:b])}
        return out


def encode_alignments(cols, sites, ref_len, min_mq, min_bq, n_threads=0):
    """
    AlignmentColumns -> EncodedReads: everything process_reads + reduce_reads + the row ordering of build_em_input do
    (module docstring), in the library's host encoder.  sites: the sorted 0-based variant sites
    (phylo.get_variant_pos() = HapVarTables.sites); ref_len: length of the reference sequence.
    Raises NeedsSlowPath when the encoder hands the input back.
    """
    import ctypes
    lib = _lib.load()
    sites = numpy.ascontiguousarray(sites, dtype=numpy.int64)
    if len(sites) > 65536:
        raise ValueError("more than 65536 variant sites (%d)" % len(sites))
    ref_len = int(max(ref_len, (int(sites.max()) + 1) if len(sites) else 1))
    site_of_pos = numpy.full(ref_len, -1, dtype=numpy.int32)
    site_of_pos[sites] = numpy.arange(len(sites), dtype=numpy.int32)
    st, keep = cols.struct()
    handle = ctypes.c_void_p()
    rc = lib.mxm_aln_encode(ctypes.byref(st), site_of_pos.ctypes.data, ref_len, sites.ctypes.data, len(sites),
                            int(min_mq), int(min_bq), int(n_threads), ctypes.byref(handle))
    del keep
    if rc == -4:
        raise NeedsSlowPath(lib.mxm_last_error().decode("utf-8", "replace"))
    _lib.check(rc, "mxm_aln_encode")
    try:
        sz = _lib.AlnSizes()
        _lib.check(lib.mxm_aln_sizes_of(handle, ctypes.byref(sz)), "mxm_aln_sizes_of")
        row_ptr = numpy.empty(sz.n_rows + 1, dtype=numpy.int64)
        site = numpy.empty(sz.nnz, dtype=numpy.uint16)
        obs = numpy.empty(sz.nnz, dtype=numpy.uint8)
        weights = numpy.empty(sz.n_rows, dtype=numpy.int64)
        group_ptr = numpy.empty(sz.n_rows + 1, dtype=numpy.int64)
        group_frag = numpy.empty(sz.n_grouped, dtype=numpy.int64)
        dropped = numpy.empty(sz.n_dropped, dtype=numpy.int64)
        text = None                                   # the signature strings stay in the library until someone asks
        text_off = numpy.empty(sz.n_rows + 1, dtype=numpy.int64)
        _lib.check(lib.mxm_aln_fetch(handle, row_ptr.ctypes.data, site.ctypes.data, obs.ctypes.data, weights.ctypes.data,
                                     group_ptr.ctypes.data, group_frag.ctypes.data, dropped.ctypes.data, None,
                                     text_off.ctypes.data), "mxm_aln_fetch")
    except Exception:
        lib.mxm_aln_free(handle)
        raise
    return EncodedReads(row_ptr, site, obs, weights, ReadIdGroups(group_ptr, group_frag, cols.names),
                        [cols.names[f] for f in dropped], int(sz.n_frag_seen), text, text_off, handle, int(sz.frag_nnz))
