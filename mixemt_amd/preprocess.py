"""
EM input construction on the GPU: the drop-in for
`mixemt.preprocess.build_em_matrix` (reference: mixemt/preprocess.py:177-198)
plus the host-side encoders that turn the reference's dictionaries and
signature strings into the flat tables the kernel reads.

Host (numpy, one-time, small):
    HapVarTables.build   <- HapVarBaseMatrix.__init__/add_hap_markers (:39-67)
    encode_signatures    <- pos_obs_from_sig (:151-160); parsed by the library's host function
                            mxm_encode_signatures, item by item in Python for anything unusual
Device (libmixemt_hip.so):
    mxm_build_em_matrix  <- the R x H x k loop (:188-191) with _prob (:69-84)
                            and prob_for_vars (:86-96) inlined

The expected-base table stores the ASCII code of the base a haplogroup should
show at a site, observations are ASCII codes too, so "hit" is byte equality --
exactly the reference's string comparison, for any alphabet ('N' included).
"""

import collections
import math
import os
import sys
import time

import numpy

from . import _lib
from . import phylotree
from ._dev import as_device, current_stream, device_empty, require_gpu, to_host, torch

MUT_WT = 0.01       # preprocess.py:39 defaults; not CLI flags (build_em_matrix :182)
MUT_MAX = 0.5


class _LazyLut(object):
    """lut() of HapVarTables: obsmap at once, the 22 MB `ecode` table only if someone asks for it on the host (the
    device path gathers it on the device)."""

    def __init__(self, expected, n_haps, code_of, obsmap):
        self._expected, self._n_haps, self.code_of = expected, n_haps, code_of
        self._items = {"obsmap": obsmap}

    def __getitem__(self, key):
        if key == "ecode" and "ecode" not in self._items:
            ecode = self.code_of[self._expected]            # pad byte 0 -> 0: never equals an observation code
            ecode[:, self._n_haps:] = 0
            self._items["ecode"] = numpy.ascontiguousarray(ecode)
        return self._items[key]

    def __contains__(self, key):
        return key in ("ecode", "obsmap")

    def keys(self):
        return ["ecode", "obsmap"]


class HapVarTables(object):
    """
    Flat form of HapVarBaseMatrix for a fixed haplogroup (column) order.

        sites[S]      sorted 0-based variant sites (phylotree.py:221-229)
        expected[S][lde] uint8: ord(base) haplogroup h is expected to show at
                      site s -- its marker where it carries a non-reference
                      allele (:60-66), else the reference base (:80-82);
                      columns [H, lde) are padding (0 never equals a base)
        lhit[S]       math.log(1 - mu_s)        (:77, :82 with :94)
        lmiss[S]      math.log(mu_s / 3.0)      (:84 with :94)
        mu_s = min(mut_max, mut_wt * sum(variants[pos].values()))  (:48-51)
    """

    def __init__(self, sites, expected, lhit, lmiss, haplogroups, n_haps):
        self.sites = sites
        self.expected = expected
        self.lhit = lhit
        self.lmiss = lmiss
        self.haplogroups = haplogroups
        self.n_haps = n_haps
        self.site_index = {int(p): k for k, p in enumerate(sites)}
        # dense position -> site index table for the library's host parser (-1 = not a site)
        self.site_of_pos = numpy.full((int(max(sites)) + 1) if len(sites) else 1, -1, dtype=numpy.int32)
        for k, p in enumerate(sites):
            if int(p) >= 0:
                self.site_of_pos[int(p)] = k
        self._marker_flat = None
        self._ref_codes = None
        self._dev = None
        self._lut = None
        self._lut_dev = None
        self._sparse = None
        self._sparse_dev = None

    @classmethod
    def build(cls, refseq, phylo, haplogroups, mut_wt=MUT_WT, mut_max=MUT_MAX):
        sites = numpy.array(sorted(phylo.variants.keys()), dtype=numpy.int64)
        n_sites, n_haps = len(sites), len(haplogroups)
        if n_sites > 65536:
            raise ValueError("more than 65536 variant sites (%d)" % n_sites)
        where = {int(p): k for k, p in enumerate(sites)}
        lde = (n_haps + 15) // 16 * 16
        expected = numpy.zeros((n_sites, lde), dtype=numpy.uint8)
        ref_codes = numpy.array([ord(refseq[int(p)]) for p in sites], dtype=numpy.uint8)
        expected[:, :n_haps] = ref_codes[:, None]
        # markers: (site index, haplogroup index, derived base) in the reference's iteration order.  A tree has a few
        # thousand DISTINCT variant strings for 280 000 (haplogroup, variant) pairs: each string is parsed once.
        parsed = {}
        flat, der_codes = [], []
        for j, hap in enumerate(haplogroups):
            for var in phylo.hap_var[hap]:
                ent = parsed.get(var)
                if ent is None:
                    pos = phylotree.pos_from_var(var)
                    der = phylotree.der_allele(var)
                    k = where.get(pos)
                    # a derived allele equal to the reference base leaves no marker (:63-66): the expected base stays
                    # whatever it was
                    ent = (k * lde, ord(der)) if (k is not None and der != refseq[pos]) else False
                    parsed[var] = ent
                if ent:
                    flat.append(ent[0] + j)
                    der_codes.append(ent[1])
        flat = numpy.asarray(flat, dtype=numpy.int64)
        if flat.size:
            # repeated (site, haplogroup) pairs: the last assignment stands, as in the reference's dict (:60-66)
            expected.reshape(-1)[flat] = numpy.asarray(der_codes, dtype=numpy.uint8)
        lhit = numpy.empty(n_sites)
        lmiss = numpy.empty(n_sites)
        for k, pos in enumerate(sites):
            mu = min(mut_max, mut_wt * sum(phylo.variants[int(pos)].values()))
            lhit[k] = math.log(1.0 - mu)
            lmiss[k] = math.log(mu / 3.0)
        tables = cls(sites, expected, lhit, lmiss, list(haplogroups), n_haps)
        # where the table differs from the reference base, as flat indexes (sorted: site-major, haplogroup ascending):
        # sparse() and lut() work from these 264 000 cells instead of sweeping all 22 million
        tables._marker_flat = numpy.unique(flat)
        tables._ref_codes = ref_codes
        return tables

    def lut(self):
        """
        The lookup-table kernel's encoding of the same tables (include/mixemt_hip.h,
        mxm_build_em_matrix_lut), or None if they do not qualify (more than 14 distinct
        expected bases, or a table beyond one 2 GiB buffer descriptor):
            ecode[S][lde] uint8   code of the expected base << 3 (codes 1..14; pad bytes 0)
            obsmap[256]   uint8   observation byte -> code << 3; 15 << 3 = equals no expected base
        """
        if self._lut is not None:
            return self._lut or None
        n_sites, n_haps = len(self.sites), self.n_haps
        if self._marker_flat is not None:                   # build(): reference bases + marker bases are all there is
            alphabet = numpy.unique(numpy.concatenate([self._ref_codes, self.expected.reshape(-1)[self._marker_flat]]))
        else:
            alphabet = numpy.flatnonzero(numpy.bincount(self.expected[:, :n_haps].reshape(-1), minlength=256)).astype(numpy.uint8)
        alphabet = alphabet[alphabet != 0]
        if (len(alphabet) > 14 or n_sites == 0 or n_haps > 8192
                or n_sites * self.expected.shape[1] >= (1 << 31)):
            self._lut = False
            return None
        code_of = numpy.zeros(256, dtype=numpy.uint8)
        code_of[alphabet] = numpy.arange(1, len(alphabet) + 1, dtype=numpy.uint8) << 3
        obsmap = numpy.full(256, 15 << 3, dtype=numpy.uint8)
        obsmap[alphabet] = code_of[alphabet]
        self._lut = _LazyLut(self.expected, n_haps, code_of, obsmap)
        return self._lut

    def sparse(self):
        """
        The marker form of the expected-base table (include/mixemt_hip.h, mxm_build_em_matrix_sparse):
            maj[S] uint8          the base most haplogroups expect at the site
            mk_ptr[S+1] int32, mk_hap[] uint16, mk_base[] uint8
                                  CSR over sites of the (haplogroup, expected base) pairs that differ from maj
        """
        if self._sparse is None:
            exp = self.expected[:, :self.n_haps]
            n_sites = exp.shape[0]
            if self._marker_flat is not None and n_sites:
                # from the marker cells alone (build()): per site a histogram of the marker bases, the reference base
                # holding every other haplogroup; the most frequent byte wins, ties to the smallest (as
                # bincount().argmax() over the row gives)
                lde = self.expected.shape[1]
                k_u, j_u = self._marker_flat // lde, self._marker_flat % lde
                v_u = self.expected.reshape(-1)[self._marker_flat]
                # (the bytes that occur -- a handful -- numbered in ascending order: a histogram of n_sites x 8 instead of
                # n_sites x 256 counters, and argmax still breaks ties towards the smallest byte)
                present = numpy.zeros(256, dtype=bool)
                present[v_u] = True
                present[self._ref_codes] = True
                alphabet = numpy.flatnonzero(present)
                small = numpy.zeros(256, dtype=numpy.int64)
                small[alphabet] = numpy.arange(alphabet.size)
                n_a = int(alphabet.size)
                cnt = numpy.bincount(k_u * n_a + small[v_u], minlength=n_sites * n_a).reshape(n_sites, n_a)
                cnt[numpy.arange(n_sites), small[self._ref_codes]] += self.n_haps - numpy.bincount(k_u, minlength=n_sites)
                maj = alphabet[cnt.argmax(axis=1)].astype(numpy.uint8)
                keep = v_u != maj[k_u]
                site_i, hap_i = k_u[keep], j_u[keep]
                special = numpy.flatnonzero(maj != self._ref_codes)  # sites where most haplogroups carry a marker
                if special.size:
                    # their lists are the haplogroups that differ from the new majority; both lists are sorted by (site,
                    # haplogroup) and a site is wholly in one of them: merged by counting, without a sort
                    stay = ~numpy.isin(site_i, special)
                    site_n, hap_n = site_i[stay], hap_i[stay]
                    xs, xh = numpy.nonzero(exp[special] != maj[special, None])
                    site_x = special[xs]
                    before_n = numpy.zeros(n_sites + 1, dtype=numpy.int64)     # special-site entries in sites below s
                    numpy.cumsum(numpy.bincount(site_x, minlength=n_sites), out=before_n[1:])
                    before_x = numpy.zeros(n_sites + 1, dtype=numpy.int64)     # ordinary entries in sites below s
                    numpy.cumsum(numpy.bincount(site_n, minlength=n_sites), out=before_x[1:])
                    total = site_n.size + site_x.size
                    site_i = numpy.empty(total, dtype=site_n.dtype)
                    hap_i = numpy.empty(total, dtype=hap_n.dtype)
                    at_n = numpy.arange(site_n.size) + before_n[site_n]
                    at_x = numpy.arange(site_x.size) + before_x[site_x]
                    site_i[at_n], hap_i[at_n] = site_n, hap_n
                    site_i[at_x], hap_i[at_x] = site_x, xh
            else:
                maj = numpy.zeros(n_sites, dtype=numpy.uint8)
                for s in range(n_sites):
                    maj[s] = numpy.bincount(exp[s], minlength=256).argmax()
                site_i, hap_i = numpy.nonzero(exp != maj[:, None])       # row-major: sorted by site
            mk_ptr = numpy.zeros(n_sites + 1, dtype=numpy.int32)
            numpy.cumsum(numpy.bincount(site_i, minlength=n_sites), out=mk_ptr[1:])
            self._sparse = {"maj": maj, "mk_ptr": mk_ptr, "mk_hap": hap_i.astype(numpy.uint16),
                            "mk_base": numpy.ascontiguousarray(exp[site_i, hap_i])}
        return self._sparse

    def _vectors_device(self):
        """lhit / lmiss on the device (uploaded once)."""
        if getattr(self, "_vec_dev", None) is None:
            dev = require_gpu()
            self._vec_dev = (torch.from_numpy(self.lhit).to(dev), torch.from_numpy(self.lmiss).to(dev))
        return self._vec_dev

    def sparse_device(self):
        """Device copies of sparse() (plus lhit / lmiss), uploaded once."""
        if self._sparse_dev is None:
            dev = require_gpu()
            enc = self.sparse()
            lhit_d, lmiss_d = self._vectors_device()
            self._sparse_dev = {key: torch.from_numpy(val).to(dev) for key, val in enc.items()}
            if self._sparse_dev["mk_hap"].numel() == 0:                  # keep the pointers valid
                self._sparse_dev["mk_hap"] = torch.zeros(1, dtype=torch.uint16, device=dev)
                self._sparse_dev["mk_base"] = torch.zeros(1, dtype=torch.uint8, device=dev)
            self._sparse_dev["lhit"], self._sparse_dev["lmiss"] = lhit_d, lmiss_d
        return self._sparse_dev

    def _expand_on_device(self, code_map):
        """[S][lde] uint8 table from the marker form, on the device (mxm_expand_tables): the expected bases themselves
        (code_map None) or their lookup codes.  0.4 MB of markers go up instead of the 22 MB table."""
        lib = _lib.load()
        dev = require_gpu()
        sp = self.sparse_device()
        out = torch.empty(self.expected.shape, dtype=torch.uint8, device=dev)
        map_d = None if code_map is None else torch.from_numpy(numpy.ascontiguousarray(code_map, dtype=numpy.uint8)).to(dev)
        _lib.check(lib.mxm_expand_tables(sp["maj"].data_ptr(), sp["mk_ptr"].data_ptr(), sp["mk_hap"].data_ptr(),
                                         sp["mk_base"].data_ptr(), map_d.data_ptr() if map_d is not None else None,
                                         len(self.sites), self.n_haps, self.expected.shape[1], out.data_ptr(), current_stream()),
                   "mxm_expand_tables")
        return out

    def lut_device(self):
        """Device form of lut() (plus lhit / lmiss), made once; None if the tables do not qualify."""
        enc = self.lut()
        if enc is None:
            return None
        if self._lut_dev is None:
            dev = require_gpu()
            lhit_d, lmiss_d = self._vectors_device()
            # the code table straight from the marker form on the device (round 5: it used to be a torch gather over the
            # uploaded 22 MB table -- most of a cold process's "tables" stage was those operators' first uses)
            ecode_d = self._expand_on_device(enc.code_of) if len(self.sites) else torch.zeros(self.expected.shape, dtype=torch.uint8, device=dev)
            self._lut_dev = {"ecode": ecode_d, "obsmap": torch.from_numpy(enc["obsmap"]).to(dev),
                             "lhit": lhit_d, "lmiss": lmiss_d}
        return self._lut_dev

    def device(self):
        """(expected, lhit, lmiss) as device tensors, made once: the table is expanded from the marker form on the device
        (mxm_expand_tables) rather than uploaded."""
        if self._dev is None:
            dev = require_gpu()
            lhit_d, lmiss_d = self._vectors_device()
            exp_d = self._expand_on_device(None) if (len(self.sites) and self.n_haps <= 65535) else torch.from_numpy(self.expected).to(dev)
            self._dev = (exp_d, lhit_d, lmiss_d)
        return self._dev


def _encode_signatures_py(reads, tables):
    """The reference's own parsing expressions (preprocess.py:151-160), item by item: the path
    that raises the reference's exceptions; ~18 s per 10^6 reads."""
    row_ptr = numpy.zeros(len(reads) + 1, dtype=numpy.int64)
    site, obs = [], []
    index = tables.site_index
    for i, sig in enumerate(reads):
        if sig == "":
            # the reference dies in int('') here (preprocess.py:156-160)
            raise ValueError("empty read signature at row %d" % i)
        for item in sig.split(","):
            pos, base = item.split(":")
            try:
                site.append(index[int(pos)])
            except KeyError:
                raise KeyError(int(pos))      # reference: self.mut_prob[pos] (:84)
            obs.append(ord(base) if len(base) == 1 else 0)
        row_ptr[i + 1] = len(site)
    return (row_ptr, numpy.array(site, dtype=numpy.uint16),
            numpy.array(obs, dtype=numpy.uint8))


def _encode_signatures_native(reads, tables):
    """
    The same CSR through the library's host parser (mxm_encode_signatures): one pass over the
    joined text, ~0.5 s per 10^6 reads.  Returns None whenever anything is out of the ordinary
    (non-ASCII text, a signature the C parser hands back): the caller then takes the
    item-by-item path, which also produces the reference's exceptions.
    """
    n_reads = len(reads)
    if n_reads == 0:
        return None
    lib = _lib.load()                                 # raises if the library is not built
    try:
        text = ("\n".join(reads) + "\n").encode("ascii")
    except (UnicodeEncodeError, TypeError):
        return None
    off = numpy.zeros(n_reads + 1, dtype=numpy.int64)
    numpy.cumsum(numpy.fromiter((len(sig) + 1 for sig in reads), dtype=numpy.int64, count=n_reads), out=off[1:])
    if int(off[-1]) != len(text):
        return None
    cap = text.count(b":")
    row_ptr = numpy.empty(n_reads + 1, dtype=numpy.int64)
    site = numpy.empty(cap, dtype=numpy.uint16)
    obs = numpy.empty(cap, dtype=numpy.uint8)
    lut = tables.site_of_pos
    got = lib.mxm_encode_signatures(text, off.ctypes.data, n_reads, lut.ctypes.data, len(lut),
                                            row_ptr.ctypes.data, site.ctypes.data, obs.ctypes.data, cap)
    if got < 0:
        return None
    return row_ptr, site[:got], obs[:got]


def encode_signatures(reads, tables):
    """
    Signature strings 'pos:base,pos:base,...' (preprocess.py:142-160) -> CSR
        row_ptr[R+1] int64, site[nnz] uint16 (index into tables.sites),
        obs[nnz] uint8 (ASCII; a multi-character observation can never equal a
        base and is stored as 0)
    Order inside a row is the signature's order: the kernel adds in that order.
    """
    fast = _encode_signatures_native(reads, tables)
    if fast is not None:
        return fast
    return _encode_signatures_py(reads, tables)


SORT_ROWS_FROM = 50000       # rows from which "auto" hands the lookup-table kernel a position-sorted row order


def row_order_by_position(row_ptr_d, site_d):
    """
    Permutation of the rows by the first variant site they observe (stable).  Rows that start at
    nearby positions read the same rows of the expected-base table, so the kernel that takes
    them in this order keeps its table traffic in L2.  One small device sort over R keys
    (set-up of the build, like the uploads; the EM loop has no library op).
    """
    n_rows = row_ptr_d.numel() - 1
    if site_d.numel() == 0:
        return torch.arange(n_rows, dtype=torch.int64, device=row_ptr_d.device)
    first = row_ptr_d[:-1].clamp(max=site_d.numel() - 1)
    keys = site_d[first].to(torch.int32) & 0xFFFF          # uint16 bit patterns stored as int16
    return torch.argsort(keys, stable=True)


def build_em_matrix_device(tables, row_ptr, site, obs, out=None, kernel="auto", sort_rows="auto"):
    """
    CSR observations (numpy or device tensors) -> device tensor M[R][H] float64.
    `out` may supply a preallocated [R][>=H] tensor.  kernel:
      "lut"    hit / miss by LDS lookup (mxm_build_em_matrix_lut): the fastest cell-by-cell kernel
      "bytes"  the byte-table kernel (mxm_build_em_matrix), any alphabet, any width
      "sparse" the marker kernel (mxm_build_em_matrix_sparse): one in-order sum per distinct cell value of a
               row instead of one per cell; rows with more than 128 observations (or 5120 marker entries) go through "lut"
      "auto"   "sparse" where the tables qualify for "lut" (at most 14 distinct bases, H <= 8192), else "bytes"
    All give the same bits (profiles/r02/build_kernels.txt has the timings).
    sort_rows: take the rows in position order ("lut" only; True / False / "auto" = from
    SORT_ROWS_FROM rows); results do not depend on it.
    """
    lib = _lib.load()
    dev = require_gpu()
    if kernel == "auto":
        kernel = "bytes" if tables.lut() is None else "sparse"
    if kernel not in ("sparse", "lut", "bytes"):
        raise ValueError("kernel must be 'auto', 'sparse', 'lut' or 'bytes'")
    if kernel == "sparse":
        if tables.lut() is None or tables.n_haps > 8192:
            raise ValueError("tables do not qualify for the marker kernel (its leftover rows need the lookup-table kernel)")
        enc = tables.sparse_device()
        row_ptr_d = as_device(row_ptr, torch.int64, dev)
        site_d = as_device(site, torch.uint16, dev)
        obs_d = as_device(obs, torch.uint8, dev)
        n_rows = row_ptr_d.numel() - 1
        n_haps = tables.n_haps
        if out is None:
            out = device_empty((n_rows, n_haps), torch.float64, dev, "the EM input matrix")
        if n_rows == 0:
            return out
        fallback = torch.empty(n_rows, dtype=torch.int64, device=dev)
        n_fallback = torch.zeros(1, dtype=torch.int64, device=dev)
        _lib.check(lib.mxm_build_em_matrix_sparse(
            enc["maj"].data_ptr(), enc["lhit"].data_ptr(), enc["lmiss"].data_ptr(), enc["mk_ptr"].data_ptr(),
            enc["mk_hap"].data_ptr(), enc["mk_base"].data_ptr(), row_ptr_d.data_ptr(), site_d.data_ptr(),
            obs_d.data_ptr(), 0, n_rows, n_haps, len(tables.sites), out.data_ptr(), out.stride(0),
            fallback.data_ptr(), n_fallback.data_ptr(), current_stream()), "mxm_build_em_matrix_sparse")
        left = int(n_fallback.item())
        build_em_matrix_device.last_fallback = left
        if left:
            # rows with more than 128 observations (or hundreds of distinct values, or thousands of marker entries): cell by cell
            lut = tables.lut_device()
            rows = fallback[:left].sort().values          # any order gives the same rows; sorted = reproducible launch
            _lib.check(lib.mxm_build_em_matrix_lut(
                lut["ecode"].data_ptr(), lut["ecode"].stride(0), lut["lhit"].data_ptr(), lut["lmiss"].data_ptr(),
                lut["obsmap"].data_ptr(), row_ptr_d.data_ptr(), site_d.data_ptr(), obs_d.data_ptr(),
                rows.data_ptr(), left, n_haps, len(tables.sites), out.data_ptr(), out.stride(0),
                current_stream()), "mxm_build_em_matrix_lut")
        return out
    if kernel == "lut":
        enc = tables.lut_device()
        if enc is None:
            raise ValueError("tables do not qualify for the lookup-table kernel")
        row_ptr_d = as_device(row_ptr, torch.int64, dev)
        site_d = as_device(site, torch.uint16, dev)
        obs_d = as_device(obs, torch.uint8, dev)
        n_rows = row_ptr_d.numel() - 1
        n_haps = tables.n_haps
        if out is None:
            out = device_empty((n_rows, n_haps), torch.float64, dev, "the EM input matrix")
        if n_rows == 0:
            return out
        order = None
        if sort_rows is True or (sort_rows == "auto" and n_rows >= SORT_ROWS_FROM):
            order = row_order_by_position(row_ptr_d, site_d)
        _lib.check(lib.mxm_build_em_matrix_lut(
            enc["ecode"].data_ptr(), enc["ecode"].stride(0), enc["lhit"].data_ptr(), enc["lmiss"].data_ptr(),
            enc["obsmap"].data_ptr(), row_ptr_d.data_ptr(), site_d.data_ptr(), obs_d.data_ptr(),
            0 if order is None else order.data_ptr(), n_rows, n_haps, len(tables.sites),
            out.data_ptr(), out.stride(0), current_stream()), "mxm_build_em_matrix_lut")
        return out
    exp_d, lhit_d, lmiss_d = tables.device()
    row_ptr_d = as_device(row_ptr, torch.int64, dev)
    site_d = as_device(site, torch.uint16, dev)
    obs_d = as_device(obs, torch.uint8, dev)
    n_rows = row_ptr_d.numel() - 1
    n_haps = tables.n_haps
    if out is None:
        out = device_empty((n_rows, n_haps), torch.float64, dev, "the EM input matrix")
    if n_rows == 0:
        return out
    _lib.check(lib.mxm_build_em_matrix(
        exp_d.data_ptr(), exp_d.stride(0), lhit_d.data_ptr(), lmiss_d.data_ptr(),
        row_ptr_d.data_ptr(), site_d.data_ptr(), obs_d.data_ptr(),
        n_rows, n_haps, len(tables.sites), out.data_ptr(), out.stride(0),
        current_stream()), "mxm_build_em_matrix")
    return out


def build_em_matrix(refseq, phylo, reads, haplogroups, args, as_device_tensor=False):
    """
    Drop-in for mixemt.preprocess.build_em_matrix (preprocess.py:177-198):
    returns the float64 [len(reads), len(haplogroups)] log-likelihood matrix
    (C-order numpy array; a device tensor if as_device_tensor=True).
    """
    verbose = getattr(args, "verbose", False)
    if verbose:
        sys.stderr.write("Building EM input matrix...\n")
    tables = HapVarTables.build(refseq, phylo, haplogroups)
    row_ptr, site, obs = encode_signatures(reads, tables)
    mat = build_em_matrix_device(tables, row_ptr, site, obs)
    if verbose:
        sys.stderr.write("  processed %d fragments...\nDone.\n\n" % len(reads))
    if as_device_tensor:
        return mat
    return to_host(mat)


# --------------------------------------------------------------------------
# host front end: alignments -> signatures -> EM input  
# --------------------------------------------------------------------------

def process_reads(alns, var_pos, min_mq, min_bq):
    """
    Alignments -> {fragment id: {0-based variant site: observed base}}
    (reference: preprocess.py:99-139).  `alns` yields pysam.AlignedSegment-like
    objects: mapping_quality, query_name, query_sequence, query_qualities and
    get_aligned_pairs(matches_only=True).  Mates share a fragment id; a site seen
    with two different bases becomes 'N' and is dropped at the end.
    """
    wanted = set(var_pos)
    frags = collections.defaultdict(dict)
    for aln in alns:
        if aln.mapping_quality < min_mq:
            continue
        quals = aln.query_qualities
        seq = aln.query_sequence
        for qpos, rpos in aln.get_aligned_pairs(matches_only=True):
            qpos, rpos = int(qpos), int(rpos)
            if rpos not in wanted:
                continue
            if quals is not None and quals[qpos] < min_bq:
                continue
            base = seq[qpos].upper()
            seen = frags[aln.query_name]
            if rpos in seen and seen[rpos] != base:
                base = "N"
            seen[rpos] = base
    return {name: {pos: base for pos, base in obs.items() if base != "N"}
            for name, obs in frags.items()}


def read_signature(obs_by_pos):
    """{site: base} -> 'site:base,site:base' by ascending site (preprocess.py:142-148)."""
    return ",".join("%d:%s" % (pos, obs_by_pos[pos]) for pos in sorted(obs_by_pos))


def pos_obs_from_sig(read_sig):
    """'site:base,...' -> [(int site, base)] (preprocess.py:151-160)."""
    out = []
    for item in read_sig.split(","):
        pos, base = item.split(":")
        out.append((int(pos), base))
    return out


def reduce_reads(read_obs):
    """{fragment id: observations} -> {signature: [fragment ids]} (preprocess.py:163-174)."""
    by_sig = collections.defaultdict(list)
    for read_id, obs in read_obs.items():
        by_sig[read_signature(obs)].append(read_id)
    return by_sig


class CodedMatrix(object):
    """
    build_em_matrix's output as row-dictionary records (include/mixemt_hip.h, mxm_build_em_records):
        rec, rec_off[R], ndist[R], rowmax[R]   the records (ndist[r] = 0: row r has none)
        used                                   bytes of `rec` in use
        rest_rows [R_rest] int64 (sorted), m_rest [R_rest][H] float64   the rows without a record, dense
    em.EmPlan(None, weights, records=this) iterates it.
    """

    def __init__(self, n_rows, n_haps, rec, rec_off, ndist, rowmax, used, rest_rows, m_rest):
        self.n_rows, self.n_haps = n_rows, n_haps
        self.rec, self.rec_off, self.ndist, self.rowmax = rec, rec_off, ndist, rowmax
        self.used, self.rest_rows, self.m_rest = used, rest_rows, m_rest

    def ndist_host(self):
        """ndist on the host (4 bytes per row, fetched once): the list of wide rows and the byte counts are formed from it with
        numpy -- the first torch.nonzero / sum of a process loads their code objects (10-15 ms in a 20 ms plan)."""
        if getattr(self, "_ndist_host", None) is None:
            self._ndist_host = self.ndist.cpu().numpy()
        return self._ndist_host

    def wide_rows(self):
        """Rows whose record has 16-bit codes (257..1024 distinct values), in row order (device int64)."""
        if getattr(self, "_wide", None) is None:
            self._wide = torch.from_numpy(numpy.flatnonzero(self.ndist_host() > 256)).to(self.rec.device)
        return self._wide

    def rows(self, lo, hi):
        """The rows [lo, hi) as a CodedMatrix of their own (views: nothing is copied but the re-based list of its
        dense rows)."""
        lo, hi = max(0, int(lo)), min(self.n_rows, int(hi))
        if self.rest_rows.numel():
            bounds = torch.searchsorted(self.rest_rows, torch.tensor([lo, hi], dtype=torch.int64, device=self.rest_rows.device))
            a, b = int(bounds[0]), int(bounds[1])
        else:
            a = b = 0
        return CodedMatrix(hi - lo, self.n_haps, self.rec, self.rec_off[lo:hi], self.ndist[lo:hi], self.rowmax[lo:hi],
                           self.used, self.rest_rows[a:b] - lo, self.m_rest[a:b])

    def dense(self, lo=0, hi=None):
        """
        The log matrix rows [lo, hi) as a dense [hi - lo][H] float64 device tensor -- build_em_matrix's own bits (a
        record's log table holds the row's distinct sums; mxm_gather_columns_coded with every column).  For
        io.save_matrix (`-s`) and for tests; the EM and its consumers never need it.
        """
        import ctypes
        part = self.rows(lo, self.n_rows if hi is None else hi)
        dev = self.rec.device
        out = device_empty((part.n_rows, self.n_haps), torch.float64, dev, "rows of the EM matrix decoded from records")
        if part.n_rows:
            cols_d = torch.arange(self.n_haps, dtype=torch.int32, device=dev)
            coded = part.struct()
            n_rest = int(part.rest_rows.numel())
            _lib.check(_lib.load().mxm_gather_columns_coded(
                ctypes.byref(coded), self.n_haps, cols_d.data_ptr(), self.n_haps,
                part.m_rest.data_ptr() if n_rest else 0, part.m_rest.stride(0) if n_rest else 0,
                part.rest_rows.data_ptr() if n_rest else 0, n_rest, out.data_ptr(), out.stride(0), current_stream()),
                "mxm_gather_columns_coded")
        return out

    @property
    def shape(self):
        return (self.n_rows, self.n_haps)

    def struct(self):
        """mxm_coded view of the records alone (the consumers that take it handle coded rows only)."""
        wide = self.wide_rows()
        return _lib.Coded(self.rec.data_ptr(), self.rec_off.data_ptr(), self.ndist.data_ptr(), self.n_rows,
                          None, 0, None, 0, wide.data_ptr() if wide.numel() else None, int(wide.numel()))


def _gather_csr(row_ptr_d, site_d, obs_d, rows):
    """CSR of the given rows only (device tensors; rows int64, any order)."""
    starts = row_ptr_d.index_select(0, rows)
    lens = row_ptr_d.index_select(0, rows + 1) - starts
    new_ptr = torch.zeros(rows.numel() + 1, dtype=torch.int64, device=rows.device)
    torch.cumsum(lens, 0, out=new_ptr[1:])
    total = int(new_ptr[-1].item())
    src = torch.repeat_interleave(starts - new_ptr[:-1], lens, output_size=total) + torch.arange(total, device=rows.device)
    return new_ptr, site_d.index_select(0, src), obs_d.index_select(0, src)


RECORD_BYTES_GUESS = 16 * 96          # table bytes per row the first record buffer allows for (synth-v1 needs 16 x 61 on average,
                                      # wide records included: 6.39 KB per row at H = 5408)
REST_SLAB_ROWS = None                 # rows without a marker-kernel record are built densely and coded a slab at a time: this
                                      # many rows (tests), or by default as many as REST_SLAB_BYTES of dense rows hold
REST_SLAB_BYTES = 2 << 30             # (46 000 rows at 5408 columns: one slab at 10^6 reads, 8 at 10^7 -- a slab ends in a
                                      # host read-back of its byte count, so fewer, larger slabs: 8192 rows cost 41 round trips at 10^7)


def record_buffer_bytes(n_rows, n_haps, mean_sites=None):
    """Size of the record buffer build_em_records_device asks for first (a guess; an overflow repeats the build once).
    mean_sites: observed sites per row, when known -- a row's distinct values grow with them (37 sites: 61 table entries'
    worth of bytes per row, wide records included; 72 sites, merged mates: ~150), so longer rows get more room at once
    instead of paying for the build twice."""
    ldc = (n_haps + 7) // 8 * 8
    one = 2 * ldc + 16 * 1024
    worst = int(n_rows) * one
    guess = RECORD_BYTES_GUESS
    if mean_sites is not None and mean_sites > 40.0:
        guess = int(RECORD_BYTES_GUESS * (mean_sites / 40.0) ** 1.5) // 16 * 16
    return max(one, min(worst, int(n_rows) * (ldc + guess) + (1 << 20)))


def build_em_records_device(tables, row_ptr, site, obs, dense=False, cap=None, rec=None):
    """
    CSR observations -> CodedMatrix: the marker kernel writes each row as a row-dictionary record
    (one byte per haplogroup + the row's distinct values) -- what em.EmPlan(storage="coded") otherwise
    makes from the dense matrix with a pass of its own.  dense=False: NO dense matrix is written (5.5 GB
    instead of 43 GB at 10^6 x 5408; 10^7 rows fit one GPU); rows the marker kernel cannot take (more than
    128 observations, more than 5120 marker entries, more than 704 distinct values: 2 % of 150-bp reads, 4 % of merged mates) are built densely by the lookup-table kernel and
    coded from there (mxm_encode_rows: byte codes up to 256 values, 16-bit codes up to 1024) a slab (REST_SLAB_BYTES) at a
    time, so the detour never holds more than one slab of dense rows (round 4 built all of them at once: 1.4 GB at
    10^6 rows, 14 GB at 10^7); what remains (more than 1024 values: none on build_em_matrix's rows) stays dense in `m_rest`.
    dense=True: returns (CodedMatrix, M) with the full dense matrix as well.
    cap: bytes of the record buffer.  Default: room for RECORD_BYTES_GUESS table bytes per row instead of the
    worst case mxm_record_bytes (27 KB per row: a 27 GB hipMalloc at 10^6 rows for 6 GB of records); the kernels
    count what they would have needed, and an overflow repeats the build once with exactly that much.
    rec: a uint8 device tensor to build into (a buffer the caller already holds); too small a one is only the first attempt.
    (The allocation itself is not what a cold call costs -- torch.empty of 7 GB takes 0.3 ms, profiles/r05/alloc_cost.txt;
    what round 4 read as "the hipMalloc" was code objects being loaded: see the comment in the loop below.)
    """
    lib = _lib.load()
    dev = require_gpu()
    if tables.lut() is None or tables.n_haps > 8192 or not lib.mxm_linear_supported(tables.n_haps):
        raise ValueError("records need tables that qualify for the lookup-table kernel and H in [65, 8192]")
    enc = tables.sparse_device()
    lut = tables.lut_device()
    row_ptr_d = as_device(row_ptr, torch.int64, dev)
    site_d = as_device(site, torch.uint16, dev)
    obs_d = as_device(obs, torch.uint8, dev)
    n_rows, n_haps, n_sites = row_ptr_d.numel() - 1, tables.n_haps, len(tables.sites)
    if n_rows <= 0:
        raise ValueError("no rows")
    ldc = (n_haps + 7) // 8 * 8
    worst = lib.mxm_record_bytes(n_rows, n_haps)
    one = 2 * ldc + 16 * 1024                       # a full wide record: the smallest buffer the library accepts
    if rec is not None and (rec.dtype != torch.uint8 or not rec.is_cuda or rec.numel() < one or rec.data_ptr() % 16):
        rec = None
    if rec is not None:
        cap = rec.numel()
    elif cap is None:
        cap = record_buffer_bytes(n_rows, n_haps, mean_sites=site_d.numel() / float(n_rows))
    cap = max(int(cap), one)
    mat = device_empty((n_rows, n_haps), torch.float64, dev, "the EM input matrix") if dense else None
    rec_off = torch.empty(n_rows, dtype=torch.int64, device=dev)
    ndist = torch.empty(n_rows, dtype=torch.int32, device=dev)
    rowmax = torch.empty(n_rows, dtype=torch.float64, device=dev)
    counters = torch.zeros(5, dtype=torch.int64, device=dev)        # [0:2] the build's stats, [2] n_fallback, [3:5] a slab's stats
    stats, n_fallback, sub_stats = counters[0:2], counters[2:3], counters[3:5]
    fallback = torch.empty(n_rows, dtype=torch.int64, device=dev)
    stream = current_stream()
    timing = {} if os.environ.get("MXM_PIPELINE_TIMING") else None     # (synchronises: measurement only)
    build_em_records_device.last_timing = timing

    def lap(name, t0):
        if timing is not None:
            torch.cuda.synchronize()
            timing[name] = round(timing.get(name, 0.0) + (time.perf_counter() - t0) * 1e3, 2)
        return time.perf_counter()

    def lut_rows(rows_d, count, out):
        """rows rows_d[0 .. count) of the CSR, cell by cell, into the COMPACT matrix out[count][H]"""
        _lib.check(lib.mxm_build_em_matrix_lut_rows(
            lut["ecode"].data_ptr(), lut["ecode"].stride(0), lut["lhit"].data_ptr(), lut["lmiss"].data_ptr(),
            lut["obsmap"].data_ptr(), row_ptr_d.data_ptr(), site_d.data_ptr(), obs_d.data_ptr(), rows_d.data_ptr(), count,
            n_haps, n_sites, out.data_ptr(), out.stride(0), stream), "mxm_build_em_matrix_lut_rows")

    # Everything between the kernels below is pointer arithmetic and three small copies.  HIP loads code objects lazily:
    # the first launch of this library costs 18 ms (profiles/r05/first_launch.txt; _lib.load() now does it: mxm_preload)
    # and the first use of EVERY torch operator loads that operator's module (5-20 ms each).  A version of this function
    # that compacted lists, gathered the long rows' observations and scattered results with torch spent ~60 ms of a cold
    # 105 ms call there -- round 4 took it for the buffer's hipMalloc (profiles/r05/experiments.md section 3).
    for attempt in (0, 1):
        t_lap = lap("uploads", time.perf_counter()) if attempt == 0 and timing is not None else time.perf_counter()
        if rec is None:
            rec = device_empty((cap,), torch.uint8, dev, "the coded matrix")
        _lib.check(lib.mxm_build_em_records(
            enc["maj"].data_ptr(), enc["lhit"].data_ptr(), enc["lmiss"].data_ptr(), enc["mk_ptr"].data_ptr(),
            enc["mk_hap"].data_ptr(), enc["mk_base"].data_ptr(), row_ptr_d.data_ptr(), site_d.data_ptr(), obs_d.data_ptr(),
            0, n_rows, n_haps, n_sites, mat.data_ptr() if dense else 0, mat.stride(0) if dense else 0,
            rec.data_ptr(), cap, rec_off.data_ptr(), ndist.data_ptr(), rowmax.data_ptr(), stats.data_ptr(),
            fallback.data_ptr(), n_fallback.data_ptr(), stream), "mxm_build_em_records")
        used, n_rest, left = (int(v) for v in counters[:3].cpu())
        t_lap = lap("marker kernel", t_lap)
        need = used
        m_rest = None
        rest_rows = fallback[:0]
        if used <= cap:
            if dense:
                if left:
                    rows = fallback[:left].sort().values
                    _lib.check(lib.mxm_build_em_matrix_lut(
                        lut["ecode"].data_ptr(), lut["ecode"].stride(0), lut["lhit"].data_ptr(), lut["lmiss"].data_ptr(),
                        lut["obsmap"].data_ptr(), row_ptr_d.data_ptr(), site_d.data_ptr(), obs_d.data_ptr(), rows.data_ptr(), left,
                        n_haps, n_sites, mat.data_ptr(), mat.stride(0), stream), "mxm_build_em_matrix_lut")
                rest_rows = torch.nonzero(ndist == 0).flatten()      # (also the rows of more than 256 values: dense above)
                assert rest_rows.numel() == n_rest
            else:
                assert left == n_rest             # without a dense matrix every row without a record is on the list
                if n_rest:                        # the list as the kernel appended it -> ascending (host: 8 bytes per row)
                    rest_host = numpy.sort(fallback[:left].cpu().numpy())
                    rest_rows = torch.from_numpy(rest_host).to(dev)
                    t_lap = lap("list sort", t_lap)
            if n_rest:
                # the dense rows of long reads mostly hold few distinct values too: built (or taken from the dense matrix)
                # a slab at a time and coded from their dense form (mxm_encode_rows: bytes, then 16-bit codes) into the
                # tail of the same record buffer; what still has no record afterwards is kept, dense
                slab = min(n_rest, REST_SLAB_ROWS if REST_SLAB_ROWS else max(1024, REST_SLAB_BYTES // (8 * n_haps)))
                m_slab = None if dense else torch.empty((slab, n_haps), dtype=torch.float64, device=dev)
                sub_off = torch.empty(slab, dtype=torch.int64, device=dev)
                sub_nd = torch.empty(slab, dtype=torch.int32, device=dev)
                sub_rm = torch.empty(slab, dtype=torch.float64, device=dev)
                kept_rows, kept_dense, kept_host = [], [], []
                for lo in range(0, n_rest, slab):
                    rows_s = rest_rows[lo:lo + slab]
                    n_s = int(rows_s.numel())
                    if dense:
                        m_s = mat.index_select(0, rows_s)
                    else:
                        m_s = m_slab[:n_s]
                        lut_rows(rows_s, n_s, m_s)
                        t_lap = lap("slab build", t_lap)
                    base = (need + 15) // 16 * 16
                    room = cap - base >= one
                    if room:
                        _lib.check(lib.mxm_encode_rows(m_s.data_ptr(), m_s.stride(0), n_s, n_haps, rec.data_ptr() + base,
                                                       cap - base, sub_off.data_ptr(), sub_nd.data_ptr(), sub_rm.data_ptr(),
                                                       sub_stats.data_ptr(), stream), "mxm_encode_rows")
                        sub_used, sub_left = (int(v) for v in sub_stats.cpu())
                        t_lap = lap("slab encode", t_lap)
                    else:
                        sub_used, sub_left = n_s * one, 0     # no room left at all: ask for the worst case of these rows
                    if room and base + sub_used <= cap:
                        _lib.check(lib.mxm_scatter_records(rows_s.data_ptr(), n_s, sub_off.data_ptr(), sub_nd.data_ptr(),
                                                           sub_rm.data_ptr(), base, rec_off.data_ptr(), ndist.data_ptr(),
                                                           rowmax.data_ptr(), stream), "mxm_scatter_records")
                        if sub_left:                          # more than 1024 values (long reads, random matrices): stays dense
                            if dense:
                                got = sub_nd[:n_s] > 0
                                kept_rows.append(rows_s[~got])
                                kept_dense.append(m_s[~got].clone())
                            else:                             # which ones: on the host (a copy, no torch operator -- see above)
                                kept_host.append(rest_host[lo:lo + n_s][sub_nd[:n_s].cpu().numpy() == 0])
                        t_lap = lap("slab scatter / keep", t_lap)
                    need = base + sub_used
                if need <= cap:
                    used = need
                    if dense:
                        rest_rows = torch.cat(kept_rows) if kept_rows else rest_rows[:0]
                        m_rest = torch.cat(kept_dense) if kept_dense else None
                    elif kept_host:                           # built once more, compactly, straight into their final place
                        kept = numpy.concatenate(kept_host)
                        rest_rows = torch.from_numpy(kept).to(dev)
                        m_rest = torch.empty((len(kept), n_haps), dtype=torch.float64, device=dev)
                        lut_rows(rest_rows, len(kept), m_rest)
                        t_lap = lap("kept rows", t_lap)
                    else:
                        rest_rows = rest_rows[:0]
                    n_rest = int(rest_rows.numel())
        if need <= cap:
            break
        if attempt == 1:
            raise ValueError("build_em_records_device: record buffer of %d bytes overflowed twice (%d needed)" % (cap, need))
        rec = None
        cap = min(worst, need + (n_rest + 1) * one)       # exact for the marker kernel's records, generous for the rest
    build_em_matrix_device.last_fallback = left
    cm = CodedMatrix(n_rows, n_haps, rec, rec_off, ndist, rowmax, used,
                     rest_rows, m_rest if m_rest is not None else torch.empty((0, n_haps), dtype=torch.float64, device=dev))
    cm.capacity = cap
    return (cm, mat) if dense else cm


build_em_matrix_device.last_fallback = 0       # rows the marker kernel handed to the lookup-table kernel, last call


def _warn_dropped(dropped):
    # never silent: weights and read ids no longer cover these fragments (the reference would have died on
    # them), so the accounting must show it whatever the verbosity
    sys.stderr.write("Warning: skipped %d fragment(s) whose every site was conflicted away "
                     "(empty signature; the reference stops at int('') there): %s%s\n"
                     % (len(dropped), ", ".join(str(x) for x in dropped[:5]), ", ..." if len(dropped) > 5 else ""))


def build_em_input(bamfile, refseq, phylo, args, as_device_tensor=False, as_records=False, frontend="auto"):
    """
    Drop-in for mixemt.preprocess.build_em_input (preprocess.py:201-227):
    (em_matrix, weights, haplogroups, read_ids) with rows = sorted distinct
    signatures, weights = fragments per signature, columns = sorted(hap_var).
    A fragment whose every site was conflicted away has the empty signature, on
    which the reference dies (int('') at :156-160); it is skipped here.
    as_records=True: the first element is a CodedMatrix (build_em_records_device: no dense matrix on the
    device or the host) for em.run_em_ex(None, weights, args, records=...).
    frontend: "batched" = the alignments are read into columns and encoded by ONE call of the library
    (alignments.encode_alignments: no interpreter step per aligned base, no signature strings; read_ids is then an
    alignments.ReadIdGroups -- the reference's list of lists, materialised row by row on demand); "python" = the
    reference's own object-by-object walk (process_reads / reduce_reads below); "auto" = batched, falling back to
    python for input the encoder hands back (so that the exception raised is the reference's).
    `bamfile` may also be an alignments.AlignmentColumns (a columnar reader's output) or the PATH of a BAM file, which
    the library's own reader turns into columns (alignments.read_bam: no pysam, no object per alignment; batched only).
    An open handle that names its file (`pysam.AlignmentFile.filename`) is read the same way -- fetch() is only called
    for what the reader does not take or the encoder hands back (`build_em_input.last_source` says which it was).
    """
    from . import alignments
    if frontend not in ("auto", "batched", "python"):
        raise ValueError("frontend must be 'auto', 'batched' or 'python'")
    var_pos = phylo.get_variant_pos()
    haplogroups = sorted(phylo.hap_var)
    verbose = getattr(args, "verbose", False)
    enc = None
    columns = bamfile if isinstance(bamfile, alignments.AlignmentColumns) else None
    if isinstance(bamfile, (str, bytes, os.PathLike)):
        if frontend == "python":
            raise ValueError("build_em_input: a BAM path goes through the batched front end (open it with pysam for 'python')")
        columns = alignments.read_bam(bamfile)
    build_em_input.last_source = "columns" if columns is not None else "objects"
    if frontend != "python":
        alns = None
        handle_file = None
        if columns is None:
            # an open pysam.AlignmentFile (what bin/mixemt:139-147 passes) knows its file: read THAT with the library's own
            # reader instead of taking one Python object per alignment from fetch() (4.3 s per 10^6 against 0.2 s) -- the
            # same records in the same order for a coordinate-sorted BAM; anything the reader does not take (CRAM, SAM
            # text, a stream) falls back to the objects
            name = getattr(bamfile, "filename", None)
            if isinstance(name, (str, bytes, os.PathLike)) and name not in ("-", b"-"):
                try:
                    columns = alignments.read_bam(os.fsdecode(name))
                    handle_file = name
                    build_em_input.last_source = "file of the handle"
                except (OSError, ValueError):
                    columns = None
        try:
            if columns is None:
                alns = list(bamfile.fetch())          # (kept: the python path below must see the same alignments)
                columns = alignments.AlignmentColumns.from_alignments(alns)
            enc = alignments.encode_alignments(columns, var_pos, len(refseq), args.min_mq, args.min_bq)
        except alignments.NeedsSlowPath:
            if frontend == "batched" or (alns is None and handle_file is None):
                raise
            if alns is None:                          # the encoder handed a FILE's alignments back: take the objects after all
                alns = list(bamfile.fetch())
            enc = None
    if enc is not None:
        dropped = enc.dropped
        if dropped:
            _warn_dropped(dropped)
        build_em_input.last_dropped = list(dropped)
        if verbose:
            sys.stderr.write("Using %d aligned fragments (MQ>=%d) (%d distinct sub-haplotypes)\n\n"
                             % (enc.n_fragments - len(dropped), args.min_mq, enc.n_rows))
        tables = HapVarTables.build(refseq, phylo, haplogroups)
        if not numpy.array_equal(tables.sites, numpy.asarray(var_pos, dtype=numpy.int64)):
            raise ValueError("build_em_input: the tables' sites are not phylo.get_variant_pos()")
        if verbose:
            sys.stderr.write("Building EM input matrix...\n")
        if as_records:
            em_matrix = build_em_records_device(tables, enc.row_ptr, enc.site, enc.obs)
        else:
            em_matrix = build_em_matrix_device(tables, enc.row_ptr, enc.site, enc.obs)
            if not as_device_tensor:
                em_matrix = to_host(em_matrix)
        if verbose:
            sys.stderr.write("  processed %d fragments...\nDone.\n\n" % enc.n_rows)
        build_em_input.last_frontend = "batched"
        return em_matrix, enc.weights, haplogroups, enc.read_ids
    build_em_input.last_frontend = "python"
    read_obs = process_reads(alns if frontend != "python" else bamfile.fetch(), var_pos, args.min_mq, args.min_bq)
    read_sigs = reduce_reads(read_obs)
    dropped = read_sigs.pop("", None)
    if dropped:
        _warn_dropped(dropped)
        build_em_input.last_dropped = list(dropped)
    else:
        build_em_input.last_dropped = []
    if verbose:
        sys.stderr.write("Using %d aligned fragments (MQ>=%d) (%d distinct sub-haplotypes)\n\n"
                         % (len(read_obs) - len(dropped or ()), args.min_mq, len(read_sigs)))
    reads = sorted(read_sigs)
    weights = numpy.array([len(read_sigs[r]) for r in reads])
    if as_records:
        tables = HapVarTables.build(refseq, phylo, haplogroups)
        row_ptr, site, obs = encode_signatures(reads, tables)
        em_matrix = build_em_records_device(tables, row_ptr, site, obs)
    else:
        em_matrix = build_em_matrix(refseq, phylo, reads, haplogroups, args,
                                    as_device_tensor=as_device_tensor)
    read_ids = [read_sigs[r] for r in reads]
    return em_matrix, weights, haplogroups, read_ids


build_em_input.last_dropped = []
build_em_input.last_frontend = None


def reduce_em_matrix(em_mat, haplogroups, contrib_props):
    """
    Column subset for the refinement EM (preprocess.py:230-251); works on numpy
    arrays and device tensors alike.
    """
    keep = {con[1] for con in contrib_props}
    idx = [i for i, hap in enumerate(haplogroups) if hap in keep]
    names = [haplogroups[i] for i in idx]
    if torch is not None and isinstance(em_mat, torch.Tensor):
        return gather_columns_device(em_mat, idx), names
    return em_mat[:, idx], names


def reduce_em_records(cm, haplogroups, contrib_props):
    """
    reduce_em_matrix (preprocess.py:230-251) for a matrix that exists only as records (CodedMatrix):
    [R][#contributors] float64 on the device -- coded rows through mxm_gather_columns_coded (the records'
    log tables), the rows without a record from their dense copies.
    """
    import ctypes
    lib = _lib.load()
    keep = {con[1] for con in contrib_props}
    idx = [i for i, hap in enumerate(haplogroups) if hap in keep]
    names = [haplogroups[i] for i in idx]
    dev = cm.rec.device
    out = device_empty((cm.n_rows, len(idx)), torch.float64, dev, "the reduced EM matrix")
    if cm.n_rows and idx:
        cols_d = torch.tensor(idx, dtype=torch.int32, device=dev)
        coded = cm.struct()
        n_rest = int(cm.rest_rows.numel())
        _lib.check(lib.mxm_gather_columns_coded(ctypes.byref(coded), cm.n_haps, cols_d.data_ptr(), len(idx),
                                                cm.m_rest.data_ptr() if n_rest else 0, cm.m_rest.stride(0) if n_rest else 0,
                                                cm.rest_rows.data_ptr() if n_rest else 0, n_rest,
                                                out.data_ptr(), out.stride(0), current_stream()),
                   "mxm_gather_columns_coded")
    return out, names


def gather_columns_device(em_mat, idx):
    """em_mat[:, idx] on the device (mxm_gather_columns): [R][len(idx)] float64, contiguous."""
    lib = _lib.load()
    n_rows, n_haps = em_mat.shape
    if em_mat.dtype != torch.float64 or em_mat.stride(1) != 1:
        raise ValueError("reduce_em_matrix: the device matrix must be float64 with unit column stride")
    cols = numpy.asarray(idx, dtype=numpy.int32)
    if cols.size and (cols.min() < 0 or cols.max() >= n_haps):
        raise ValueError("reduce_em_matrix: column index out of range")
    out = device_empty((n_rows, len(cols)), torch.float64, em_mat.device, "the reduced EM matrix")
    if n_rows and len(cols):
        cols_d = torch.from_numpy(cols).to(em_mat.device)
        _lib.check(lib.mxm_gather_columns(em_mat.data_ptr(), em_mat.stride(0), n_rows, n_haps,
                                          cols_d.data_ptr(), len(cols), out.data_ptr(), out.stride(0),
                                          current_stream()), "mxm_gather_columns")
    return out
