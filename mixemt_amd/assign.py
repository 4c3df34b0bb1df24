"""
Consumers of the EM result that reduce the [R][H] posterior matrix to one small
value per read -- the "next" rows f-1..f-3 of SURVEY.md section 8.  On the
device they are single streaming / gather passes over a matrix that is already
resident, and they remove the 43 GB device->host copy of `read_mix` the
reference's NumPy versions would need:

    find_contribs_from_reads   <- assemble._find_contribs_from_reads  assemble.py:103-123
    read_votes / report_read_votes <- stats.report_read_votes          stats.py:34-45
    update_contribs            <- assemble.update_contribs             assemble.py:211-230
    assign_read_indexes        <- assemble.assign_read_indexes         assemble.py:284-334
                                  (with _find_best_n_for_read :267-281)

Matrices may be numpy arrays (uploaded) or ROCm tensors.
"""

import collections
import collections.abc
import sys

import numpy

from . import _lib
from ._dev import as_device, current_stream, ptr, require_gpu, torch


def _contributors_on_device(best_d, votes_d, n_haps, min_reads):
    """
    Columns with at least min_reads votes in the order their haplogroup first won a row (assemble.py:115-123): the
    first-seen row of every haplogroup is formed on the device (mxm_first_seen), so 2 x H values come back instead of
    best[R] (4 MB and a numpy.unique over R at 10^6 rows: 23 ms for a 1.5 ms pass).
    """
    lib = _lib.load()
    n_rows = best_d.numel()
    first = torch.empty(n_haps, dtype=torch.int64, device=best_d.device)
    _lib.check(lib.mxm_first_seen(best_d.data_ptr(), n_rows, n_haps, first.data_ptr(), current_stream()), "mxm_first_seen")
    first_h, votes_h = first.cpu().numpy(), votes_d.cpu().numpy()
    seen = numpy.flatnonzero(first_h < n_rows)
    order = seen[numpy.argsort(first_h[seen], kind="stable")]
    return [int(h) for h in order if votes_h[h] >= min_reads], order, votes_h


def _row_argmax_votes_device(read_hap_mat, wts):
    lib = _lib.load()
    dev = require_gpu()
    mat = as_device(read_hap_mat, torch.float64, dev)
    n_rows, n_haps = mat.shape
    w_d = None if wts is None else as_device(wts, torch.float64, dev)
    best = torch.empty(n_rows, dtype=torch.int32, device=dev)
    votes = torch.zeros(n_haps, dtype=torch.float64, device=dev)
    if n_rows:
        nbytes = lib.mxm_workspace_bytes(n_rows, n_haps, 1)       # per-workgroup vote rows (no float atomics)
        ws = torch.empty(nbytes // 8 + 1, dtype=torch.float64, device=dev)
        _lib.check(lib.mxm_row_argmax_votes(mat.data_ptr(), mat.stride(0), ptr(w_d), n_rows, n_haps,
                                            best.data_ptr(), votes.data_ptr(), ws.data_ptr(), nbytes,
                                            current_stream()), "mxm_row_argmax_votes")
    return best, votes


def row_argmax_votes(read_hap_mat, wts=None):
    """
    best[r] = first index of the row maximum (numpy.argmax semantics) and
    votes[h] = sum of wts over the rows that picked h (mxm_row_argmax_votes).
    Returns (best int32[R], votes float64[H]) as numpy arrays.
    """
    best, votes = _row_argmax_votes_device(read_hap_mat, wts)
    return best.cpu().numpy(), votes.cpu().numpy()


def row_argmax_votes_records(cm, ln_theta_k, wts=None):
    """
    row_argmax_votes of run_em's returned posterior for a matrix that exists only as records
    (preprocess.CodedMatrix) -- neither the dense matrix nor the posterior matrix is made
    (mxm_row_argmax_votes_coded).  ln_theta_k: the log theta_k of the run ([H]) or of every run of a multi-run
    ([n_multi][H], em.run_em_ex's "ln_theta_k"): the reference votes on the logaddexp fold of the runs' posteriors
    (em.py:156 -> assemble.py:115-123), in which each run's row normaliser weighs that run's columns -- with one
    run it drops out and best = argmax_h (ln_theta[h] + M[r][h]).  Rows without a record are read from their
    dense copies by the same entry point; votes are summed without float atomics.
    Returns (best int32[R], votes float64[H]) as numpy arrays.
    """
    best, votes = _row_argmax_votes_records_device(cm, ln_theta_k, wts)
    return best.cpu().numpy(), votes.cpu().numpy()


def _row_argmax_votes_records_device(cm, ln_theta_k, wts=None):
    import ctypes
    lib = _lib.load()
    dev = cm.rec.device
    if isinstance(ln_theta_k, torch.Tensor):
        lnp = as_device(ln_theta_k, torch.float64, dev)
        lnp = (lnp.reshape(1, -1) if lnp.dim() == 1 else lnp).contiguous()
        props = torch.exp(lnp)
    else:
        # H values per run: exponentiated on the host (the process's first torch.exp on the device loads torch's
        # elementwise kernels -- 18 ms that landed in this 3 ms stage)
        host = numpy.atleast_2d(numpy.ascontiguousarray(ln_theta_k, dtype=numpy.float64))
        lnp = torch.from_numpy(host).to(dev)
        props = torch.from_numpy(numpy.exp(host)).to(dev)
    n_runs = lnp.shape[0]
    if lnp.dim() != 2 or lnp.shape[1] != cm.n_haps:
        raise ValueError("ln_theta_k does not match the matrix width")
    best = torch.zeros(cm.n_rows, dtype=torch.int32, device=dev)
    votes = torch.zeros(cm.n_haps, dtype=torch.float64, device=dev)
    w_d = None if wts is None else as_device(wts, torch.float64, dev)
    nbytes = lib.mxm_workspace_bytes(cm.n_rows, cm.n_haps, 1)
    ws = torch.empty(nbytes // 8 + 1, dtype=torch.float64, device=dev)
    coded = cm.struct()
    n_rest = int(cm.rest_rows.numel())
    _lib.check(lib.mxm_row_argmax_votes_coded(
        ctypes.byref(coded), cm.n_haps, n_runs, lnp.data_ptr(), props.data_ptr(), cm.rowmax.data_ptr(),
        cm.m_rest.data_ptr() if n_rest else 0, cm.m_rest.stride(0) if n_rest else 0,
        cm.rest_rows.data_ptr() if n_rest else 0, n_rest, ptr(w_d), best.data_ptr(), votes.data_ptr(),
        ws.data_ptr(), nbytes, current_stream()), "mxm_row_argmax_votes_coded")
    return best, votes


def find_contribs_from_records(cm, ln_theta_k, wts, args):
    """find_contribs_from_reads (assemble.py:103-123) from records and the EM's log theta_k ([H], or [n_multi][H]
    for a multi-run: see row_argmax_votes_records)."""
    best, votes = _row_argmax_votes_records_device(cm, ln_theta_k, wts)
    return _contributors_on_device(best, votes, cm.n_haps, args.min_reads)[0]


def vote_table_from_records(cm, ln_theta_k, wts=None):
    """(columns in first-seen order, votes[H]) of the records' vote (stats.py:34-45 / assemble.py:115-123); H-sized
    arrays only leave the device."""
    best, votes = _row_argmax_votes_records_device(cm, ln_theta_k, wts)
    _, order, votes_h = _contributors_on_device(best, votes, cm.n_haps, 0)
    return order, votes_h


def _first_seen_order(best):
    """Column indexes in the order they first appear in `best` (dict insertion order
    of the reference's vote table, assemble.py:116-119)."""
    uniq, first = numpy.unique(best, return_index=True)
    return uniq[numpy.argsort(first, kind="stable")]


def find_contribs_from_reads(read_hap_mat, wts, args):
    """
    Haplogroup columns that are the most probable source of at least
    args.min_reads fragments (assemble.py:103-123), in the reference's order
    (first appearance among the rows).
    """
    best, votes = _row_argmax_votes_device(read_hap_mat, wts)
    return _contributors_on_device(best, votes, votes.numel(), args.min_reads)[0]


def read_votes(read_hap_mat):
    """Counter {column: number of rows voting for it}, unweighted (stats.py:39-40),
    with the reference's insertion order so that most_common() breaks ties alike."""
    best, votes = _row_argmax_votes_device(read_hap_mat, None)
    _, order, votes_h = _contributors_on_device(best, votes, votes.numel(), 0)
    counter = collections.Counter()
    for h in order:
        counter[int(h)] = int(votes_h[h])
    return counter


def report_read_votes(haplogroups, read_hap_mat, top_n=10):
    """stats.report_read_votes (stats.py:34-45), same text on stderr."""
    sys.stderr.write("\nTop 10 haplogroups by read probabilities...\n")
    for hap_i, count in read_votes(read_hap_mat).most_common(top_n):
        sys.stderr.write("%s\t%d\n" % (haplogroups[hap_i], count))
    sys.stderr.write("\n")


def update_contribs(contribs, em_results, haps):
    """assemble.update_contribs (assemble.py:211-230): refined proportions by name."""
    props, _ = em_results
    by_hap = {haps[i]: props[i] for i in range(len(haps))}
    for con in contribs:
        con[2] = by_hap[con[1]]
    return contribs


class AssignedReads(collections.abc.Mapping):
    """
    assign_read_indexes' result -- contributor name -> set of row indexes, plus 'unassigned' (assemble.py:284-334) -- held
    as ONE small integer per row: the sets are only formed when someone looks at them (a 10^6-row table of Python
    integers costs 40 ms to build and nothing downstream of the EM needs it before a writer asks).  Behaves like the
    reference's defaultdict(set) for reading: keys are the names that got at least one row (contributors in their
    order, then 'unassigned'), table[name] is a set (empty for a name without rows), dict(table) the reference's dict.
        count(name)   rows of `name` without forming the set
        rows(name)    their indexes as a numpy array (ascending)
    """

    def __init__(self, assigned, names):
        self._assigned = numpy.asarray(assigned)             # ordinal of the contributor, -1 = unassigned
        self._names = list(names)
        counts = numpy.bincount(self._assigned[self._assigned >= 0], minlength=len(self._names)) if self._assigned.size \
            else numpy.zeros(len(self._names), dtype=numpy.int64)
        self._counts = {name: int(counts[i]) for i, name in enumerate(self._names)}
        n_un = int((self._assigned < 0).sum())
        self._keys = [name for name in self._names if self._counts[name] > 0]
        if n_un:
            self._counts["unassigned"] = n_un
            self._keys.append("unassigned")
        self._sets = {}

    def rows(self, name):
        if name == "unassigned" and "unassigned" not in self._names:
            return numpy.flatnonzero(self._assigned < 0)
        if name not in self._names:
            return numpy.zeros(0, dtype=numpy.int64)
        return numpy.flatnonzero(self._assigned == self._names.index(name))

    def count(self, name):
        return self._counts.get(name, 0)

    def __getitem__(self, name):
        got = self._sets.get(name)
        if got is None:
            got = set(self.rows(name).tolist())
            if name in self._keys:
                self._sets[name] = got
        return got

    def __iter__(self):
        return iter(self._keys)

    def __len__(self):
        return len(self._keys)

    def __contains__(self, name):
        return name in self._keys

    def __eq__(self, other):
        try:
            return dict(self) == dict(other)
        except (TypeError, ValueError):
            return NotImplemented

    def __repr__(self):
        return "AssignedReads(%s)" % ", ".join("%s: %d" % (k, self._counts[k]) for k in self._keys)


def assign_read_indexes(contribs, em_results, haps, reads, min_fold):
    """
    assemble.assign_read_indexes (assemble.py:284-334): contributor name -> set
    of row indexes, plus 'unassigned'; a row goes to its best contributor when,
    after dividing out the mixture proportions, it beats the runner-up by
    min_fold.  One gather kernel (mxm_assign_reads) instead of an argsort of
    all H columns per row; the result is an AssignedReads (one integer per row, sets formed on demand).
    """
    props, read_hap_mat = em_results
    n_rows = len(reads)
    names = [hap_n for hap_n, _, _ in contribs]
    if len(contribs) <= 1:
        return AssignedReads(numpy.zeros(n_rows, dtype=numpy.int32), names[:1])
    lib = _lib.load()
    dev = require_gpu()
    mat = as_device(read_hap_mat, torch.float64, dev)
    with numpy.errstate(divide="ignore"):
        log_props = numpy.log(numpy.asarray(props, dtype=numpy.float64))
    cols = numpy.array([haps.index(group) for _, group, _ in contribs], dtype=numpy.int32)
    lp_d = torch.from_numpy(log_props).to(dev)
    cols_d = torch.from_numpy(cols).to(dev)
    assigned = torch.empty(n_rows, dtype=torch.int32, device=dev)
    if n_rows:
        _lib.check(lib.mxm_assign_reads(mat.data_ptr(), mat.stride(0), lp_d.data_ptr(), cols_d.data_ptr(),
                                        len(cols), n_rows, mat.shape[1], float(numpy.log(min_fold)),
                                        assigned.data_ptr(), current_stream()), "mxm_assign_reads")
    return AssignedReads(assigned.cpu().numpy(), names)
