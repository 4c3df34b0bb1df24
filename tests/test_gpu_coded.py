"""
Row-dictionary storage (mxm_encode_rows / mxm_em_iter_coded / mxm_em_loop_coded; EmPlan(storage="coded")):
lossless by construction, so the bar is the dense fp64 path's own -- decoded rows equal mxm_linearize's
output bit for bit, and the EM run reproduces the reference's goldens (same iteration counts, same
haplogroup calls, proportions within 1e-9).
"""
import ctypes

import numpy
import pytest

from conftest import em_args, golden
from oracle import c_oracle, em_oracle

pytestmark = pytest.mark.gpu

PROPS_ATOL = 1e-9


def _b17_matrix(tables, g, n_haps):
    return c_oracle.build_em_matrix(tables.expected, tables.lhit, tables.lmiss, g["row_ptr"],
                                    g["site"], g["obs"], n_haps)


def _decode(plan):
    """The coded plan's rows back as a dense P (coded rows only; the others stay NaN)."""
    import torch
    from mixemt_amd import _lib
    from mixemt_amd._dev import current_stream
    out = torch.full((plan.n_rows, plan.n_haps), float("nan"), dtype=torch.float64, device=plan.dev)
    _lib.check(plan.lib.mxm_decode_rows(ctypes.byref(plan.coded), plan.n_haps, out.data_ptr(), out.stride(0),
                                        current_stream()), "mxm_decode_rows")
    return out


def _ndist(plan):
    return plan.coded_ndist.cpu().numpy()


@pytest.mark.parametrize("name", ["g4_run_em", "g9_run_em_2400"])
def test_encoded_rows_decode_to_the_linearised_matrix_bit_for_bit(b17, name):
    import torch
    from mixemt_amd import em
    refseq, phy, haps, tables = b17
    g = golden(name)
    mat = _b17_matrix(tables, g, len(haps))
    dense = em.EmPlan(mat, g["wts"], storage="f64")
    coded = em.EmPlan(mat, g["wts"], storage="coded")
    assert coded.coded is not None and coded.lin is None
    nd = _ndist(coded)
    assert (nd > 0).sum() + coded.coded_rest == len(nd) and (nd <= 1024).all()
    assert (nd > 0).mean() > 0.9                      # build_em_matrix rows hold few distinct sums
    assert coded.coded_rest == 0                      # ... and never more than 1024: no row stays dense (round 4)
    assert coded.coded_wide == (nd > 256).sum()
    # the dictionary of a row has exactly as many entries as the row has distinct values -- byte-coded or wide
    for r in list(numpy.flatnonzero(nd > 0)[:50]) + list(numpy.flatnonzero(nd > 256)):
        assert nd[r] == len(numpy.unique(mat[r]))
    dec = _decode(coded)
    rows = torch.from_numpy(nd > 0).to(dec.device)
    assert torch.equal(dec[rows].view(torch.int64), dense.lin[rows][:, :len(haps)].view(torch.int64))
    assert torch.equal(coded.rowmax, dense.rowmax)
    assert coded.coded_bytes < 0.25 * mat.size * 8    # ~8x smaller on these matrices


def test_one_iteration_equals_the_dense_pass(b17):
    """Same proportions in -> same column sums out (summation order differs: 1e-13 relative), three restarts."""
    import torch
    from mixemt_amd import em
    refseq, phy, haps, tables = b17
    g = golden("g9_run_em_2400")
    mat = _b17_matrix(tables, g, len(haps))
    dense = em.EmPlan(mat, g["wts"])
    coded = em.EmPlan(mat, g["wts"], storage="coded")
    rng = numpy.random.default_rng(1)
    init = rng.dirichlet([1.0] * len(haps), size=3)
    props = torch.from_numpy(init).to(dense.dev)
    lnp = torch.log(props)
    state = em.new_state(3, dense.dev)
    a, b = torch.zeros_like(props), torch.zeros_like(props)
    dense.em_iter(props, lnp, state, a)
    coded.em_iter(props, lnp, state, b)
    assert float(((a - b).abs() / a.abs()).max()) < 1e-12
    # and against the oracle's M-step: sum_r w_r posterior = p_h T_h
    buf = numpy.empty_like(mat)
    _, theta = em_oracle.em_step(mat, g["wts"], numpy.log(init[0]), buf)
    mine = (props[0] * b[0]).cpu().numpy()
    assert numpy.allclose(mine / mine.sum(), numpy.exp(theta), rtol=0, atol=1e-13)


def test_run_em_goldens_in_coded_storage(b17):
    """g4 (600 rows), g9 (2400 rows, repeat weights, 1209 iterations), g5 (three restarts): the reference's
    iteration counts, haplogroup calls and proportions from the coded loop."""
    from mixemt_amd import assign, em
    refseq, phy, haps, tables = b17
    for name, seed, n_multi in (("g4_run_em", 7, 1), ("g9_run_em_2400", 17, 1), ("g5_run_em_multi", 11, 3)):
        g = golden(name)
        mat = _b17_matrix(tables, g, len(haps))
        numpy.random.seed(seed)
        res = em.run_em_ex(mat, g["wts"], em_args(n_multi=n_multi), storage="coded")
        assert numpy.array_equal(res["inits"], g["inits"])
        assert res["iters"] == list(g["iters"]), name
        assert numpy.abs(res["props"] - g["props"]).max() < PROPS_ATOL
        best, votes = assign.row_argmax_votes(res["read_mix"], g["wts"])
        assert numpy.array_equal(best, g["mix_argmax"])
        if "votes" in g.files:
            assert numpy.array_equal(votes, g["votes"])


def test_rows_that_do_not_code_stay_dense_and_count():
    """Random matrices have H distinct values per row: every row takes the dense rest, the result is the
    dense path's; a mixed matrix (half dictionary rows) sums both parts."""
    from mixemt_amd import em
    rng = numpy.random.default_rng(5)
    n_rows, n_haps = 300, 1500                                       # 1500 distinct values per random row: beyond wide records
    mat = rng.normal(-25.0, 8.0, size=(n_rows, n_haps))
    few = rng.normal(-25.0, 8.0, size=(n_rows, 7))
    pick = rng.integers(0, 7, size=(n_rows, n_haps))
    half = numpy.arange(n_rows) % 2 == 0
    mat[half] = numpy.take_along_axis(few, pick, axis=1)[half]       # 7 distinct values per even row
    mat[4, :] = -3.0                                                 # one value
    mat[6, :300] = -numpy.inf                                        # -inf is a value like any other
    wts = rng.integers(1, 5, size=n_rows)
    init = rng.dirichlet([1.0] * n_haps)
    plan = em.EmPlan(mat, wts, storage="coded")
    nd = _ndist(plan)
    assert plan.coded_rest == n_rows // 2 and (nd[~half] == 0).all() and (nd[half] > 0).all() and nd[4] == 1
    res = em.run_em_ex(mat, wts, em_args(max_iter=6, tolerance=0.0), inits=init[None, :], storage="coded")
    theta = numpy.log(init)
    buf = numpy.empty_like(mat)
    for _ in range(6):
        buf, theta = em_oracle.em_step(mat, wts, theta, buf)
    assert res["iters"] == [6]
    assert numpy.abs(res["props"] - numpy.exp(theta)).max() < 1e-12
    # all rows dense / all rows coded
    for sub in (mat[~half], mat[half]):
        r2 = em.run_em_ex(sub, wts[:len(sub)], em_args(max_iter=3, tolerance=0.0), inits=init[None, :], storage="coded")
        theta = numpy.log(init)
        buf = numpy.empty_like(sub)
        for _ in range(3):
            buf, theta = em_oracle.em_step(sub, wts[:len(sub)], theta, buf)
        assert numpy.abs(r2["props"] - numpy.exp(theta)).max() < 1e-12


@pytest.mark.parametrize("n_rows,n_haps", [(1, 66), (5, 5408), (700, 130), (513, 8192), (40, 2050)])
def test_shapes_and_ragged_ends(n_rows, n_haps):
    """Widths across the kernel's chunk counts (H % 4 == 2 included), fewer rows than workgroups."""
    from mixemt_amd import em
    rng = numpy.random.default_rng(n_rows + n_haps)
    few = rng.normal(-20.0, 6.0, size=(n_rows, 40))
    mat = numpy.take_along_axis(few, rng.integers(0, 40, size=(n_rows, n_haps)), axis=1)
    wts = rng.integers(1, 4, size=n_rows)
    init = rng.dirichlet([1.0] * n_haps)
    res = em.run_em_ex(mat, wts, em_args(max_iter=4, tolerance=0.0), inits=init[None, :], storage="coded")
    theta = numpy.log(init)
    buf = numpy.empty_like(mat)
    for _ in range(4):
        buf, theta = em_oracle.em_step(mat, wts, theta, buf)
    assert numpy.abs(res["props"] - numpy.exp(theta)).max() < 1e-12


@pytest.mark.parametrize("n_rows,n_haps,seed", [(700, 2000, 1), (37, 5408, 2), (3, 1100, 3), (1300, 1026, 4)])
def test_wide_records_sixteen_bit_codes(n_rows, n_haps, seed):
    """
    Rows of 257..1024 distinct values get records with 16-bit codes (round 4: they used to stay dense and cost a kernel
    launch of their own per iteration): decode == mxm_linearize bit for bit, the table has one entry per distinct
    value, the EM iteration over a mix of byte-coded, wide and dense rows equals the oracle, and so do the record
    consumers (posterior, column gather, argmax votes).
    """
    import torch
    from mixemt_amd import assign, em, preprocess
    rng = numpy.random.default_rng(seed)
    kinds = rng.integers(0, 4, size=n_rows)                        # 0: few values, 1: ~300, 2: exactly up to 1024, 3: too many
    kinds[:4] = [0, 1, 2, 3][: min(4, n_rows)]
    want_d = numpy.where(kinds == 0, 9, numpy.where(kinds == 1, 300, numpy.where(kinds == 2, min(1024, n_haps), n_haps)))
    mat = numpy.empty((n_rows, n_haps))
    for r in range(n_rows):
        vals = rng.normal(-20.0, 6.0, size=want_d[r])
        idx = rng.integers(0, want_d[r], size=n_haps)
        idx[rng.permutation(n_haps)[: want_d[r]]] = numpy.arange(want_d[r])     # every value occurs
        mat[r] = vals[idx]
    if n_rows > 10:
        mat[9, : n_haps // 3] = -numpy.inf                          # -inf is a value like any other
    wts = rng.integers(1, 4, size=n_rows).astype(numpy.float64)
    plan = em.EmPlan(mat, wts, storage="coded")
    nd = _ndist(plan)
    uniq = numpy.array([len(numpy.unique(row)) for row in mat])
    assert numpy.array_equal(nd, numpy.where(uniq <= 1024, uniq, 0))
    assert plan.coded_wide == ((uniq > 256) & (uniq <= 1024)).sum() and plan.coded_rest == (uniq > 1024).sum()
    dense = em.EmPlan(mat, wts, storage="f64")
    dec = _decode(plan)
    rows = torch.from_numpy(nd > 0).to(dec.device)
    assert torch.equal(dec[rows].view(torch.int64), dense.lin[rows][:, :n_haps].view(torch.int64))
    init = rng.dirichlet([1.0] * n_haps)
    res = em.run_em_ex(mat, wts, em_args(max_iter=5, tolerance=0.0), inits=init[None, :], storage="coded")
    theta = numpy.log(init)
    buf = numpy.empty_like(mat)
    with numpy.errstate(invalid="ignore"):
        for _ in range(5):
            last = theta
            buf, theta = em_oracle.em_step(mat, wts, theta, buf)
    assert res["iters"] == [5] and numpy.abs(res["props"] - numpy.exp(theta)).max() < 1e-12
    # consumers of the records' log tables: posterior under theta_k, column gather, argmax + votes
    cm = preprocess.CodedMatrix(n_rows, n_haps, *plan._coded_keep[:3], plan.rowmax, 0,
                                torch.nonzero(plan.coded_ndist == 0).flatten(),
                                dense.mat[torch.nonzero(plan.coded_ndist == 0).flatten()].contiguous())
    rec_plan = em.EmPlan(None, wts, records=cm)
    post = em.posterior(rec_plan, last).cpu().numpy()
    finite = numpy.isfinite(buf)
    assert numpy.array_equal(numpy.isfinite(post), finite) and numpy.abs(post[finite] - buf[finite]).max() < 1e-10
    cols = sorted(rng.choice(n_haps, size=5, replace=False).tolist())
    haps = ["h%d" % i for i in range(n_haps)]
    sub, _ = preprocess.reduce_em_records(cm, haps, [["x", haps[c], 0.0] for c in cols])
    assert numpy.array_equal(sub.cpu().numpy(), mat[:, cols])
    best, votes = assign.row_argmax_votes_records(cm, last, wts)
    want_best = (last[None, :] + mat).argmax(axis=1)
    assert numpy.array_equal(best, want_best)
    assert numpy.array_equal(votes, numpy.bincount(want_best, weights=wts, minlength=n_haps))
    # a NaN counts as the maximum, the first one (numpy.argmax): the kernels' general form, byte-coded and wide rows
    poisoned = last.copy()
    poisoned[[n_haps // 2, n_haps - 3]] = numpy.nan
    poisoned[: n_haps // 4] = -numpy.inf
    best_n, _ = assign.row_argmax_votes_records(cm, poisoned, wts)
    with numpy.errstate(invalid="ignore"):
        want_n = (poisoned[None, :] + mat).argmax(axis=1)
    assert numpy.array_equal(best_n, want_n) and (want_n == n_haps // 2).all()
    poisoned[[n_haps // 2, n_haps - 3]] = -numpy.inf           # ... and without it: -inf columns never win
    best_i, _ = assign.row_argmax_votes_records(cm, poisoned, wts)
    with numpy.errstate(invalid="ignore"):
        assert numpy.array_equal(best_i, (poisoned[None, :] + mat).argmax(axis=1))


def test_shapes_outside_the_record_kernels_iterate_as_fp64_and_odd_widths_do_not():
    """Fewer than 65 columns: the plan says so and runs the dense path (same as the f32 variant's rule).  An ODD width
    (round 5; it used to fall back too) is coded like any other: few distinct values per row, any row stride."""
    from mixemt_amd import em
    rng = numpy.random.default_rng(3)
    mat = rng.normal(-20.0, 5.0, size=(50, 9))
    plan = em.EmPlan(mat, numpy.ones(50), storage="coded")
    assert plan.coded is None and plan.storage == "f64"
    for n_haps in (67, 255, 1001):
        few = rng.normal(-15.0, 4.0, size=(50, 7))
        mat = numpy.take_along_axis(few, rng.integers(0, 7, size=(50, n_haps)), axis=1)
        wts = rng.integers(1, 4, size=50).astype(numpy.float64)
        init = rng.dirichlet([1.0] * n_haps)
        plan = em.EmPlan(mat, wts, storage="coded")
        assert plan.coded is not None and plan.storage == "coded" and plan.coded_rest == 0
        res = em.run_em_ex(mat, wts, em_args(max_iter=6, tolerance=0.0), inits=init[None, :], storage="coded")
        theta = numpy.log(init)
        buf = numpy.empty_like(mat)
        for _ in range(6):
            buf, theta = em_oracle.em_step(mat, wts, theta, buf)
        assert res["storage"] == "coded" and numpy.abs(res["props"] - numpy.exp(theta)).max() < 1e-12


def test_row_sharded_loop_over_coded_plans(b17):
    """dist.sharded_em_loop drives plan.em_iter / finalize: a coded plan is a drop-in there (one rank)."""
    from mixemt_amd import dist as mdist
    from mixemt_amd import em
    refseq, phy, haps, tables = b17
    g = golden("g4_run_em")
    mat = _b17_matrix(tables, g, len(haps))
    plan = em.EmPlan(mat, g["wts"], storage="coded")
    ln_cur, ln_new, states = mdist.sharded_em_loop(plan, g["inits"], 1e-4, 10000)
    assert [s[1] for s in states] == list(g["iters"])
    assert numpy.abs(numpy.exp(ln_new[0].cpu().numpy()) - g["props"]).max() < PROPS_ATOL


def test_auto_storage_codes_large_build_matrices_only(b17):
    """storage="auto": dictionary rows above the one-launch loops' range when the rows compress; dense otherwise."""
    import torch
    from mixemt_amd import em, preprocess, synth
    refseq, phy, haps, tables = b17
    g = golden("g4_run_em")
    small = _b17_matrix(tables, g, len(haps))
    assert em.EmPlan(small, g["wts"], storage="auto").storage == "f64"          # 600 rows: the one-launch loop's
    n_rows = 20000                                                              # 1.08e8 cells: above auto's threshold
    row_ptr, site, obs, _ = synth.synth_rows(tables, len(refseq), 0, n_rows, seed=3)
    big = preprocess.build_em_matrix_device(tables, row_ptr, site, obs)
    plan = em.EmPlan(big, torch.ones(n_rows, dtype=torch.float64, device=big.device), storage="auto")
    assert plan.storage == "coded" and plan.coded is not None and plan.coded_rest < 0.05 * n_rows
    noise = torch.empty((12500, 8192), dtype=torch.float64, device=big.device).normal_(-25.0, 8.0)   # 1.0e8 cells
    plan = em.EmPlan(noise, torch.ones(12500, dtype=torch.float64, device=big.device), storage="auto")
    assert plan.storage == "f64" and plan.coded is None and plan.lin is not None  # nothing to compress: dense


def test_special_values_and_weights():
    """NaN-free edge values: a row that is -inf everywhere with weight 0 is dropped like scipy drops it, with
    weight > 0 it poisons the run exactly like the dense path; fractional weights; the all-ones bit pattern
    (the hash table's empty marker) sends its row to the dense rest instead of being mis-coded."""
    import struct
    from mixemt_amd import em
    rng = numpy.random.default_rng(9)
    n_rows, n_haps = 64, 256
    few = rng.normal(-15.0, 4.0, size=(n_rows, 9))
    mat = numpy.take_along_axis(few, rng.integers(0, 9, size=(n_rows, n_haps)), axis=1)
    mat[5, :] = -numpy.inf
    wts = rng.random(n_rows) + 0.5
    wts[5] = 0.0
    init = rng.dirichlet([1.0] * n_haps)
    res = em.run_em_ex(mat, wts, em_args(max_iter=5, tolerance=0.0), inits=init[None, :], storage="coded")
    theta = numpy.log(init)
    buf = numpy.empty_like(mat)
    with numpy.errstate(invalid="ignore"):
        for _ in range(5):
            buf, theta = em_oracle.em_step(mat, wts, theta, buf)
    assert numpy.abs(res["props"] - numpy.exp(theta)).max() < 1e-12
    wts[5] = 1.0                                        # now the empty row counts: NaN everywhere, as in the reference
    bad = em.run_em_ex(mat, wts, em_args(max_iter=3, tolerance=0.0), inits=init[None, :], storage="coded")
    ref = em.run_em_ex(mat, wts, em_args(max_iter=3, tolerance=0.0), inits=init[None, :], storage="f64")
    assert numpy.isnan(bad["props"]).all() and numpy.isnan(ref["props"]).all()
    # the reserved bit pattern (a NaN payload of all ones)
    mat2 = numpy.take_along_axis(few, rng.integers(0, 9, size=(n_rows, n_haps)), axis=1)
    mat2[7, 3] = struct.unpack("<d", b"\xff" * 8)[0]
    plan = em.EmPlan(mat2, numpy.ones(n_rows), storage="coded")
    nd = plan.coded_ndist.cpu().numpy()
    assert nd[7] == 0 and plan.coded_rest == 1 and (numpy.delete(nd, 7) > 0).all()


def _decode_cm(cm):
    import torch
    from mixemt_amd import _lib
    from mixemt_amd._dev import current_stream
    lib = _lib.load()
    out = torch.full((cm.n_rows, cm.n_haps), float("nan"), dtype=torch.float64, device=cm.rec.device)
    coded = _lib.Coded(cm.rec.data_ptr(), cm.rec_off.data_ptr(), cm.ndist.data_ptr(), cm.n_rows, None, 0, None, 0, None, 0)
    _lib.check(lib.mxm_decode_rows(ctypes.byref(coded), cm.n_haps, out.data_ptr(), out.stride(0), current_stream()),
               "mxm_decode_rows")
    return out


@pytest.mark.parametrize("name,read_len", [("g9_run_em_2400", 0), ("synth", 150), ("synth", 260)])
def test_records_straight_from_the_build(b17, name, read_len):
    """mxm_build_em_records: the marker kernel's records decode to mxm_linearize's rows bit for bit, with and
    without the dense matrix; rows without a record (long rows, more than 256 values) arrive dense."""
    import torch
    from mixemt_amd import em, preprocess, synth
    refseq, phy, haps, tables = b17
    if name == "synth":
        row_ptr, site, obs, _ = synth.synth_reads(tables, len(refseq), 1500, seed=41, read_len=read_len)
    else:
        g = golden(name)
        row_ptr, site, obs = g["row_ptr"], g["site"], g["obs"]
    want = c_oracle.build_em_matrix(tables.expected, tables.lhit, tables.lmiss, row_ptr, site, obs, len(haps))
    cm, mat = preprocess.build_em_records_device(tables, row_ptr, site, obs, dense=True)
    assert numpy.array_equal(mat.cpu().numpy(), want)                          # the dense matrix, reference bits
    dense = em.EmPlan(mat, numpy.ones(len(want)))
    nd = cm.ndist.cpu().numpy()
    rest = cm.rest_rows.cpu().numpy()
    assert numpy.array_equal(rest, numpy.flatnonzero(nd == 0)) and (nd <= 1024).all()
    long_rows = numpy.flatnonzero(numpy.diff(row_ptr) > 64)                   # beyond the marker kernel's mask: built
    for r in rest:                                                             # densely, then coded from there
        assert len(numpy.unique(want[r])) > 1024                               # only rows with too many values stay dense
    n_obs = numpy.diff(row_ptr)
    for r in numpy.flatnonzero(nd > 256):        # wide records: from the encoder (rows beyond 128 observations) one entry per
        if n_obs[r] > 128:                       # distinct value; from the marker kernel (round 6) one per distinct mask
            assert nd[r] == len(numpy.unique(want[r]))
        else:
            assert nd[r] >= len(numpy.unique(want[r]))
    if read_len == 260:
        assert len(long_rows) > 100 and (nd[long_rows] > 0).sum() > 50         # long rows with records of their own
    for r in numpy.flatnonzero(nd > 0)[:40]:
        assert nd[r] >= len(numpy.unique(want[r]))        # one entry per distinct mask (equal sums may repeat)
    coded_rows = torch.from_numpy(nd > 0).to(mat.device)
    dec = _decode_cm(cm)
    assert torch.equal(dec[coded_rows].view(torch.int64), dense.lin[coded_rows][:, :len(haps)].view(torch.int64))
    assert torch.equal(cm.rowmax[coded_rows], dense.rowmax[coded_rows])
    assert numpy.array_equal(cm.m_rest.cpu().numpy(), want[rest])
    # the same without ever writing the dense matrix
    cm2 = preprocess.build_em_records_device(tables, row_ptr, site, obs)
    nd2 = cm2.ndist.cpu().numpy()
    assert numpy.array_equal(nd2 > 0, nd > 0) or set(numpy.flatnonzero(nd2 == 0)) >= set(rest)
    dec2 = _decode_cm(cm2)
    rows2 = torch.from_numpy(nd2 > 0).to(mat.device)
    assert torch.equal(dec2[rows2].view(torch.int64), dense.lin[rows2][:, :len(haps)].view(torch.int64))
    assert numpy.array_equal(cm2.m_rest.cpu().numpy(), want[cm2.rest_rows.cpu().numpy()])


def test_rows_from_alignments_keep_their_dense_rows(b17):
    """Rows made of two mates carry up to ~170 observations: past the marker kernel (built a slab at a time, coded from
    there) and, some of them, past 1024 distinct values -- those stay dense beside the records (`rest_rows` ascending,
    `m_rest` the reference's bits), found without a dense matrix anywhere."""
    import torch
    from mixemt_amd import alignments, preprocess, synth
    refseq, phy, haps, tables = b17
    cols = synth.synth_alignments(tables, refseq, 6000, seed=21, mate_share=0.9)
    enc = alignments.encode_alignments(cols, tables.sites, len(refseq), 30, 30)
    want = c_oracle.build_em_matrix(tables.expected, tables.lhit, tables.lmiss, enc.row_ptr, enc.site, enc.obs, len(haps))
    old = preprocess.REST_SLAB_ROWS
    preprocess.REST_SLAB_ROWS = 60                                             # several slabs
    try:
        cm = preprocess.build_em_records_device(tables, enc.row_ptr, enc.site, enc.obs)
    finally:
        preprocess.REST_SLAB_ROWS = old
    nd = cm.ndist.cpu().numpy()
    rest = cm.rest_rows.cpu().numpy()
    assert preprocess.build_em_matrix_device.last_fallback > 150               # (three slabs at least: the rows beyond 128 observations)
    assert len(rest) >= 5 and numpy.array_equal(rest, numpy.flatnonzero(nd == 0))
    assert all(len(numpy.unique(want[r])) > 1024 for r in rest)
    assert numpy.array_equal(cm.m_rest.cpu().numpy(), want[rest])
    dec = _decode_cm(cm)
    rows = torch.from_numpy(nd > 0).to(dec.device)
    lin = numpy.exp(want - want.max(axis=1, keepdims=True))
    got = dec[rows].cpu().numpy()
    assert numpy.allclose(got, lin[nd > 0], rtol=1e-15, atol=0)
    assert numpy.array_equal(cm.rowmax.cpu().numpy()[nd > 0], want.max(axis=1)[nd > 0])


def test_run_em_from_build_records_reproduces_the_reference(b17):
    """g9 (2400 rows, repeat weights, 1209 iterations): CSR -> records -> EM, no dense matrix anywhere."""
    from mixemt_amd import em, preprocess
    refseq, phy, haps, tables = b17
    g = golden("g9_run_em_2400")
    cm = preprocess.build_em_records_device(tables, g["row_ptr"], g["site"], g["obs"])
    numpy.random.seed(17)
    res = em.run_em_ex(None, g["wts"], em_args(), want_read_mix=False, records=cm)
    assert numpy.array_equal(res["inits"], g["inits"])
    assert res["iters"] == list(g["iters"]) and res["storage"] == "coded"
    assert numpy.abs(res["props"] - g["props"]).max() < PROPS_ATOL
    # the posterior from the records' log tables (mxm_em_step_coded) against the dense pass over the dense matrix
    numpy.random.seed(17)
    lean = em.run_em_ex(None, g["wts"], em_args(), records=cm)
    cm2, mat = preprocess.build_em_records_device(tables, g["row_ptr"], g["site"], g["obs"], dense=True)
    numpy.random.seed(17)
    full = em.run_em_ex(mat, g["wts"], em_args(), records=cm2)
    assert lean["iters"] == full["iters"] == list(g["iters"])
    a, b = lean["read_mix"].cpu().numpy(), full["read_mix"].cpu().numpy()
    assert numpy.array_equal(a.argmax(axis=1), g["mix_argmax"]) and numpy.array_equal(b.argmax(axis=1), g["mix_argmax"])
    assert numpy.abs(a - b).max() < 1e-10
    assert numpy.allclose(a[:4], g["mix_rows"], rtol=0, atol=1e-8) and numpy.allclose(a.max(axis=1), g["mix_rowmax"], rtol=0, atol=1e-8)
    # three restarts: the fold (logaddexp over runs, em.py:156) from records
    g5 = golden("g5_run_em_multi")
    cm5 = preprocess.build_em_records_device(tables, g5["row_ptr"], g5["site"], g5["obs"])
    numpy.random.seed(11)
    res5 = em.run_em_ex(None, g5["wts"], em_args(n_multi=3), records=cm5)
    assert res5["iters"] == list(g5["iters"]) and numpy.abs(res5["props"] - g5["props"]).max() < PROPS_ATOL
    mix5 = res5["read_mix"].cpu().numpy()
    assert numpy.array_equal(mix5.argmax(axis=1), g5["mix_argmax"])
    assert numpy.allclose(mix5[:16], g5["mix_rows"], rtol=0, atol=1e-8)


def test_consumers_from_records_match_the_dense_pipeline(b17):
    """Contributors from read votes, the vote table, the reduced matrix and the refinement EM from records
    alone == the same steps over the dense matrix and posterior (g9: 2400 rows, repeat weights)."""
    from mixemt_amd import assign, em, preprocess
    refseq, phy, haps, tables = b17
    g = golden("g9_run_em_2400")
    args = em_args(min_reads=10, min_fold=2.0)
    cm, mat = preprocess.build_em_records_device(tables, g["row_ptr"], g["site"], g["obs"], dense=True)
    cm_only = preprocess.build_em_records_device(tables, g["row_ptr"], g["site"], g["obs"])
    numpy.random.seed(17)
    full = em.run_em_ex(mat, g["wts"], args)                                     # dense path, posterior included
    numpy.random.seed(17)
    lean = em.run_em_ex(None, g["wts"], args, want_read_mix=False, records=cm_only)
    assert lean["iters"] == full["iters"] == list(g["iters"])
    best_d, votes_d = assign.row_argmax_votes(full["read_mix"], g["wts"])
    best_r, votes_r = assign.row_argmax_votes_records(cm_only, lean["ln_theta_k"][0], g["wts"])
    assert numpy.array_equal(best_r, best_d) and numpy.array_equal(best_r, g["mix_argmax"])
    assert numpy.array_equal(votes_r, votes_d) and numpy.array_equal(votes_r, g["votes"])
    con_d = assign.find_contribs_from_reads(full["read_mix"], g["wts"], args)
    con_r = assign.find_contribs_from_records(cm_only, lean["ln_theta_k"][0], g["wts"], args)
    assert con_r == con_d and sorted(con_r) == sorted(int(c) for c in g["contributors"])
    contribs = [["hap%d" % (i + 1), haps[c], 0.0] for i, c in enumerate(con_r)]
    sub_d, names_d = preprocess.reduce_em_matrix(mat, haps, contribs)
    sub_r, names_r = preprocess.reduce_em_records(cm_only, haps, contribs)
    assert names_r == names_d
    assert numpy.array_equal(sub_r.cpu().numpy(), sub_d.cpu().numpy())           # the log values, bit for bit


def test_votes_from_records_of_a_multi_run_follow_the_fold(b17):
    """
    ADVICE r2: with n_multi > 1 the reference votes on the logaddexp fold of the runs' posteriors (em.py:156 ->
    assemble.py:115-123); each run's row normaliser weighs its columns, so argmax_h(ln theta_0 + M) is NOT it.
    g5 (three restarts, reference run): calls and votes from records alone, all three log theta_k handed over,
    equal the reference's; fractional weights reproduce bit for bit; rows without a record are covered.
    """
    from mixemt_amd import _lib, assign, em, preprocess
    refseq, phy, haps, tables = b17
    g5 = golden("g5_run_em_multi")
    cm = preprocess.build_em_records_device(tables, g5["row_ptr"], g5["site"], g5["obs"])
    numpy.random.seed(11)
    res = em.run_em_ex(None, g5["wts"], em_args(n_multi=3), want_read_mix=False, records=cm)
    assert res["iters"] == list(g5["iters"]) and res["ln_theta_k"].shape == (3, len(haps))
    best, votes = assign.row_argmax_votes_records(cm, res["ln_theta_k"], g5["wts"])
    assert numpy.array_equal(best, g5["mix_argmax"])
    assert numpy.array_equal(votes, g5["votes"])
    # the same through the dense pipeline's posterior
    numpy.random.seed(11)
    cm2, mat = preprocess.build_em_records_device(tables, g5["row_ptr"], g5["site"], g5["obs"], dense=True)
    full = em.run_em_ex(mat, g5["wts"], em_args(n_multi=3))
    best_d, votes_d = assign.row_argmax_votes(full["read_mix"], g5["wts"])
    assert numpy.array_equal(best, best_d) and numpy.array_equal(votes, votes_d)
    frac = numpy.random.default_rng(5).random(len(best))
    v1 = assign.row_argmax_votes_records(cm, res["ln_theta_k"], frac)[1]
    v2 = assign.row_argmax_votes_records(cm, res["ln_theta_k"], frac)[1]
    assert numpy.array_equal(v1, v2) and abs(v1.sum() - frac.sum()) < 1e-9
    # every row dense-leftover (the marker kernel hands every row to the fallback list, whose rows are coded from
    # their dense form only up to 1024 values): very long reads give rows that stay dense
    from mixemt_amd import synth
    row_ptr, site, obs, _ = synth.synth_reads(tables, len(refseq), 300, seed=77, read_len=6000)
    cm3, mat3 = preprocess.build_em_records_device(tables, row_ptr, site, obs, dense=True)
    assert int(cm3.rest_rows.numel()) > 100 or int(cm3.wide_rows().numel()) > 100
    wts3 = numpy.ones(300)
    numpy.random.seed(3)
    r3 = em.run_em_ex(mat3, wts3, em_args(n_multi=2, max_iter=60))
    b_rec, v_rec = assign.row_argmax_votes_records(cm3, r3["ln_theta_k"], wts3)
    b_den, v_den = assign.row_argmax_votes(r3["read_mix"], wts3)
    assert numpy.array_equal(b_rec, b_den) and numpy.array_equal(v_rec, v_den)


def test_records_posterior_stays_finite_where_the_dense_pass_does(b17):
    """a row whose every supported haplogroup has an underflowed proportion has a zero linear row sum; the
    records posterior then redoes that row in log space and returns the dense pass's finite values."""
    import torch
    from mixemt_amd import em, preprocess
    refseq, phy, haps, tables = b17
    g = golden("g4_run_em")
    cm, mat = preprocess.build_em_records_device(tables, g["row_ptr"], g["site"], g["obs"], dense=True)
    lnp = numpy.full(len(haps), -900.0)                    # exp() underflows to 0 everywhere
    lnp[5] = -1200.0
    rec_plan = em.EmPlan(None, g["wts"], records=cm)
    dense_plan = em.EmPlan(mat, g["wts"])
    a = em.posterior(rec_plan, lnp).cpu().numpy()
    b = em.posterior(dense_plan, lnp).cpu().numpy()
    assert numpy.isfinite(b).all() and numpy.isfinite(a).all()
    assert numpy.abs(a - b).max() < 1e-9
    fold_a = em.posterior(rec_plan, lnp + 1.0, out=torch.from_numpy(a.copy()).cuda(), fold=True).cpu().numpy()
    fold_b = em.posterior(dense_plan, lnp + 1.0, out=torch.from_numpy(b.copy()).cuda(), fold=True).cpu().numpy()
    assert numpy.abs(fold_a - fold_b).max() < 1e-9


@pytest.fixture()
def lib():
    from mixemt_amd import _lib
    handle = _lib.load()
    yield handle
    handle.mxm_set_loop_fused(-1, 0)
    handle.mxm_diag_fused_force_abort(0)
    handle.mxm_set_fused_coded_grid(0)


@pytest.mark.parametrize("name,seed,n_multi", [("g4_run_em", 7, 1), ("g9_run_em_2400", 17, 1), ("g5_run_em_multi", 11, 3)])
def test_one_launch_loop_over_records_reproduces_the_reference_runs(b17, lib, name, seed, n_multi):
    """
    em_fused_coded_kernel (round 4): the whole loop over records in one persistent launch -- the reference's stopping
    iterations, proportions and calls (g4, g9 with 1209 iterations, g5 with three restarts), identical results when the
    launch is cut into chunks of 7 iterations, rounding-level agreement with the per-iteration kernels, and a launch that
    gives up is undone and finished by them.
    """
    from mixemt_amd import em
    refseq, phy, haps, tables = b17
    g = golden(name)
    mat = _b17_matrix(tables, g, len(haps))
    runs = {}
    for label, mode, chunk in (("one launch", 1, 0), ("chunks", 1, 7), ("kernels", 0, 0)):
        lib.mxm_set_loop_fused(mode, chunk)
        numpy.random.seed(seed)
        res = em.run_em_ex(mat, g["wts"], em_args(n_multi=n_multi), storage="coded")
        assert res["storage"] == "coded" and numpy.array_equal(res["inits"], g["inits"])
        assert res["iters"] == list(g["iters"]), label
        assert res["done"] == [1] * n_multi
        assert numpy.abs(res["props"] - g["props"]).max() < PROPS_ATOL
        mix = res["read_mix"].cpu().numpy()
        assert numpy.array_equal(mix.argmax(axis=1), g["mix_argmax"])
        runs[label] = res
    assert numpy.array_equal(runs["one launch"]["run_props"], runs["chunks"]["run_props"])   # a resumed launch continues bit for bit
    assert runs["one launch"]["l1"] == runs["chunks"]["l1"]
    assert numpy.abs(runs["one launch"]["run_props"] - runs["kernels"]["run_props"]).max() < 1e-12
    # starved grid: undone from the snapshot, the same call finishes through the per-iteration kernels
    lib.mxm_set_loop_fused(-1, 0)
    lib.mxm_diag_fused_force_abort(1)
    numpy.random.seed(seed)
    got = em.run_em_ex(mat, g["wts"], em_args(n_multi=n_multi), storage="coded", want_read_mix=False)
    assert got["iters"] == list(g["iters"]) and numpy.array_equal(got["run_props"], runs["kernels"]["run_props"])
    lib.mxm_set_loop_fused(1, 0)
    plan = em.EmPlan(mat, g["wts"], storage="coded")
    with pytest.raises(ValueError, match="one-launch loop"):
        em.em_loop(plan, g["inits"], 1e-4, 10000)
    lib.mxm_diag_fused_force_abort(0)
    ln_cur, ln_new, states = em.em_loop(plan, g["inits"], 1e-4, 10000)
    assert [st[1] for st in states] == list(g["iters"])


@pytest.mark.parametrize("n_rows,n_haps,seed", [(1, 66, 1), (3, 5408, 2), (700, 130, 3), (513, 8192, 4), (1400, 2050, 5),
                                                (300, 4096, 6), (40, 1026, 7)])
def test_one_launch_loop_over_records_vs_oracle_on_random_shapes(lib, n_rows, n_haps, seed):
    """Fewer rows than workgroups, every chunk count of the kernel, slices of 1 to 8 column pairs, byte-coded and wide
    rows mixed, zero weights, -inf cells, max_iter exhaustion: the oracle's run_em."""
    from mixemt_amd import em
    rng = numpy.random.default_rng(seed)
    nval = numpy.where(rng.random(n_rows) < 0.8, 30, 400)
    mat = numpy.empty((n_rows, n_haps))
    for r in range(n_rows):
        vals = rng.normal(-12.0, 4.0, size=nval[r])
        mat[r] = vals[rng.integers(0, nval[r], size=n_haps)]
    hot = rng.integers(0, 5, size=n_rows)
    mat[numpy.arange(n_rows), hot] = -1.0                          # a few haplogroups explain most rows
    if n_rows > 8:
        mat[7, : n_haps // 2] = -numpy.inf
    wts = rng.integers(0, 4, size=n_rows).astype(numpy.float64)
    wts[0] = 2.0
    for max_iter in (300, 6):
        args = em_args(max_iter=max_iter)
        trace = []
        numpy.random.seed(seed)
        props, mix = em_oracle.run_em(mat, wts, args, trace=trace)
        for mode, chunk in ((1, 0), (1, 4)):
            lib.mxm_set_loop_fused(mode, chunk)
            numpy.random.seed(seed)
            res = em.run_em_ex(mat, wts, args, storage="coded")
            assert res["storage"] == "coded" and res["iters"] == [trace[0]["iters"]], (mode, chunk)
            assert numpy.abs(res["props"] - props).max() < PROPS_ATOL
            got = res["read_mix"].cpu().numpy()
            assert numpy.array_equal(numpy.isfinite(got), numpy.isfinite(mix))
            assert numpy.abs(numpy.exp(got) - numpy.exp(mix)).max() < 1e-9


def test_one_launch_loop_over_records_poisons_like_the_reference(lib):
    """A row that is -inf everywhere: NaN proportions and max_iter, as the dense path and the reference (em.py:81-83)."""
    from mixemt_amd import em
    rng = numpy.random.default_rng(3)
    few = rng.normal(-15.0, 4.0, size=(64, 9))
    mat = numpy.take_along_axis(few, rng.integers(0, 9, size=(64, 256)), axis=1)
    mat[5, :] = -numpy.inf
    lib.mxm_set_loop_fused(1, 0)
    res = em.run_em_ex(mat, numpy.ones(64), em_args(max_iter=9), storage="coded", want_read_mix=False)
    assert res["iters"] == [9] and res["done"] == [2] and numpy.isnan(res["props"]).all()


@pytest.mark.parametrize("grid", [8, 40])
def test_one_launch_loop_over_records_with_many_rows_per_workgroup(b17, lib, grid):
    """
    The paths a 10^6-row matrix takes, at test size: with the grid held to a few workgroups (mxm_set_fused_coded_grid) a
    workgroup has more than 256 rows -- the metadata blocks are refetched inside the pass (the non-resident instance) --
    and more than 256 WIDE rows (several batches of the second loop).  g9's rows (2400, repeat weights) tiled four times
    with every fifth row replaced by a wide one: same stopping iteration and proportions as the per-iteration kernels and
    as the oracle's run on the same matrix.
    """
    from mixemt_amd import em
    refseq, phy, haps, tables = b17
    g = golden("g9_run_em_2400")
    base = _b17_matrix(tables, g, len(haps))
    rng = numpy.random.default_rng(grid)
    mat = numpy.tile(base, (4, 1))
    wts = numpy.tile(g["wts"], 4).astype(numpy.float64)
    wide = numpy.arange(0, len(mat), 5)
    vals = rng.normal(-20.0, 5.0, size=(len(wide), 700))
    pick = rng.integers(0, 700, size=(len(wide), len(haps)))
    mat[wide] = numpy.take_along_axis(vals, pick, axis=1) + base[wide % len(base)].max(axis=1, keepdims=True) - 40.0
    plan = em.EmPlan(mat, wts, storage="coded")
    assert plan.coded_wide >= len(wide) and plan.coded_rest == 0
    args = em_args(max_iter=12)                                      # (the oracle's step takes a second at this size)
    numpy.random.seed(5)
    init = em.init_props(len(haps), 1.0)[None, :]
    lib.mxm_set_loop_fused(0, 0)
    want = em.run_em_ex(mat, wts, args, inits=init, storage="coded", want_read_mix=False)
    lib.mxm_set_fused_coded_grid(grid)
    for chunk in (0, 9):
        lib.mxm_set_loop_fused(1, chunk)
        got = em.run_em_ex(mat, wts, args, inits=init, storage="coded", want_read_mix=False)
        assert got["iters"] == want["iters"] and got["done"] == want["done"]
        assert numpy.abs(got["run_props"] - want["run_props"]).max() < 1e-12
    lib.mxm_set_fused_coded_grid(0)
    theta = numpy.log(init[0])
    buf = numpy.empty_like(mat)
    for _ in range(want["iters"][0]):
        buf, theta = em_oracle.em_step(mat, wts, theta, buf)
    assert numpy.abs(want["run_props"][0] - numpy.exp(theta)).max() < 1e-11


def test_the_wide_rows_list_is_checked_where_it_is_used(b17):
    """the EM iteration skips rows with 16-bit codes in its main pass and takes them from mxm_coded.wide_rows,
    so a descriptor WITHOUT the list (or with an incomplete / unsorted / wrong one) must fail loudly instead of dropping
    2 % of the rows: mxm_em_iter_coded poisons its sums (NaN) and raises state.error on the device, mxm_em_loop_coded
    refuses on entry; the vote finds the wide rows from ndist itself and never needs the list."""
    import torch
    from mixemt_amd import _lib, assign, em, preprocess, synth
    from mixemt_amd._dev import current_stream
    refseq, phy, haps, tables = b17
    n_rows, n_haps = 3000, len(haps)
    row_ptr, site, obs, _ = synth.synth_reads(tables, len(refseq), n_rows, seed=77)
    cm = preprocess.build_em_records_device(tables, row_ptr, site, obs)
    wide = cm.wide_rows()
    assert wide.numel() >= 10 and cm.rest_rows.numel() == 0
    wts = torch.ones(n_rows, dtype=torch.float64, device="cuda")
    plan = em.EmPlan(None, wts, records=cm)
    lib = _lib.load()
    props = torch.from_numpy(numpy.random.default_rng(5).dirichlet([1.0] * n_haps)[None, :]).cuda()
    good = torch.zeros_like(props)
    plan.em_iter(props, torch.log(props), em.new_state(1, props.device), good)
    assert torch.isfinite(good).all() and abs(float((props * good).sum()) - n_rows) < 1e-9 * n_rows

    def descriptor(rows):
        return _lib.Coded(cm.rec.data_ptr(), cm.rec_off.data_ptr(), cm.ndist.data_ptr(), n_rows, None, 0, None, 0,
                          rows.data_ptr() if rows is not None and rows.numel() else None, 0 if rows is None else int(rows.numel()))

    unsorted = wide.flip(0).contiguous()
    not_wide = wide.clone()
    not_wide[3] = int(torch.nonzero(cm.ndist <= 256)[0])                       # a byte-coded row passed off as wide
    not_wide = not_wide.sort().values
    out_of_range = wide.clone()
    out_of_range[-1] = n_rows + 12345
    for label, rows in (("no list", None), ("one missing", wide[1:].contiguous()), ("unsorted", unsorted),
                        ("not wide", not_wide), ("out of range", out_of_range)):
        coded = descriptor(rows)
        state = em.new_state(1, props.device)
        colsum = torch.zeros_like(props)
        _lib.check(lib.mxm_em_iter_coded(ctypes.byref(coded), wts.data_ptr(), props.data_ptr(), n_haps, 1, state.data_ptr(),
                                         colsum.data_ptr(), plan.ws.data_ptr(), plan.ws_bytes, current_stream()), label)
        torch.cuda.synchronize()
        assert torch.isnan(colsum).all(), label                                 # never a plausible, wrong sum
        with pytest.raises(ValueError, match="wide_rows"):
            em.read_state(state)
        # without a state the sums are poisoned all the same
        colsum.zero_()
        _lib.check(lib.mxm_em_iter_coded(ctypes.byref(coded), wts.data_ptr(), props.data_ptr(), n_haps, 1, None,
                                         colsum.data_ptr(), plan.ws.data_ptr(), plan.ws_bytes, current_stream()), label)
        assert torch.isnan(colsum).all(), label
        # the blocking loop refuses on entry, the loop vectors untouched
        ln0 = torch.log(props).clone()
        ln_cur, ln_new, pc = ln0.clone(), ln0.clone(), props.clone()
        st = em.new_state(1, props.device)
        host_state = (_lib.EmState * 1)()
        rc = lib.mxm_em_loop_coded(ctypes.byref(coded), wts.data_ptr(), n_haps, 1, pc.data_ptr(), ln_cur.data_ptr(), ln_new.data_ptr(),
                                   colsum.data_ptr(), st.data_ptr(), 1e-4, 50, 8, plan.ws.data_ptr(), plan.ws_bytes, current_stream(),
                                   host_state)
        assert rc == -1 and b"wide_rows" in lib.mxm_last_error(), label
        assert torch.equal(ln_cur, ln0) and torch.equal(pc, props)
    # the right list again: same bits as before (the check itself changes nothing)
    again = torch.zeros_like(props)
    coded = descriptor(wide)
    _lib.check(lib.mxm_em_iter_coded(ctypes.byref(coded), wts.data_ptr(), props.data_ptr(), n_haps, 1, None, again.data_ptr(),
                                     plan.ws.data_ptr(), plan.ws_bytes, current_stream()), "good list")
    assert torch.equal(again, good)
    # the vote does not read the list at all: a descriptor without it gives every row its call
    ln_theta = torch.log(props)[0]
    best_ref = assign.row_argmax_votes_records(cm, ln_theta.cpu().numpy(), None)[0]
    assert (best_ref >= 0).all()
    best = torch.full((n_rows,), -7, dtype=torch.int32, device="cuda")
    coded = descriptor(None)
    _lib.check(lib.mxm_row_argmax_votes_coded(ctypes.byref(coded), n_haps, 1, ln_theta.data_ptr(), None, None, None, 0, None, 0,
                                              None, best.data_ptr(), None, None, 0, current_stream()), "votes without the list")
    assert numpy.array_equal(best.cpu().numpy(), best_ref)
    dense = cm.dense().cpu().numpy() + ln_theta.cpu().numpy()[None, :]
    assert numpy.array_equal(best_ref, dense.argmax(axis=1))
