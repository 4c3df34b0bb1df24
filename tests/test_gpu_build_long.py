"""
Rows of 65 .. 128 observations in the marker build (csrc/build_sparse_kernels.hpp, W = 2: 128-bit flip masks; round 6).
The reference merges mates into ONE fragment (preprocess.py:118-138), so two thirds of ordinary 2 x 150 paired-end
fragments -- and 38 % of 250-bp reads -- observe more than 64 variant sites; they used to leave the marker kernel for the
cell-by-cell one.  prob_for_vars (preprocess.py:86-96) adds a row's terms in signature order from 0.0: the bits must
be the reference's whatever kernel forms them.
  * synth-pe-v1 and 250-bp rows, dense and as records: bit-exact against the C oracle, <= 5 % of the rows left for the
    fallback kernel; with mxm_set_sparse_long_rows(0) (round 5's routing) the same bits and two thirds left over;
  * hand-made edge rows: exactly 64, 65, 127, 128 and 129 observations, flips only beyond observation 64, a row whose
    haplogroups all flip;
  * narrow and odd widths; the row order indirection; records that decode to the dense rows.
"""
import ctypes

import numpy
import pytest

pytestmark = pytest.mark.gpu


def _oracle(tables, row_ptr, site, obs):
    from oracle import c_oracle
    return c_oracle.build_em_matrix(tables.expected, tables.lhit, tables.lmiss, row_ptr, site, obs, tables.n_haps)


def _same_bits(a, b):
    return numpy.array_equal(numpy.ascontiguousarray(a).view(numpy.int64), numpy.ascontiguousarray(b).view(numpy.int64))


@pytest.mark.parametrize("gen,n_rows,seed", [("pairs", 3000, 3), ("pairs", 257, 4), ("reads250", 2500, 5), ("reads400", 1200, 6)])
def test_long_rows_keep_the_reference_bits(b17, gen, n_rows, seed):
    from mixemt_amd import _lib, preprocess, synth
    refseq, phy, haps, tables = b17
    if gen == "pairs":
        row_ptr, site, obs, _ = synth.synth_pairs(tables, len(refseq), n_rows, seed=seed)
    else:
        row_ptr, site, obs, _ = synth.synth_reads(tables, len(refseq), n_rows, seed=seed, read_len=int(gen[5:]))
    n = numpy.diff(row_ptr)
    assert (n > 64).mean() > 0.3
    want = _oracle(tables, row_ptr, site, obs)
    lib = _lib.load()
    got = preprocess.build_em_matrix_device(tables, row_ptr, site, obs, kernel="sparse").cpu().numpy()
    left = preprocess.build_em_matrix_device.last_fallback
    assert _same_bits(got, want)
    assert left >= (n > 128).sum()
    if gen != "reads400":
        assert left <= 0.05 * n_rows + 4, (left, n_rows)            # the marker kernel keeps what the verdict asked it to keep (+ small-sample slack)
    lib.mxm_set_sparse_long_rows(0)
    try:
        old = preprocess.build_em_matrix_device(tables, row_ptr, site, obs, kernel="sparse").cpu().numpy()
        assert preprocess.build_em_matrix_device.last_fallback >= (n > 64).sum()
    finally:
        lib.mxm_reset_tuning()
    assert _same_bits(old, want)


def test_edge_lengths_and_flips_beyond_the_first_word(b17):
    from mixemt_amd import preprocess
    refseq, phy, haps, tables = b17
    rng = numpy.random.default_rng(12)
    n_sites = len(tables.sites)
    rows = []
    for n in (1, 63, 64, 65, 66, 100, 127, 128, 129, 150, 64, 128):
        first = int(rng.integers(0, n_sites - n))
        st = numpy.arange(first, first + n)
        col = int(rng.integers(0, len(haps)))
        ob = tables.expected[st, col].copy()
        rows.append((st, ob))
    # a row whose first 64 observations agree with every haplogroup's majority and whose flips sit beyond: only the
    # second mask word is set
    st = numpy.arange(500, 500 + 120)
    # (the base most haplogroups expect at each site: the column-wise mode of the expected-base table)
    ob = numpy.array([numpy.bincount(tables.expected[s_]).argmax() for s_ in st], dtype=numpy.uint8)
    alt = {ord("A"): ord("C"), ord("C"): ord("G"), ord("G"): ord("T"), ord("T"): ord("A")}
    for j in range(70, 120, 3):
        ob[j] = alt.get(int(ob[j]), ord("A"))
    rows.append((st, ob))
    # every observation a base nobody expects: all terms miss for all haplogroups
    st = numpy.arange(2000, 2000 + 90)
    rows.append((st, numpy.full(90, ord("N"), dtype=numpy.uint8)))
    row_ptr = numpy.zeros(len(rows) + 1, dtype=numpy.int64)
    numpy.cumsum([len(r[0]) for r in rows], out=row_ptr[1:])
    site = numpy.concatenate([r[0] for r in rows]).astype(numpy.uint16)
    obs = numpy.concatenate([r[1] for r in rows]).astype(numpy.uint8)
    want = _oracle(tables, row_ptr, site, obs)
    got = preprocess.build_em_matrix_device(tables, row_ptr, site, obs, kernel="sparse").cpu().numpy()
    assert _same_bits(got, want)
    assert preprocess.build_em_matrix_device.last_fallback >= 2      # the rows of 129 and 150 observations


@pytest.mark.parametrize("n_cols", [1, 3, 255, 1025, 2050, 5407])
def test_long_rows_on_narrow_and_odd_tables(b17, n_cols):
    from mixemt_amd import preprocess, synth
    refseq, phy, haps, tables = b17
    sub = haps[300:300 + n_cols] if n_cols < 5000 else haps[:n_cols]
    sub_tables = preprocess.HapVarTables.build(refseq, phy, sub)
    if sub_tables.lut() is None:
        pytest.skip("tables do not qualify")
    contrib = (0, n_cols // 2, n_cols - 1)
    row_ptr, site, obs, _ = synth.synth_pairs(sub_tables, len(refseq), 700, seed=n_cols, contrib=contrib)
    assert (numpy.diff(row_ptr) > 64).sum() > 10
    want = _oracle(sub_tables, row_ptr, site, obs)
    got = preprocess.build_em_matrix_device(sub_tables, row_ptr, site, obs, kernel="sparse").cpu().numpy()
    assert _same_bits(got, want)


def test_records_of_long_rows_decode_to_the_dense_rows(b17):
    """mxm_build_em_records over paired-end fragments: the long rows get their records from the marker kernel too (no
    dense-slab detour for two thirds of the rows), and every record decodes to exp(M - rowmax) of the reference's row."""
    import torch
    from mixemt_amd import _lib, preprocess, synth
    from mixemt_amd._dev import current_stream
    refseq, phy, haps, tables = b17
    n_rows = 2500
    row_ptr, site, obs, _ = synth.synth_pairs(tables, len(refseq), n_rows, seed=21)
    n = numpy.diff(row_ptr)
    cm = preprocess.build_em_records_device(tables, row_ptr, site, obs)
    left = preprocess.build_em_matrix_device.last_fallback
    assert left <= 0.06 * n_rows and left >= (n > 128).sum()
    nd = cm.ndist_host()
    assert (nd > 0).all()
    lib = _lib.load()
    out = torch.full((n_rows, cm.n_haps), float("nan"), dtype=torch.float64, device=cm.rec.device)
    coded = cm.struct()
    _lib.check(lib.mxm_decode_rows(ctypes.byref(coded), cm.n_haps, out.data_ptr(), out.stride(0), current_stream()), "mxm_decode_rows")
    want = _oracle(tables, row_ptr, site, obs)
    p_want = numpy.exp(want - want.max(axis=1, keepdims=True))
    got = out.cpu().numpy()
    # P = exp(M - rowmax) as mxm_linearize forms it: compare through the log matrix the records also hold
    dense = cm.dense().cpu().numpy() if hasattr(cm, "dense") else None
    if dense is not None:
        assert _same_bits(dense, want)
    assert numpy.allclose(got, p_want, rtol=1e-14, atol=0)
    assert numpy.array_equal(cm.rowmax.cpu().numpy(), want.max(axis=1))


def test_paired_end_fragments_to_convergence_records_against_the_dense_loop(b17):
    """>= 10^5 paired-end fragments through the default route -- records from the marker kernels (long rows
    and wide records included), the quad dictionary, the per-iteration loop -- run to CONVERGENCE against the dense matrix's
    per-iteration loop: the same stopping iteration, proportions within 1e-12, identical calls and votes; the row lists
    (quad / byte / wide / dense rest) add up to all rows with no row listed twice."""
    import torch
    from mixemt_amd import _lib, assign, em, preprocess, synth
    from conftest import em_args
    refseq, phy, haps, tables = b17
    n_rows = 120000
    row_ptr, site, obs, who = synth.synth_pairs(tables, len(refseq), n_rows, seed=1)
    lens = numpy.diff(row_ptr)
    wts = torch.ones(n_rows, dtype=torch.float64, device="cuda")
    lib = _lib.load()
    cm = preprocess.build_em_records_device(tables, row_ptr, site, obs)
    left = preprocess.build_em_matrix_device.last_fallback
    assert (lens > 128).sum() <= left <= 0.05 * n_rows
    rest = cm.rest_rows.cpu().numpy()
    nd = cm.ndist_host()
    assert len(numpy.unique(rest)) == len(rest) and numpy.array_equal(numpy.flatnonzero(nd == 0), rest)
    assert (nd > 256).sum() > 0.01 * n_rows                       # wide records straight from the marker kernel
    plan = em.EmPlan(None, wts, records=cm)
    assert plan.attach_quads(True)
    c = plan.coded
    assert c.n_quad_rows + c.n_byte_rows + c.n_wide + c.R_rest == n_rows and c.n_quad_rows > 0.5 * n_rows
    del plan
    lib.mxm_set_loop_fused(0, 0)
    try:
        numpy.random.seed(7)
        rec = em.run_em_ex(None, wts, em_args(), want_read_mix=False, records=cm)
        best_r, votes_r = assign.row_argmax_votes_records(cm, rec["ln_theta_k"], wts)
        mat = preprocess.build_em_matrix_device(tables, row_ptr, site, obs)
        numpy.random.seed(7)
        dense = em.run_em_ex(mat, wts, em_args(), storage="f64")
    finally:
        lib.mxm_reset_tuning()
    best_d, votes_d = assign.row_argmax_votes(dense["read_mix"], wts)
    assert rec["iters"] == dense["iters"] and rec["iters"][0] > 50 and rec["storage"] == "coded"
    assert numpy.abs(rec["props"] - dense["props"]).max() < 1e-12
    assert numpy.array_equal(best_r, best_d) and numpy.array_equal(votes_r, votes_d)
    assert (best_r == numpy.array([10, 2000, 4000])[who]).mean() > 0.6
    p = rec["props"]
    assert numpy.allclose(p[[10, 2000, 4000]], [0.6, 0.3, 0.1], atol=0.02)
