"""
Pins the ORACLE (oracle/) to the reference: bit-for-bit against scipy's
logsumexp, and against golden vectors produced by importing the reference
(tools/gen_golden.py).  CPU only.
"""
import hashlib

import numpy
import pytest
import scipy.special

from conftest import em_args, golden
from oracle import build_oracle, c_oracle, em_oracle


def _sha(arr):
    return hashlib.sha256(numpy.ascontiguousarray(arr).tobytes()).hexdigest()


@pytest.mark.parametrize("shape,axis", [((7, 13), 1), ((7, 13), 0), ((64,), None), ((5, 200), 1),
                                        ((300, 9), 0)])
def test_logsumexp_bitwise_vs_scipy(shape, axis):
    rng = numpy.random.default_rng(5)
    a = rng.normal(-20, 15, size=shape)
    a.flat[::7] = -numpy.inf
    a.flat[3] = a.flat[4]                      # a tie at (maybe) the max
    mine = em_oracle.logsumexp(a, axis=axis)
    theirs = scipy.special.logsumexp(a, axis=axis)
    assert numpy.array_equal(mine, theirs, equal_nan=True)
    if a.ndim == 2 and axis == 0:
        b = rng.integers(0, 4, size=(shape[0], 1))
        mine = em_oracle.logsumexp(a, axis=0, b=b)
        theirs = scipy.special.logsumexp(a, axis=0, b=b)
        assert numpy.array_equal(mine, theirs, equal_nan=True)
    assert not numpy.shares_memory(mine, a)


def test_logsumexp_all_minus_inf_and_ties():
    a = numpy.full((3, 4), -numpy.inf)
    assert numpy.array_equal(em_oracle.logsumexp(a, axis=1), scipy.special.logsumexp(a, axis=1))
    t = numpy.zeros(3)
    assert em_oracle.logsumexp(t) == scipy.special.logsumexp(t) == numpy.log(3.0)


def test_em_step_exact_cases_of_reference_tests():
    """em_test.py:35-65: identity-in-log matrix with -inf, unit and [2,1,1] weights."""
    g = golden("g3_em_step")
    ident, lnp = g["ident"], g["ident_lnp"]
    for wts, mix_key, new_key, want in ((numpy.array([1, 1, 1]), "ident_mix1", "ident_new1",
                                         numpy.log(numpy.array([1.0, 1.0, 1.0]) / 3.0)),
                                        (numpy.array([2, 1, 1]), "ident_mix2", "ident_new2",
                                         numpy.log(numpy.array([2.0, 1.0, 1.0]) / 4.0))):
        buf = numpy.empty_like(ident)
        mix, new = em_oracle.em_step(ident, wts, lnp, buf)
        assert mix is buf
        assert numpy.array_equal(mix, ident)
        assert numpy.array_equal(new, want)
        assert numpy.array_equal(mix, g[mix_key]) and numpy.array_equal(new, g[new_key])


def test_converged_truth_table():
    """em_test.py:22-33."""
    prev = numpy.log(numpy.ones(10))
    cur = numpy.log(numpy.full(10, 2.0))
    assert em_oracle.converged(cur, cur) and em_oracle.converged(prev, prev)
    assert not em_oracle.converged(prev, cur) and not em_oracle.converged(cur, prev)
    close = cur.copy()
    close[3] = numpy.log(2.0001)
    assert em_oracle.converged(cur, prev, 20.0)
    assert not em_oracle.converged(cur, close)
    assert em_oracle.converged(cur, close, 0.001)


def test_build_matrix_toy_bitwise(toy):
    ref, phy, haps = toy
    g = golden("g1_toy")
    for key, mkey in (("reads", "mat"), ("reads_b", "mat_b")):
        reads = str(g[key]).split("\n")
        assert numpy.array_equal(build_oracle.build_em_matrix(ref, phy, reads, haps), g[mkey])
        assert numpy.array_equal(build_oracle.build_em_matrix_np(ref, phy, reads, haps), g[mkey])


def test_build_matrix_reference_test_values(toy):
    """preprocess_test.py:268-284: hand-computed products."""
    ref, phy, haps = toy
    reads = ["1:A,2:C", "1:T,2:C", "3:T,4:T", "2:A,4:T"]
    r1 = [(0.01 / 3) * (0.01 / 3)] + [0.99 * (0.01 / 3)] * 8
    r2 = [0.99 * (0.01 / 3)] + [(0.01 / 3) * (0.01 / 3)] * 8
    r3 = ([0.98 * (0.02 / 3)] + [(0.02 / 3) * 0.98] + [(0.02 / 3) * (0.02 / 3)] + [(0.02 / 3) * 0.98]
          + [0.98 * 0.98] + [(0.02 / 3) * 0.98] * 3 + [(0.02 / 3) * (0.02 / 3)])
    r4 = ([0.99 * (0.02 / 3)] + [(0.01 / 3) * 0.98] + [(0.01 / 3) * (0.02 / 3)]
          + [(0.01 / 3) * 0.98] * 5 + [0.99 * (0.02 / 3)])
    want = numpy.log(numpy.array([r1, r2, r3, r4]))
    assert numpy.allclose(build_oracle.build_em_matrix(ref, phy, reads, haps), want)


def test_build_matrix_b17_bitwise(b17):
    refseq, phy, haps, tables = b17
    g = golden("g2_build_b17")
    from mixemt_amd import synth
    sigs = synth.signatures(tables, g["row_ptr"][:33], g["site"], g["obs"])
    flat = build_oracle.flat_tables(refseq, phy, haps)
    mat = build_oracle.build_em_matrix_np(refseq, phy, sigs, haps, tables=flat)
    assert numpy.array_equal(mat, g["mat32"])
    # per-cell loop form on two rows (slow path, same bits)
    assert numpy.array_equal(build_oracle.build_em_matrix(refseq, phy, sigs[:2], haps), g["mat32"][:2])
    # C form on all 1032 rows: digest of the whole matrix
    full = c_oracle.build_em_matrix(flat[1], flat[2], flat[3], g["row_ptr"], g["site"], g["obs"],
                                    len(haps))
    assert _sha(full) == str(g["mat_sha256"])
    assert numpy.array_equal(full.argmax(axis=1), g["row_argmax"])


def test_em_step_b17(b17):
    refseq, phy, haps, tables = b17
    g = golden("g3_em_step")
    flat = build_oracle.flat_tables(refseq, phy, haps)
    mat = c_oracle.build_em_matrix(flat[1], flat[2], flat[3], g["row_ptr"], g["site"], g["obs"],
                                   len(haps))
    assert _sha(mat) == str(g["mat_sha256"])
    mix, new = em_oracle.em_step(mat, g["wts"], g["lnp"], numpy.empty_like(mat))
    assert _sha(mix) == str(g["mix_sha256"])
    assert numpy.array_equal(new, g["new_props"])
    cmix, cnew = c_oracle.em_step(mat, g["wts"], g["lnp"])
    assert numpy.allclose(cmix, mix, rtol=0, atol=1e-12)
    assert numpy.allclose(cnew, new, rtol=0, atol=1e-12)


@pytest.mark.parametrize("n_multi", [1, 3])
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_run_em_toy_bitwise(n_multi, seed):
    g = golden("g1_toy")
    key = "m%d_s%d" % (n_multi, seed)
    trace = []
    numpy.random.seed(seed)
    props, mix = em_oracle.run_em(g["mat"], numpy.ones(10), em_args(n_multi=n_multi, max_iter=1000),
                                  trace=trace)
    assert numpy.array_equal(props, g[key + "_props"])
    assert numpy.array_equal(mix, g[key + "_mix"])
    assert [t["iters"] for t in trace] == list(g[key + "_iters"])
    assert numpy.array_equal(numpy.stack([t["init"] for t in trace]), g[key + "_inits"])


def test_run_em_toy_reference_tolerances():
    """em_test.py:104-116: props ~ [0,.8,0,0,.2,...] atol .02, posteriors atol .05."""
    g = golden("g1_toy")
    true_props = numpy.array([0.0, 0.8, 0.0, 0.0, 0.2, 0.0, 0.0, 0.0, 0.0])
    true_haps = numpy.full((10, 9), -numpy.inf)
    true_haps[0:8, 1] = 0.0
    true_haps[8:10, 4] = 0.0
    for n_multi in (1, 10):
        props, mix = em_oracle.run_em(g["mat"], numpy.ones(10), em_args(n_multi=n_multi, max_iter=1000))
        assert numpy.allclose(props, true_props, atol=0.02)
        assert numpy.allclose(numpy.exp(mix), numpy.exp(true_haps), atol=0.05)


def test_run_em_config1_bitwise():
    g = golden("g7_config1")
    trace = []
    numpy.random.seed(7)
    props, mix = em_oracle.run_em(g["mat"], numpy.ones(1000, dtype=numpy.int64), em_args(), trace=trace)
    assert trace[0]["iters"] == int(g["iters"][0])
    assert numpy.array_equal(props, g["props"])
    assert numpy.array_equal(mix, g["mix"])


def test_run_em_refinement_shape_bitwise(b17):
    refseq, phy, haps, tables = b17
    g = golden("g6_refine")
    flat = build_oracle.flat_tables(refseq, phy, haps)
    mat = c_oracle.build_em_matrix(flat[1], flat[2], flat[3], g["row_ptr"], g["site"], g["obs"],
                                   len(haps))
    assert _sha(mat) == str(g["mat_sha256"])
    sub = mat[:, g["cols"]]
    trace = []
    numpy.random.seed(5)
    props, mix = em_oracle.run_em(sub, g["wts"], em_args(), trace=trace)
    assert trace[0]["iters"] == int(g["iters"][0])
    assert numpy.array_equal(props, g["props"]) and numpy.array_equal(mix, g["mix"])


def test_run_em_b17_stopped_by_max_iter_bitwise(b17):
    """
    g15: the reference's run_em on the g4 inputs (600 x 5408) stopped by max_iter = 5 and 25 -- its loop's "never
    converged" exit (em.py:140-142).  The oracle's loop reproduces proportions and posterior bit for bit: a Build-17-size
    pin of the LOOP in the default suite (the runs to convergence, g4 / g5 / g9, are `-m slow`).
    """
    refseq, phy, haps, tables = b17
    g = golden("g15_run_em_max_iter")
    flat = build_oracle.flat_tables(refseq, phy, haps)
    mat = c_oracle.build_em_matrix(flat[1], flat[2], flat[3], g["row_ptr"], g["site"], g["obs"], len(haps))
    for k in (5, 25):
        trace = []
        numpy.random.seed(7)
        props, mix = em_oracle.run_em(mat, g["wts"], em_args(max_iter=k), trace=trace)
        assert [t["iters"] for t in trace] == [k]
        assert numpy.array_equal(props, g["props_%d" % k])
        assert numpy.array_equal(mix[:4], g["mix_rows_%d" % k])
        assert numpy.array_equal(mix.max(axis=1), g["mix_rowmax_%d" % k])
        assert numpy.array_equal(mix.argmax(axis=1), g["mix_argmax_%d" % k])


@pytest.mark.slow
@pytest.mark.parametrize("name,seed,n_multi", [("g4_run_em", 7, 1), ("g5_run_em_multi", 11, 3)])
def test_run_em_b17_bitwise_slow(b17, name, seed, n_multi):
    refseq, phy, haps, tables = b17
    g = golden(name)
    flat = build_oracle.flat_tables(refseq, phy, haps)
    mat = c_oracle.build_em_matrix(flat[1], flat[2], flat[3], g["row_ptr"], g["site"], g["obs"],
                                   len(haps))
    trace = []
    numpy.random.seed(seed)
    props, mix = em_oracle.run_em(mat, g["wts"], em_args(n_multi=n_multi), trace=trace)
    assert [t["iters"] for t in trace] == list(g["iters"])
    assert numpy.array_equal(props, g["props"])
    assert numpy.array_equal(mix.argmax(axis=1), g["mix_argmax"])


@pytest.mark.slow
def test_run_em_b17_2400_rows_bitwise_slow(b17):
    """g9 (2400 x 5408, repeat weights up to 400): the oracle reproduces the reference bit for bit."""
    refseq, phy, haps, tables = b17
    g = golden("g9_run_em_2400")
    flat = build_oracle.flat_tables(refseq, phy, haps)
    mat = c_oracle.build_em_matrix(flat[1], flat[2], flat[3], g["row_ptr"], g["site"], g["obs"],
                                   len(haps))
    assert _sha(mat) == str(g["mat_sha256"])
    trace = []
    numpy.random.seed(17)
    props, mix = em_oracle.run_em(mat, g["wts"], em_args(), trace=trace)
    assert [t["iters"] for t in trace] == list(g["iters"])
    assert numpy.array_equal(props, g["props"])
    assert numpy.array_equal(mix.argmax(axis=1), g["mix_argmax"])


def test_g9_matrix_bits_from_the_c_oracle(b17):
    """The build half of g9 is cheap enough for every CPU run: the C oracle's matrix has the
    reference's sha256 (2400 x 5408)."""
    refseq, phy, haps, tables = b17
    g = golden("g9_run_em_2400")
    flat = build_oracle.flat_tables(refseq, phy, haps)
    mat = c_oracle.build_em_matrix(flat[1], flat[2], flat[3], g["row_ptr"], g["site"], g["obs"],
                                   len(haps))
    assert _sha(mat) == str(g["mat_sha256"])
    assert numpy.array_equal(mat.sum(axis=1), g["mat_row_sum"])


def test_g10_oracle_build_equals_the_reference_matrix_at_20000_rows(b17):
    """g10 (reference run at 20 000 x 5408): the C restatement of build_em_matrix gives the reference's bits on every
    one of its 1.08e8 cells (sha256), from the generator's seed (its CSR digest is pinned too)."""
    import hashlib
    from mixemt_amd import synth
    from oracle import c_oracle
    refseq, phy, haps, tables = b17
    g = golden("g10_run_em_20k")
    row_ptr, site, obs, _ = synth.synth_reads(tables, len(refseq), int(g["n_rows"]), seed=int(g["synth_seed"]))
    sha = lambda a: hashlib.sha256(numpy.ascontiguousarray(a).tobytes()).hexdigest()
    assert sha(row_ptr) + sha(site) + sha(obs) == str(g["csr_sha256"])
    mat = c_oracle.build_em_matrix(tables.expected, tables.lhit, tables.lmiss, row_ptr, site, obs, len(haps))
    assert sha(mat) == str(g["mat_sha256"])
    assert numpy.array_equal(mat.sum(axis=1), g["mat_row_sum"])
