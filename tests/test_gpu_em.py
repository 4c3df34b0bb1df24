"""
GPU parity of the EM core (em_step / converged / run_em through the drop-in
wrappers over the C ABI) against the oracle and the reference-derived goldens.

Tolerances (north_star: identical haplogroup calls, proportions within 1e-6):
the tests hold the HIP path to 1e-9 on proportions (linear-space loop vs the
reference's log-space loop differ by rounding only), identical iteration
counts, identical row argmax.
"""
import numpy
import pytest

from conftest import em_args, golden
from oracle import c_oracle, em_oracle

pytestmark = pytest.mark.gpu

PROPS_ATOL = 1e-9          # << the 1e-6 parity bar
LOGMIX_ATOL = 1e-9


def _b17_matrix(tables, g, n_haps):
    return c_oracle.build_em_matrix(tables.expected, tables.lhit, tables.lmiss, g["row_ptr"],
                                    g["site"], g["obs"], n_haps)


def test_converged_truth_table():
    """em_test.py:22-33."""
    from mixemt_amd import em
    prev = numpy.log(numpy.ones(10))
    cur = numpy.log(numpy.full(10, 2.0))
    assert em.converged(cur, cur) and em.converged(prev, prev)
    assert not em.converged(prev, cur) and not em.converged(cur, prev)
    close = cur.copy()
    close[3] = numpy.log(2.0001)
    assert em.converged(cur, prev, 20.0)
    assert not em.converged(cur, close)
    assert em.converged(cur, close, 0.001)


def test_init_props_uses_numpy_global_stream():
    """em_test.py:17-20 + the seeding contract of bin/mixemt:507-508."""
    from mixemt_amd import em
    numpy.random.seed(3)
    mine = em.init_props(10)
    numpy.random.seed(3)
    assert numpy.array_equal(mine, numpy.random.dirichlet([1.0] * 10))
    assert abs(mine.sum() - 1.0) < 1e-9
    assert numpy.array_equal(em.init_props(4, float("inf")), numpy.full(4, 0.25))


@pytest.mark.parametrize("wts,want", [([1, 1, 1], [1.0, 1.0, 1.0]), ([2, 1, 1], [2.0, 1.0, 1.0])])
def test_em_step_identity_with_minus_inf(wts, want):
    """em_test.py:35-65: output matrix == input, props == log(w / sum w)."""
    from mixemt_amd import em
    inf = float("inf")
    in_mat = numpy.array([[0.0, -inf, -inf], [-inf, 0.0, -inf], [-inf, -inf, 0.0]])
    props = numpy.log(numpy.array([0.6, 0.2, 0.2]))
    mix = numpy.empty_like(in_mat)
    res_mat, res_props = em.em_step(in_mat, numpy.array(wts), props, mix)
    assert res_mat is mix
    assert numpy.all(in_mat == mix)
    want = numpy.log(numpy.array(want) / sum(want))
    assert numpy.allclose(res_props, want, rtol=0, atol=1e-15)


def test_em_step_b17_golden(b17):
    from mixemt_amd import em
    refseq, phy, haps, tables = b17
    g = golden("g3_em_step")
    mat = _b17_matrix(tables, g, len(haps))
    mix = numpy.empty_like(mat)
    res_mat, new_props = em.em_step(mat, g["wts"], g["lnp"], mix)
    assert numpy.allclose(mix[:8], g["mix_rows"], rtol=0, atol=LOGMIX_ATOL)
    assert numpy.array_equal(mix.argmax(axis=1), g["mix_argmax"])
    assert numpy.allclose(mix.max(axis=1), g["mix_rowmax"], rtol=0, atol=LOGMIX_ATOL)
    assert numpy.allclose(new_props, g["new_props"], rtol=0, atol=1e-10)
    assert numpy.allclose(numpy.exp(new_props), numpy.exp(g["new_props"]), rtol=0, atol=1e-13)
    # rows of the posterior are normalised
    assert numpy.allclose(numpy.exp(mix).sum(axis=1), 1.0, atol=1e-12)


@pytest.mark.parametrize("n_multi", [1, 3])
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_run_em_toy_golden(n_multi, seed):
    """9 haplogroups: the generic log-space kernel path (H below the streaming kernel's range)."""
    from mixemt_amd import em
    g = golden("g1_toy")
    key = "m%d_s%d" % (n_multi, seed)
    numpy.random.seed(seed)
    res = em.run_em_ex(g["mat"], numpy.ones(10), em_args(n_multi=n_multi, max_iter=1000))
    assert numpy.array_equal(res["inits"], g[key + "_inits"])
    assert res["iters"] == list(g[key + "_iters"])
    assert numpy.allclose(res["props"], g[key + "_props"], rtol=0, atol=PROPS_ATOL)
    mix = res["read_mix"].cpu().numpy()
    assert numpy.allclose(numpy.exp(mix), numpy.exp(g[key + "_mix"]), rtol=0, atol=1e-9)
    assert numpy.array_equal(mix.argmax(axis=1), g[key + "_mix"].argmax(axis=1))


def test_run_em_reference_test_tolerances():
    """em_test.py:104-116 verbatim expectations (n_multi 1 and 10)."""
    from mixemt_amd import em
    g = golden("g1_toy")
    true_props = numpy.array([0.0, 0.8, 0.0, 0.0, 0.2, 0.0, 0.0, 0.0, 0.0])
    true_haps = numpy.full((10, 9), -numpy.inf)
    true_haps[0:8, 1] = 0.0
    true_haps[8:10, 4] = 0.0
    for n_multi in (1, 10):
        props, read_mix = em.run_em(g["mat"], numpy.ones(10), em_args(n_multi=n_multi, max_iter=1000))
        assert isinstance(read_mix, numpy.ndarray) and read_mix.shape == (10, 9)
        assert numpy.allclose(props, true_props, atol=0.02)
        assert numpy.allclose(numpy.exp(read_mix), numpy.exp(true_haps), atol=0.05)


def test_run_em_config1_golden():
    """BASELINE config 1: 1000 x 100 (the streaming linear-space kernel, NCH = 1)."""
    from mixemt_amd import em
    g = golden("g7_config1")
    numpy.random.seed(7)
    res = em.run_em_ex(g["mat"], numpy.ones(1000, dtype=numpy.int64), em_args())
    assert res["iters"] == [int(g["iters"][0])]
    assert numpy.allclose(res["props"], g["props"], rtol=0, atol=PROPS_ATOL)
    mix = res["read_mix"].cpu().numpy()
    assert numpy.array_equal(mix.argmax(axis=1), g["mix"].argmax(axis=1))
    assert numpy.allclose(numpy.exp(mix), numpy.exp(g["mix"]), rtol=0, atol=1e-9)


def test_run_em_b17_golden(b17):
    """600 x 5408, default flags: iteration count, proportions, calls, votes."""
    from mixemt_amd import em
    refseq, phy, haps, tables = b17
    g = golden("g4_run_em")
    mat = _b17_matrix(tables, g, len(haps))
    numpy.random.seed(7)
    res = em.run_em_ex(mat, g["wts"], em_args())
    assert numpy.array_equal(res["inits"], g["inits"])
    assert res["iters"] == list(g["iters"])
    assert numpy.abs(res["props"] - g["props"]).max() < PROPS_ATOL
    mix = res["read_mix"].cpu().numpy()
    best = mix.argmax(axis=1)
    assert numpy.array_equal(best, g["mix_argmax"])                 # identical haplogroup calls
    votes = numpy.zeros(len(haps))
    numpy.add.at(votes, best, g["wts"])
    assert numpy.array_equal(votes, g["votes"])
    assert numpy.array_equal(numpy.flatnonzero(votes >= 10), g["contributors"])
    assert numpy.allclose(mix[:4], g["mix_rows"], rtol=0, atol=1e-8)
    assert numpy.allclose(mix.max(axis=1), g["mix_rowmax"], rtol=0, atol=1e-8)


@pytest.mark.parametrize("storage", ["f64", "coded"])
@pytest.mark.parametrize("fused", [-1, 0])
def test_run_em_b17_stopped_by_max_iter_golden(b17, storage, fused):
    """
    g15: the reference's loop stopped by max_iter after 5 and 25 iterations (em.py:140-142) -- a mid-run pin: exactly
    k iterations, state "max_iter reached", the LAST step's proportions and posterior; dense and records, one launch and
    per-iteration kernels.
    """
    from mixemt_amd import _lib, em
    refseq, phy, haps, tables = b17
    g = golden("g15_run_em_max_iter")
    mat = _b17_matrix(tables, g, len(haps))
    _lib.load().mxm_set_loop_fused(fused, 0)                # (conftest resets every knob after the test)
    for k in (5, 25):
        numpy.random.seed(7)
        res = em.run_em_ex(mat, g["wts"], em_args(max_iter=k), storage=storage)
        assert res["iters"] == [k] and res["done"] == [2] and res["storage"] == storage
        assert numpy.abs(res["props"] - g["props_%d" % k]).max() < PROPS_ATOL
        mix = res["read_mix"].cpu().numpy()
        assert numpy.array_equal(mix.argmax(axis=1), g["mix_argmax_%d" % k])
        assert numpy.allclose(mix[:4], g["mix_rows_%d" % k], rtol=0, atol=1e-8)
        assert numpy.allclose(mix.max(axis=1), g["mix_rowmax_%d" % k], rtol=0, atol=1e-8)


def test_run_em_b17_golden_2400_rows_with_repeat_weights(b17):
    """
    g9: the reference's run_em on 2400 x 5408 with the weights reduce_reads leaves
    (preprocess.py:218-220: fragments per distinct signature, repeats up to 400) -- 4x the rows of
    g4.  Bit-exact matrix, identical init / iteration count / haplogroup calls / votes /
    contributor set, proportions within 1e-9, posterior within 1e-8.
    """
    import hashlib
    from mixemt_amd import assign, em, preprocess
    refseq, phy, haps, tables = b17
    g = golden("g9_run_em_2400")
    mat = preprocess.build_em_matrix_device(tables, g["row_ptr"], g["site"], g["obs"])      # built on the device
    assert hashlib.sha256(mat.cpu().numpy().tobytes()).hexdigest() == str(g["mat_sha256"])  # the reference's bits
    assert int(g["wts"].max()) >= 100
    numpy.random.seed(17)
    res = em.run_em_ex(mat, g["wts"], em_args())
    assert numpy.array_equal(res["inits"], g["inits"])
    assert res["iters"] == list(g["iters"])
    assert numpy.abs(res["props"] - g["props"]).max() < PROPS_ATOL
    best, votes = assign.row_argmax_votes(res["read_mix"], g["wts"])
    assert numpy.array_equal(best, g["mix_argmax"])                 # identical haplogroup calls, all 2400 rows
    assert hashlib.sha256(numpy.ascontiguousarray(best.astype(g["mix_argmax"].dtype)).tobytes()).hexdigest() \
        == str(g["mix_argmax_sha256"])
    assert numpy.array_equal(votes, g["votes"])
    assert numpy.array_equal(numpy.flatnonzero(votes >= 10), g["contributors"])
    mix = res["read_mix"].cpu().numpy()
    assert numpy.allclose(mix[:4], g["mix_rows"], rtol=0, atol=1e-8)
    assert numpy.allclose(mix.max(axis=1), g["mix_rowmax"], rtol=0, atol=1e-8)


def test_run_em_b17_multi_golden(b17):
    """n_multi = 3: sequential init draws, geometric-mean proportions (sum != 1), folded posterior."""
    from mixemt_amd import em
    refseq, phy, haps, tables = b17
    g = golden("g5_run_em_multi")
    mat = _b17_matrix(tables, g, len(haps))
    numpy.random.seed(11)
    res = em.run_em_ex(mat, g["wts"], em_args(n_multi=3))
    assert numpy.array_equal(res["inits"], g["inits"])
    assert res["iters"] == list(g["iters"])
    assert numpy.abs(res["props"] - g["props"]).max() < PROPS_ATOL
    assert abs(res["props"].sum() - g["props"].sum()) < 1e-9
    mix = res["read_mix"].cpu().numpy()
    assert numpy.array_equal(mix.argmax(axis=1), g["mix_argmax"])
    assert numpy.allclose(mix[:16], g["mix_rows"], rtol=0, atol=1e-8)


def test_run_em_refinement_shape_golden(b17):
    """600 x 5 contributor columns (bin/mixemt:311-320): tiny H."""
    from mixemt_amd import em, preprocess
    refseq, phy, haps, tables = b17
    g = golden("g6_refine")
    mat = _b17_matrix(tables, g, len(haps))
    contribs = [["hap%d" % (i + 1), haps[c], 0.0] for i, c in enumerate(g["cols"])]
    sub, names = preprocess.reduce_em_matrix(mat, haps, contribs)
    assert names == [haps[c] for c in g["cols"]]
    numpy.random.seed(5)
    res = em.run_em_ex(sub, g["wts"], em_args())
    assert res["iters"] == list(g["iters"])
    assert numpy.abs(res["props"] - g["props"]).max() < PROPS_ATOL
    mix = res["read_mix"].cpu().numpy()
    assert numpy.allclose(numpy.exp(mix), numpy.exp(g["mix"]), rtol=0, atol=1e-9)


@pytest.mark.parametrize("n_rows,n_haps,seed", [(37, 66, 1), (500, 513, 2), (300, 1000, 3), (64, 2049, 4),
                                                (33, 8192, 5), (129, 5408, 6), (20, 1, 7), (300, 2, 8),
                                                (70, 16, 9), (70, 17, 10), (50, 32, 11), (50, 33, 12),
                                                (50, 64, 13), (50, 65, 14)])
def test_em_iterations_vs_oracle_random_shapes(n_rows, n_haps, seed):
    """A few EM steps on random matrices across the streaming kernel's NCH range, odd H included,
    and across the narrow kernels' widths (thread per row up to 32 columns, workgroup per row to 64)."""
    from mixemt_amd import em
    rng = numpy.random.default_rng(seed)
    mat = rng.normal(-25.0, 8.0, size=(n_rows, n_haps))
    if n_haps > 1:                                  # (a row that is -inf everywhere is its own test below)
        mat[rng.random(mat.shape) < 0.01] = -numpy.inf
    wts = rng.integers(1, 5, size=n_rows)
    init = rng.dirichlet([1.0] * n_haps)
    res = em.run_em_ex(mat, wts, em_args(max_iter=5, tolerance=0.0), inits=init[None, :],
                       want_read_mix=True)
    assert res["iters"] == [5] and res["done"] == [2]
    theta = numpy.log(init)
    buf = numpy.empty_like(mat)
    for _ in range(5):
        buf, theta = em_oracle.em_step(mat, wts, theta, buf)
    assert numpy.abs(res["props"] - numpy.exp(theta)).max() < 1e-12
    mix = res["read_mix"].cpu().numpy()
    fin = numpy.isfinite(buf)
    assert numpy.array_equal(numpy.isfinite(mix), fin)
    assert numpy.allclose(mix[fin], buf[fin], rtol=0, atol=1e-9)


def test_max_iter_exhaustion_matches_reference_semantics():
    """for-else at em.py:141-143: result is (theta_{k+1}, E-step under theta_k) also when not converged."""
    from mixemt_amd import em
    g = golden("g7_config1")
    mat, wts = g["mat"], numpy.ones(1000)
    numpy.random.seed(21)
    res = em.run_em_ex(mat, wts, em_args(max_iter=7))
    numpy.random.seed(21)
    props, mix = em_oracle.run_em(mat, wts, em_args(max_iter=7))
    assert res["iters"] == [7] and res["done"] == [2]
    assert numpy.abs(res["props"] - props).max() < 1e-12
    assert numpy.allclose(numpy.exp(res["read_mix"].cpu().numpy()), numpy.exp(mix), rtol=0, atol=1e-10)


def test_device_resident_pipeline(b17):
    """CSR -> device matrix -> run_em on tensors: nothing but proportions leaves the GPU."""
    import torch
    from mixemt_amd import em, preprocess
    refseq, phy, haps, tables = b17
    g = golden("g4_run_em")
    mat = preprocess.build_em_matrix_device(tables, g["row_ptr"], g["site"], g["obs"])
    assert mat.is_cuda
    keep = mat.clone()
    numpy.random.seed(7)
    props, read_mix = em.run_em(mat, torch.from_numpy(g["wts"]).cuda(), em_args())
    assert isinstance(read_mix, torch.Tensor) and read_mix.is_cuda
    assert torch.equal(mat, keep)                                   # caller's matrix untouched
    assert numpy.abs(props - g["props"]).max() < PROPS_ATOL
    assert numpy.array_equal(read_mix.argmax(dim=1).cpu().numpy(), g["mix_argmax"])


def test_row_argmax_votes_kernel(b17):
    """assemble.py:115-123 / stats.py:39-40 on device (first max wins, weighted votes)."""
    import torch
    from mixemt_amd import _lib
    from mixemt_amd._dev import current_stream
    rng = numpy.random.default_rng(12)
    mat = rng.normal(size=(300, 777))
    mat[5, 100] = mat[5, 600] = 9.0          # tie -> first index
    mat[6, :] = -numpy.inf
    wts = rng.integers(1, 4, size=300).astype(numpy.float64)
    x = torch.from_numpy(mat).cuda()
    w = torch.from_numpy(wts).cuda()
    best = torch.empty(300, dtype=torch.int32, device="cuda")
    votes = torch.zeros(777, dtype=torch.float64, device="cuda")
    lib = _lib.load()
    nbytes = lib.mxm_workspace_bytes(300, 777, 1)
    ws = torch.empty(nbytes // 8 + 1, dtype=torch.float64, device="cuda")
    _lib.check(lib.mxm_row_argmax_votes(x.data_ptr(), x.stride(0), w.data_ptr(), 300, 777,
                                        best.data_ptr(), votes.data_ptr(), ws.data_ptr(), nbytes,
                                        current_stream()), "argmax")
    want = mat.argmax(axis=1)
    assert numpy.array_equal(best.cpu().numpy(), want)
    want_votes = numpy.zeros(777)
    numpy.add.at(want_votes, want, wts)
    assert numpy.array_equal(votes.cpu().numpy(), want_votes)


@pytest.mark.parametrize("n_rows,n_haps,n_runs,seed", [(200, 5408, 3, 1), (150, 1000, 5, 2), (90, 8192, 3, 3),
                                                      (64, 6200, 4, 4), (300, 66, 2, 5), (130, 5408, 4, 6),
                                                      (77, 5408, 10, 7), (50, 4096, 7, 8)])
def test_batched_restarts_share_matrix_reads(n_rows, n_haps, n_runs, seed):
    """
    Restarts advance in tiles of up to 4 that share one pass over the matrix
    (proportions in LDS, the fourth restart's in registers).  Every restart must come out as if run alone:
    compared with the oracle and with the unbatched schedule (tile = 1).
    """
    from mixemt_amd import _lib, em
    lib = _lib.load()
    rng = numpy.random.default_rng(seed)
    mat = rng.normal(-25.0, 8.0, size=(n_rows, n_haps))
    wts = rng.integers(1, 5, size=n_rows)
    inits = rng.dirichlet([1.0] * n_haps, size=n_runs)
    args = em_args(max_iter=4, tolerance=0.0, n_multi=n_runs)
    try:
        lib.mxm_set_batch_tile(4)
        batched = em.run_em_ex(mat, wts, args, inits=inits, want_read_mix=False)
        lib.mxm_set_batch_tile(3)
        three = em.run_em_ex(mat, wts, args, inits=inits, want_read_mix=False)
        lib.mxm_set_batch_tile(1)
        single = em.run_em_ex(mat, wts, args, inits=inits, want_read_mix=False)
    finally:
        lib.mxm_set_batch_tile(4)
    assert batched["iters"] == [4] * n_runs
    for run in range(n_runs):
        theta = numpy.log(inits[run])
        buf = numpy.empty_like(mat)
        for _ in range(4):
            buf, theta = em_oracle.em_step(mat, wts, theta, buf)
        assert numpy.abs(batched["run_props"][run] - numpy.exp(theta)).max() < 1e-12
    assert numpy.abs(batched["run_props"] - single["run_props"]).max() < 1e-13
    assert numpy.abs(three["run_props"] - single["run_props"]).max() < 1e-13


def test_batched_restarts_stop_independently(b17):
    """Restarts of one tile converge on different iterations; finished ones freeze (g5: 391/446/454)."""
    from mixemt_amd import _lib, em
    refseq, phy, haps, tables = b17
    g = golden("g5_run_em_multi")
    mat = _b17_matrix(tables, g, len(haps))
    lib = _lib.load()
    out = {}
    for tile in (3, 2, 1):
        lib.mxm_set_batch_tile(tile)
        try:
            numpy.random.seed(11)
            out[tile] = em.run_em_ex(mat, g["wts"], em_args(n_multi=3), want_read_mix=False)
        finally:
            lib.mxm_set_batch_tile(4)
        assert out[tile]["iters"] == list(g["iters"])
        assert numpy.abs(out[tile]["props"] - g["props"]).max() < PROPS_ATOL


@pytest.mark.parametrize("n_rows,n_haps,seed", [(50, 66, 1), (257, 512, 2), (100, 2560, 3), (64, 5408, 4),
                                                (40, 6656, 5), (33, 8192, 6), (20, 1001, 7), (300, 1, 8), (1, 1, 9),
                                                (70, 2, 10), (70, 3, 11), (90, 31, 12), (90, 33, 13), (90, 64, 14)])
def test_em_step_wide_and_generic_kernels_vs_oracle(n_rows, n_haps, seed):
    """em_step across the register-resident (wide) and the generic E-step kernels, -inf included."""
    from mixemt_amd import em
    rng = numpy.random.default_rng(seed)
    mat = rng.normal(-30.0, 10.0, size=(n_rows, n_haps))
    mat[rng.random(mat.shape) < 0.02] = -numpy.inf
    wts = rng.integers(1, 4, size=n_rows)
    lnp = numpy.log(rng.dirichlet([0.5] * n_haps))
    want_mix, want_new = em_oracle.em_step(mat, wts, lnp, numpy.empty_like(mat))
    got_mix = numpy.empty_like(mat)
    _, got_new = em.em_step(mat, wts, lnp, got_mix)
    fin = numpy.isfinite(want_mix)
    assert numpy.array_equal(numpy.isfinite(got_mix), fin)
    assert numpy.allclose(got_mix[fin], want_mix[fin], rtol=0, atol=1e-10)
    # (one column and a -inf cell: the reference's own step is NaN there -- em.py:81-83 -- and so is this one)
    assert numpy.allclose(numpy.exp(got_new), numpy.exp(want_new), rtol=0, atol=1e-13, equal_nan=True)


def test_posterior_fold_matches_logaddexp():
    """mode 1 of mxm_em_step: out = logaddexp(out, E-step) (em.py:156), wide kernel."""
    import torch
    from mixemt_amd import em
    rng = numpy.random.default_rng(8)
    mat = rng.normal(-30.0, 10.0, size=(70, 5408))
    plan = em.EmPlan(mat, numpy.ones(70))
    lnp_a = numpy.log(rng.dirichlet([1.0] * 5408))
    lnp_b = numpy.log(rng.dirichlet([1.0] * 5408))
    out = em.posterior(plan, lnp_a)
    first = out.cpu().numpy().copy()
    out = em.posterior(plan, lnp_b, out=out, fold=True)
    second, _ = em_oracle.em_step(mat, numpy.ones(70), lnp_b, numpy.empty_like(mat))
    want_first, _ = em_oracle.em_step(mat, numpy.ones(70), lnp_a, numpy.empty_like(mat))
    assert numpy.allclose(first, want_first, rtol=0, atol=1e-10)
    assert numpy.allclose(out.cpu().numpy(), numpy.logaddexp(want_first, second), rtol=0, atol=1e-10)


@pytest.mark.parametrize("name,seed,n_multi", [("g4_run_em", 7, 1), ("g5_run_em_multi", 11, 3)])
def test_f32_storage_variant_stays_inside_the_parity_bar(b17, name, seed, n_multi):
    """
    Opt-in fp32 STORAGE of the streamed matrix (fp64 arithmetic): the north-star
    bar is 1e-6 on proportions and identical haplogroup calls; iteration counts
    are reported by the reference, so they are checked too.
    """
    from mixemt_amd import em
    refseq, phy, haps, tables = b17
    g = golden(name)
    mat = _b17_matrix(tables, g, len(haps))
    numpy.random.seed(seed)
    res = em.run_em_ex(mat, g["wts"], em_args(n_multi=n_multi), storage="f32")
    assert numpy.array_equal(res["inits"], g["inits"])
    assert numpy.abs(res["props"] - g["props"]).max() < 1e-6
    assert res["iters"] == list(g["iters"])
    assert numpy.array_equal(res["read_mix"].argmax(dim=1).cpu().numpy(), g["mix_argmax"])


@pytest.mark.parametrize("n_rows,n_haps,seed", [(40, 66, 1), (300, 1023, 2), (64, 4097, 3), (50, 8192, 4)])
def test_f32_storage_iterations_vs_oracle(n_rows, n_haps, seed):
    from mixemt_amd import em
    rng = numpy.random.default_rng(seed)
    mat = rng.normal(-25.0, 8.0, size=(n_rows, n_haps))
    wts = rng.integers(1, 5, size=n_rows)
    init = rng.dirichlet([1.0] * n_haps)
    res = em.run_em_ex(mat, wts, em_args(max_iter=4, tolerance=0.0), inits=init[None, :], want_read_mix=False,
                       storage="f32")
    theta = numpy.log(init)
    buf = numpy.empty_like(mat)
    for _ in range(4):
        buf, theta = em_oracle.em_step(mat, wts, theta, buf)
    assert numpy.abs(res["props"] - numpy.exp(theta)).max() < 2e-7      # float storage of P: ~6e-8 relative


def test_other_tree_end_to_end_vs_oracle():
    """Build 16 (different H and S than the goldens' Build 17): matrix bit-exact, run within 1e-9."""
    from mixemt_amd import em, phylotree, preprocess, synth
    from oracle import build_oracle
    refseq = phylotree.load_rsrs()
    phy = phylotree.load_build16(refseq)
    haps = sorted(phy.hap_var)
    tables = preprocess.HapVarTables.build(refseq, phy, haps)
    row_ptr, site, obs, _ = synth.synth_reads(tables, len(refseq), 300, seed=21, contrib=(5, 1500, 3000))
    mat = preprocess.build_em_matrix_device(tables, row_ptr, site, obs)
    flat = build_oracle.flat_tables(refseq, phy, haps)
    want = c_oracle.build_em_matrix(flat[1], flat[2], flat[3], row_ptr, site, obs, len(haps))
    assert numpy.array_equal(mat.cpu().numpy(), want)
    wts = numpy.ones(300)
    numpy.random.seed(3)
    res = em.run_em_ex(mat, wts, em_args(max_iter=60))
    trace = []
    numpy.random.seed(3)
    props, mix = em_oracle.run_em(want, wts, em_args(max_iter=60), trace=trace)
    assert res["iters"] == [trace[0]["iters"]]
    assert numpy.abs(res["props"] - props).max() < 1e-9
    assert numpy.array_equal(res["read_mix"].argmax(dim=1).cpu().numpy(), mix.argmax(axis=1))


@pytest.mark.parametrize("n_haps", [64, 1024, 1026, 2048, 3074, 4096, 5120, 5408, 6146, 7168, 7170, 8192])
def test_streaming_kernel_shape_grid(n_haps):
    """
    One fused E+M step for every column-chunk count of the streaming kernel (H = 1024 k and
    just past it) x row counts around the grid size (1, fewer rows than workgroups, one more
    than a multiple) x 1..5 restarts per call (tiles of 1-4 + remainder), against the oracle's
    em_step on the same inputs: colsum_b = exp(new_props_b) * sum(w).
    """
    from mixemt_amd import em
    rng = numpy.random.default_rng(n_haps)
    for n_rows in (1, 2, 7, 255, 257, 1030):
        mat = rng.normal(-20.0, 6.0, size=(n_rows, n_haps))
        mat[rng.random(mat.shape) < 0.002] = -numpy.inf
        wts = rng.integers(1, 4, size=n_rows).astype(numpy.float64)
        n_runs = 1 + (n_rows % 5)
        plan = em.EmPlan(mat, wts, n_runs=n_runs)
        inits = rng.dirichlet([1.0] * n_haps, size=n_runs)
        ln0, p0 = em.log_inits(inits)
        props, lnp = plan.alloc_props(p0), plan.alloc_props(ln0)
        colsum = plan.alloc_props(numpy.zeros_like(p0))
        plan.em_iter(props, lnp, None, colsum)
        got = (colsum * props).cpu().numpy()
        for run in range(n_runs):
            with numpy.errstate(divide="ignore"):
                _, new = em_oracle.em_step(mat, wts, ln0[run], numpy.empty_like(mat))
            want = numpy.exp(new) * wts.sum()
            assert numpy.abs(got[run] - want).max() <= 1e-12 * wts.sum(), (n_rows, n_haps, run)


@pytest.mark.parametrize("n_rows,n_haps,n_runs,seed", [(240, 1000, 9, 1), (90, 5408, 6, 2), (300, 66, 5, 3)])
def test_running_restarts_are_packed_without_changing_results(n_rows, n_haps, n_runs, seed):
    """
    mxm_em_loop packs the restarts that still run into the leading slots (fewer passes per
    iteration once some have stopped) and restores the order before returning: per-run
    iteration counts, proportions and the folded posterior must equal the unpacked schedule's,
    in the caller's run order.
    """
    from mixemt_amd import _lib, em
    lib = _lib.load()
    rng = numpy.random.default_rng(seed)
    mat = rng.normal(-30.0, 10.0, size=(n_rows, n_haps))
    truth = rng.choice(n_haps, size=3, replace=False)
    for r in range(n_rows):
        mat[r, truth[r % 3]] += 22.0
    wts = rng.integers(1, 4, size=n_rows)
    inits = rng.dirichlet([0.3] * n_haps, size=n_runs)
    args = em_args(n_multi=n_runs, max_iter=300, tolerance=1e-5)
    out = {}
    for on in (1, 0):                        # (the library default is 2; conftest resets every knob after the test)
        lib.mxm_set_compact_restarts(on)
        out[on] = em.run_em_ex(mat, wts, args, inits=inits, want_read_mix=True)
    assert len(set(out[0]["iters"])) > 1, "the case must have restarts stopping on different iterations"
    assert out[1]["iters"] == out[0]["iters"]
    assert numpy.abs(out[1]["run_props"] - out[0]["run_props"]).max() < 1e-13
    assert numpy.abs(out[1]["props"] - out[0]["props"]).max() < 1e-13
    a, b = out[1]["read_mix"].cpu().numpy(), out[0]["read_mix"].cpu().numpy()
    fin = numpy.isfinite(a) & numpy.isfinite(b)
    assert numpy.array_equal(numpy.isfinite(a), numpy.isfinite(b))
    assert numpy.abs(a[fin] - b[fin]).max() < 1e-9
    for run in range(n_runs):        # and each run is the oracle's run from the same init
        with numpy.errstate(divide="ignore"):
            theta, _, n_iter = em_oracle._one_run(mat, wts, numpy.log(inits[run]), numpy.empty_like(mat),
                                                  300, 1e-5, False)
        assert out[1]["iters"][run] == n_iter
        assert numpy.abs(out[1]["run_props"][run] - numpy.exp(theta)).max() < 1e-9


def test_posterior_reuses_the_linearised_matrix_storage():
    """run_em's posterior lands in the buffer the loop streamed from (no second R x H allocation);
    the caller's matrix is untouched and the values equal a freshly allocated posterior's."""
    import torch
    from mixemt_amd import em
    rng = numpy.random.default_rng(12)
    mat = torch.from_numpy(rng.normal(-25.0, 8.0, size=(300, 1000))).cuda()
    keep = mat.clone()
    wts = torch.ones(300, dtype=torch.float64, device="cuda")
    inits = rng.dirichlet([1.0] * 1000, size=2)
    args = em_args(n_multi=2, max_iter=6, tolerance=0.0)
    plan = em.EmPlan(mat, wts, n_runs=2)
    lin_ptr = plan.lin.data_ptr()
    ln_cur, ln_new, states = em.em_loop(plan, inits, args.tolerance, args.max_iter)
    fresh = em.collect_result(plan, inits, ln_cur, ln_new, states)["read_mix"]
    assert fresh.data_ptr() != lin_ptr and plan.lin is not None
    reused = em.collect_result(plan, inits, ln_cur, ln_new, states, reuse_linear=True)["read_mix"]
    assert reused.data_ptr() == lin_ptr and plan.lin is None
    assert torch.equal(reused, fresh) and torch.equal(mat, keep)
    with pytest.raises(ValueError):
        plan.em_iter(ln_cur, ln_cur, None, ln_cur.clone())      # a spent plan fails loudly


@pytest.mark.parametrize("n_rows,n_haps", [(40, 3), (60, 20), (50, 48), (50, 66), (30, 1001), (70, 5408)])
def test_row_without_any_possible_haplogroup_poisons_like_the_reference(n_rows, n_haps):
    """
    A row that is -inf in every column: the reference's E-step forms -inf - (-inf) = NaN
    (em.py:81-83) and the weighted column logsumexp (em.py:87) spreads it to every proportion;
    the loop then never converges and stops at max_iter with NaN proportions.  With weight 0 on
    that row scipy's logsumexp drops it and everything stays finite.  Same here, on every kernel
    family (thread per row, workgroup per row, linear-space streaming, em_step's wide kernel).
    """
    from mixemt_amd import em
    rng = numpy.random.default_rng(n_haps)
    mat = rng.normal(-25.0, 8.0, size=(n_rows, n_haps))
    mat[7, :] = -numpy.inf
    init = rng.dirichlet([1.0] * n_haps)
    for w7 in (2.0, 0.0):
        wts = rng.integers(1, 4, size=n_rows).astype(numpy.float64)
        wts[7] = w7
        with numpy.errstate(invalid="ignore", divide="ignore"):
            want_mix, want_new = em_oracle.em_step(mat, wts, numpy.log(init), numpy.empty_like(mat))
            theta = numpy.log(init)
            for _ in range(3):
                _, theta = em_oracle.em_step(mat, wts, theta, numpy.empty_like(mat))
        assert numpy.isnan(want_new).all() == (w7 != 0.0)          # what the reference does
        mix, new = em.em_step(mat, wts, numpy.log(init), numpy.empty_like(mat))
        assert numpy.array_equal(numpy.isnan(new), numpy.isnan(want_new))
        assert numpy.array_equal(numpy.isnan(mix), numpy.isnan(want_mix))
        res = em.run_em_ex(mat, wts, em_args(max_iter=3, tolerance=1e-4), inits=init[None, :], want_read_mix=False)
        assert numpy.array_equal(numpy.isnan(res["props"]), numpy.isnan(theta))
        if w7 != 0.0:
            assert res["iters"] == [3] and res["done"] == [2]      # NaN never passes the convergence test
        else:
            assert numpy.abs(res["props"] - numpy.exp(theta)).max() < 1e-12
            assert numpy.abs(new - want_new).max() < 1e-12


def test_single_contributor_refinement_golden(b17):
    """
    The commonest real input, an unmixed sample: one contributor -> reduce_em_matrix keeps ONE column
    (preprocess.py:247-251) and bin/mixemt:311-320 runs run_em on R x 1, then update_contribs and assign_read_indexes with
    a single contributor.  Golden g14 is the reference's own run of exactly that (600 x 1, n_multi 1 and 3) -- through the
    dense matrix and through records (reduce_em_records), VERDICT r3 #2.
    """
    from mixemt_amd import assign, em, preprocess
    refseq, phy, haps, tables = b17
    g = golden("g14_single_contributor")
    contribs = [["hap1", haps[int(g["col"][0])], 1.0]]
    mat = preprocess.build_em_matrix_device(tables, g["row_ptr"], g["site"], g["obs"])
    sub, names = preprocess.reduce_em_matrix(mat, haps, contribs)
    assert names == [haps[int(g["col"][0])]] and tuple(sub.shape) == (600, 1)
    assert numpy.array_equal(sub.cpu().numpy(), g["sub"])                       # the reference's column, bit for bit
    cm = preprocess.build_em_records_device(tables, g["row_ptr"], g["site"], g["obs"])
    sub_r, names_r = preprocess.reduce_em_records(cm, haps, contribs)
    assert names_r == names and numpy.array_equal(sub_r.cpu().numpy(), g["sub"])
    for label, inp in (("dense", sub), ("records", sub_r), ("numpy", g["sub"])):
        numpy.random.seed(13)
        res = em.run_em_ex(inp, g["wts"], em_args())
        assert numpy.array_equal(res["inits"], g["inits"]), label
        assert res["iters"] == list(g["iters"]) and res["done"] == [1]
        assert numpy.array_equal(res["props"], g["props"])                     # exactly [1.0]
        mix = res["read_mix"].cpu().numpy()
        assert mix.shape == (600, 1) and numpy.abs(mix - g["mix"]).max() < 1e-12
        numpy.random.seed(13)
        res3 = em.run_em_ex(inp, g["wts"], em_args(n_multi=3))
        assert res3["iters"] == list(g["iters3"]) and numpy.abs(res3["props"] - g["props3"]).max() < 1e-15
        assert numpy.abs(res3["read_mix"].cpu().numpy() - g["mix3"]).max() < 1e-12
    refined = assign.update_contribs([list(c) for c in contribs], (res["props"], res["read_mix"]), names)
    assert [c[2] for c in refined] == list(g["refined_props"])
    table = assign.assign_read_indexes(refined, (res["props"], res["read_mix"]), names, [[str(i)] for i in range(600)], 2.0)
    assert list(table) == ["hap1"] and sorted(table["hap1"]) == list(g["assigned"])
    # the drop-in signature on host arrays
    numpy.random.seed(13)
    props, mix = em.run_em(g["sub"], g["wts"], em_args())
    assert isinstance(mix, numpy.ndarray) and numpy.array_equal(props, g["props"]) and numpy.abs(mix - g["mix"]).max() < 1e-12


def test_verbose_progress_text_of_a_multi_run_is_live_and_the_references(capsys):
    """-v with n_multi = 3: 'Starting EM run i...', dots, 'Converged! (n)' per run, in run order, written while
    the loop is going on (em.py:119-135).  Round 5: the restarts still advance in ONE batched loop, so -v changes
    no bit of the result -- checked on the toy tree and on a Build-17 matrix of the size real inputs have."""
    from mixemt_amd import em
    g = golden("g1_toy")
    key = "m3_s2"
    mat, wts = g["mat"], numpy.ones(10)
    numpy.random.seed(2)
    quiet = em.run_em_ex(mat, wts, em_args(n_multi=3, max_iter=1000), want_read_mix=False)
    capsys.readouterr()
    numpy.random.seed(2)
    loud = em.run_em_ex(mat, wts, em_args(n_multi=3, max_iter=1000, verbose=True), want_read_mix=False)
    want = "".join("Starting EM run %d...\n" % (i + 1) + "." * (n // 10) + "\nConverged! (%d)\n" % n
                   for i, n in enumerate(loud["iters"]))
    assert capsys.readouterr().err == want
    assert loud["iters"] == quiet["iters"] and numpy.array_equal(loud["props"], quiet["props"])
    assert loud["iters"] == list(g[key + "_iters"]) and numpy.abs(loud["props"] - g[key + "_props"]).max() < PROPS_ATOL
    # the same at a realistic size (2400 x 5408, two restarts): identical bits, the reference's text
    g9 = golden("g9_run_em_2400")
    from mixemt_amd import phylotree, preprocess
    refseq = phylotree.load_rsrs()
    phy = phylotree.load_build17(refseq)
    tables = preprocess.HapVarTables.build(refseq, phy, sorted(phy.hap_var))
    big = preprocess.build_em_matrix_device(tables, g9["row_ptr"], g9["site"], g9["obs"])
    runs = []
    for verbose in (False, True):
        numpy.random.seed(5)
        runs.append(em.run_em_ex(big, g9["wts"], em_args(n_multi=2, max_iter=60, verbose=verbose), want_read_mix=False))
    text = capsys.readouterr().err
    assert runs[0]["iters"] == runs[1]["iters"] == [60, 60] and numpy.array_equal(runs[0]["props"], runs[1]["props"])
    assert text == "".join("Starting EM run %d...\n" % (i + 1) + "." * 6 for i in range(2))


@pytest.mark.parametrize("fused", [1, 2, 0])
def test_verbose_progress_text_is_the_references(capsys, fused):
    """-v: 'Starting EM run 1...', a dot per 10 iterations WHILE the loop runs, 'Converged! (n)'
    (em.py:119-135) -- through both one-launch loops (chunks of 10) and the per-iteration kernels."""
    from mixemt_amd import _lib, em
    g = golden("g7_config1")
    lib = _lib.load()
    lib.mxm_set_loop_fused(fused, 0)
    try:
        numpy.random.seed(7)
        res = em.run_em_ex(g["mat"], numpy.ones(1000, dtype=numpy.int64), em_args(verbose=True), want_read_mix=False)
    finally:
        lib.mxm_set_loop_fused(-1, 0)
    n = int(g["iters"][0])
    assert res["iters"] == [n]
    assert capsys.readouterr().err == "Starting EM run 1...\n" + "." * (n // 10) + "\nConverged! (%d)\n" % n
    assert numpy.abs(res["props"] - g["props"]).max() < PROPS_ATOL


def test_two_host_threads_on_two_streams_while_a_third_turns_the_knobs(b17):
    """
    VERDICT r2 #9: the library's tuning state is process-wide but guarded -- setters lock, every entry point works on
    the snapshot it took when it started.  Two host threads run EM (one-launch loop and per-iteration kernels,
    each on a stream of its own) while a third keeps changing knobs: every run ends on the reference's iteration
    with the reference's proportions (goldens g4 and g9), and nothing deadlocks.
    """
    import threading
    import torch
    from mixemt_amd import _lib, em
    refseq, phy, haps, tables = b17
    lib = _lib.load()
    jobs = []
    for name, seed in (("g4_run_em", 7), ("g9_run_em_2400", 17)):
        g = golden(name)
        mat = c_oracle.build_em_matrix(tables.expected, tables.lhit, tables.lmiss, g["row_ptr"], g["site"], g["obs"], len(haps))
        jobs.append((g, torch.from_numpy(mat).cuda(), torch.from_numpy(g["wts"].astype(numpy.float64)).cuda()))
    torch.cuda.synchronize()
    out, errors, stop = {}, [], threading.Event()

    def worker(k):
        try:
            g, mat, wts = jobs[k]
            with torch.cuda.stream(torch.cuda.Stream()):
                for rep in range(3):
                    res = em.run_em_ex(mat, wts, em_args(), inits=g["inits"], want_read_mix=False)
                    out[(k, rep)] = (res["iters"], res["props"])
        except Exception as exc:                      # pragma: no cover
            errors.append(repr(exc))

    def fiddler():
        i = 0
        while not stop.is_set():
            lib.mxm_set_batch_tile(1 + i % 4)
            lib.mxm_set_compact_restarts(i % 3)
            lib.mxm_set_loop_graph(i % 2)
            lib.mxm_set_min_rows_per_wg(1 + i % 8)
            i += 1
        lib.mxm_reset_tuning()

    threads = [threading.Thread(target=worker, args=(k,)) for k in (0, 1)] + [threading.Thread(target=fiddler)]
    for th in threads:
        th.start()
    for th in threads[:2]:
        th.join(timeout=300)
    stop.set()
    threads[2].join(timeout=30)
    assert not errors, errors
    assert not any(th.is_alive() for th in threads)
    for k in (0, 1):
        g = jobs[k][0]
        for rep in range(3):
            iters, props = out[(k, rep)]
            assert iters == list(g["iters"])
            assert numpy.abs(props - g["props"]).max() < 1e-9
