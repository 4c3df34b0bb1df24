"""
The one-launch EM loop (em_fused_loop_kernel: the whole run_em inner loop, em.py:126-143, of a
cache-resident matrix inside one persistent grid with grid barriers) against the reference's
goldens, against the oracle, and against the per-iteration kernels it replaces at these sizes.
"""
import numpy
import pytest

from conftest import em_args, golden
from oracle import c_oracle, em_oracle

pytestmark = pytest.mark.gpu

PROPS_ATOL = 1e-9


@pytest.fixture()
def lib():
    from mixemt_amd import _lib
    handle = _lib.load()
    yield handle
    handle.mxm_set_loop_fused(-1, 0)


def _b17_matrix(tables, g, n_haps):
    return c_oracle.build_em_matrix(tables.expected, tables.lhit, tables.lmiss, g["row_ptr"],
                                    g["site"], g["obs"], n_haps)


@pytest.mark.parametrize("name,seed,n_multi", [("g4_run_em", 7, 1), ("g5_run_em_multi", 11, 3)])
def test_fused_loop_reproduces_the_reference_runs(b17, lib, name, seed, n_multi):
    """600 x 5408: same init draws, same stopping iterations, proportions within 1e-9, identical
    calls -- through the one-launch loop, through the per-iteration kernels, and in launches of 7."""
    from mixemt_amd import em
    refseq, phy, haps, tables = b17
    g = golden(name)
    mat = _b17_matrix(tables, g, len(haps))
    runs = {}
    # 600 rows: mode 1 takes the transposed one-launch loop (columns split over the workgroups, matrix in
    # registers), mode 2 the row-split one-launch loop, mode 0 the per-iteration kernels
    for label, mode, chunk in (("fused", 1, 0), ("rows", 2, 0), ("kernels", 0, 0), ("chunks", 1, 7),
                               ("rows-chunks", 2, 5)):
        lib.mxm_set_loop_fused(mode, chunk)
        numpy.random.seed(seed)
        res = em.run_em_ex(mat, g["wts"], em_args(n_multi=n_multi))
        assert numpy.array_equal(res["inits"], g["inits"])
        assert res["iters"] == list(g["iters"]), label
        assert res["done"] == [1] * n_multi
        assert numpy.abs(res["props"] - g["props"]).max() < PROPS_ATOL
        mix = res["read_mix"].cpu().numpy()
        assert numpy.array_equal(mix.argmax(axis=1), g["mix_argmax"])
        assert numpy.allclose(mix.max(axis=1), g["mix_rowmax"], rtol=0, atol=1e-8)
        runs[label] = res
    # a resumed launch continues from the saved proportions: chunking changes nothing at all
    assert numpy.array_equal(runs["fused"]["run_props"], runs["chunks"]["run_props"])
    assert runs["fused"]["l1"] == runs["chunks"]["l1"]
    assert numpy.array_equal(runs["rows"]["run_props"], runs["rows-chunks"]["run_props"])
    # the two one-launch forms differ by summation order and by how the normaliser is formed
    assert numpy.abs(runs["fused"]["run_props"] - runs["rows"]["run_props"]).max() < 1e-12
    # against the per-iteration kernels: another summation order, nothing more
    assert numpy.abs(runs["fused"]["run_props"] - runs["kernels"]["run_props"]).max() < 1e-12


@pytest.mark.parametrize("n_rows,n_haps,seed", [(1, 5408, 1), (5, 777, 2), (300, 66, 3), (257, 6144, 4),
                                                (1500, 1001, 5), (64, 5408, 6), (4000, 512, 7), (1024, 5408, 8),
                                                (1025, 5408, 9), (1536, 3072, 10), (1537, 3000, 11), (513, 6100, 12)])
def test_fused_loop_matches_oracle_on_random_shapes(lib, n_rows, n_haps, seed):
    """Fewer rows than workgroups, odd widths, the widest instance: iteration count and result of
    the oracle's run_em (weights with repeats, a zero weight, -inf entries)."""
    from mixemt_amd import em
    rng = numpy.random.default_rng(seed)
    mat = rng.normal(size=(n_rows, n_haps)) * 3.0 - 10.0
    hot = rng.integers(0, n_haps, size=n_rows)
    mat[numpy.arange(n_rows), hot % 7] += 12.0                 # a few haplogroups explain most rows
    mat[rng.random(mat.shape) < 0.01] = -numpy.inf
    mat[numpy.arange(n_rows), hot % 7] = numpy.maximum(mat[numpy.arange(n_rows), hot % 7], -5.0)
    wts = rng.integers(0, 5, size=n_rows).astype(numpy.float64)
    wts[0] = 2.0
    args = em_args(max_iter=300)
    trace = []
    numpy.random.seed(seed)
    props, mix = em_oracle.run_em(mat, wts, args, trace=trace)
    finite = numpy.isfinite(mix)
    for mode in (1, 2):                          # transposed form where it applies / rows split
        lib.mxm_set_loop_fused(mode, 0)
        numpy.random.seed(seed)
        res = em.run_em_ex(mat, wts, args)
        assert res["iters"] == [trace[0]["iters"]], mode
        assert numpy.abs(res["props"] - props).max() < PROPS_ATOL
        got = res["read_mix"].cpu().numpy()
        assert numpy.array_equal(numpy.isfinite(got), finite)
        assert numpy.abs(numpy.exp(got) - numpy.exp(mix)).max() < 1e-9


def test_fused_loop_runs_out_of_iterations_like_the_reference(lib):
    """max_iter exhausted (em.py:141-143: the for-else swaps back): theta_{k+1} with the posterior
    under theta_k, done = 2, and the iteration count is max_iter."""
    from mixemt_amd import em
    rng = numpy.random.default_rng(9)
    mat = rng.normal(size=(200, 900)) * 2.0
    trace = []
    numpy.random.seed(4)
    props, mix = em_oracle.run_em(mat, numpy.ones(200), em_args(max_iter=9), trace=trace)
    for mode in (1, 2):
        lib.mxm_set_loop_fused(mode, 0)
        numpy.random.seed(4)
        res = em.run_em_ex(mat, numpy.ones(200), em_args(max_iter=9))
        assert res["iters"] == [9] and res["done"] == [2]
        assert numpy.abs(res["props"] - props).max() < PROPS_ATOL
        assert numpy.abs(res["read_mix"].cpu().numpy() - mix).max() < 1e-9


def test_fused_loop_poisons_like_the_reference(lib):
    """A row that is -inf in every column makes every proportion NaN (em.py:81-83, :87) and the loop
    runs to max_iter; with weight 0 scipy drops the row and nothing happens."""
    from mixemt_amd import em
    rng = numpy.random.default_rng(10)
    mat = rng.normal(size=(50, 300))
    mat[7, :] = -numpy.inf
    for mode in (1, 2):
        lib.mxm_set_loop_fused(mode, 0)
        wts = numpy.ones(50)
        numpy.random.seed(1)
        res = em.run_em_ex(mat, wts, em_args(max_iter=12), want_read_mix=False)
        assert res["iters"] == [12] and res["done"] == [2] and numpy.isnan(res["props"]).all()
        wts[7] = 0.0
        numpy.random.seed(1)
        res = em.run_em_ex(mat, wts, em_args(max_iter=500), want_read_mix=False)
        trace = []
        numpy.random.seed(1)
        with numpy.errstate(invalid="ignore"):
            props, _ = em_oracle.run_em(mat, wts, em_args(max_iter=500), trace=trace)
        assert res["iters"] == [trace[0]["iters"]] and numpy.abs(res["props"] - props).max() < PROPS_ATOL


def test_auto_selection_by_size(b17, lib):
    """Automatic mode: the 600-row golden takes the one-launch loop, and says so by its speed-independent
    trace -- identical bits to forcing it; a matrix above the cache-resident bound does not."""
    from mixemt_amd import em
    refseq, phy, haps, tables = b17
    g = golden("g4_run_em")
    mat = _b17_matrix(tables, g, len(haps))
    lib.mxm_set_loop_fused(-1, 0)
    numpy.random.seed(7)
    auto = em.run_em_ex(mat, g["wts"], em_args(), want_read_mix=False)
    lib.mxm_set_loop_fused(1, 0)
    numpy.random.seed(7)
    forced = em.run_em_ex(mat, g["wts"], em_args(), want_read_mix=False)
    assert numpy.array_equal(auto["props"], forced["props"]) and auto["iters"] == forced["iters"]


@pytest.mark.parametrize("rows,chunk", [(600, 0), (600, 7), (2400, 0)])
def test_a_one_launch_loop_that_gives_up_is_undone_and_finished_by_the_kernels(b17, lib, rows, chunk):
    """
    ADVICE r2: if the persistent grid cannot run to its end (not co-resident / starved at a grid barrier) the
    call must not fail with the loop vectors left mid-run.  mxm_diag_fused_force_abort raises the abort flag
    before the launch -- exactly what a workgroup that waited in vain does: in automatic mode the launch is
    undone from the snapshot and the SAME call finishes through the per-iteration kernels with the reference's
    result (g4: 600 rows -> transposed loop; g9: 2400 rows -> row-split loop); with mxm_set_loop_fused(1) it is
    the documented error -3, and the loop vectors are intact even then.
    """
    from mixemt_amd import em
    refseq, phy, haps, tables = b17
    g = golden("g4_run_em" if rows == 600 else "g9_run_em_2400")
    seed = 7 if rows == 600 else 17
    mat = _b17_matrix(tables, g, len(haps))
    lib.mxm_set_loop_fused(0, 0)
    numpy.random.seed(seed)
    want = em.run_em_ex(mat, g["wts"], em_args())                  # the per-iteration kernels, undisturbed
    lib.mxm_set_loop_fused(-1, chunk)
    lib.mxm_diag_fused_force_abort(1)
    numpy.random.seed(seed)
    got = em.run_em_ex(mat, g["wts"], em_args())
    assert got["iters"] == list(g["iters"]) == want["iters"] and got["done"] == [1]
    assert numpy.array_equal(got["run_props"], want["run_props"])   # the very same kernels ran from the very same state
    assert numpy.abs(got["props"] - g["props"]).max() < PROPS_ATOL
    # forced mode: the error is reported, nothing is left half-done
    lib.mxm_set_loop_fused(1, chunk)
    plan = em.EmPlan(mat, g["wts"])
    with pytest.raises(ValueError, match="one-launch loop"):
        em.em_loop(plan, g["inits"], 1e-4, 10000)
    lib.mxm_diag_fused_force_abort(0)
    ln_cur, ln_new, states = em.em_loop(plan, g["inits"], 1e-4, 10000)
    assert [s[1] for s in states] == list(g["iters"])


@pytest.mark.parametrize("n_rows,n_haps", [(1500, 66), (1536, 96), (1100, 80)])
def test_transposed_loop_on_a_narrow_matrix_stays_inside_its_workspace(lib, n_rows, n_haps):
    """
    ADVICE r3 (medium): the transposed one-launch loop's z partials are [workgroups][R] doubles -- 3.2 MB at 1500 rows
    on 256 CUs -- while the workspace used to be sized by H alone (2.2 MB at H = 66), so they ran past its end and
    over the give-up snapshot.  The workspace query now covers that layout; the run must match the oracle, and a
    launch that gives up must be undone from an intact snapshot (same bits as the per-iteration kernels).
    """
    from mixemt_amd import _lib, em
    rng = numpy.random.default_rng(n_rows + n_haps)
    mat = rng.normal(size=(n_rows, n_haps)) * 3.0 - 8.0
    mat[numpy.arange(n_rows), rng.integers(0, 5, size=n_rows)] += 10.0
    wts = rng.integers(1, 4, size=n_rows).astype(numpy.float64)
    args = em_args(max_iter=400)
    trace = []
    numpy.random.seed(3)
    props, _ = em_oracle.run_em(mat, wts, args, trace=trace)
    # the query covers sync block + z partials + c + L1 partials + snapshot for a 1024-workgroup grid
    need = 1025 * ((n_rows + 1) // 2 * 2) * 8
    assert _lib.load().mxm_workspace_bytes(n_rows, n_haps, 1) > need
    lib.mxm_set_loop_fused(1, 0)
    numpy.random.seed(3)
    res = em.run_em_ex(mat, wts, args, want_read_mix=False)
    assert res["iters"] == [trace[0]["iters"]]
    assert numpy.abs(res["props"] - props).max() < PROPS_ATOL
    lib.mxm_set_loop_fused(0, 0)
    numpy.random.seed(3)
    want = em.run_em_ex(mat, wts, args, want_read_mix=False)
    lib.mxm_set_loop_fused(-1, 0)
    lib.mxm_diag_fused_force_abort(1)
    numpy.random.seed(3)
    got = em.run_em_ex(mat, wts, args, want_read_mix=False)
    lib.mxm_diag_fused_force_abort(0)
    assert got["iters"] == want["iters"] and numpy.array_equal(got["run_props"], want["run_props"])
