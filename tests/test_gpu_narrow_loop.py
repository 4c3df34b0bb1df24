"""
The one-launch loop of the refinement EM (em_fused_narrow_kernel, csrc/fused_narrow_kernels.hpp): the contributors'
columns of the matrix (preprocess.py:230-251 -> bin/mixemt:311-320, H' = 1 .. 16) iterate inside one persistent grid
with the matrix in registers.  Pinned to the reference's own runs (goldens g6: 600 x 3/5 columns, g14: 600 x 1) through
test_gpu_em.py / test_gpu_consumers.py, which now take this loop by default; here: the oracle on random narrow
matrices, and the per-iteration kernels (mxm_set_loop_fused(0)) as the second witness at sizes the oracle is slow at.
"""
import numpy
import pytest

from conftest import em_args
from oracle import em_oracle

pytestmark = pytest.mark.gpu


@pytest.fixture()
def lib():
    from mixemt_amd import _lib
    return _lib.load()


def _run(mat, wts, inits, lib, fused, **kw):
    from mixemt_amd import em
    lib.mxm_set_loop_fused(1 if fused else 0, kw.pop("chunk", 0))
    try:
        return em.run_em_ex(mat, wts, em_args(n_multi=len(inits), **kw), inits=inits)
    finally:
        lib.mxm_reset_tuning()


@pytest.mark.parametrize("n_rows,n_haps,seed", [(1, 1, 1), (7, 2, 2), (600, 3, 3), (513, 4, 4), (1500, 5, 5), (2000, 8, 6),
                                                (900, 9, 7), (700, 16, 8)])
def test_narrow_loop_matches_the_oracle(lib, n_rows, n_haps, seed):
    rng = numpy.random.default_rng(seed)
    mat = rng.normal(-20.0, 6.0, size=(n_rows, n_haps))
    mat[rng.random(mat.shape) < 0.05] = -numpy.inf
    mat[:, 0] = rng.normal(-15.0, 3.0, size=n_rows)                 # no row is -inf everywhere
    wts = rng.integers(1, 50, size=n_rows).astype(numpy.float64)
    inits = numpy.stack([rng.dirichlet([1.0] * n_haps) for _ in range(3)])
    got = _run(mat, wts, inits, lib, fused=True, tolerance=1e-6, max_iter=400)
    for run in range(3):
        theta = numpy.log(inits[run])
        buf = numpy.empty_like(mat)
        iters = 0
        while True:
            with numpy.errstate(divide="ignore"):
                buf, new = em_oracle.em_step(mat, wts, theta, buf)
            iters += 1
            if em_oracle.converged(new, theta, 1e-6) or iters >= 400:
                break
            theta = new
        assert got["iters"][run] == iters, (run, got["iters"], iters)
        assert numpy.abs(got["run_props"][run] - numpy.exp(new)).max() < 1e-12
        assert numpy.abs(got["ln_theta_k"][run] - theta)[numpy.isfinite(theta)].max(initial=0.0) < 1e-10


@pytest.mark.parametrize("n_rows,n_haps", [(70000, 3), (300000, 4), (1000000, 3), (400000, 7), (100000, 12), (120000, 17),
                                           (1200000, 3)])
def test_narrow_loop_equals_the_per_iteration_kernels(lib, n_rows, n_haps):
    """Same stopping iteration and proportions as mxm_em_iter + mxm_m_finalize (another summation order: rounding only),
    for one and several restarts; 17 columns and 1.2 * 10^6 rows are beyond the one-launch loop (per-iteration path
    either way: equal bits)."""
    import torch
    rng = numpy.random.default_rng(n_rows + n_haps)
    truth = rng.dirichlet([2.0] * n_haps)
    who = rng.choice(n_haps, size=n_rows, p=truth)
    mat = torch.from_numpy(rng.normal(-9.0, 2.0, size=(n_rows, n_haps))).cuda()
    mat[torch.arange(n_rows), torch.from_numpy(who).cuda()] += 6.0
    wts = torch.from_numpy(rng.integers(1, 4, size=n_rows).astype(numpy.float64)).cuda()
    inits = numpy.stack([rng.dirichlet([1.0] * n_haps) for _ in range(2)])
    a = _run(mat, wts, inits, lib, fused=True)
    b = _run(mat, wts, inits, lib, fused=False)
    assert a["iters"] == b["iters"] and a["done"] == b["done"] == [1, 1]
    assert numpy.abs(a["props"] - b["props"]).max() < 1e-13
    assert numpy.abs(a["run_props"] - b["run_props"]).max() < 1e-13
    assert torch.allclose(a["read_mix"], b["read_mix"], rtol=0, atol=1e-10)
    assert numpy.abs(a["run_props"][0] - truth).max() < 0.05          # and it is the planted mixture
    if n_haps > 16 or n_rows > 1048576:
        assert numpy.array_equal(a["props"], b["props"])              # not eligible: the same kernels ran


def test_narrow_loop_chunks_max_iter_and_resume(lib):
    """Launches cut into chunks give the bits of one launch; max_iter ends a run with done = 2 and theta_k / theta_{k+1}
    as the reference leaves them (em.py:137-143); the verbose path (one restart at a time, progress callback) agrees."""
    rng = numpy.random.default_rng(9)
    mat = rng.normal(-12.0, 3.0, size=(5000, 3))
    wts = numpy.ones(5000)
    inits = rng.dirichlet([1.0] * 3)[None, :]
    whole = _run(mat, wts, inits, lib, fused=True, tolerance=1e-9)
    for chunk in (1, 7):
        part = _run(mat, wts, inits, lib, fused=True, tolerance=1e-9, chunk=chunk)
        assert part["iters"] == whole["iters"] and numpy.array_equal(part["props"], whole["props"])
        assert numpy.array_equal(part["ln_theta_k"], whole["ln_theta_k"])
    capped = _run(mat, wts, inits, lib, fused=True, tolerance=1e-9, max_iter=5)
    ref = _run(mat, wts, inits, lib, fused=False, tolerance=1e-9, max_iter=5)
    assert capped["iters"] == [5] and capped["done"] == [2] and ref["done"] == [2]
    assert numpy.abs(capped["props"] - ref["props"]).max() < 1e-14
    assert numpy.abs(capped["ln_theta_k"] - ref["ln_theta_k"]).max() < 1e-13


def test_narrow_loop_poisons_like_the_reference(lib):
    rng = numpy.random.default_rng(4)
    mat = rng.normal(-12.0, 3.0, size=(300, 4))
    mat[17, :] = -numpy.inf
    inits = rng.dirichlet([1.0] * 4)[None, :]
    for w17 in (3.0, 0.0):
        wts = numpy.ones(300)
        wts[17] = w17
        got = _run(mat, wts, inits, lib, fused=True, max_iter=4)
        want = _run(mat, wts, inits, lib, fused=False, max_iter=4)
        assert numpy.isnan(got["props"]).all() == (w17 != 0.0) and numpy.isnan(want["props"]).all() == (w17 != 0.0)
        if w17 == 0.0:
            assert numpy.abs(got["props"] - want["props"]).max() < 1e-14
