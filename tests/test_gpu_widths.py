"""
Widths the reference takes and the fast paths used to refuse (the reference's matrix is whatever
Phylotree build plus custom haplogroups it is given: phylotree.py:231-250, bin/mixemt:104-136):
  * an ODD number of haplogroups (Build 17 + one custom haplogroup = 5409) in row-dictionary records -- straight from
    the build, encoded from a dense matrix, and through every consumer;
  * more than 8192 / 9600 / 10 240 columns: the any-width forms of the E-step and of the finalize.
"""
import numpy
import pytest

from conftest import em_args
from oracle import c_oracle, em_oracle

pytestmark = pytest.mark.gpu


def _oracle_run(mat, wts, init, tol, max_iter):
    theta = numpy.log(init)
    buf = numpy.empty_like(mat)
    iters = 0
    while True:
        with numpy.errstate(divide="ignore"):
            buf, new = em_oracle.em_step(mat, wts, theta, buf)
        iters += 1
        if em_oracle.converged(new, theta, tol) or iters >= max_iter:
            return theta, new, iters, buf
        theta = new


def test_odd_width_takes_the_records_route_and_reproduces_the_oracle():
    """Build 17 + one custom haplogroup (H = 5409): records from the build, storage="auto" from the dense matrix, the
    loop, the posterior, the vote and the column gather -- against the C oracle's matrix and the numpy oracle's loop."""
    import torch
    from mixemt_amd import assign, em, phylotree, preprocess, synth
    refseq = phylotree.load_rsrs()
    phy = phylotree.load_build17(refseq)
    phy.add_custom_hap("zz_custom", ["A73G", "C150T", "T16189C", "G8994A"])
    haps = sorted(phy.hap_var)
    assert len(haps) == 5409 and len(haps) % 2 == 1
    tables = preprocess.HapVarTables.build(refseq, phy, haps)
    n_rows = 700
    row_ptr, site, obs, _ = synth.synth_reads(tables, len(refseq), n_rows, seed=21, contrib=(10, 2000, haps.index("zz_custom")))
    want = c_oracle.build_em_matrix(tables.expected, tables.lhit, tables.lmiss, row_ptr, site, obs, len(haps))
    cm = preprocess.build_em_records_device(tables, row_ptr, site, obs)
    assert cm.n_haps == 5409 and cm.rest_rows.numel() == 0
    assert numpy.array_equal(cm.dense().cpu().numpy(), want)                       # the records hold the reference's bits
    wts = numpy.random.default_rng(2).integers(1, 5, size=n_rows).astype(numpy.float64)
    init = numpy.random.default_rng(3).dirichlet([1.0] * len(haps))
    theta_k, theta_next, iters, post = _oracle_run(want, wts, init, 1e-4, 25)
    runs = {
        "records": em.run_em_ex(None, wts, em_args(max_iter=25), inits=init[None, :], records=cm, want_read_mix=True),
        "auto": em.run_em_ex(torch.from_numpy(want).cuda(), wts, em_args(max_iter=25), inits=init[None, :], storage="coded"),
        "dense": em.run_em_ex(want, wts, em_args(max_iter=25), inits=init[None, :], storage="f64"),
    }
    assert runs["auto"]["storage"] == "coded" and runs["records"]["storage"] == "coded"     # no silent 7 x larger route
    for label, res in runs.items():
        assert res["iters"] == [iters], label
        assert numpy.abs(res["props"] - numpy.exp(theta_next)).max() < 1e-12, label
        mix = res["read_mix"].cpu().numpy() if hasattr(res["read_mix"], "cpu") else res["read_mix"]
        assert mix.shape == (n_rows, 5409) and numpy.abs(mix - post).max() < 1e-9, label
        assert numpy.array_equal(mix.argmax(axis=1), post.argmax(axis=1)), label
    best, votes = assign.row_argmax_votes_records(cm, runs["records"]["ln_theta_k"], wts)
    assert numpy.array_equal(best, post.argmax(axis=1))
    assert numpy.array_equal(votes, numpy.bincount(best, weights=wts, minlength=5409))
    sub, names = preprocess.reduce_em_records(cm, haps, [[None, haps[5408], 0.0], [None, haps[10], 0.0]])
    assert names == [haps[10], haps[5408]] and numpy.array_equal(sub.cpu().numpy(), want[:, [10, 5408]])


@pytest.mark.parametrize("n_rows,n_haps", [(40, 9001), (30, 9601), (24, 12000), (12, 20011)])
def test_any_width_em_against_the_oracle(n_rows, n_haps):
    """Beyond the streaming kernels (8192 columns), the log-space kernel's LDS vectors (9600) and the finalize kernel's
    block table (10 240): same em_step, same stopping iteration, same proportions as the oracle."""
    from mixemt_amd import em
    rng = numpy.random.default_rng(n_haps)
    mat = rng.normal(-30.0, 8.0, size=(n_rows, n_haps))
    mat[rng.random(mat.shape) < 0.01] = -numpy.inf
    mat[:, 7] = rng.normal(-20.0, 2.0, size=n_rows)
    wts = rng.integers(1, 6, size=n_rows).astype(numpy.float64)
    init = rng.dirichlet([1.0] * n_haps)
    with numpy.errstate(divide="ignore"):
        want_mix, want_new = em_oracle.em_step(mat, wts, numpy.log(init), numpy.empty_like(mat))
    mix, new = em.em_step(mat, wts, numpy.log(init), numpy.empty_like(mat))
    fin = numpy.isfinite(want_mix)
    assert numpy.array_equal(fin, numpy.isfinite(mix)) and numpy.abs(mix[fin] - want_mix[fin]).max() < 1e-10
    assert numpy.abs(new - want_new).max() < 1e-10
    theta_k, theta_next, iters, post = _oracle_run(mat, wts, init, 1e-3, 12)
    res = em.run_em_ex(mat, wts, em_args(tolerance=1e-3, max_iter=12), inits=init[None, :])
    assert res["iters"] == [iters] and res["storage"] == "f64"
    assert numpy.abs(res["props"] - numpy.exp(theta_next)).max() < 1e-12
    got = res["read_mix"].cpu().numpy() if hasattr(res["read_mix"], "cpu") else res["read_mix"]
    fin = numpy.isfinite(post)
    assert numpy.abs(got[fin] - post[fin]).max() < 1e-9 and numpy.array_equal(got.argmax(axis=1), post.argmax(axis=1))
    assert em.converged(theta_next, theta_k, 10.0) and not em.converged(theta_next, numpy.log(init), 1e-12)
