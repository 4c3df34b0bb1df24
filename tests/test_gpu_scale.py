"""
Size-independent properties of the hot path at the bench's shape (full Build-17
width, 10^5 rows by default; MXM_TEST_ROWS raises it), where the CPU oracle is
too slow to be the checker: conservation, determinism, shard additivity,
row-permutation invariance, and a sampled comparison with the oracle.
"""
import os

import numpy
import pytest

from conftest import em_args
from oracle import c_oracle, em_oracle

pytestmark = pytest.mark.gpu

N_ROWS = int(os.environ.get("MXM_TEST_ROWS", "100000"))


@pytest.fixture(scope="module")
def big(b17):
    import torch
    from mixemt_amd import em, preprocess, synth
    refseq, phy, haps, tables = b17
    row_ptr, site, obs, who = synth.synth_reads(tables, len(refseq), N_ROWS, seed=1)
    mat = preprocess.build_em_matrix_device(tables, row_ptr, site, obs)
    wts = torch.ones(N_ROWS, dtype=torch.float64, device="cuda")
    return dict(tables=tables, row_ptr=row_ptr, site=site, obs=obs, who=who, mat=mat, wts=wts,
                plan=em.EmPlan(mat, wts), n_haps=len(haps))


def _one_iter(plan, props_host):
    """M-step sums sum_r w_r posterior_rh of one fused iteration (= p_h times the kernel's
    unscaled column sums)."""
    import torch
    props = torch.from_numpy(numpy.ascontiguousarray(props_host[None, :])).cuda()
    ln_props = torch.log(props)
    colsum = torch.zeros_like(props)
    plan.em_iter(props, ln_props, None, colsum)
    torch.cuda.synchronize()
    return (colsum[0] * props[0]).cpu().numpy()


def test_matrix_rows_match_oracle_on_a_sample(big):
    rows = numpy.random.default_rng(3).choice(N_ROWS, size=64, replace=False)
    rp, si, ob = big["row_ptr"], big["site"], big["obs"]
    sub_ptr = numpy.zeros(65, dtype=numpy.int64)
    sub_site, sub_obs = [], []
    for i, r in enumerate(rows):
        sub_site.append(si[rp[r]:rp[r + 1]])
        sub_obs.append(ob[rp[r]:rp[r + 1]])
        sub_ptr[i + 1] = sub_ptr[i] + (rp[r + 1] - rp[r])
    t = big["tables"]
    want = c_oracle.build_em_matrix(t.expected, t.lhit, t.lmiss, sub_ptr, numpy.concatenate(sub_site),
                                    numpy.concatenate(sub_obs), big["n_haps"])
    import torch
    got = big["mat"][torch.from_numpy(rows).cuda()].cpu().numpy()
    assert numpy.array_equal(got, want)


def test_colsum_conserves_total_weight_and_is_deterministic(big):
    props = numpy.random.default_rng(1).dirichlet([1.0] * big["n_haps"])
    a = _one_iter(big["plan"], props)
    b = _one_iter(big["plan"], props)
    assert numpy.array_equal(a, b)                         # no atomics: bitwise reproducible
    assert abs(a.sum() - N_ROWS) < 1e-6 * N_ROWS * 1e-3    # sum_h colsum = sum_r w_r
    assert (a >= 0).all()


def test_shard_additivity(big):
    """colsum(all rows) == colsum(first part) + colsum(second part): the multi-GPU identity."""
    from mixemt_amd import em
    props = numpy.random.default_rng(2).dirichlet([1.0] * big["n_haps"])
    full = _one_iter(big["plan"], props)
    cut = N_ROWS // 3
    lo = _one_iter(em.EmPlan(big["mat"][:cut], big["wts"][:cut]), props)
    hi = _one_iter(em.EmPlan(big["mat"][cut:], big["wts"][cut:]), props)
    assert numpy.allclose(lo + hi, full, rtol=1e-12, atol=1e-12)


def test_em_iteration_matches_oracle_on_a_slab(big):
    """One fused iteration restricted to 2000 rows against the numpy oracle's em_step."""
    from mixemt_amd import em
    props = numpy.random.default_rng(4).dirichlet([1.0] * big["n_haps"])
    sub = big["mat"][:2000]
    got = _one_iter(em.EmPlan(sub, big["wts"][:2000]), props)
    host = sub.cpu().numpy()
    _, new = em_oracle.em_step(host, numpy.ones(2000), numpy.log(props), numpy.empty_like(host))
    assert numpy.allclose(got / got.sum(), numpy.exp(new), rtol=0, atol=1e-13)


def test_full_run_recovers_the_mixture(big):
    """End to end at scale: the three contributors come back with their proportions and calls."""
    import torch
    from mixemt_amd import em
    numpy.random.seed(7)
    res = em.run_em_ex(big["mat"], big["wts"], em_args())
    assert res["done"] == [1]
    props = res["props"]
    top = numpy.argsort(props)[::-1][:3]
    assert sorted(top) == [10, 2000, 4000]
    assert numpy.allclose(props[[10, 2000, 4000]], [0.6, 0.3, 0.1], atol=0.02)
    assert abs(props.sum() - 1.0) < 1e-9
    best = res["read_mix"].argmax(dim=1).cpu().numpy()
    truth = numpy.array([10, 2000, 4000])[big["who"]]
    assert (best == truth).mean() > 0.5       # many reads carry no discriminating site
    # rerun: same iteration count and bit-identical proportions
    numpy.random.seed(7)
    again = em.run_em_ex(big["mat"], big["wts"], em_args(), want_read_mix=False)
    assert again["iters"] == res["iters"] and numpy.array_equal(again["props"], props)
