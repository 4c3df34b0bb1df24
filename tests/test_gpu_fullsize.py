"""
The BASELINE configurations at FULL size on the device (BASELINE.json configs 2-3: 10^6 synth-v1
reads x 5408 haplogroups, one MI355X).  The CPU oracle cannot run at this size, so the checks are
(i) the oracle on sampled rows / row slabs of the very same device buffers, (ii) the batched
kernels against the one-restart-per-pass schedule on the whole matrix, (iii) size-independent
identities (mass conservation, votes adding up, bitwise determinism).

MXM_FULL_ROWS lowers the row count for a quick run on a smaller card; the default IS the
configuration the metric is quoted on.
"""
import os

import numpy
import pytest

from oracle import c_oracle, em_oracle

pytestmark = pytest.mark.gpu

FULL_ROWS = int(os.environ.get("MXM_FULL_ROWS", "1000000"))
N_RESTARTS = 10            # config 3
N_ITERS = 20


@pytest.fixture(scope="module")
def full(b17):
    import torch
    from mixemt_amd import em, preprocess, synth
    refseq, phy, haps, tables = b17
    need = 3.3 * FULL_ROWS * len(haps) * 8
    if torch.cuda.mem_get_info()[0] < need:
        pytest.skip("needs %.0f GB of free HBM" % (need / 1e9))
    row_ptr, site, obs, who = synth.synth_rows(tables, len(refseq), 0, FULL_ROWS, seed=1)
    mat = preprocess.build_em_matrix_device(tables, row_ptr, site, obs)
    wts = torch.ones(FULL_ROWS, dtype=torch.float64, device="cuda")
    plan = em.EmPlan(mat, wts, n_runs=N_RESTARTS)
    numpy.random.seed(7)
    inits = numpy.stack([em.init_props(len(haps), 1.0) for _ in range(N_RESTARTS)])   # sequential draws
    held = dict(tables=tables, row_ptr=row_ptr, site=site, obs=obs, who=who, mat=mat, wts=wts, plan=plan,
                inits=inits, n_haps=len(haps))
    yield held
    # 87 GB back to the device before the next module (configs 4 and 5 at size) starts its own processes
    held.clear()
    del mat, wts, plan
    import gc
    gc.collect()
    torch.cuda.empty_cache()


def _sample_csr(row_ptr, site, obs, rows):
    sub_ptr = numpy.zeros(len(rows) + 1, dtype=numpy.int64)
    sub_site, sub_obs = [], []
    for i, r in enumerate(rows):
        sub_site.append(site[row_ptr[r]:row_ptr[r + 1]])
        sub_obs.append(obs[row_ptr[r]:row_ptr[r + 1]])
        sub_ptr[i + 1] = sub_ptr[i] + (row_ptr[r + 1] - row_ptr[r])
    return sub_ptr, numpy.concatenate(sub_site), numpy.concatenate(sub_obs)


def test_build_is_bit_exact_on_256_sampled_rows(full):
    """build_em_matrix at 10^6 rows: 256 rows drawn over the whole matrix equal the C oracle bit for bit."""
    import torch
    rows = numpy.sort(numpy.random.default_rng(31).choice(FULL_ROWS, size=256, replace=False))
    rows[0], rows[-1] = 0, FULL_ROWS - 1                      # the ends of the grid included
    sub_ptr, sub_site, sub_obs = _sample_csr(full["row_ptr"], full["site"], full["obs"], rows)
    t = full["tables"]
    want = c_oracle.build_em_matrix(t.expected, t.lhit, t.lmiss, sub_ptr, sub_site, sub_obs, full["n_haps"])
    got = full["mat"][torch.from_numpy(rows).cuda()].cpu().numpy()
    assert numpy.array_equal(got, want)


def test_config3_ten_batched_restarts_match_one_per_pass(full):
    """
    Config 3 (em.py:117-161 with n_multi = 10): twenty iterations of all ten restarts with four
    restarts sharing each pass over the matrix, against the same twenty iterations one restart per
    pass; then each restart's next M-step sums on a 2 000-row slab against the oracle's em_step.
    """
    import torch
    from mixemt_amd import _lib, em
    lib = _lib.load()
    plan, inits = full["plan"], full["inits"]
    try:
        lib.mxm_set_batch_tile(4)
        cur4, new4, st4 = em.em_loop(plan, inits, 0.0, N_ITERS)
        lib.mxm_set_batch_tile(1)
        cur1, new1, st1 = em.em_loop(plan, inits, 0.0, N_ITERS)
    finally:
        lib.mxm_set_batch_tile(4)
    assert [s[:2] for s in st4] == [(2, N_ITERS)] * N_RESTARTS == [s[:2] for s in st1]
    p4, p1 = torch.exp(new4).cpu().numpy(), torch.exp(new1).cpu().numpy()
    assert numpy.abs(p4 - p1).max() <= 1e-13
    assert numpy.abs(p4.sum(axis=1) - 1.0).max() < 1e-12
    assert numpy.abs(numpy.array([s[2] for s in st4]) - numpy.array([s[2] for s in st1])).max() < 1e-12
    # restarts really differ (each has its own proportion vector in the batched kernel)
    assert numpy.abs(p4[0] - p4[1]).max() > 1e-6

    # mass conservation over the whole matrix, every restart: sum_h p_h T_h = sum_r w_r
    props = torch.exp(cur4)
    colsum = torch.zeros_like(props)
    state = em.new_state(N_RESTARTS, props.device)
    plan.em_iter(props, cur4, state, colsum)
    mass = (props * colsum).sum(dim=1).cpu().numpy()
    assert numpy.abs(mass - FULL_ROWS).max() < 1e-6 * FULL_ROWS * 1e-3

    # the same batched kernel on a slab the oracle can follow: rows [500000, 502000)
    lo = min(500000, FULL_ROWS - 2000)
    slab = full["mat"][lo:lo + 2000]
    sub = em.EmPlan(slab, full["wts"][lo:lo + 2000], n_runs=N_RESTARTS)
    sub_cs = torch.zeros_like(props)
    sub.em_iter(props, cur4, em.new_state(N_RESTARTS, props.device), sub_cs)
    got = (props * sub_cs).cpu().numpy()
    host = slab.cpu().numpy()
    ln_cur = cur4.cpu().numpy()
    for b in range(N_RESTARTS):
        _, new = em_oracle.em_step(host, numpy.ones(2000), ln_cur[b], numpy.empty_like(host))
        assert numpy.allclose(got[b] / got[b].sum(), numpy.exp(new), rtol=0, atol=1e-13), b
        assert abs(got[b].sum() - 2000.0) < 1e-8


def test_posterior_argmax_and_votes_at_full_size(full):
    """
    Row f-1 at 10^6 rows (assemble.py:115-123): the posterior of one restart, its row argmax against
    numpy.argmax on sampled rows, the weighted votes against a count of the device's own calls, and
    bit-identical votes for fractional weights on a rerun (no float atomics).
    """
    import torch
    from mixemt_amd import _lib, assign, em
    plan = full["plan"]
    ln_theta = numpy.log(full["inits"][0])
    post = em.posterior(plan, ln_theta)                        # [R][H] log posterior under the init draw
    best, votes = assign.row_argmax_votes(post, full["wts"])
    assert votes.sum() == FULL_ROWS
    assert numpy.array_equal(votes, numpy.bincount(best, minlength=full["n_haps"]).astype(numpy.float64))
    rows = numpy.sort(numpy.random.default_rng(32).choice(FULL_ROWS, size=512, replace=False))
    sample = post[torch.from_numpy(rows).cuda()].cpu().numpy()
    assert numpy.array_equal(best[rows], sample.argmax(axis=1))
    # and the sampled posterior rows themselves against the oracle's E-step on the same input rows
    host_rows = full["mat"][torch.from_numpy(rows[:64]).cuda()].cpu().numpy()
    want, _ = em_oracle.em_step(host_rows, numpy.ones(64), ln_theta, numpy.empty_like(host_rows))
    assert numpy.allclose(sample[:64], want, rtol=0, atol=1e-9)
    frac = torch.from_numpy(numpy.random.default_rng(33).random(FULL_ROWS)).cuda()
    _, v1 = assign.row_argmax_votes(post, frac)
    _, v2 = assign.row_argmax_votes(post, frac)
    assert numpy.array_equal(v1, v2)
    assert abs(v1.sum() - float(frac.sum().item())) < 1e-6
