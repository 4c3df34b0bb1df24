"""
Three restarts through ONE pass over the records beside a quad dictionary (csrc/quad_batched_kernels.hpp,
em_iter_quad_batched_kernel; mxm_em_iter_coded / mxm_em_loop_coded take full tiles of mxm_restart_tile_coded() = 3).
The reference runs its restarts one after another over the same matrix (em.py:117-161); here they share the row loads,
the tables in LDS and the lookups:
  * every restart's column sums equal the one-restart-per-pass kernel's (another summation order: 3e-15), for tiles and
    remainders (3, 4, 7 restarts), reruns bit for bit, against the oracle's M-step sums;
  * all three classes of rows in one kernel: quad rows, byte-coded rows without quads, wide rows -- including matrices
    where a class is empty, narrow tables (one or two code bytes per thread) and an odd width;
  * faulty row lists poison the sums of every restart of the tile;
  * restarts that stop on different iterations (g5: the reference's three runs), a tile with a finished member.
"""
import ctypes

import numpy
import pytest

from conftest import em_args, golden

pytestmark = pytest.mark.gpu


def _records(tables, ref_len, n_rows, seed, read_len=150, contrib=None):
    from mixemt_amd import preprocess, synth
    kw = {} if contrib is None else {"contrib": contrib}
    row_ptr, site, obs, _ = synth.synth_reads(tables, ref_len, n_rows, seed=seed, read_len=read_len, **kw)
    return preprocess.build_em_records_device(tables, row_ptr, site, obs), (row_ptr, site, obs)


def _iterate(plan, props, tile):
    """One mxm_em_iter_coded over all restarts of `props` ([B][H]) with `tile` restarts per pass (3 or 1)."""
    import torch
    from mixemt_amd import em
    lib = plan.lib
    assert lib.mxm_set_coded_batch_tile(tile) == 0
    try:
        p = torch.from_numpy(numpy.ascontiguousarray(props)).to(plan.dev)
        colsum = torch.zeros_like(p)
        state = em.new_state(p.shape[0], plan.dev)
        plan.em_iter(p, p.log(), state, colsum)
        torch.cuda.synchronize()
        em.read_state(state)                                         # raises if a list check fired
        return colsum.cpu().numpy()
    finally:
        lib.mxm_reset_tuning()


def _plan(cm, wts, quads=True, n_runs=8):
    from mixemt_amd import em
    plan = em.EmPlan(None, wts, n_runs=n_runs, records=cm)
    if quads:
        assert plan.attach_quads(True) and plan.coded.n_quad_rows > 0
    return plan


@pytest.mark.parametrize("n_rows,seed,read_len,n_runs", [(4000, 7, 150, 3), (2500, 8, 260, 4), (37, 9, 150, 7), (1200, 10, 400, 6)])
def test_a_tile_of_three_equals_three_passes(b17, n_rows, seed, read_len, n_runs):
    from oracle import c_oracle, em_oracle
    refseq, phy, haps, tables = b17
    cm, (row_ptr, site, obs) = _records(tables, len(refseq), n_rows, seed, read_len)
    rng = numpy.random.default_rng(seed)
    wts = rng.integers(1, 6, size=n_rows).astype(numpy.float64)
    props = rng.dirichlet([0.5] * len(haps), size=n_runs)
    plan = _plan(cm, wts)
    assert plan.restart_tile() == 3
    nd = cm.ndist_host()
    if read_len >= 260:
        assert (nd > 256).sum() > 5 and plan.coded.n_byte_rows >= 0      # wide rows in the third class
    one = _iterate(plan, props, 1)
    got = _iterate(plan, props, 3)
    again = _iterate(plan, props, 3)
    assert numpy.array_equal(got, again)                                 # reruns: the same bits
    for b in range(n_runs):
        rel = numpy.abs(got[b] - one[b]) / numpy.abs(one[b]).max()
        assert rel.max() < 3e-15, (b, rel.max())
    # the remainder (B mod 3) went through the one-restart kernel: identical bits there
    for b in range(n_runs - n_runs % 3, n_runs):
        assert numpy.array_equal(got[b], one[b])
    # ... and the oracle's M-step sums on the reference's matrix, restart by restart
    mat = c_oracle.build_em_matrix(tables.expected, tables.lhit, tables.lmiss, row_ptr, site, obs, len(haps))
    for b in range(min(n_runs, 3)):
        with numpy.errstate(divide="ignore"):
            post, _ = em_oracle.em_step(mat, wts, numpy.log(props[b]), numpy.empty_like(mat))
        want = (numpy.exp(post) * wts[:, None]).sum(axis=0)
        assert numpy.abs(got[b] * props[b] - want).max() < 1e-9 * wts.sum()


def test_without_a_quad_dictionary_every_restart_has_its_own_pass(b17):
    refseq, phy, haps, tables = b17
    cm, _ = _records(tables, len(refseq), 900, 3)
    plan = _plan(cm, numpy.ones(900), quads=False)
    assert plan.restart_tile() == 1
    props = numpy.random.default_rng(2).dirichlet([1.0] * len(haps), size=3)
    assert numpy.array_equal(_iterate(plan, props, 3), _iterate(plan, props, 1))
    lib = plan.lib
    assert lib.mxm_set_coded_batch_tile(2) < 0 and lib.mxm_set_coded_batch_tile(0) < 0
    assert plan.attach_quads(True) and plan.restart_tile() == 3
    assert lib.mxm_set_coded_batch_tile(1) == 0
    try:
        assert plan.restart_tile() == 1
    finally:
        lib.mxm_reset_tuning()


@pytest.mark.parametrize("n_cols,n_rows,read_len", [(2050, 3000, 150), (130, 900, 600), (1024, 2000, 150), (700, 1500, 260)])
def test_narrower_tables(n_cols, n_rows, read_len):
    """Sub-trees of Build 17: one or two code bytes per thread of 512, matrices whose byte-coded rows all have quads."""
    from mixemt_amd import phylotree, preprocess
    refseq = phylotree.load_rsrs()
    phy = phylotree.load_build17(refseq)
    haps = sorted(phy.hap_var)
    sub = haps[700:700 + n_cols]
    tables = preprocess.HapVarTables.build(refseq, phy, sub)
    cm, _ = _records(tables, len(refseq), n_rows, n_cols + n_rows, read_len, contrib=(0, n_cols // 2, n_cols - 1))
    rng = numpy.random.default_rng(n_cols)
    wts = rng.integers(1, 4, size=n_rows).astype(numpy.float64)
    props = rng.dirichlet([1.0] * n_cols, size=3)
    plan = _plan(cm, wts)
    one = _iterate(plan, props, 1)
    got = _iterate(plan, props, 3)
    assert numpy.isfinite(got).all()
    for b in range(3):
        assert (numpy.abs(got[b] - one[b]) / numpy.abs(one[b]).max()).max() < 3e-15


def test_odd_width(monkeypatch):
    from mixemt_amd import phylotree, preprocess
    refseq = phylotree.load_rsrs()
    phy = phylotree.load_build17(refseq)
    phy.add_custom_hap("zz_custom", ["A73G", "C150T", "T16189C", "G8994A"])
    haps = sorted(phy.hap_var)
    tables = preprocess.HapVarTables.build(refseq, phy, haps)
    assert len(haps) % 2 == 1
    cm, _ = _records(tables, len(refseq), 1200, 31, contrib=(10, 2000, haps.index("zz_custom")))
    props = numpy.random.default_rng(4).dirichlet([1.0] * len(haps), size=3)
    plan = _plan(cm, numpy.ones(1200))
    one = _iterate(plan, props, 1)
    got = _iterate(plan, props, 3)
    for b in range(3):
        assert (numpy.abs(got[b] - one[b]) / numpy.abs(one[b]).max()).max() < 3e-15


def test_faulty_lists_poison_every_restart_of_the_tile(b17):
    import torch
    from mixemt_amd import em
    refseq, phy, haps, tables = b17
    n_rows = 3000
    cm, _ = _records(tables, len(refseq), n_rows, 11, 260)
    props = numpy.random.default_rng(1).dirichlet([1.0] * len(haps), size=3)
    nd = cm.ndist_host()
    assert (nd > 256).sum() > 5

    def run(mutate):
        plan = _plan(cm, numpy.ones(n_rows))
        qrec, qoff, nquad, quad_rows, byte_rows = plan._quad_keep
        wide = plan._coded_keep[-1]
        keep = mutate(plan, quad_rows, byte_rows, wide)
        p = torch.from_numpy(props).to(plan.dev)
        colsum = torch.zeros_like(p)
        state = em.new_state(3, plan.dev)
        plan.em_iter(p, p.log(), state, colsum)
        torch.cuda.synchronize()
        del keep
        with pytest.raises(ValueError) if mutate is not None and mutate.__name__ != "nothing" else _nullcontext():
            em.read_state(state)
        return colsum.cpu().numpy()

    def nothing(plan, q, b, wd):
        return None

    def swap_two(plan, q, b, wd):                                    # not ascending
        bad = q.clone()
        bad[[3, 4]] = bad[[4, 3]]
        plan.coded.quad_rows = bad.data_ptr()
        return bad

    def quad_row_in_bytes(plan, q, b, wd):                           # a row the first class takes listed for the second too
        assert b.numel() > 0
        bad = b.clone()
        qn, bn = q.cpu().numpy(), b.cpu().numpy()
        cand = qn[qn > bn[0]]
        bad[0] = int(cand[0]) if len(cand) and (b.numel() == 1 or cand[0] < bn[1]) else int(qn[0])
        plan.coded.byte_rows = bad.data_ptr()
        return bad

    def byte_row_among_the_wide(plan, q, b, wd):                     # a byte-coded row must not be read as 16-bit codes
        bad = wd.clone()
        qn = q.cpu().numpy()
        bad[0] = int(qn[qn < int(wd[1])][-1]) if wd.numel() > 1 else int(qn[0])
        plan.coded.wide_rows = bad.data_ptr()
        return bad

    def out_of_range(plan, q, b, wd):
        bad = q.clone()
        bad[-1] = n_rows + 5
        plan.coded.quad_rows = bad.data_ptr()
        return bad

    good = run(nothing)
    assert numpy.isfinite(good).all()
    for mutate in (swap_two, quad_row_in_bytes, byte_row_among_the_wide, out_of_range):
        got = run(mutate)
        assert numpy.isnan(got).all(), mutate.__name__


class _nullcontext(object):
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


def test_three_restarts_of_the_reference_through_shared_passes(b17, monkeypatch):
    """g5 (the reference's three runs: they stop on different iterations, so tiles shrink to single passes on the way) and
    the same with one restart per pass: the same iteration counts, proportions within 1e-13 of each other."""
    from mixemt_amd import _lib, em, preprocess
    refseq, phy, haps, tables = b17
    g = golden("g5_run_em_multi")
    cm = preprocess.build_em_records_device(tables, g["row_ptr"], g["site"], g["obs"])
    monkeypatch.setattr(em, "QUADS", True)
    lib = _lib.load()
    out = {}
    for tile in (3, 1):
        lib.mxm_set_loop_fused(0, 0)                     # (at this size the one-launch loop over the records would run)
        lib.mxm_set_coded_batch_tile(tile)
        try:
            numpy.random.seed(11)
            out[tile] = em.run_em_ex(None, g["wts"], em_args(n_multi=3), records=cm)
        finally:
            lib.mxm_reset_tuning()
    res = out[3]
    assert numpy.array_equal(res["inits"], g["inits"])
    assert res["iters"] == list(g["iters"]) == out[1]["iters"]
    assert numpy.abs(res["props"] - g["props"]).max() < 1e-12
    assert numpy.abs(res["props"] - out[1]["props"]).max() < 1e-13
    mix = res["read_mix"].cpu().numpy()
    assert numpy.array_equal(mix.argmax(axis=1), g["mix_argmax"])


def test_seven_restarts_to_convergence_with_every_schedule(b17):
    """mxm_em_loop_coded's three restart schedules around tiles of three (7 restarts: 3 + 3 + 1, tiles that lose members):
    the same iteration counts and proportions as one restart per pass."""
    from mixemt_amd import _lib, em
    refseq, phy, haps, tables = b17
    n_rows = 2000
    cm, _ = _records(tables, len(refseq), n_rows, 41)
    wts = numpy.ones(n_rows)
    lib = _lib.load()
    rng = numpy.random.default_rng(3)
    inits = rng.dirichlet([1.0] * len(haps), size=7)
    ref = None
    for tile, sched in ((1, 2), (3, 0), (3, 1), (3, 2)):
        plan = _plan(cm, wts)
        lib.mxm_set_loop_fused(0, 0)
        lib.mxm_set_coded_batch_tile(tile)
        lib.mxm_set_compact_restarts(sched)
        try:
            ln_cur, ln_new, states = em.em_loop(plan, inits, 2e-3, 600, check_every=5)
        finally:
            lib.mxm_reset_tuning()
        got = (ln_new.cpu().numpy(), [s[1] for s in states], [s[0] for s in states])
        if ref is None:
            ref = got
            assert len(set(got[1])) > 1                              # they do stop on different iterations
            continue
        assert got[1] == ref[1] and got[2] == ref[2], (tile, sched)
        assert numpy.abs(numpy.exp(got[0]) - numpy.exp(ref[0])).max() < 1e-12


def test_several_restarts_take_the_shared_passes_from_twenty_thousand_rows(b17, monkeypatch):
    """`"auto"` (profiles/r06/multi_restart_routes.txt): with three or more restarts the quad dictionary is attached from 2*10^4
    byte-coded rows and mxm_em_loop_coded leaves the one-launch loop (restarts one after another) for tiles of three --
    45 against 55 us per restart-iteration at that size; with one restart the plan stays without quads.  Same iteration
    counts and proportions as the one-launch loop."""
    from mixemt_amd import _lib, em
    refseq, phy, haps, tables = b17
    n_rows = 24000
    cm, _ = _records(tables, len(refseq), n_rows, 51)
    wts = numpy.ones(n_rows)
    lib = _lib.load()
    assert lib.mxm_quad_loop_min_rows(1) == 300000 and lib.mxm_quad_loop_min_rows(3) < 20000
    made = []
    real = em.EmPlan.attach_quads

    def spy(self, mode=None, cap=None, min_rows=None):
        ok = real(self, mode, cap, min_rows)
        made.append((self.n_runs, bool(ok)))
        return ok

    monkeypatch.setattr(em.EmPlan, "attach_quads", spy)
    monkeypatch.setattr(em, "QUADS", "auto")
    args3 = em_args(n_multi=4, max_iter=40)
    numpy.random.seed(5)
    shared = em.run_em_ex(None, wts, args3, want_read_mix=False, records=cm)
    assert made and made[-1] == (4, True)
    em.QUADS = False                                                 # the one-launch loop over the records alone
    numpy.random.seed(5)
    plain = em.run_em_ex(None, wts, args3, want_read_mix=False, records=cm)
    em.QUADS = "auto"
    assert shared["iters"] == plain["iters"] == [40, 40, 40, 40]
    assert numpy.abs(shared["run_props"] - plain["run_props"]).max() < 1e-12
    numpy.random.seed(5)
    one = em.run_em_ex(None, wts, em_args(n_multi=1, max_iter=10), want_read_mix=False, records=cm)
    assert made[-1] == (1, False) and one["iters"] == [10]
