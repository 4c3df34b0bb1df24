"""
The quad dictionary beside the records (csrc/quad_kernels.hpp, mxm_build_quads; em.EmPlan.attach_quads): an acceleration
structure for the EM iteration alone -- the records stay complete and every other consumer reads them.
  * a quad record decodes to the record's own P row, bit for bit (codes thread-contiguous, 32-byte table entries);
  * the iteration with quads gives the column sums of the iteration without (3e-15), reruns bit for bit, and runs of
    run_em over goldens g9 / g10 stop at the reference's iteration with its proportions;
  * the three row lists are checked where they are used: a faulty list poisons the sums and raises the error flag;
  * odd widths (pad columns inside the last quad), a buffer that overflows once, rows without quads.
"""
import ctypes

import numpy
import pytest

from conftest import em_args, golden

pytestmark = pytest.mark.gpu


def _records(b17, n_rows, seed, read_len=150):
    from mixemt_amd import preprocess, synth
    refseq, phy, haps, tables = b17
    row_ptr, site, obs, _ = synth.synth_reads(tables, len(refseq), n_rows, seed=seed, read_len=read_len)
    return preprocess.build_em_records_device(tables, row_ptr, site, obs), (row_ptr, site, obs)


def _decode(cm):
    import torch
    from mixemt_amd import _lib
    from mixemt_amd._dev import current_stream
    out = torch.full((cm.n_rows, cm.n_haps), float("nan"), dtype=torch.float64, device=cm.rec.device)
    coded = cm.struct()
    _lib.check(_lib.load().mxm_decode_rows(ctypes.byref(coded), cm.n_haps, out.data_ptr(), out.stride(0), current_stream()),
               "mxm_decode_rows")
    return out.cpu().numpy()


def _quad_rows_decoded(plan):
    """{row: its P row rebuilt from the quad record on the host}"""
    qrec, qoff, nquad, quad_rows, byte_rows = (x.cpu().numpy() for x in plan._quad_keep)
    H = plan.n_haps
    ldc = (H + 7) // 8 * 8
    nqc = ldc // 4
    out = {}
    for r in quad_rows:
        o, n = int(qoff[r]), int(nquad[r])
        assert 1 <= n <= 256 and o % 32 == 0
        codes = qrec[o:o + 2048].reshape(256, 8)                     # [thread][j]: quad t + 256 j
        table = qrec[o + 2048:o + 2048 + 32 * n].view(numpy.float64).reshape(n, 4)
        quad_code = numpy.zeros(nqc, dtype=numpy.int64)
        for j in range(8):
            idx = numpy.arange(256) + 256 * j
            ok = idx < nqc
            quad_code[idx[ok]] = codes[ok, j]
            assert (codes[~ok, j] == 0).all()
        assert quad_code.max() < n
        out[int(r)] = table[quad_code].reshape(-1)[:H]
    return out, quad_rows, byte_rows, nquad


@pytest.mark.parametrize("n_rows,seed,read_len", [(1500, 5, 150), (900, 6, 260)])
def test_quad_records_decode_to_the_records_rows(b17, n_rows, seed, read_len):
    from mixemt_amd import em
    cm, _ = _records(b17, n_rows, seed, read_len)
    plan = em.EmPlan(None, numpy.ones(n_rows), records=cm)
    assert plan._quad_keep is None                                   # "auto": far too few rows
    assert plan.attach_quads(True)
    want = _decode(cm)
    rows, quad_rows, byte_rows, nquad = _quad_rows_decoded(plan)
    nd = cm.ndist_host()
    assert len(quad_rows) > 0.9 * ((nd > 0) & (nd <= 256)).sum()
    for r, row in rows.items():
        assert numpy.array_equal(row.view(numpy.int64), want[r].view(numpy.int64)), r
    # the lists partition the rows with a record
    byte_coded = numpy.flatnonzero((nd > 0) & (nd <= 256))
    assert numpy.array_equal(numpy.sort(numpy.concatenate([quad_rows, byte_rows])), byte_coded)
    assert (nquad[byte_rows] == 0).all() and (nquad[nd > 256] == 0).all() and (nquad[nd == 0] == 0).all()
    # a row's codes are the ranks of its distinct quads in ascending order of the quad (4 code bytes as one integer):
    # rebuilding the dictionary twice gives the same bytes apart from where the allocator put them
    first = {int(r): plan._quad_keep[0].cpu().numpy()[int(plan._quad_keep[1][r]):int(plan._quad_keep[1][r]) + 2048 + 32 * int(nquad[r])].copy()
             for r in quad_rows[:50]}
    plan._quad_keep = None
    plan.coded.qrec = None
    assert plan.attach_quads(True)
    q2, off2 = plan._quad_keep[0].cpu().numpy(), plan._quad_keep[1].cpu().numpy()
    for r, blob in first.items():
        assert numpy.array_equal(q2[int(off2[r]):int(off2[r]) + len(blob)], blob)


def _iterate(plan, props, reps=1):
    import torch
    from mixemt_amd import em
    dev = plan.dev
    p = torch.from_numpy(props[None, :]).to(dev)
    out = []
    for _ in range(reps):
        colsum = torch.zeros((1, plan.n_haps), dtype=torch.float64, device=dev)
        state = em.new_state(1, dev)
        plan.em_iter(p, p.log(), state, colsum)
        torch.cuda.synchronize()
        out.append((colsum[0].cpu().numpy(), em.read_state(state) if hasattr(em, "read_state") else None))
    return out


@pytest.mark.parametrize("n_rows,seed,read_len", [(4000, 7, 150), (2500, 8, 260), (37, 9, 150)])
def test_iteration_with_quads_equals_the_iteration_without(b17, n_rows, seed, read_len):
    from mixemt_amd import em
    from oracle import c_oracle, em_oracle
    refseq, phy, haps, tables = b17
    cm, (row_ptr, site, obs) = _records(b17, n_rows, seed, read_len)
    rng = numpy.random.default_rng(seed)
    wts = rng.integers(1, 6, size=n_rows).astype(numpy.float64)
    props = rng.dirichlet([0.5] * len(haps))
    plain = em.EmPlan(None, wts, records=cm)
    base = _iterate(plain, props)[0][0]
    quad = em.EmPlan(None, wts, records=cm)
    assert quad.attach_quads(True) and quad.coded.n_quad_rows > 0
    got = _iterate(quad, props, reps=3)
    for colsum, _ in got:
        assert numpy.array_equal(colsum, got[0][0])                                  # reruns: the same bits
    rel = numpy.abs(got[0][0] - base) / numpy.abs(base).max()
    assert rel.max() < 3e-15
    # ... and the oracle's M-step sums on the reference's matrix
    mat = c_oracle.build_em_matrix(tables.expected, tables.lhit, tables.lmiss, row_ptr, site, obs, len(haps))
    with numpy.errstate(divide="ignore"):
        post, new = em_oracle.em_step(mat, wts, numpy.log(props), numpy.empty_like(mat))
    t = got[0][0] * props                                                            # T_h p_h = sum_r w_r posterior_rh
    want = (numpy.exp(post) * wts[:, None]).sum(axis=0)
    assert numpy.abs(t - want).max() < 1e-9 * wts.sum()


@pytest.mark.parametrize("n_cols,n_rows,read_len", [(2050, 3000, 150), (2050, 600, 3000), (130, 900, 600), (1024, 2000, 150)])
def test_narrower_tables_and_no_byte_rows_left(n_cols, n_rows, read_len):
    """Sub-trees of Build 17 (two to three code words per thread instead of six): among them matrices where EVERY byte-coded
    row gets quads and only wide rows are left to the records' pass -- its row list is then EMPTY, which is not the same
    as "no list" (found by tools/stress_parity.py: the pass took every row a second time, and the in-kernel check said so)."""
    from mixemt_amd import em, phylotree, preprocess, synth
    refseq = phylotree.load_rsrs()
    phy = phylotree.load_build17(refseq)
    haps = sorted(phy.hap_var)
    sub = haps[700:700 + n_cols]
    tables = preprocess.HapVarTables.build(refseq, phy, sub)
    row_ptr, site, obs, _ = synth.synth_reads(tables, len(refseq), n_rows, seed=n_cols + n_rows, read_len=read_len,
                                              contrib=(0, n_cols // 2, n_cols - 1))
    cm = preprocess.build_em_records_device(tables, row_ptr, site, obs)
    rng = numpy.random.default_rng(n_cols)
    wts = rng.integers(1, 4, size=n_rows).astype(numpy.float64)
    props = rng.dirichlet([1.0] * n_cols)
    base = _iterate(em.EmPlan(None, wts, records=cm), props)[0][0]
    plan = em.EmPlan(None, wts, records=cm)
    if not plan.attach_quads(True):                                  # long reads: no row with at most 256 quads -- nothing attached
        assert read_len >= 3000 and plan.coded.qrec is None and plan._quad_keep is None
        assert numpy.array_equal(_iterate(plan, props)[0][0], base)
        return
    got = _iterate(plan, props, reps=2)
    assert numpy.isfinite(got[0][0]).all() and numpy.array_equal(got[0][0], got[1][0])
    assert (numpy.abs(got[0][0] - base) / numpy.abs(base).max()).max() < 3e-15


def test_an_empty_byte_list_is_still_a_list(b17):
    """Only quad rows and wide rows: the records' pass has wide rows to do and an EMPTY list of byte-coded rows -- a null
    list pointer would mean "every row" to it (tools/stress_parity.py found exactly that; the in-kernel check caught it)."""
    from mixemt_amd import em, preprocess, synth
    refseq, phy, haps, tables = b17
    n_rows = 3000
    row_ptr, site, obs, _ = synth.synth_reads(tables, len(refseq), n_rows, seed=77, read_len=260)
    cm0 = preprocess.build_em_records_device(tables, row_ptr, site, obs)
    plan0 = em.EmPlan(None, numpy.ones(n_rows), records=cm0)
    assert plan0.attach_quads(True)
    nquad = plan0._quad_keep[2].cpu().numpy()
    nd = cm0.ndist_host()
    keep = numpy.flatnonzero((nquad > 0) | (nd > 256))
    assert (nd[keep] > 256).sum() > 10 and len(keep) < n_rows
    lens = numpy.diff(row_ptr)[keep]
    new_ptr = numpy.zeros(len(keep) + 1, dtype=numpy.int64)
    numpy.cumsum(lens, out=new_ptr[1:])
    idx = numpy.concatenate([numpy.arange(row_ptr[r], row_ptr[r + 1]) for r in keep])
    cm = preprocess.build_em_records_device(tables, new_ptr, site[idx], obs[idx])
    rng = numpy.random.default_rng(5)
    wts = rng.integers(1, 4, size=len(keep)).astype(numpy.float64)
    props = rng.dirichlet([1.0] * len(haps))
    base = _iterate(em.EmPlan(None, wts, records=cm), props)[0][0]
    plan = em.EmPlan(None, wts, records=cm)
    assert plan.attach_quads(True)
    assert int(plan.coded.n_byte_rows) == 0 and int(plan.coded_wide) > 10 and not plan.coded.byte_rows
    got = _iterate(plan, props, reps=2)
    assert numpy.isfinite(got[0][0]).all() and numpy.array_equal(got[0][0], got[1][0])
    assert (numpy.abs(got[0][0] - base) / numpy.abs(base).max()).max() < 3e-15


@pytest.mark.parametrize("name,seed", [("g9_run_em_2400", 17), ("g10_run_em_20k", 23)])
def test_run_em_with_quads_reproduces_the_reference(b17, name, seed, monkeypatch):
    """Goldens through records + quads (the per-iteration kernels and the batched loop around them): the reference's
    inits, iteration counts and proportions."""
    from mixemt_amd import em, preprocess
    refseq, phy, haps, tables = b17
    g = golden(name)
    if name.startswith("g10"):
        from test_gpu_g10 import _inputs
        row_ptr, site, obs, wts = _inputs(tables, len(refseq), g)
    else:
        row_ptr, site, obs, wts = g["row_ptr"], g["site"], g["obs"], g["wts"]
    cm = preprocess.build_em_records_device(tables, row_ptr, site, obs)
    monkeypatch.setattr(em, "QUADS", True)
    made = []
    real = em.EmPlan.attach_quads

    def spy(self, mode=None, cap=None):
        ok = real(self, mode, cap)
        made.append((ok, int(self.coded.n_quad_rows), int(self.coded.n_byte_rows)))
        return ok

    monkeypatch.setattr(em.EmPlan, "attach_quads", spy)
    from mixemt_amd import _lib
    lib = _lib.load()
    lib.mxm_set_loop_fused(0, 0)                         # (at this size the one-launch loop, which reads the records, would run)
    try:
        numpy.random.seed(seed)
        res = em.run_em_ex(None, wts, em_args(), want_read_mix=False, records=cm)
    finally:
        lib.mxm_reset_tuning()
    assert made and made[0][0] is True and made[0][1] > 0.8 * len(wts)
    assert numpy.array_equal(res["inits"], g["inits"])
    assert res["iters"] == list(g["iters"]) and res["storage"] == "coded"
    assert numpy.abs(res["props"] - g["props"]).max() < 1e-12


def test_three_restarts_with_quads_reproduce_the_reference(b17, monkeypatch):
    """g5 (three restarts, the fold over runs, em.py:145-165) through records + quads: what `"auto"` attaches for several
    restarts over many rows, here forced at the golden's size, with the per-iteration kernels that serve it."""
    from mixemt_amd import _lib, em, preprocess
    refseq, phy, haps, tables = b17
    g = golden("g5_run_em_multi")
    cm = preprocess.build_em_records_device(tables, g["row_ptr"], g["site"], g["obs"])
    monkeypatch.setattr(em, "QUADS", True)
    lib = _lib.load()
    lib.mxm_set_loop_fused(0, 0)
    try:
        numpy.random.seed(11)
        res = em.run_em_ex(None, g["wts"], em_args(n_multi=3), records=cm)
    finally:
        lib.mxm_reset_tuning()
    assert numpy.array_equal(res["inits"], g["inits"])
    assert res["iters"] == list(g["iters"]) and numpy.abs(res["props"] - g["props"]).max() < 1e-12
    mix = res["read_mix"].cpu().numpy()
    assert numpy.array_equal(mix.argmax(axis=1), g["mix_argmax"])
    assert numpy.allclose(mix[:16], g["mix_rows"], rtol=0, atol=1e-8)


def test_faulty_lists_poison_the_sums(b17):
    import torch
    from mixemt_amd import em
    refseq, phy, haps, tables = b17
    n_rows = 3000
    cm, _ = _records(b17, n_rows, 11, 260)
    props = numpy.random.default_rng(1).dirichlet([1.0] * len(haps))
    nd = cm.ndist_host()
    assert (nd > 256).sum() > 5

    def run(mutate):
        plan = em.EmPlan(None, numpy.ones(n_rows), records=cm)
        assert plan.attach_quads(True)
        qrec, qoff, nquad, quad_rows, byte_rows = plan._quad_keep
        keep = mutate(plan, quad_rows, byte_rows)
        dev = plan.dev
        p = torch.from_numpy(props[None, :]).to(dev)
        colsum = torch.zeros((1, plan.n_haps), dtype=torch.float64, device=dev)
        state = em.new_state(1, dev)
        try:
            plan.em_iter(p, p.log(), state, colsum)
        except Exception as exc:                                     # (refused on the host: counts do not add up)
            return "refused: %s" % exc, None
        torch.cuda.synchronize()
        del keep
        return colsum[0].cpu().numpy(), state.cpu().numpy() if hasattr(state, "cpu") else state

    good, _ = run(lambda plan, q, b: None)
    assert numpy.isfinite(good).all()

    def swap_two(plan, q, b):                                        # not ascending
        bad = q.clone()
        bad[[3, 4]] = bad[[4, 3]]
        plan.coded.quad_rows = bad.data_ptr()
        return bad

    def wide_in_quads(plan, q, b):                                   # a row without quads among the quad rows
        bad = q.clone()
        wide = int(numpy.flatnonzero(nd > 256)[-1])
        bad[-1] = max(wide, int(bad[-2]) + 1) if wide > int(bad[-2]) else int(bad[-1])
        if int(bad[-1]) == int(q[-1]):
            bad[0] = -5                                              # (fallback: out of range)
        plan.coded.quad_rows = bad.data_ptr()
        return bad

    def quad_row_in_bytes(plan, q, b):                               # a row the quad pass takes listed for the byte pass too
        if b.numel() == 0:
            return None
        bad = b.clone()
        qn = q.cpu().numpy()
        bn = b.cpu().numpy()
        cand = qn[(qn > bn[0])]
        bad[0] = int(cand[0]) if len(cand) and (b.numel() == 1 or cand[0] < bn[1]) else int(qn[0])
        plan.coded.byte_rows = bad.data_ptr()
        return bad

    def short_count(plan, q, b):                                     # counts that do not add up: refused on the host
        plan.coded.n_quad_rows = plan.coded.n_quad_rows - 1
        return None

    for mutate in (swap_two, wide_in_quads, quad_row_in_bytes):
        got, state = run(mutate)
        assert not isinstance(got, str), got
        assert numpy.isnan(got).all(), mutate.__name__
    got, _ = run(short_count)
    assert isinstance(got, str) and "must be all" in got


def test_odd_width_and_a_buffer_that_overflows_once(monkeypatch):
    """H = 5409: the last quad holds pad columns; a first buffer that is too small is replaced by one of the counted size."""
    import torch
    from mixemt_amd import _dev, em, phylotree, preprocess, synth
    refseq = phylotree.load_rsrs()
    phy = phylotree.load_build17(refseq)
    phy.add_custom_hap("zz_custom", ["A73G", "C150T", "T16189C", "G8994A"])
    haps = sorted(phy.hap_var)
    tables = preprocess.HapVarTables.build(refseq, phy, haps)
    n_rows = 1200
    row_ptr, site, obs, _ = synth.synth_reads(tables, len(refseq), n_rows, seed=31, contrib=(10, 2000, haps.index("zz_custom")))
    cm = preprocess.build_em_records_device(tables, row_ptr, site, obs)
    wts = numpy.ones(n_rows)
    props = numpy.random.default_rng(4).dirichlet([1.0] * len(haps))
    base = _iterate(em.EmPlan(None, wts, records=cm), props)[0][0]
    sizes = []
    real = em.device_empty

    def watch(shape, dtype, dev, what):
        if what == "the quad dictionary":
            sizes.append(shape[0])
        return real(shape, dtype, dev, what)

    monkeypatch.setattr(em, "device_empty", watch)
    plan = em.EmPlan(None, wts, records=cm)
    assert plan.attach_quads(True, cap=200000)                       # room for ~40 of 1200 rows
    assert len(sizes) == 2 and sizes[0] == 200000 and sizes[1] > 20 * sizes[0]
    assert plan.coded.n_quad_rows > 1000
    got = _iterate(plan, props)[0][0]
    assert numpy.abs(got - base).max() / numpy.abs(base).max() < 3e-15
    rows, quad_rows, byte_rows, nquad = _quad_rows_decoded(plan)
    want = _decode(cm)
    for r in list(rows)[:200]:
        assert numpy.array_equal(rows[r].view(numpy.int64), want[r].view(numpy.int64))


def test_the_bare_reader_of_the_quad_records(b17):
    """mxm_diag_stream_quads (the counter calibration's reader, tools/pmc_calibrate_coded.py --quads): runs over a plan's
    quad records and leaves everything as it was; without a quad dictionary it says so instead of reading."""
    import torch
    from mixemt_amd import _lib, em
    from mixemt_amd._dev import current_stream
    lib = _lib.load()
    cm, _ = _records(b17, 3000, 21)
    plan = em.EmPlan(None, numpy.ones(3000), records=cm)
    sink = torch.zeros(4, dtype=torch.int32, device=plan.dev)
    assert lib.mxm_diag_stream_quads(ctypes.byref(plan.coded), plan.n_haps, 2, sink.data_ptr(), current_stream()) == -1
    assert b"quad dictionary" in lib.mxm_last_error()
    assert plan.attach_quads(True)
    props = numpy.random.default_rng(1).dirichlet([0.5] * plan.n_haps)
    before = _iterate(plan, props)[0][0]
    qrec = plan._quad_keep[0].clone()
    for wg in (1, 2, 4):
        _lib.check(lib.mxm_diag_stream_quads(ctypes.byref(plan.coded), plan.n_haps, wg, sink.data_ptr(), current_stream()),
                   "mxm_diag_stream_quads")
    torch.cuda.synchronize()
    assert torch.equal(qrec, plan._quad_keep[0]) and int(sink.abs().sum()) == 0
    assert numpy.array_equal(_iterate(plan, props)[0][0], before)
    assert lib.mxm_diag_stream_quads(ctypes.byref(plan.coded), plan.n_haps, 0, sink.data_ptr(), current_stream()) == -1


def _quad_blobs(plan):
    qrec, qoff, nquad = (x.cpu().numpy() for x in plan._quad_keep[:3])
    return nquad, {int(r): qrec[int(qoff[r]):int(qoff[r]) + 2048 + 32 * int(nquad[r])].copy() for r in numpy.flatnonzero(nquad > 0)}


@pytest.mark.parametrize("n_cols,n_rows,read_len", [(None, 2600, 150), (None, 700, 260), (2050, 900, 150), (130, 300, 600), (1021, 5, 150)])
def test_the_two_encoders_write_the_same_records(b17, n_cols, n_rows, read_len):
    """mxm_set_quad_encoder: a wave per row (default, round 6) against a workgroup per row -- nquad for every row and every
    quad record byte for byte (codes = ranks by value: nothing depends on which lane won a hash slot); full width, narrow
    and odd tables (rounds past the row's last quad, pad columns inside the last quad), fewer rows than waves."""
    from mixemt_amd import _lib, em, phylotree, preprocess, synth
    refseq, phy, haps, tables = b17
    if n_cols is not None:
        tables = preprocess.HapVarTables.build(refseq, phy, haps[400:400 + n_cols])
    width = len(haps) if n_cols is None else n_cols
    row_ptr, site, obs, _ = synth.synth_reads(tables, len(refseq), n_rows, seed=n_rows, read_len=read_len,
                                              contrib=(0, width // 2, width - 1))
    cm = preprocess.build_em_records_device(tables, row_ptr, site, obs)
    lib = _lib.load()
    got = {}
    try:
        for kind in (0, 1):
            _lib.check(lib.mxm_set_quad_encoder(kind), "mxm_set_quad_encoder")
            plan = em.EmPlan(None, numpy.ones(n_rows), records=cm)
            attached = plan.attach_quads(True)
            got[kind] = _quad_blobs(plan) if attached else (None, {})
    finally:
        lib.mxm_set_quad_encoder(1)
    assert lib.mxm_set_quad_encoder(2) != 0
    assert (got[0][0] is None) == (got[1][0] is None)
    if got[0][0] is not None:
        assert numpy.array_equal(got[0][0], got[1][0])
        assert len(got[0][1]) > 0 and got[0][1].keys() == got[1][1].keys()
        for r, blob in got[0][1].items():
            assert numpy.array_equal(blob, got[1][1][r]), r
