"""
BASELINE configs 4 and 5 and the default route of config 3's matrix at their WHOLE size on one MI355X -- possible
because the matrix build leaves rows as row-dictionary records (6.4 KB per row instead of 43 KB):

  config 4   10^7 reads x 5408: build_em_records_device -> run_em_ex(records=...) on ONE rank (64 GB resident), then the
             same rows as FOUR ranks (2.5 * 10^6 rows each, gloo: the box has one GPU and RCCL refuses two ranks on a
             device) through dist.run_em_sharded -- the all-reduced sums and the stop state must be the one-rank run's
  config 5   64 restarts on the replicated 10^6 x 5408 records through dist.run_em_restart_parallel: one rank x 64
             and four ranks x 16, the 64 sequential init draws dealt by global run index
  default    10^6 x 5408 through run_em's own choice (storage="auto" -> records, the one-launch loop) to CONVERGENCE
             against the dense per-iteration loop: same stopping iteration, proportions, haplogroup calls

The oracle cannot run at these sizes; it checks sampled rows (matrix build, bit-exact) and row slabs (em_step) of the
very device buffers, next to size-independent identities (mass conservation, bitwise reruns, shard additivity).
Reference loop being exercised: em.py:94-165.  Every leg runs in processes of its own (spawned), so that its memory
is given back when it ends.  MXM_CFG4_ROWS / MXM_FULL_ROWS scale the cases down for a smaller card.
"""
import os
import socket

import numpy
import pytest

pytestmark = pytest.mark.gpu

CFG4_ROWS = int(os.environ.get("MXM_CFG4_ROWS", "10000000"))
FULL_ROWS = int(os.environ.get("MXM_FULL_ROWS", "1000000"))
CFG4_ITERS = 40
CFG4_RANKS = 4
CFG5_RESTARTS = 64
CFG5_ITERS = 30
CFG5_RANKS = 4


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _paths():
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (here, os.path.dirname(here)):
        if p not in sys.path:
            sys.path.insert(0, p)


def _tables():
    from mixemt_amd import phylotree, preprocess
    refseq = phylotree.load_rsrs()
    phy = phylotree.load_build17(refseq)
    haps = sorted(phy.hap_var)
    return refseq, haps, preprocess.HapVarTables.build(refseq, phy, haps)


def _group(backend, rank, world, port):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    if world > 1:
        # several rank PROCESSES share the box's one GPU: two persistent one-launch loops at once could starve each
        # other (include/mixemt_hip.h, mxm_em_loop), so shared-GPU runs take the per-iteration kernels
        from mixemt_amd import _lib
        _lib.load().mxm_set_loop_fused(0, 0)


def _probe_props(n_haps):
    """A fixed, rank-independent proportion vector for the shard-additivity probe."""
    return numpy.random.default_rng(2025).dirichlet([1.0] * n_haps)


def _probe_colsum(plan, n_haps, reduce=False):
    """p_h * T_h of ONE fused iteration under _probe_props (all-reduced over the group when asked)."""
    import torch
    import torch.distributed as dist
    from mixemt_amd import em
    props = torch.from_numpy(_probe_props(n_haps)[None, :]).cuda()
    colsum = torch.zeros_like(props)
    plan.em_iter(props, torch.log(props), em.new_state(1, props.device), colsum)
    if reduce:
        dist.all_reduce(colsum)
    return (props * colsum)[0].cpu().numpy()


def _sample_blocks(n_rows, n_blocks, width, seed):
    rng = numpy.random.default_rng(seed)
    starts = numpy.sort(rng.choice(max(1, n_rows - width), size=n_blocks, replace=False))
    return [(int(a), int(min(a + width, n_rows))) for a in starts]


def _oracle_rows(tables, n_haps, row_ptr, site, obs, lo, hi):
    from oracle import c_oracle
    a, b = int(row_ptr[lo]), int(row_ptr[hi])
    return c_oracle.build_em_matrix(tables.expected, tables.lhit, tables.lmiss, row_ptr[lo:hi + 1] - row_ptr[lo],
                                    site[a:b], obs[a:b], n_haps)


# ---------------------------------------------------------------------------------------------------------------
# config 4
# ---------------------------------------------------------------------------------------------------------------
def _cfg4_one_rank(rank, out_path):
    _paths()
    import argparse
    import torch
    from mixemt_amd import em, preprocess, synth
    from oracle import em_oracle
    torch.cuda.set_device(0)
    refseq, haps, tables = _tables()
    n_haps = len(haps)
    out = {"skipped": 0}
    if torch.cuda.mem_get_info()[0] < CFG4_ROWS * 9.5e3:          # records 6.4 KB + the long rows' dense detour
        out["skipped"] = 1
        numpy.savez(out_path, **out)
        return
    row_ptr, site, obs, _ = synth.synth_rows(tables, len(refseq), 0, CFG4_ROWS, seed=1)
    cm = preprocess.build_em_records_device(tables, row_ptr, site, obs)
    out["rest"] = int(cm.rest_rows.numel())
    out["record_gb"] = cm.used / 1e9
    out["dense_gb"] = CFG4_ROWS * n_haps * 8 / 1e9
    # 256 sampled rows decode to build_em_matrix's own bits (C oracle)
    bad = 0
    for lo, hi in _sample_blocks(CFG4_ROWS, 64, 4, seed=41):
        got = cm.dense(lo, hi).cpu().numpy()
        bad += int(not numpy.array_equal(got, _oracle_rows(tables, n_haps, row_ptr, site, obs, lo, hi)))
    out["decode_mismatches"] = bad
    wts = torch.ones(CFG4_ROWS, dtype=torch.float64, device="cuda")
    args = argparse.Namespace(init_alpha=1.0, tolerance=1e-4, max_iter=CFG4_ITERS, n_multi=1, verbose=False)
    runs = []
    for _ in range(2):                                            # twice: the whole loop reproduces bit for bit
        numpy.random.seed(7)
        runs.append(em.run_em_ex(None, wts, args, want_read_mix=False, records=cm))
    res = runs[0]
    out.update(storage=res["storage"], iters=numpy.array(res["iters"]), done=numpy.array(res["done"]),
               props=res["props"], ln_theta_k=res["ln_theta_k"], inits=res["inits"], loop_s=res["loop_s"],
               deterministic=int(numpy.array_equal(runs[0]["props"], runs[1]["props"]) and runs[0]["iters"] == runs[1]["iters"]))
    plan = em.EmPlan(None, wts, records=cm)
    # mass conservation over all 10^7 rows: sum_h p_h T_h = sum_r w_r; the same vector is the 4-rank leg's yardstick
    probe = _probe_colsum(plan, n_haps)
    out["probe"] = probe
    out["mass"] = float(probe.sum())
    # a 2000-row slab in the middle against the oracle's em_step under the run's last proportions
    a = CFG4_ROWS // 2
    part = cm.rows(a, a + 2000)
    sub = em.EmPlan(None, wts[a:a + 2000], records=part)
    ln_k = torch.from_numpy(res["ln_theta_k"]).cuda()
    props_k = torch.exp(ln_k)
    sub_cs = torch.zeros_like(props_k)
    sub.em_iter(props_k, ln_k, em.new_state(1, props_k.device), sub_cs)
    got = (props_k * sub_cs)[0].cpu().numpy()
    host = cm.dense(a, a + 2000).cpu().numpy()
    _, new = em_oracle.em_step(host, numpy.ones(2000), res["ln_theta_k"][0], numpy.empty_like(host))
    out["slab_err"] = float(numpy.abs(got / got.sum() - numpy.exp(new)).max())
    out["slab_mass"] = float(got.sum())
    numpy.savez(out_path, **out)


def _cfg4_ranks(rank, world, port, out_dir):
    _paths()
    import argparse
    import torch
    import torch.distributed as dist
    _group("gloo", rank, world, port)
    try:
        from mixemt_amd import dist as mdist, em, preprocess, synth
        refseq, haps, tables = _tables()
        n_haps = len(haps)
        lo, hi = mdist.shard_bounds(CFG4_ROWS, rank, world)
        row_ptr, site, obs, _ = synth.synth_rows(tables, len(refseq), lo, hi, seed=1)
        cm = preprocess.build_em_records_device(tables, row_ptr, site, obs)
        wts = torch.ones(hi - lo, dtype=torch.float64, device="cuda")
        args = argparse.Namespace(init_alpha=1.0, tolerance=1e-4, max_iter=CFG4_ITERS, n_multi=1, verbose=False)
        numpy.random.seed(7 if rank == 0 else 1000 + rank)        # only rank 0's stream may matter (broadcast_inits)
        res = mdist.run_em_sharded(None, wts, args, want_read_mix=False, records=cm, check_every=8)
        plan = em.EmPlan(None, wts, records=cm)
        probe = _probe_colsum(plan, n_haps, reduce=True)
        numpy.savez(os.path.join(out_dir, "rank%d.npz" % rank), props=res["props"], iters=numpy.array(res["iters"]),
                    done=numpy.array(res["done"]), inits=res["inits"], ln_theta_k=res["ln_theta_k"], probe=probe,
                    lo=lo, hi=hi)
    finally:
        dist.destroy_process_group()


def test_config4_whole_size_one_rank_then_four_ranks(tmp_path):
    """10^7 x 5408 as records on one GPU (a dense matrix would be 433 GB), 40 EM iterations; then the same rows over 4 ranks."""
    import gc
    import torch
    import torch.multiprocessing as mp
    gc.collect()
    torch.cuda.empty_cache()
    one = str(tmp_path / "one.npz")
    mp.spawn(_cfg4_one_rank, args=(one,), nprocs=1, join=True)
    r = numpy.load(one)
    if int(r["skipped"]):
        pytest.skip("needs %.0f GB of free HBM" % (CFG4_ROWS * 9.5e3 / 1e9))
    assert str(r["storage"]) == "coded"
    assert int(r["decode_mismatches"]) == 0
    assert r["iters"].tolist() == [CFG4_ITERS] and r["done"].tolist() == [2]
    assert int(r["deterministic"]) == 1
    assert abs(float(r["props"].sum()) - 1.0) < 1e-12
    assert abs(float(r["mass"]) - CFG4_ROWS) < 1e-9 * CFG4_ROWS
    assert float(r["slab_err"]) < 1e-13 and abs(float(r["slab_mass"]) - 2000.0) < 1e-8
    assert float(r["record_gb"]) < 0.2 * float(r["dense_gb"])
    p = r["props"]
    assert int(numpy.argmax(p)) == 10 and (p[[10, 2000, 4000]] > 1.0 / 5408).all()
    numpy.random.seed(7)
    from mixemt_amd import em
    assert numpy.array_equal(r["inits"][0], em.init_props(len(p), 1.0))
    # ---- the same 10^7 rows as four ranks sharing the GPU ----
    mp.spawn(_cfg4_ranks, args=(CFG4_RANKS, _free_port(), str(tmp_path)), nprocs=CFG4_RANKS, join=True)
    res = [numpy.load(str(tmp_path / ("rank%d.npz" % k))) for k in range(CFG4_RANKS)]
    assert int(res[0]["lo"]) == 0 and int(res[-1]["hi"]) == CFG4_ROWS
    for k, q in enumerate(res):
        if k:
            assert int(q["lo"]) == int(res[k - 1]["hi"])
        assert numpy.array_equal(q["inits"], r["inits"])                       # rank 0 drew, all received
        assert q["iters"].tolist() == [CFG4_ITERS] and q["done"].tolist() == [2]   # the one-rank run's stop state
        assert numpy.array_equal(q["props"], res[0]["props"])                  # ranks agree bit for bit
        assert numpy.array_equal(q["probe"], res[0]["probe"])
        # all-reduced sums of the four shards = the one-rank sums (another summation order: rounding only)
        assert numpy.abs(q["probe"] - r["probe"]).max() < 1e-12 * CFG4_ROWS
        assert numpy.abs(q["probe"] / r["probe"] - 1.0)[r["probe"] > 1e-3].max() < 1e-11
        assert numpy.abs(q["props"] - r["props"]).max() < 1e-12
        assert numpy.abs(q["ln_theta_k"] - r["ln_theta_k"]).max() < 1e-9


# ---------------------------------------------------------------------------------------------------------------
# config 5
# ---------------------------------------------------------------------------------------------------------------
def _cfg5_worker(rank, world, port, out_dir):
    _paths()
    import argparse
    import torch
    import torch.distributed as dist
    _group("nccl" if world == 1 else "gloo", rank, world, port)
    try:
        from mixemt_amd import dist as mdist, em, preprocess, synth
        from oracle import em_oracle
        refseq, haps, tables = _tables()
        n_haps = len(haps)
        row_ptr, site, obs, _ = synth.synth_rows(tables, len(refseq), 0, FULL_ROWS, seed=1)
        cm = preprocess.build_em_records_device(tables, row_ptr, site, obs)       # the replicated matrix
        wts = torch.ones(FULL_ROWS, dtype=torch.float64, device="cuda")
        args = argparse.Namespace(init_alpha=1.0, tolerance=1e-4, max_iter=CFG5_ITERS, n_multi=CFG5_RESTARTS, verbose=False)
        numpy.random.seed(7 if rank == 0 else 1000 + rank)
        timing = {}
        res = mdist.run_em_restart_parallel(None, wts, args, want_read_mix=False, timing=timing, records=cm)
        out = dict(props=res["props"], run_props=res["run_props"], iters=numpy.array(res["iters"]),
                   done=numpy.array(res["done"]), inits=res["inits"], ln_theta_k=res["ln_theta_k"],
                   loop_s=timing["loop_s"], rows=numpy.array(res["rows"]))
        if world == 1:
            # the next M-step of four of the 64 restarts on a 2000-row slab against the oracle
            a = FULL_ROWS // 2
            host = cm.dense(a, a + 2000).cpu().numpy()
            sub = em.EmPlan(None, wts[a:a + 2000], records=cm.rows(a, a + 2000))
            errs = []
            for b in (0, 21, 42, 63):
                ln_k = torch.from_numpy(res["ln_theta_k"][b:b + 1]).cuda()
                props_k = torch.exp(ln_k)
                cs = torch.zeros_like(props_k)
                sub.em_iter(props_k, ln_k, em.new_state(1, props_k.device), cs)
                got = (props_k * cs)[0].cpu().numpy()
                _, new = em_oracle.em_step(host, numpy.ones(2000), res["ln_theta_k"][b], numpy.empty_like(host))
                errs.append(float(numpy.abs(got / got.sum() - numpy.exp(new)).max()))
            out["slab_errs"] = numpy.array(errs)
            # run 5 again, alone, through the per-iteration kernels (mxm_em_iter_coded + mxm_m_finalize)
            from mixemt_amd import _lib
            _lib.load().mxm_set_loop_fused(0, 0)
            plan = em.EmPlan(None, wts, records=cm)
            _, ln_new, st = em.em_loop(plan, res["inits"][5:6], args.tolerance, CFG5_ITERS)
            _lib.load().mxm_reset_tuning()
            out["alone_err"] = float(numpy.abs(torch.exp(ln_new)[0].cpu().numpy() - res["run_props"][5]).max())
            out["alone_iters"] = int(st[0][1])
        numpy.savez(os.path.join(out_dir, "w%d_rank%d.npz" % (world, rank)), **out)
    finally:
        dist.destroy_process_group()


def test_config5_sixty_four_restarts_one_rank_and_four_ranks(tmp_path):
    """64 restarts on the 10^6 x 5408 records: one rank x 64 (the one-launch loop, restarts back to back), then
    4 ranks x 16 sharing the GPU (per-iteration kernels), dealt round-robin by global run index."""
    import gc
    import torch
    import torch.multiprocessing as mp
    from mixemt_amd import em
    gc.collect()
    torch.cuda.empty_cache()
    if torch.cuda.mem_get_info()[0] < CFG5_RANKS * FULL_ROWS * 9.5e3:
        pytest.skip("needs %.0f GB of free HBM" % (CFG5_RANKS * FULL_ROWS * 9.5e3 / 1e9))
    mp.spawn(_cfg5_worker, args=(1, _free_port(), str(tmp_path)), nprocs=1, join=True)
    one = numpy.load(str(tmp_path / "w1_rank0.npz"))
    n_haps = one["props"].shape[0]
    numpy.random.seed(7)
    want_inits = numpy.stack([em.init_props(n_haps, 1.0) for _ in range(CFG5_RESTARTS)])
    assert numpy.array_equal(one["inits"], want_inits)                        # the reference's sequential draws (em.py:123)
    assert one["iters"].tolist() == [CFG5_ITERS] * CFG5_RESTARTS and one["done"].tolist() == [2] * CFG5_RESTARTS
    assert numpy.abs(one["run_props"].sum(axis=1) - 1.0).max() < 1e-12
    # geometric mean over the runs, not renormalised (em.py:155-163): a host fold of the runs' proportions
    with numpy.errstate(divide="ignore"):
        fold = numpy.exp(numpy.log(one["run_props"]).sum(axis=0) / CFG5_RESTARTS)
    assert numpy.abs(one["props"] - fold).max() < 1e-15
    assert one["rows"].tolist() == [0, FULL_ROWS]
    assert float(one["slab_errs"].max()) < 1e-13
    assert int(one["alone_iters"]) == CFG5_ITERS and float(one["alone_err"]) < 1e-13
    # distinct inits really give distinct states this early
    assert numpy.abs(one["run_props"][0] - one["run_props"][1]).max() > 1e-6
    mp.spawn(_cfg5_worker, args=(CFG5_RANKS, _free_port(), str(tmp_path)), nprocs=CFG5_RANKS, join=True)
    for k in range(CFG5_RANKS):
        q = numpy.load(str(tmp_path / ("w%d_rank%d.npz" % (CFG5_RANKS, k))))
        assert numpy.array_equal(q["inits"], want_inits)
        assert q["iters"].tolist() == one["iters"].tolist() and q["done"].tolist() == one["done"].tolist()
        assert numpy.abs(q["run_props"] - one["run_props"]).max() < 1e-13      # run by run, whichever rank ran it
        assert numpy.abs(q["ln_theta_k"] - one["ln_theta_k"]).max() < 1e-9
        assert numpy.abs(q["props"] - one["props"]).max() < 1e-13
        from mixemt_amd import dist as mdist
        assert tuple(q["rows"]) == mdist.shard_bounds(FULL_ROWS, k, CFG5_RANKS)


# ---------------------------------------------------------------------------------------------------------------
# the default route at 10^6 rows, to convergence
# ---------------------------------------------------------------------------------------------------------------
def _default_route_worker(rank, out_path):
    _paths()
    import argparse
    import torch
    from mixemt_amd import _lib, assign, em, preprocess, synth
    torch.cuda.set_device(0)
    refseq, haps, tables = _tables()
    n_haps = len(haps)
    out = {"skipped": 0}
    if torch.cuda.mem_get_info()[0] < 3.4 * FULL_ROWS * n_haps * 8:
        out["skipped"] = 1
        numpy.savez(out_path, **out)
        return
    row_ptr, site, obs, who = synth.synth_rows(tables, len(refseq), 0, FULL_ROWS, seed=1)
    mat = preprocess.build_em_matrix_device(tables, row_ptr, site, obs)
    wts = torch.ones(FULL_ROWS, dtype=torch.float64, device="cuda")
    args = argparse.Namespace(init_alpha=1.0, tolerance=1e-4, max_iter=10000, n_multi=1, verbose=False)
    numpy.random.seed(7)
    auto = em.run_em_ex(mat, wts, args)                               # storage defaults to "auto"
    best_a, votes_a = assign.row_argmax_votes(auto["read_mix"], wts)
    lse = torch.logsumexp(auto["read_mix"][::997], dim=1)
    out.update(storage=auto["storage"], iters=numpy.array(auto["iters"]), done=numpy.array(auto["done"]),
               props=auto["props"], votes=votes_a, lse_err=float(lse.abs().max().item()), loop_s=auto["loop_s"])
    auto["read_mix"] = None
    torch.cuda.empty_cache()
    # the dense matrix through the per-iteration kernels (mxm_em_iter + mxm_m_finalize)
    _lib.load().mxm_set_loop_fused(0, 0)
    numpy.random.seed(7)
    dense = em.run_em_ex(mat, wts, args, storage="f64")
    _lib.load().mxm_reset_tuning()
    best_d, votes_d = assign.row_argmax_votes(dense["read_mix"], wts)
    out.update(dense_storage=dense["storage"], dense_iters=numpy.array(dense["iters"]), dense_props=dense["props"],
               calls_equal=int(numpy.array_equal(best_a, best_d)), votes_equal=int(numpy.array_equal(votes_a, votes_d)),
               truth_share=float((best_a == numpy.array([10, 2000, 4000])[who]).mean()), dense_loop_s=dense["loop_s"])
    # the same run from records straight out of the build (no dense matrix): same loop, same bits as "auto"'s records?
    # (the encoder and the marker build write the same tables, so the one-launch loop sees the same records)
    del dense
    torch.cuda.empty_cache()
    cm = preprocess.build_em_records_device(tables, row_ptr, site, obs)
    numpy.random.seed(7)
    rec = em.run_em_ex(None, wts, args, want_read_mix=False, records=cm)
    best_r, votes_r = assign.row_argmax_votes_records(cm, rec["ln_theta_k"], wts)
    out.update(rec_iters=numpy.array(rec["iters"]), rec_props=rec["props"],
               rec_calls_equal=int(numpy.array_equal(best_r, best_a)), rec_votes_equal=int(numpy.array_equal(votes_r, votes_a)))
    numpy.savez(out_path, **out)


def test_default_route_at_a_million_rows_to_convergence(tmp_path):
    """run_em's own choice at 10^6 x 5408 (records, one launch) run to convergence = the dense per-iteration loop."""
    import gc
    import torch
    import torch.multiprocessing as mp
    gc.collect()
    torch.cuda.empty_cache()
    out_path = str(tmp_path / "auto.npz")
    mp.spawn(_default_route_worker, args=(out_path,), nprocs=1, join=True)
    r = numpy.load(out_path)
    if int(r["skipped"]):
        pytest.skip("needs %.0f GB of free HBM" % (3.4 * FULL_ROWS * 5408 * 8 / 1e9))
    assert str(r["storage"]) == "coded" and str(r["dense_storage"]) == "f64"
    assert r["done"].tolist() == [1] and int(r["iters"][0]) > 100
    assert r["iters"].tolist() == r["dense_iters"].tolist() == r["rec_iters"].tolist()
    assert numpy.abs(r["props"] - r["dense_props"]).max() < 1e-12
    assert numpy.abs(r["props"] - r["rec_props"]).max() < 1e-12
    assert int(r["calls_equal"]) == 1 and int(r["votes_equal"]) == 1
    assert int(r["rec_calls_equal"]) == 1 and int(r["rec_votes_equal"]) == 1
    assert float(r["lse_err"]) < 1e-9
    p = r["props"]
    assert sorted(numpy.argsort(p)[::-1][:3].tolist()) == [10, 2000, 4000]
    assert numpy.allclose(p[[10, 2000, 4000]], [0.6, 0.3, 0.1], atol=0.02) and abs(p.sum() - 1.0) < 1e-9
    assert float(r["truth_share"]) > 0.5
