"""
BASELINE configs 4 and 5 at their PER-GPU workload, through the multi-GPU code paths with the RCCL calls in place
(a one-rank "nccl" group: the box has one GPU; the 2-rank transport itself is covered by test_gpu_dist.py /
test_dist_cpu.py):

  config 4  10^7 reads x 5408 sharded over 8 GPUs  ->  this rank's shard, 1.25 * 10^6 x 5408 (54 GB + its linearised
            copy), 20 iterations of dist.sharded_em_loop with the all-reduce between mxm_em_iter and mxm_m_finalize
  config 5  64 restarts over 8 GPUs on the replicated 10^6 x 5408 matrix  ->  this rank's 8 restarts through
            dist.run_em_restart_parallel (mxm_em_loop with full tiles, posterior fold, end-of-run combine)

The oracle cannot run at these sizes; checks are the oracle on row slabs / sampled rows of the same device buffers,
the unbatched schedule on the whole matrix, and size-independent identities (reference loop being sharded:
em.py:117-161).  Each case runs in a process of its own (spawned) so that its ~110 GB are given back when it ends.
MXM_FULL_ROWS scales both down for a smaller card.
"""
import os
import socket

import numpy
import pytest

pytestmark = pytest.mark.gpu

FULL_ROWS = int(os.environ.get("MXM_FULL_ROWS", "1000000"))
SHARD_ROWS = FULL_ROWS * 5 // 4          # config 4: 10^7 / 8
N_ITERS = 20


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _setup(port):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    sys.path.insert(0, os.path.dirname(here))
    import torch
    import torch.distributed as dist
    from mixemt_amd import phylotree, preprocess
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    refseq = phylotree.load_rsrs()
    phy = phylotree.load_build17(refseq)
    haps = sorted(phy.hap_var)
    return refseq, haps, preprocess.HapVarTables.build(refseq, phy, haps)


def _config4_worker(rank, port, out_path):
    import torch
    import torch.distributed as dist
    refseq, haps, tables = _setup(port)
    try:
        from mixemt_amd import dist as mdist, em, preprocess, synth
        from oracle import em_oracle
        n_haps = len(haps)
        out = {"skipped": 0}
        need = 2.15 * SHARD_ROWS * n_haps * 8
        if torch.cuda.mem_get_info()[0] < need:
            out["skipped"] = 1
            numpy.savez(out_path, **out)
            return
        # rank 3's shard of the 10^7-row problem (rows [3 750 000, 5 000 000) of the global synth-v1 read set)
        lo, hi = mdist.shard_bounds(8 * SHARD_ROWS, 3, 8)
        assert hi - lo == SHARD_ROWS
        row_ptr, site, obs, _ = synth.synth_rows(tables, len(refseq), lo, hi, seed=1)
        mat = preprocess.build_em_matrix_device(tables, row_ptr, site, obs)
        wts = torch.ones(SHARD_ROWS, dtype=torch.float64, device="cuda")
        plan = em.EmPlan(mat, wts, n_runs=1)
        out["bytes_in_use"] = float(torch.cuda.memory_allocated())
        numpy.random.seed(7)
        init = em.init_props(n_haps, 1.0)[None, :]
        runs = []
        for _ in range(2):                                         # twice: bitwise determinism of the whole loop
            ln_cur, ln_new, states = mdist.sharded_em_loop(plan, init, 0.0, N_ITERS, check_every=8)
            runs.append((ln_cur.clone(), ln_new.clone(), states))
        out["states"] = numpy.array([[s[0], s[1]] for s in runs[0][2]])
        out["deterministic"] = int(torch.equal(runs[0][1], runs[1][1]) and torch.equal(runs[0][0], runs[1][0])
                                   and runs[0][2] == runs[1][2])
        # the plain single-process driver (mxm_em_loop, no collective) must give the same bits: a one-rank all-reduce is the identity
        c2, n2, st2 = em.em_loop(plan, init, 0.0, N_ITERS)
        out["same_as_local_loop"] = int(torch.equal(n2, runs[0][1]))
        ln_cur, ln_new, _ = runs[0]
        props = torch.exp(ln_new)
        out["props_sum"] = float(props.sum().item())
        # mass conservation over the whole shard: sum_h p_h T_h = sum_r w_r
        colsum = torch.zeros_like(props)
        state = em.new_state(1, props.device)
        plan.em_iter(props, ln_new, state, colsum)
        dist.all_reduce(colsum)
        out["mass"] = float((props * colsum).sum().item())
        # a 2 000-row slab in the middle of the shard against the oracle's em_step under the same proportions
        a = SHARD_ROWS // 2
        slab = mat[a:a + 2000]
        sub = em.EmPlan(slab, wts[a:a + 2000], n_runs=1)
        sub_cs = torch.zeros_like(props)
        sub.em_iter(props, ln_new, em.new_state(1, props.device), sub_cs)
        got = (props * sub_cs)[0].cpu().numpy()
        host = slab.cpu().numpy()
        _, new = em_oracle.em_step(host, numpy.ones(2000), ln_new[0].cpu().numpy(), numpy.empty_like(host))
        out["slab_err"] = float(numpy.abs(got / got.sum() - numpy.exp(new)).max())
        out["slab_mass"] = float(got.sum())
        # after 20 iterations the proportions already lean the planted way (0.6 / 0.3 / 0.1 at columns 10, 2000, 4000;
        # close relatives of a contributor still hold part of its share this early)
        p = props[0].cpu().numpy()
        out["top1"] = int(numpy.argmax(p))
        out["planted"] = p[[10, 2000, 4000]]
        numpy.savez(out_path, **out)
    finally:
        dist.destroy_process_group()


def test_config4_shard_at_size_through_the_sharded_loop(tmp_path):
    """Config 4's per-GPU share: 1.25 * 10^6 x 5408 resident, dist.sharded_em_loop over a one-rank RCCL group."""
    import gc
    import torch
    import torch.multiprocessing as mp
    gc.collect()
    torch.cuda.empty_cache()
    out_path = str(tmp_path / "cfg4.npz")
    mp.spawn(_config4_worker, args=(_free_port(), out_path), nprocs=1, join=True)
    r = numpy.load(out_path)
    if int(r["skipped"]):
        pytest.skip("needs %.0f GB of free HBM" % (2.15 * SHARD_ROWS * 5408 * 8 / 1e9))
    assert r["states"].tolist() == [[2, N_ITERS]]
    assert int(r["deterministic"]) == 1 and int(r["same_as_local_loop"]) == 1
    assert abs(float(r["props_sum"]) - 1.0) < 1e-12
    assert abs(float(r["mass"]) - SHARD_ROWS) < 1e-9 * SHARD_ROWS
    assert float(r["slab_err"]) < 1e-13 and abs(float(r["slab_mass"]) - 2000.0) < 1e-8
    assert int(r["top1"]) == 10 and float(r["planted"][0]) > 0.3 and (r["planted"] > 1.0 / 5408).all()
    assert float(r["bytes_in_use"]) > 2.0 * SHARD_ROWS * 5408 * 8          # M and P really were resident


N_RESTARTS = 8           # config 5: 64 restarts / 8 GPUs
MAX_ITER = 40


def _config5_worker(rank, port, out_path):
    import argparse
    import torch
    import torch.distributed as dist
    refseq, haps, tables = _setup(port)
    try:
        from mixemt_amd import _lib, dist as mdist, em, preprocess, synth
        from oracle import em_oracle
        n_haps = len(haps)
        out = {"skipped": 0}
        if torch.cuda.mem_get_info()[0] < 2.3 * FULL_ROWS * n_haps * 8:
            out["skipped"] = 1
            numpy.savez(out_path, **out)
            return
        row_ptr, site, obs, _ = synth.synth_rows(tables, len(refseq), 0, FULL_ROWS, seed=1)
        mat = preprocess.build_em_matrix_device(tables, row_ptr, site, obs)
        wts = torch.ones(FULL_ROWS, dtype=torch.float64, device="cuda")
        args = argparse.Namespace(init_alpha=1.0, tolerance=1e-4, max_iter=MAX_ITER, n_multi=N_RESTARTS, verbose=False)
        numpy.random.seed(7)
        timing = {}
        res = mdist.run_em_restart_parallel(mat, wts, args, timing=timing)
        out.update(iters=numpy.array(res["iters"]), done=numpy.array(res["done"]), rows=numpy.array(res["rows"]),
                   props=res["props"], run_sums=res["run_props"].sum(axis=1), loop_s=timing["loop_s"])
        # the inits are the sequential draws of the reference (em.py:123)
        numpy.random.seed(7)
        want_inits = numpy.stack([em.init_props(n_haps, 1.0) for _ in range(N_RESTARTS)])
        out["inits_equal"] = int(numpy.array_equal(res["inits"], want_inits))
        # geometric mean over the runs, not renormalised (em.py:155-163)
        out["geo_err"] = float(numpy.abs(res["props"] - numpy.exp(numpy.log(res["run_props"]).mean(axis=0))).max())
        # the same eight restarts one per pass (no batching, no round-robin tiles): same proportions up to summation order
        lib = _lib.load()
        lib.mxm_set_batch_tile(1)
        plan = em.EmPlan(mat, wts, n_runs=N_RESTARTS)
        _, new1, st1 = em.em_loop(plan, res["inits"], args.tolerance, MAX_ITER)
        lib.mxm_reset_tuning()
        out["unbatched_err"] = float(numpy.abs(torch.exp(new1).cpu().numpy() - res["run_props"]).max())
        out["unbatched_iters"] = numpy.array([s[1] for s in st1])
        # every restart's next M-step on a 2 000-row slab against the oracle
        a = FULL_ROWS // 2
        slab = mat[a:a + 2000]
        sub = em.EmPlan(slab, wts[a:a + 2000], n_runs=N_RESTARTS)
        ln_k = torch.from_numpy(res["ln_theta_k"]).cuda()
        props_k = torch.exp(ln_k)
        sub_cs = torch.zeros_like(props_k)
        sub.em_iter(props_k, ln_k, em.new_state(N_RESTARTS, props_k.device), sub_cs)
        got = (props_k * sub_cs).cpu().numpy()
        host = slab.cpu().numpy()
        errs = []
        for b in range(N_RESTARTS):
            _, new = em_oracle.em_step(host, numpy.ones(2000), res["ln_theta_k"][b], numpy.empty_like(host))
            errs.append(float(numpy.abs(got[b] / got[b].sum() - numpy.exp(new)).max()))
        out["slab_errs"] = numpy.array(errs)
        # the folded posterior (em.py:156-161): row-normalised, and sampled rows equal the oracle's fold over the 8 runs
        mix = res["read_mix"]
        assert tuple(mix.shape) == (FULL_ROWS, n_haps)
        lse = torch.logsumexp(mix[::97], dim=1)
        out["fold_lse_err"] = float(lse.abs().max().item())
        rows = numpy.sort(numpy.random.default_rng(55).choice(FULL_ROWS, size=48, replace=False))
        idx = torch.from_numpy(rows).cuda()
        host_rows = mat[idx].cpu().numpy()
        want = None
        for b in range(N_RESTARTS):
            post, _ = em_oracle.em_step(host_rows, numpy.ones(len(rows)), res["ln_theta_k"][b], numpy.empty_like(host_rows))
            want = post.copy() if want is None else numpy.logaddexp(want, post)
        want -= numpy.log(N_RESTARTS)
        out["fold_err"] = float(numpy.abs(mix[idx].cpu().numpy() - want).max())
        numpy.savez(out_path, **out)
    finally:
        dist.destroy_process_group()


def test_config5_share_at_size_through_restart_parallel(tmp_path):
    """Config 5's per-GPU share: 8 restarts on the replicated 10^6 x 5408 matrix through dist.run_em_restart_parallel
    (one-rank RCCL group), 40 iterations each."""
    import gc
    import torch
    import torch.multiprocessing as mp
    gc.collect()
    torch.cuda.empty_cache()
    out_path = str(tmp_path / "cfg5.npz")
    mp.spawn(_config5_worker, args=(_free_port(), out_path), nprocs=1, join=True)
    r = numpy.load(out_path)
    if int(r["skipped"]):
        pytest.skip("needs %.0f GB of free HBM" % (2.3 * FULL_ROWS * 5408 * 8 / 1e9))
    assert r["iters"].tolist() == [MAX_ITER] * N_RESTARTS and r["done"].tolist() == [2] * N_RESTARTS
    assert r["unbatched_iters"].tolist() == [MAX_ITER] * N_RESTARTS
    assert r["rows"].tolist() == [0, FULL_ROWS]
    assert int(r["inits_equal"]) == 1
    assert numpy.abs(r["run_sums"] - 1.0).max() < 1e-12 and float(r["geo_err"]) < 1e-15
    assert float(r["unbatched_err"]) < 1e-13
    assert float(r["slab_errs"].max()) < 1e-13
    assert float(r["fold_lse_err"]) < 1e-9 and float(r["fold_err"]) < 1e-9
