"""
The optional one-shot exchange of a row-sharded loop (include/mixemt_hip.h mxm_exchange_*, csrc/exchange.hpp,
dist.OneShotExchange; SURVEY.md section 8 e): every rank writes its M-step sums into every rank's buffer, each rank adds
the slots up in rank order.  One GPU here: the ranks are PROCESSES sharing it (gloo carries the 64-byte handles and the
loop's agreement checks), which exercises the protocol -- slots, flags, epochs, parities, the bounded wait -- but not
xGMI; over several GPUs it is unmeasured.
  * world of one, no process group: the loop with the exchange in it equals the loop without, bit for bit, eager and
    replayed from a captured hipGraph (the exchange count lives on the device);
  * golden g10 over two ranks: the reference's iteration counts and proportions, bit-identical to the all-reduce run;
  * three ranks: colsum is the sum of the ranks' sums IN RANK ORDER, the same bits on every rank, over 40 exchanges;
  * a rank that never pushes: the others' pull times out, poisons the sums and raises the error flag.
"""
import os
import socket

import numpy
import pytest

from conftest import em_args, golden

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        return sock.getsockname()[1]


def _paths():
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    sys.path.insert(0, os.path.dirname(here))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return here


def test_world_of_one_equals_the_loop_without_an_exchange(b17):
    import torch
    from mixemt_amd import dist as mdist, em, preprocess
    refseq, phy, haps, tables = b17
    g = golden("g5_run_em_multi")
    mat = preprocess.build_em_matrix_device(tables, g["row_ptr"], g["site"], g["obs"])
    plan = em.EmPlan(mat, torch.from_numpy(g["wts"]).cuda(), n_runs=3)
    plain = mdist.sharded_em_loop(plan, g["inits"], 1e-4, 10000, check_every=8)
    one = mdist.sharded_em_loop(plan, g["inits"], 1e-4, 10000, check_every=8, exchange="oneshot", graph=False)
    assert [s[1] for s in one[2]] == list(g["iters"]) and one[2] == plain[2]
    assert torch.equal(one[0], plain[0]) and torch.equal(one[1], plain[1])
    graphed = mdist.sharded_em_loop(plan, g["inits"], 1e-4, 10000, check_every=8, exchange="oneshot", graph=True)
    assert mdist.sharded_em_loop.last_graph_bursts > 10
    assert torch.equal(graphed[1], plain[1]) and graphed[2] == plain[2]
    x = mdist.OneShotExchange(3 * len(haps))
    import ctypes
    fine, nbytes = ctypes.c_int32(-1), ctypes.c_int64(0)
    assert x.lib.mxm_exchange_info(x.handle, ctypes.byref(fine), ctypes.byref(nbytes)) == 0
    assert nbytes.value >= 2 * 3 * len(haps) * 8 and fine.value in (0, 1)
    print("exchange buffer: %d bytes, fine-grained: %d" % (nbytes.value, fine.value))
    x.close()


def _g10_worker(rank, world, port, out_dir):
    here = _paths()
    import torch
    import torch.distributed as dist
    from mixemt_amd import _lib, dist as mdist, phylotree, preprocess
    from test_gpu_g10 import _inputs
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    _lib.load().mxm_set_loop_fused(0, 0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = numpy.load(os.path.join(here, "golden", "g10_run_em_20k.npz"))
        refseq = phylotree.load_rsrs()
        phy = phylotree.load_build17(refseq)
        tables = preprocess.HapVarTables.build(refseq, phy, sorted(phy.hap_var))
        row_ptr, site, obs, wts = _inputs(tables, len(refseq), g)
        lo, hi = mdist.shard_bounds(len(wts), rank, world)
        a, b = int(row_ptr[lo]), int(row_ptr[hi])
        shard = preprocess.build_em_matrix_device(tables, row_ptr[lo:hi + 1] - row_ptr[lo], site[a:b], obs[a:b])
        w = torch.from_numpy(wts[lo:hi]).cuda()
        import argparse
        args = argparse.Namespace(init_alpha=1.0, tolerance=1e-4, max_iter=10000, n_multi=1, verbose=False)
        out = {}
        for label in ("rccl", "oneshot"):
            numpy.random.seed(23 if rank == 0 else 999)
            res = mdist.run_em_sharded(shard, w, args, check_every=8, want_read_mix=False, exchange=label)
            out[label + "_props"] = res["props"]
            out[label + "_iters"] = numpy.array(res["iters"])
            out[label + "_inits"] = res["inits"]
        # the same loop with its bursts replayed from a captured hipGraph: kernels only, so capturable over a gloo group too;
        # the exchange count lives on the device, so a replay pushes the right epoch
        from mixemt_amd import em
        plan = em.EmPlan(shard, w, n_runs=1, storage="f64")
        eager = mdist.sharded_em_loop(plan, out["oneshot_inits"], 1e-4, 10000, check_every=8, exchange="oneshot", graph=False)
        graphed = mdist.sharded_em_loop(plan, out["oneshot_inits"], 1e-4, 10000, check_every=8, exchange="oneshot", graph=True)
        out["graph_bursts"] = mdist.sharded_em_loop.last_graph_bursts
        out["graph_equal"] = int(torch.equal(eager[1], graphed[1]) and eager[2] == graphed[2])
        out["graph_iters"] = numpy.array([st[1] for st in graphed[2]])
        numpy.savez(os.path.join(out_dir, "rank%d.npz" % rank), **out)
    finally:
        dist.destroy_process_group()


def test_g10_over_two_ranks_is_the_all_reduce_run_bit_for_bit(tmp_path):
    import torch
    import torch.multiprocessing as mp
    torch.cuda.empty_cache()
    g = golden("g10_run_em_20k")
    mp.spawn(_g10_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    res = [numpy.load(str(tmp_path / ("rank%d.npz" % r))) for r in range(2)]
    for r in res:
        assert numpy.array_equal(r["oneshot_inits"], g["inits"])
        assert list(r["oneshot_iters"]) == list(g["iters"]) == list(r["rccl_iters"])
        assert numpy.abs(r["oneshot_props"] - g["props"]).max() < 1e-12
        assert numpy.array_equal(r["oneshot_props"], r["rccl_props"])          # a + b in either order: the same bits
        assert numpy.array_equal(r["oneshot_props"], res[0]["oneshot_props"])
        assert int(r["graph_equal"]) == 1 and int(r["graph_bursts"]) > 10 and list(r["graph_iters"]) == list(g["iters"])


def _sum_worker(rank, world, port, out_dir, silent_rank):
    _paths()
    import torch
    import torch.distributed as dist
    from mixemt_amd import dist as mdist, em
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n_runs, width = 2, 5408
        x = mdist.OneShotExchange(n_runs * width)
        rng = numpy.random.default_rng(100 + rank)
        ok, same = True, True
        state = em.new_state(n_runs, "cuda")
        if silent_rank < 0:
            for it in range(40):
                mine = rng.standard_normal((n_runs, width)) * 10.0 ** rng.integers(-8, 8)
                every = [None] * world
                dist.all_gather_object(every, mine)
                want = every[0].copy()
                for other in every[1:]:
                    want = want + other                              # rank order
                lead = 1 + it % n_runs                               # the loop exchanges the iterating restarts only
                col = torch.from_numpy(mine).cuda()
                x.reduce(col[:lead], state[:lead])
                torch.cuda.synchronize()
                got = col.cpu().numpy()
                ok = ok and numpy.array_equal(got[:lead], want[:lead]) and numpy.array_equal(got[lead:], mine[lead:])
                every = [None] * world
                dist.all_gather_object(every, got[:lead].tobytes())
                same = same and all(e == every[0] for e in every)
            numpy.savez(os.path.join(out_dir, "rank%d.npz" % rank), ok=int(ok), same=int(same))
        else:
            col = torch.ones((n_runs, width), dtype=torch.float64, device="cuda")
            if rank != silent_rank:
                x.reduce(col, state)                                 # the silent rank never pushes: a bounded wait
                torch.cuda.synchronize()
                raised = 0
                try:
                    em.read_state(state)
                except ValueError as exc:
                    raised = int("timed out" in str(exc))
                numpy.savez(os.path.join(out_dir, "rank%d.npz" % rank), nan=int(bool(torch.isnan(col).all())), raised=raised)
            dist.barrier()
        x.close()
    finally:
        dist.destroy_process_group()


def test_three_ranks_sum_in_rank_order(tmp_path):
    import torch.multiprocessing as mp
    mp.spawn(_sum_worker, args=(3, _free_port(), str(tmp_path), -1), nprocs=3, join=True)
    for r in range(3):
        res = numpy.load(str(tmp_path / ("rank%d.npz" % r)))
        assert int(res["ok"]) == 1 and int(res["same"]) == 1


def test_a_silent_rank_times_the_others_out(tmp_path):
    import torch.multiprocessing as mp
    mp.spawn(_sum_worker, args=(2, _free_port(), str(tmp_path), 1), nprocs=2, join=True)
    res = numpy.load(str(tmp_path / "rank0.npz"))
    assert int(res["nan"]) == 1 and int(res["raised"]) == 1


def _fallback_worker(rank, world, port, out_dir, mode):
    """mode "fail": rank 1's mxm_exchange_create fails (fault injection) -- ALL ranks must take the all-reduce together;
    mode "coarse": the runtime "refuses" to export the fine-grained buffer -- the library retries with ordinary device memory."""
    _paths()
    import warnings
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if mode == "fail":
        os.environ["MXM_EXCHANGE_FAIL_RANK"] = "1"
    else:
        os.environ["MXM_EXCHANGE_REFUSE_FINE"] = "1"
    os.environ["MXM_EXCHANGE_TIMEOUT_MS"] = "5000"
    from mixemt_amd import _lib, dist as mdist, em
    torch.cuda.set_device(0)
    _lib.load().mxm_set_loop_fused(0, 0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = numpy.random.default_rng(7)                                # the same matrix on every rank; each takes its rows
        n_rows, n_haps = 600, 130
        mat = numpy.log(rng.dirichlet([0.3] * n_haps, size=n_rows))
        wts = rng.integers(1, 4, size=n_rows).astype(numpy.float64)
        inits = rng.dirichlet([1.0] * n_haps, size=2)
        lo, hi = mdist.shard_bounds(n_rows, rank, world)
        plan = em.EmPlan(torch.from_numpy(mat[lo:hi]).cuda(), torch.from_numpy(wts[lo:hi]).cuda(), n_runs=2)
        plain = mdist.sharded_em_loop(plan, inits, 1e-5, 400, check_every=8)
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            got = mdist.sharded_em_loop(plan, inits, 1e-5, 400, check_every=8, exchange="oneshot", graph=False)
        warned = int(any("one-shot exchange" in str(w.message) and "all-reduce" in str(w.message) for w in caught))
        fine = -1
        if mode == "coarse":
            x = mdist.OneShotExchange(2 * n_haps)
            fine = int(x.fine_grained)
            x.close()
        numpy.savez(os.path.join(out_dir, "rank%d.npz" % rank), warned=warned, fine=fine,
                    equal=int(torch.equal(got[1], plain[1]) and got[2] == plain[2]), iters=numpy.array([s[1] for s in got[2]]),
                    ln=got[1].cpu().numpy(), route=str(mdist.sharded_em_loop.last_exchange))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["fail", "coarse"])
def test_a_rank_that_cannot_set_the_exchange_up_takes_every_rank_to_the_all_reduce(tmp_path, mode):
    """the first 8-GPU attempt must not hang -- one rank raising while its peers wait in a barrier would.  With
    the failure injected on rank 1 both ranks warn, run the loop over the group's all-reduce and get the plain run's bits;
    with the fine-grained export refused the exchange itself still runs, on ordinary device memory."""
    import torch.multiprocessing as mp
    mp.spawn(_fallback_worker, args=(2, _free_port(), str(tmp_path), mode), nprocs=2, join=True)
    res = [numpy.load(str(tmp_path / ("rank%d.npz" % r))) for r in range(2)]
    for r in res:
        assert int(r["equal"]) == 1 and list(r["iters"]) == list(res[0]["iters"]) and numpy.array_equal(r["ln"], res[0]["ln"])
        if mode == "fail":
            assert int(r["warned"]) == 1 and str(r["route"]).startswith("rccl")
        else:
            assert int(r["warned"]) == 0 and str(r["route"]) == "oneshot" and int(r["fine"]) == 0
