"""
world_size-2 gloo runs of the row-sharded EM loop (mixemt_amd.dist) on CPU:
the orchestration -- shard bounds, one all-reduce of the column sums per
iteration, identical stop decision on every rank, init broadcast -- with the
oracle-backed CpuPlan standing in for the device kernels.
"""
import os
import socket

import numpy
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import em_args, golden
from mixemt_amd import dist as mdist


def test_shard_bounds_cover_rows_exactly():
    for n, world in ((10, 3), (7, 8), (1000000, 8), (0, 2), (5, 1)):
        spans = [mdist.shard_bounds(n, r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        sizes = [hi - lo for lo, hi in spans]
        assert max(sizes) - min(sizes) <= 1


def test_single_process_loop_matches_oracle_run():
    """world = 1 (no process group): the loop alone reproduces the oracle's run_em."""
    from _cpu_plan import CpuPlan
    from oracle import em_oracle
    g = golden("g7_config1")
    mat, wts = g["mat"][:200], numpy.ones(200)
    numpy.random.seed(3)
    init = numpy.random.dirichlet([1.0] * mat.shape[1])[None, :]
    cur, new, states = mdist.sharded_em_loop(CpuPlan(mat, wts), init, 1e-4, 10000, check_every=5)
    trace = []
    numpy.random.seed(3)
    props, _ = em_oracle.run_em(mat, wts, em_args(), trace=trace)
    assert states[0][0] == 1 and states[0][1] == trace[0]["iters"]
    assert numpy.abs(numpy.exp(new[0].numpy()) - props).max() < 1e-12


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from _cpu_plan import CpuPlan
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = numpy.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g7_config1.npz"))
        mat = g["mat"][:301]
        wts = (numpy.arange(301) % 3 + 1).astype(numpy.float64)
        lo, hi = mdist.shard_bounds(301, rank, world)
        plan = CpuPlan(mat[lo:hi], wts[lo:hi])
        numpy.random.seed(100 + rank)            # ranks deliberately disagree: rank 0's draw must win
        inits = mdist.broadcast_inits(2, mat.shape[1], 1.0, torch.device("cpu"))
        cur, new, states = mdist.sharded_em_loop(plan, inits, 1e-4, 10000, check_every=7)
        numpy.savez(os.path.join(out_dir, "rank%d.npz" % rank), inits=inits, cur=cur.numpy(),
                    new=new.numpy(), states=numpy.array(states), calls=plan.calls)
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_matches_single_process(tmp_path):
    from _cpu_plan import CpuPlan
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    res = [numpy.load(str(tmp_path / ("rank%d.npz" % r))) for r in range(world)]
    # every rank saw rank 0's init draws and took identical decisions
    numpy.random.seed(100)
    want_inits = numpy.stack([numpy.random.dirichlet([1.0] * 100) for _ in range(2)])
    for r in res:
        assert numpy.array_equal(r["inits"], want_inits)
        assert numpy.array_equal(r["states"], res[0]["states"])
        assert numpy.array_equal(r["new"], res[0]["new"]) and numpy.array_equal(r["cur"], res[0]["cur"])
        assert int(r["calls"]) == int(res[0]["calls"])
    # and the sharded result equals the unsharded one
    g = golden("g7_config1")
    mat = g["mat"][:301]
    wts = (numpy.arange(301) % 3 + 1).astype(numpy.float64)
    cur, new, states = mdist.sharded_em_loop(CpuPlan(mat, wts), want_inits, 1e-4, 10000, check_every=7)
    assert [s[1] for s in states] == [int(s[1]) for s in res[0]["states"]]
    assert all(s[0] == 1 for s in states)
    assert numpy.abs(numpy.exp(new.numpy()) - numpy.exp(res[0]["new"])).max() < 1e-12
    assert numpy.abs(numpy.exp(cur.numpy()) - numpy.exp(res[0]["cur"])).max() < 1e-12


def test_running_restarts_are_packed_in_the_sharded_loop():
    """
    Restarts stop on different iterations; the loop keeps the running ones in the leading
    slots (fewer restarts per em_iter / all-reduce afterwards) and hands the results back in
    the caller's run order -- same states and vectors as the unpacked schedule.
    """
    from _cpu_plan import CpuPlan
    g = golden("g7_config1")
    mat = g["mat"][:150]
    wts = (numpy.arange(150) % 2 + 1).astype(numpy.float64)
    numpy.random.seed(5)
    inits = numpy.stack([numpy.random.dirichlet([0.5] * mat.shape[1]) for _ in range(5)])
    packed_plan, plain_plan = CpuPlan(mat, wts), CpuPlan(mat, wts)
    cur_a, new_a, st_a = mdist.sharded_em_loop(packed_plan, inits, 1e-4, 10000, check_every=4, compact=True)
    cur_b, new_b, st_b = mdist.sharded_em_loop(plain_plan, inits, 1e-4, 10000, check_every=4, compact=False)
    iters = [s[1] for s in st_b]
    assert len(set(iters)) > 1                       # the case exercises the packing
    assert st_a == st_b
    assert numpy.array_equal(new_a.numpy(), new_b.numpy()) and numpy.array_equal(cur_a.numpy(), cur_b.numpy())
    assert packed_plan.calls == plain_plan.calls
    assert packed_plan.restart_steps < plain_plan.restart_steps


def test_window_runs_one_tile_at_a_time_with_identical_results():
    """
    window = 2 of 5 restarts: only two iterate at a time, a stopped one is replaced by a waiting
    one; every restart counts its own iterations, so states and vectors equal the all-together
    schedule bit for bit while no em_iter call ever carries more than the window.
    """
    from _cpu_plan import CpuPlan
    g = golden("g7_config1")
    mat = g["mat"][:120]
    wts = (numpy.arange(120) % 3 + 1).astype(numpy.float64)
    numpy.random.seed(6)
    inits = numpy.stack([numpy.random.dirichlet([0.5] * mat.shape[1]) for _ in range(5)])
    win_plan, all_plan = CpuPlan(mat, wts), CpuPlan(mat, wts)
    cur_a, new_a, st_a = mdist.sharded_em_loop(win_plan, inits, 1e-4, 10000, check_every=4, window=2)
    cur_b, new_b, st_b = mdist.sharded_em_loop(all_plan, inits, 1e-4, 10000, check_every=4, compact=False)
    assert st_a == st_b and all(s[0] == 1 for s in st_a)
    assert numpy.array_equal(new_a.numpy(), new_b.numpy()) and numpy.array_equal(cur_a.numpy(), cur_b.numpy())
    assert win_plan.widest == 2 and all_plan.widest == 5


def test_max_iter_is_per_restart_under_the_window():
    from _cpu_plan import CpuPlan
    g = golden("g7_config1")
    mat = g["mat"][:60]
    numpy.random.seed(8)
    inits = numpy.stack([numpy.random.dirichlet([1.0] * mat.shape[1]) for _ in range(3)])
    cur, new, states = mdist.sharded_em_loop(CpuPlan(mat, numpy.ones(60)), inits, 0.0, 7, check_every=3, window=1)
    assert [s[:2] for s in states] == [(2, 7)] * 3           # each one ran its own 7 iterations


def _numpy_fold(acc, pieces, delta):
    out = acc.numpy()
    for piece in pieces:
        numpy.logaddexp(out, piece.numpy(), out=out)
    out += delta


def _exchange_worker(rank, world, port, out_dir, n_rows, n_haps, n_with_fold, chunk_bytes):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = numpy.random.default_rng(500 + rank)
        has_fold = [r < n_with_fold for r in range(world)]
        fold = None
        if has_fold[rank]:
            host = rng.normal(size=(n_rows, n_haps)) * 30.0 - 800.0      # far below exp(-745): log space matters
            fold = torch.from_numpy(host.copy())
            numpy.save(os.path.join(out_dir, "fold%d.npy" % rank), host)
        block = mdist.exchange_fold_blocks(fold, has_fold, n_rows, n_haps, -0.25, torch.device("cpu"),
                                           chunk_bytes=chunk_bytes, fold_fn=_numpy_fold)
        numpy.save(os.path.join(out_dir, "block%d.npy" % rank), block.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_with_fold,n_rows,chunk_bytes", [(2, 2, 37, 1 << 30), (3, 3, 50, 7 * 11 * 8 * 2),
                                                                  (3, 2, 20, 5 * 11 * 8), (2, 1, 9, 1 << 30)])
def test_row_block_exchange_is_a_log_space_reduce_scatter(tmp_path, world, n_with_fold, n_rows, chunk_bytes):
    """
    Config 5's end-of-run combine: every rank ends with its row block of logaddexp over all
    ranks' folds (+ delta), also when the blocks move in several chunks, when a rank has no
    fold of its own, and for values whose exp() underflows.
    """
    n_haps = 11
    mp.spawn(_exchange_worker, args=(world, _free_port(), str(tmp_path), n_rows, n_haps, n_with_fold, chunk_bytes),
             nprocs=world, join=True)
    folds = [numpy.load(str(tmp_path / ("fold%d.npy" % r))) for r in range(n_with_fold)]
    want = folds[0].copy()
    for other in folds[1:]:
        want = numpy.logaddexp(want, other)
    want -= 0.25
    assert numpy.isfinite(want).all() and (numpy.exp(want) == 0).any()
    for r in range(world):
        lo, hi = mdist.shard_bounds(n_rows, r, world)
        got = numpy.load(str(tmp_path / ("block%d.npy" % r)))
        assert got.shape == (hi - lo, n_haps)
        assert numpy.allclose(got, want[lo:hi], rtol=0, atol=1e-12)


def test_sharded_loop_raises_instead_of_spinning_when_finalize_never_stops_a_restart():
    """sharded_em_loop bounds its bursts; a plan whose finalize ignores max_iter is an error, not a hang."""
    from _cpu_plan import CpuPlan
    from mixemt_amd import dist as mdist

    class NeverStops(CpuPlan):
        def finalize(self, colsum, ln_cur, ln_new, props_cur, state, tol, max_iter):
            CpuPlan.finalize(self, colsum, ln_cur, ln_new, props_cur, state, -1.0, 1 << 30)   # never converged, never capped

    rng = numpy.random.default_rng(3)
    mat = rng.normal(-20.0, 5.0, size=(30, 12))
    inits = rng.dirichlet([1.0] * 12, size=3)
    plan = NeverStops(mat, numpy.ones(30))
    with pytest.raises(RuntimeError, match="still running"):
        mdist.sharded_em_loop(plan, inits, 1e-4, 20, check_every=4, window=2)
    assert plan.calls <= (2 * (5 + 1) + 1) * 4
    # and the cap never bites a well-behaved plan: three restarts, window 1, per-restart max_iter
    good = CpuPlan(mat, numpy.ones(30))
    _, _, states = mdist.sharded_em_loop(good, inits, 0.0, 9, check_every=4, window=1)
    assert [s[:2] for s in states] == [(2, 9)] * 3
