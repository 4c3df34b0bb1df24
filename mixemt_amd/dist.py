"""
Multi-GPU EM: one process per GPU, `torch.distributed` (backend "nccl" = RCCL
over xGMI on ROCm; "gloo" in the CPU tests).

The reference is single-process (SURVEY.md section 8 e); what shards is the
matrix's rows: the E-step is row-local (em.py:80-83) and the M-step is a
weighted column sum over rows (em.py:87-88).  So each rank keeps a contiguous
row block of the matrix / weights / observations resident, and an EM
iteration has exactly ONE exchange step:

    colsum_local[b][h] = sum_{r in shard} w_r P[r][h] / Z_b[r]     (mxm_em_iter)
    all-reduce(SUM, fp64, B*H values)                              (RCCL)
    ln p' = ln p + ln colsum - ln sum_h p colsum ; L1 test ; state (mxm_m_finalize)

The all-reduce hands every rank the same bytes, so every rank takes the same
stop decision on the same iteration; kernels of a finished restart are no-ops,
which lets the host look at the loop state only every `check_every`
iterations without overshooting the reference's stopping iteration.

Two modes:
    run_em_sharded            rows sharded, every restart on every rank (configs 2-4)
    run_em_restart_parallel   matrix replicated, restarts dealt round-robin
                              over ranks, no per-iteration traffic (config 5)
"""

import math

import numpy

from . import em as _em

try:
    import torch
    import torch.distributed as dist
except ImportError:          # pragma: no cover
    torch = None
    dist = None


def shard_bounds(n_rows, rank, world):
    """Contiguous row block [lo, hi) of rank `rank`; sizes differ by at most 1."""
    base, extra = divmod(int(n_rows), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def _world(group):
    if dist is None or not dist.is_available() or not dist.is_initialized():
        return 0, 1
    return dist.get_rank(group), dist.get_world_size(group)


def _collective(group):
    """True when a process group exists: the collectives are then issued even for a world of one
    (a no-op exchange), so that a single-GPU box drives the same calls as an 8-GPU node."""
    return dist is not None and dist.is_available() and dist.is_initialized()


def broadcast_inits(n_runs, n_haps, alpha, device, group=None, src=0):
    """
    The restarts' initial proportions are sequential draws from numpy's global
    legacy RNG on ONE process (em.py:36, :123); rank `src` draws, all receive.
    """
    rank, world = _world(group)
    if rank == src:
        host = numpy.stack([_em.init_props(n_haps, alpha=alpha) for _ in range(n_runs)])
    else:
        host = numpy.empty((n_runs, n_haps))
    buf = torch.from_numpy(numpy.ascontiguousarray(host)).to(device)
    if _collective(group):
        dist.broadcast(buf, src=src, group=group)
    return buf.cpu().numpy()


def sharded_em_loop(plan, inits, tolerance, max_iter, group=None, check_every=8, compact=True):
    """
    The EM loop over a row-sharded matrix.  `plan` is the rank-local EmPlan (or
    any object with its em_iter / finalize / alloc / read_state surface -- the
    CPU tests pass a numpy-backed double).  Returns
    (ln_cur = log theta_k, ln_new = log theta_{k+1}, [(done, iters, l1)]) --
    identical on every rank.

    compact: restarts stop on different iterations; the ones still running are
    kept packed in the leading slots of the loop vectors (as mxm_em_loop does on
    one GPU), so an iteration takes ceil(running / tile) passes over the shard
    and all-reduces only the running restarts' sums.  Every rank sees the same
    state, hence takes the same packing decisions.
    """
    exchange = _collective(group)
    ln0, p0 = _em.log_inits(inits)
    n_runs = ln0.shape[0]
    props_cur = plan.alloc_props(p0)
    ln_cur = plan.alloc_props(ln0)
    ln_new = plan.alloc_props(ln0)
    colsum = plan.alloc_props(numpy.zeros_like(ln0))
    state = plan.alloc_state(n_runs).view(n_runs, -1)          # one row per restart
    vectors = (props_cur, ln_cur, ln_new, colsum, state)
    slot_run = list(range(n_runs))                             # slot -> caller's run index
    lead = n_runs
    issued = 0
    states = plan.read_state(state)
    while issued < max_iter:
        running = [s for s in range(lead) if states[s][0] == 0]
        if not running:
            break
        if compact and len(running) < lead:
            order = running + [s for s in range(n_runs) if s not in set(running)]
            idx = torch.as_tensor(order, device=props_cur.device)
            for vec in vectors:
                vec.copy_(vec[idx])
            slot_run = [slot_run[s] for s in order]
            states = [states[s] for s in order]
            lead = len(running)
        burst = min(check_every, max_iter - issued)
        for _ in range(burst):
            plan.em_iter(props_cur[:lead], ln_cur[:lead], state[:lead], colsum[:lead])
            if exchange:
                dist.all_reduce(colsum[:lead], op=dist.ReduceOp.SUM, group=group)
            plan.finalize(colsum[:lead], ln_cur[:lead], ln_new[:lead], props_cur[:lead], state[:lead],
                          tolerance, max_iter)
        issued += burst
        states = plan.read_state(state)
    if slot_run != list(range(n_runs)):                        # back to the caller's run order
        back = [0] * n_runs
        for slot, run in enumerate(slot_run):
            back[run] = slot
        idx = torch.as_tensor(back, device=props_cur.device)
        for vec in vectors:
            vec.copy_(vec[idx])
        states = [states[s] for s in back]
    return ln_cur, ln_new, states


def run_em_sharded(local_mat, local_weights, args, inits=None, group=None, want_read_mix=True,
                   check_every=8, storage=None):
    """
    run_em (em.py:94-165) over a matrix whose rows are spread over the ranks of
    `group`; each rank passes ITS row block and gets back the global proportions
    plus ITS block of the posterior matrix.  Same dict as em.run_em_ex.
    """
    n_multi = int(args.n_multi)
    plan = _em.EmPlan(local_mat, local_weights, n_runs=n_multi,
                      storage=storage or getattr(args, "storage", "f64"))
    if inits is None:
        inits = broadcast_inits(n_multi, plan.n_haps, args.init_alpha, plan.dev, group)
    inits = numpy.ascontiguousarray(inits, dtype=numpy.float64)
    ln_cur, ln_new, states = sharded_em_loop(plan, inits, args.tolerance, args.max_iter,
                                             group=group, check_every=check_every)
    return _em.collect_result(plan, inits, ln_cur, ln_new, states, want_read_mix, reuse_linear=True)


def run_em_restart_parallel(mat, weights, args, inits=None, group=None, want_read_mix=True):
    """
    Config 5: the matrix is replicated, the n_multi restarts are dealt
    round-robin over the ranks (run i -> rank i % world) and run with no
    per-iteration communication.  At the end the per-run log-proportions are
    summed with one all-reduce; the posterior fold (em.py:156, log-mean-exp over
    runs) is completed in linear space with one all-reduce of the local folds.
    Every rank returns the full result.
    """
    rank, world = _world(group)
    n_multi = int(args.n_multi)
    plan = _em.EmPlan(mat, weights, n_runs=max(1, (n_multi + world - 1) // world))
    if inits is None:
        inits = broadcast_inits(n_multi, plan.n_haps, args.init_alpha, plan.dev, group)
    inits = numpy.ascontiguousarray(inits, dtype=numpy.float64)
    mine = list(range(rank, n_multi, world))
    n_haps = plan.n_haps
    ln_sum = torch.zeros(n_haps, dtype=torch.float64, device=plan.dev)
    iters = torch.zeros(n_multi, dtype=torch.int64, device=plan.dev)
    run_props = torch.zeros((n_multi, n_haps), dtype=torch.float64, device=plan.dev)
    fold = None
    if mine:
        ln_cur, ln_new, states = _em.em_loop(plan, inits[mine], args.tolerance, args.max_iter)
        ln_k = ln_cur.cpu().numpy()
        ln_sum += ln_new.sum(dim=0)
        if want_read_mix:
            fold = plan.release_linear()              # the loop is over: reuse its 43 GB
        for j, run in enumerate(mine):
            iters[run] = states[j][1]
            run_props[run] = torch.exp(ln_new[j])
            if want_read_mix:
                fold = _em.posterior(plan, ln_k[j], out=fold, fold=(j > 0))
    if _collective(group):
        dist.all_reduce(ln_sum, group=group)
        dist.all_reduce(iters, group=group)
        dist.all_reduce(run_props, group=group)
    read_mix = None
    if want_read_mix:
        # in place: at 10^6 x 5408 every extra copy of the posterior is 43 GB
        read_mix = fold.exp_() if fold is not None else torch.zeros(
            (plan.n_rows, n_haps), dtype=torch.float64, device=plan.dev)
        if _collective(group):
            dist.all_reduce(read_mix, group=group)
        read_mix.log_()
        if n_multi > 1:
            read_mix -= math.log(n_multi)
    props = torch.exp(ln_sum / n_multi).cpu().numpy() if n_multi > 1 else run_props[0].cpu().numpy()
    return {"props": props, "read_mix": read_mix, "iters": [int(x) for x in iters.cpu()],
            "run_props": run_props.cpu().numpy(), "inits": inits}
