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
                              over ranks, no per-iteration traffic (config 5); at the
                              end the folded posteriors are combined by a direct
                              row-block exchange (every rank sends block j straight to
                              rank j -- xGMI is point to point, all links carry one
                              block each) and a log-space fold kernel, so that each
                              rank ends up with ITS row block of the result
"""

import math
import time
import warnings

import numpy

from . import _lib
from . import em as _em
from ._dev import require_gpu

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


class ExchangeUnavailable(RuntimeError):
    """The one-shot exchange could not be set up on at least one rank; EVERY rank of the group raises this together
    (the outcome of create / connect is agreed with a MIN all-reduce), so a caller may fall back to the all-reduce."""


def _agree(ok, group, device):
    """True iff `ok` holds on every rank of the group (one tiny MIN all-reduce; every rank must call it)."""
    if not _collective(group):
        return bool(ok)
    on_gpu = dist.get_backend(group) == "nccl"
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device if on_gpu else "cpu")
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    return bool(int(flag.item()))


class OneShotExchange(object):
    """
    The library's one-shot exchange (include/mixemt_hip.h, mxm_exchange_*; csrc/exchange.hpp) for a process group: every
    rank's buffer mapped into every rank, the handles all-gathered over the group (any backend: 64 bytes per rank, once).
    reduce(colsum, state): push + pull on the current stream -- colsum becomes the sum over the ranks in rank order, the
    same bits everywhere.  Opt-in: sharded_em_loop(exchange="oneshot"); RCCL's all-reduce is the default.  Exercised with
    several processes on one GPU; UNMEASURED over xGMI.
    Setting it up cannot leave the group half way (round 6): a rank whose mxm_exchange_create / _connect fails does not
    raise on its own while its peers wait in a collective -- after each of the two steps all ranks agree on the outcome
    (MIN all-reduce) and, if any failed, ALL release what they hold and raise ExchangeUnavailable (sharded_em_loop then
    warns and uses the all-reduce).  The library itself retries with ordinary device memory when the runtime will not
    export a fine-grained buffer (`fine_grained` says which one it got).  A pull waits at most 3 s for a peer
    (MXM_EXCHANGE_TIMEOUT_MS in the environment sets another bound), then poisons the sums and raises error 2.
    """

    def __init__(self, n_doubles, group=None, split=False):
        import ctypes
        self.split = bool(split)
        from . import _lib
        self.lib = _lib.load()
        self.handle = None
        rank, world = _world(group)
        self.rank, self.world = rank, world
        dev = require_gpu()
        hb = int(self.lib.mxm_exchange_handle_bytes())
        mine = ctypes.create_string_buffer(hb)
        handle = ctypes.c_void_p()
        rc = self.lib.mxm_exchange_create(world, rank, int(n_doubles), ctypes.byref(handle), mine)
        why = self.lib.mxm_last_error().decode("utf-8", "replace") if rc != 0 else ""
        if rc == 0:
            self.handle = handle
        if not _agree(rc == 0, group, dev):
            self.close()
            raise ExchangeUnavailable("one-shot exchange: mxm_exchange_create failed on %s" % ("this rank (%d): %s" % (rank, why) if rc != 0 else "another rank"))
        every = [None] * world
        if world > 1:
            dist.all_gather_object(every, bytes(mine.raw), group=group)
        else:
            every = [bytes(mine.raw)]
        blob = ctypes.create_string_buffer(b"".join(every), hb * world)
        rc = self.lib.mxm_exchange_connect(handle, blob)
        why = self.lib.mxm_last_error().decode("utf-8", "replace") if rc != 0 else ""
        if not _agree(rc == 0, group, dev):
            if world > 1:
                dist.barrier(group=group)              # (nobody unmaps a buffer a peer is still mapping)
            self.close()
            raise ExchangeUnavailable("one-shot exchange: mxm_exchange_connect failed on %s" % ("this rank (%d): %s" % (rank, why) if rc != 0 else "another rank"))
        if world > 1:
            dist.barrier(group=group)                  # nobody pushes into a buffer its owner has not finished setting up
        self.n_doubles = int(n_doubles)
        fine, nbytes = ctypes.c_int32(0), ctypes.c_int64(0)
        self.lib.mxm_exchange_info(handle, ctypes.byref(fine), ctypes.byref(nbytes))
        self.fine_grained, self.bytes = bool(fine.value), int(nbytes.value)

    def reduce(self, colsum, state=None):
        from . import _lib
        from ._dev import current_stream
        n = colsum.numel()
        nb = colsum.shape[0] if colsum.dim() > 1 else 1
        if not colsum.is_contiguous() or n > self.n_doubles:
            raise ValueError("one-shot exchange: a contiguous block of at most %d doubles" % self.n_doubles)
        stream = current_stream()
        state_ptr, nb = (state.data_ptr(), nb) if state is not None else (None, 0)
        if self.split:                                     # (two launches: kept for A/B and as the documented primitive pair)
            _lib.check(self.lib.mxm_exchange_push(self.handle, colsum.data_ptr(), n, stream), "mxm_exchange_push")
            _lib.check(self.lib.mxm_exchange_pull(self.handle, colsum.data_ptr(), n, state_ptr, nb, stream), "mxm_exchange_pull")
        else:
            _lib.check(self.lib.mxm_exchange_reduce(self.handle, colsum.data_ptr(), n, state_ptr, nb, stream), "mxm_exchange_reduce")

    def close(self):
        handle, self.handle = getattr(self, "handle", None), None
        if handle:
            self.lib.mxm_exchange_destroy(handle)

    def __del__(self):
        try:
            self.close()
        except Exception:                                  # (interpreter shutdown: the runtime may be gone already)
            pass


GRAPH_AUTO_ISSUE_SHARE = 0.5       # graph="auto": replay bursts from a hipGraph once the host needs more than this share
                                   # of a burst's wall time just to ENQUEUE it (the device would otherwise wait for Python)


SHARD_QUADS_MIN_ROWS = 60000        # sharded_em_loop: attach a quad dictionary to a records shard of at least this many rows


def sharded_em_loop(plan, inits, tolerance, max_iter, group=None, check_every=8, compact=True,
                    window=None, verify=True, graph="auto", exchange="rccl"):
    """
    The EM loop over a row-sharded matrix.  `plan` is the rank-local EmPlan (or
    any object with its em_iter / finalize / alloc / read_state surface -- the
    CPU tests pass a numpy-backed double).  Returns
    (ln_cur = log theta_k, ln_new = log theta_{k+1}, [(done, iters, l1)]) --
    identical on every rank.  The plan's finalize() must mark a restart done (1 converged, 2 at max_iter) -- as
    mxm_m_finalize does; the loop raises instead of spinning if it never does.

    compact: restarts stop on different iterations; the unfinished ones are
    kept packed in the leading slots of the loop vectors.  window: at most that
    many of them iterate at a time (default: the plan's restart tile, i.e. one
    full pass over the shard per iteration), dealt round-robin over the running
    restarts chunk by chunk like mxm_em_loop's schedule 2; each restart counts
    its own iterations (mxm_m_finalize), so when it runs changes nothing in its
    result.  Only the iterating restarts' sums are all-reduced.  Every rank sees
    the same state, hence takes the same scheduling decisions.
    verify: at every state check the ranks compare (done, iters) of all restarts
    (two tiny all-reduces, MIN and MAX) and fail loudly if they ever disagree --
    the loop's correctness rests on bit-identical all-reduce results on all ranks.
    graph: replay each burst of `check_every` iterations -- streaming kernel, column reduce, all-reduce,
    finalize -- from ONE captured hipGraph instead of enqueueing 4 x check_every operations from Python, for shards so
    small that the host would otherwise be what a step waits for.  "auto" (default) decides by a rule the loop applies
    itself: the first TWO bursts run eagerly and the SECOND is timed (the first is cold: lazy code-object loading of the
    kernels and, without a broadcast before the loop, RCCL's communicator set-up sit inside it and say nothing about the
    steady state) -- how long the host took to ENQUEUE it against how long
    it took until its state came back; above GRAPH_AUTO_ISSUE_SHARE (bench.py: ~40 us of issue per step against
    0.83 ms at 125 000 dense rows per GPU -> eager; against 0.27 ms of a records shard it is close) the next bursts are
    captured.  Every rank must take the same decision (a captured collective and an eager one do not pair up), so the
    shares are max-reduced over the ranks first.  True / False force it.  A burst whose set of iterating restarts
    changed is re-captured; whenever capture is not possible (gloo, CPU tensors, a backend that refuses collectives
    under capture) the loop logs once and stays eager.  Results are bit-identical either way (same kernels, same order).
    exchange: "rccl" (default) = torch.distributed's all-reduce of the iterating restarts' sums; "oneshot" = the library's
    one-shot exchange (OneShotExchange: every rank writes its sums into every rank's buffer, each sums them in rank order;
    kernels only, so a burst is capturable whatever the group's backend) -- an OneShotExchange may also be passed in, to be
    reused over several loops.  Opt-in; exercised with several processes on one GPU, unmeasured over xGMI.
    If the exchange cannot be set up on ANY rank (hipIpc refused, a device without peer access), every rank falls back to
    the all-reduce together, with a RuntimeWarning (OneShotExchange / ExchangeUnavailable; sharded_em_loop.last_exchange
    says which one ran).  A pull waits at most 3 s for a peer's sums before it poisons them and the loop raises
    (MXM_EXCHANGE_TIMEOUT_MS in the environment, read when the exchange is created, sets another bound -- a rank under a
    debugger, a host stalled in a graph instantiation); an exchange this call created is released on every exit, after
    draining the stream and a bounded wait for the peers.
    """
    oneshot, own_exchange = None, False
    if exchange == "oneshot":
        if not hasattr(plan, "lib"):
            raise ValueError("the one-shot exchange needs a device plan")
        try:
            oneshot = OneShotExchange(int(numpy.asarray(inits).shape[0]) * int(numpy.asarray(inits).shape[1]), group)
            own_exchange = True
        except ExchangeUnavailable as exc:             # raised on EVERY rank together: all take the all-reduce instead
            warnings.warn("sharded EM loop: %s; using the group's all-reduce" % (exc,), RuntimeWarning, stacklevel=2)
            exchange = "rccl"
        sharded_em_loop.last_exchange = "oneshot" if oneshot is not None else "rccl (one-shot exchange unavailable)"
    elif isinstance(exchange, OneShotExchange):
        oneshot = exchange
    if exchange != "rccl" and oneshot is None:
        raise ValueError("exchange must be 'rccl', 'oneshot' or an OneShotExchange")
    grouped = _collective(group)                       # a process group exists: decisions are agreed over it
    exchange = grouped or oneshot is not None
    try:
        # This loop always runs the per-iteration kernels (the exchange sits between the row pass and the finalize), so over
        # records a quad dictionary has no one-launch loop to beat and pays from far fewer rows than in run_em's own loop
        # (em.EmPlan.attach_quads; SHARD_QUADS_MIN_ROWS byte-coded rows in the shard)
        if getattr(plan, "coded", None) is not None and hasattr(plan, "attach_quads") and _em.QUADS == "auto":
            # (three or more restarts: tiles of three share a pass -- worth the dictionary from far fewer rows, em.QUADS_MIN_ROWS_MULTI)
            few = int(numpy.asarray(inits).shape[0]) < 3
            plan.attach_quads("auto", min_rows=SHARD_QUADS_MIN_ROWS if few else min(SHARD_QUADS_MIN_ROWS, _em.QUADS_MIN_ROWS_MULTI))
        ln0, p0 = _em.log_inits(inits)
        n_runs = ln0.shape[0]
        if window is None:
            window = plan.restart_tile() if (compact and hasattr(plan, "restart_tile")) else n_runs
        window = max(1, int(window))
        props_cur = plan.alloc_props(p0)
        ln_cur = plan.alloc_props(ln0)
        ln_new = plan.alloc_props(ln0)
        colsum = plan.alloc_props(numpy.zeros_like(ln0))
        state = plan.alloc_state(n_runs).view(n_runs, -1)          # one row per restart
        vectors = (props_cur, ln_cur, ln_new, colsum, state)
        slot_run = list(range(n_runs))                             # slot -> caller's run index
        states = plan.read_state(state)
        if max_iter <= 0:
            if own_exchange:
                oneshot.close()
            return ln_cur, ln_new, states
        first = True
        eager_bursts = 0                                   # graph="auto" takes its decision from the second eager burst
        captured = None                                    # None: nothing captured yet; False: capture refused; (lead, graph)
        graph_bursts = 0
        sharded_em_loop.last_issue_share = None
        sharded_em_loop.last_issue_burst = None
        # The loop ends when `finalize` has marked every restart done (converged, or at its own max_iter).  A plan whose
        # finalize never does that must not spin for ever: a restart needs at most ceil(max_iter / check_every) bursts, and
        # at most ceil(n_runs / window) groups of restarts take turns.
        bursts_left = -(-n_runs // window) * (-(-int(max_iter) // int(check_every)) + 1) + 1
        while True:
            running = [s for s in range(n_runs) if states[s][0] == 0]        # slot order = round-robin order
            if not running:
                break
            bursts_left -= 1
            if bursts_left < 0:
                raise RuntimeError("sharded EM loop: restarts %s are still running after every burst max_iter=%d allows "
                                   "(the plan's finalize() must stop a restart at its max_iter)"
                                   % ([slot_run[s] for s in running], max_iter))
            lead = min(len(running), window) if compact else n_runs
            if compact:
                # one full tile per iteration, dealt round-robin chunk by chunk: the tile that just ran goes
                # to the back of the queue, so all restarts advance at the same rate and every pass over the
                # shard carries a full tile until fewer than a tile's worth are left (as mxm_em_loop does)
                if not first and len(running) > window:
                    was = [s for s in range(window) if states[s][0] == 0]   # the tile that just ran, still running
                    rest = [s for s in running if s >= window]
                    running = rest + was
                order = running + [s for s in range(n_runs) if states[s][0] != 0]
                if order != list(range(n_runs)):
                    idx = torch.as_tensor(order, device=props_cur.device)
                    for vec in vectors:
                        vec.copy_(vec[idx])
                    slot_run = [slot_run[s] for s in order]
                    states = [states[s] for s in order]
            def burst():
                for _ in range(check_every):
                    plan.em_iter(props_cur[:lead], ln_cur[:lead], state[:lead], colsum[:lead])
                    if oneshot is not None:
                        oneshot.reduce(colsum[:lead], state[:lead])
                    elif exchange:
                        dist.all_reduce(colsum[:lead], op=dist.ReduceOp.SUM, group=group)
                    plan.finalize(colsum[:lead], ln_cur[:lead], ln_new[:lead], props_cur[:lead], state[:lead],
                                  tolerance, max_iter)

            use_graph = (graph is True and not first and captured is not False and props_cur.is_cuda
                         and (not exchange or oneshot is not None or dist.get_backend(group) == "nccl"))
            if use_graph and (captured is None or captured[0] != lead):
                # (re)capture: the burst's launches are recorded, not run; a failure leaves the loop eager for good
                #  only what a refused CAPTURE raises is taken as "not capturable here" -- torch reports those as
                # RuntimeError (hipErrorStreamCapture*), RCCL as DistBackendError (a RuntimeError); argument / library errors
                # of the plan (ValueError from _lib.check) are real and propagate.  The fallback is logged once.
                try:
                    torch.cuda.synchronize()
                    cg = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(cg, capture_error_mode="thread_local"):
                        burst()
                    captured = (lead, cg, _burst_ptrs(vectors, plan))
                except RuntimeError as exc:
                    captured = False
                    torch.cuda.synchronize()
                    warnings.warn("sharded EM loop: hipGraph capture of a burst was refused (%s); staying eager" % (exc,),
                                  RuntimeWarning, stacklevel=2)
            first = False
            if use_graph and captured:
                # a captured burst replays raw pointers: the loop vectors and the plan's buffers must still be the ones
                # it recorded (they are never reallocated inside this loop; this makes the convention a check)
                if captured[2] != _burst_ptrs(vectors, plan):
                    raise RuntimeError("sharded EM loop: a buffer of the captured burst was reallocated between bursts")
                captured[1].replay()
                graph_bursts += 1
                states = plan.read_state(state)
            else:
                t_burst = time.perf_counter()
                burst()
                t_issue = time.perf_counter() - t_burst
                states = plan.read_state(state)                  # (synchronises: the burst's state is back)
                t_wall = time.perf_counter() - t_burst
                eager_bursts += 1
                if eager_bursts == 2 and graph == "auto":
                    share = t_issue / max(t_wall, 1e-9)
                    if grouped:                                  # one decision for all ranks
                        agreed = torch.tensor([share], dtype=torch.float64, device=props_cur.device)
                        dist.all_reduce(agreed, op=dist.ReduceOp.MAX, group=group)
                        share = float(agreed.item())
                    graph = bool(share > GRAPH_AUTO_ISSUE_SHARE and props_cur.is_cuda
                                 and (not grouped or oneshot is not None or dist.get_backend(group) == "nccl"))
                    sharded_em_loop.last_issue_share = share
                    sharded_em_loop.last_issue_burst = eager_bursts
            if grouped and verify:
                _assert_ranks_agree(states, props_cur.device, group)
        if slot_run != list(range(n_runs)):                        # back to the caller's run order
            back = [0] * n_runs
            for slot, run in enumerate(slot_run):
                back[run] = slot
            idx = torch.as_tensor(back, device=props_cur.device)
            for vec in vectors:
                vec.copy_(vec[idx])
            states = [states[s] for s in back]
        sharded_em_loop.last_graph_bursts = graph_bursts
        if own_exchange:                                   # made here: unmapped here, once every rank is through its last pull
            torch.cuda.synchronize()
            if grouped:
                dist.barrier(group=group)
            oneshot.close()
        return ln_cur, ln_new, states
    except BaseException:
        #  an exchange made here is never left to __del__: that would unmap this rank's buffer, with no
        # barrier, while peers may still push into it -- a fault on THEIR side.  Best effort: drain this rank's stream,
        # meet the peers if they can still be met (bounded), then release.
        if own_exchange and oneshot is not None:
            _release_exchange(oneshot, grouped, group)
        raise


def _release_exchange(oneshot, grouped, group, seconds=5.0):
    import datetime
    try:
        torch.cuda.synchronize()
    except Exception:
        pass
    if grouped:
        try:
            work = dist.barrier(group=group, async_op=True)
            work.wait(timeout=datetime.timedelta(seconds=seconds))
        except Exception:
            pass
    oneshot.close()


sharded_em_loop.last_exchange = None         # "oneshot" / "rccl (one-shot exchange unavailable)" when the last call asked for the one-shot exchange
sharded_em_loop.last_graph_bursts = 0        # bursts the last call replayed from a captured graph (diagnostic)
sharded_em_loop.last_issue_share = None      # graph="auto": host enqueue time / wall time of the burst it decided on (max over ranks)
sharded_em_loop.last_issue_burst = None      # ... which eager burst that was (2: the first warm one); None = the loop ended before


def _burst_ptrs(vectors, plan):
    """Addresses a captured burst bakes in: the loop vectors and the plan's scratch / matrix buffers."""
    ptrs = [v.data_ptr() for v in vectors]
    for name in ("ws", "lin", "mat", "wts"):
        buf = getattr(plan, name, None)
        if buf is not None and hasattr(buf, "data_ptr"):
            ptrs.append(buf.data_ptr())
    return tuple(ptrs)


def _assert_ranks_agree(states, device, group):
    """Every rank must hold the same (done, iters) for every restart: they all received the same
    all-reduced bytes.  A mismatch means the collective did not deliver identical results."""
    mine = torch.tensor([[s[0], s[1]] for s in states], dtype=torch.int64, device=device)
    low, high = mine.clone(), mine.clone()
    dist.all_reduce(low, op=dist.ReduceOp.MIN, group=group)
    dist.all_reduce(high, op=dist.ReduceOp.MAX, group=group)
    if not bool(torch.equal(low, high)):
        raise RuntimeError("sharded EM loop: ranks disagree on the loop state (done, iters): min %s max %s"
                           % (low.tolist(), high.tolist()))


def run_em_sharded(local_mat, local_weights, args, inits=None, group=None, want_read_mix=True,
                   check_every=8, storage=None, records=None, exchange="rccl"):
    """
    run_em (em.py:94-165) over a matrix whose rows are spread over the ranks of
    `group`; each rank passes ITS row block and gets back the global proportions
    plus ITS block of the posterior matrix.  Same dict as em.run_em_ex.
    records: the shard as a preprocess.CodedMatrix (build_em_records_device on the rank's own rows);
    local_mat may then be None.  exchange: see sharded_em_loop ("rccl" | "oneshot").
    """
    n_multi = int(args.n_multi)
    plan = _em.EmPlan(local_mat, local_weights, n_runs=n_multi,
                      storage=storage or getattr(args, "storage", "auto"), records=records)
    if inits is None:
        inits = broadcast_inits(n_multi, plan.n_haps, args.init_alpha, plan.dev, group)
    inits = numpy.ascontiguousarray(inits, dtype=numpy.float64)
    ln_cur, ln_new, states = sharded_em_loop(plan, inits, args.tolerance, args.max_iter,
                                             group=group, check_every=check_every, exchange=exchange)
    return _em.collect_result(plan, inits, ln_cur, ln_new, states, want_read_mix, reuse_linear=True)


EXCHANGE_CHUNK_BYTES = 2 << 30          # receive staging per chunk of the row-block exchange (x2: double buffer)


def _global_rank(group, r):
    return dist.get_global_rank(group, r) if group is not None else r


def exchange_fold_blocks(fold, has_fold, n_rows, n_haps, delta, dev, group=None,
                         chunk_bytes=EXCHANGE_CHUNK_BYTES, fold_fn=None):
    """
    End-of-run combine of config 5 (SURVEY.md section 8 e; em.py:156-161 across ranks).
    Every rank r that ran restarts holds `fold` = logaddexp over ITS runs of the full
    [R][H] log posterior.  Rank j must end up with rows shard_bounds(R, j, world) of
    logaddexp over ALL ranks' folds, plus `delta` (= -log n_multi).

    Transport: a direct exchange, not a ring -- xGMI links are point to point, so rank r
    sends its copy of block j straight to rank j for every j at once (grouped RCCL
    send/recv) and all 7 links of a GPU carry one block each: (world - 1) / world of the
    matrix leaves each GPU once, 1/world of it per link.  The blocks move in row chunks,
    double buffered: chunk c + 1 is in flight while mxm_fold_logaddexp folds chunk c
    (own rows first, then the peers' in rank order -> deterministic) in log space, so
    entries below exp(-745) keep the finite values numpy.logaddexp gives them.

    fold: [R][H] device tensor, or None on a rank without restarts (has_fold[rank] False).
    fold_fn(acc, pieces, delta): stand-in for the device kernel (the gloo CPU tests drive the
    transport with a numpy logaddexp); None = mxm_fold_logaddexp.
    Returns this rank's [hi - lo][H] block (a view of `fold` where there is one).
    """
    import ctypes
    rank, world = _world(group)
    bounds = [shard_bounds(n_rows, r, world) for r in range(world)]
    lo, hi = bounds[rank]

    def fold_into(acc, pieces, dlt):
        """acc = logaddexp(acc, *pieces) + dlt, at most 8 inputs per launch"""
        if fold_fn is not None:
            return fold_fn(acc, list(pieces), dlt)
        lib = _lib.load()
        stream = torch.cuda.current_stream().cuda_stream
        pieces = list(pieces)
        first = True
        while pieces or first:
            now, pieces = pieces[:8], pieces[8:]
            ptrs = (ctypes.c_void_p * max(len(now), 1))(*[t.data_ptr() for t in now])
            lds = (ctypes.c_int64 * max(len(now), 1))(*[t.stride(0) for t in now])
            _lib.check(lib.mxm_fold_logaddexp(acc.data_ptr(), acc.stride(0), ptrs, lds, len(now),
                                                  acc.shape[0], n_haps, dlt if not pieces else 0.0, stream),
                           "mxm_fold_logaddexp")
            first = False

    if has_fold[rank]:
        mine = fold[lo:hi]
    else:
        mine = torch.full((hi - lo, n_haps), float("-inf"), dtype=torch.float64, device=dev)
    senders = [r for r in range(world) if r != rank and has_fold[r]]      # who sends me pieces of my block
    if world == 1 or (not senders and not has_fold[rank]):
        if hi > lo and delta != 0.0:
            fold_into(mine, [], delta)
        return mine
    direct = dist.get_backend(group) == "nccl"          # RCCL moves device memory; gloo (tests) is staged on the host
    blocks = [b[1] - b[0] for b in bounds]
    # the chunking must be the same on every rank (a send's size has to match its receive):
    # size it by the most pieces any rank can receive, not by this rank's own senders
    most = max(1, sum(1 for f in has_fold if f))
    chunk_rows = max(1, min(max(blocks), int(chunk_bytes // (n_haps * 8 * most))))
    n_chunks = (max(blocks) + chunk_rows - 1) // chunk_rows
    stage_dev = dev if direct else torch.device("cpu")
    tmp = [torch.empty((max(1, len(senders)), chunk_rows, n_haps), dtype=torch.float64, device=stage_dev)
           for _ in range(2 if n_chunks > 1 else 1)]

    def post(c, slot):
        ops, keep = [], []
        if has_fold[rank]:
            for p in range(world):
                if p == rank:
                    continue
                a = bounds[p][0] + c * chunk_rows
                b = min(a + chunk_rows, bounds[p][1])
                if b > a:
                    src = fold[a:b]
                    if direct and not src.is_contiguous():
                        src = src.contiguous()          # RCCL sends a flat buffer: rows with a padded stride are staged
                    elif not direct:
                        src = src.cpu()
                    keep.append(src)
                    ops.append(dist.P2POp(dist.isend, src, _global_rank(group, p), group))
        a = lo + c * chunk_rows
        b = min(a + chunk_rows, hi)
        if b > a:
            for k, p in enumerate(senders):
                ops.append(dist.P2POp(dist.irecv, tmp[slot][k, :b - a], _global_rank(group, p), group))
        return (dist.batch_isend_irecv(ops) if ops else []), keep, (a, b)

    pending = post(0, 0)
    for c in range(n_chunks):
        nxt = post(c + 1, (c + 1) % len(tmp)) if c + 1 < n_chunks else None
        reqs, keep, (a, b) = pending
        for req in reqs:
            req.wait()
        if b > a:
            slot = c % len(tmp)
            pieces = [tmp[slot][k, :b - a] for k in range(len(senders))]
            if not direct:
                pieces = [t.to(dev) for t in pieces]
            fold_into(mine[a - lo:b - lo], pieces, delta)
        del keep
        pending = nxt
    return mine


def run_em_restart_parallel(mat, weights, args, inits=None, group=None, want_read_mix=True, timing=None,
                            records=None):
    """
    Config 5: the matrix is replicated, the n_multi restarts are dealt
    round-robin over the ranks (run i -> rank i % world) and run with no
    per-iteration communication; on each rank mxm_em_loop keeps one full tile of
    its restarts iterating and refills a slot when its restart stops.  The deal is
    static: restarts of one matrix take about the same number of iterations
    (396-563 at 10^6 x 5408), a rank runs its 8 of 64 in two generations of one
    tile either way, and the job ends with the slowest restart wherever it runs
    (bench.py --mode restarts reports each rank's idle tail).

    At the end: the per-run log-proportions are summed with one all-reduce [H]
    (em.py:155, geometric mean); the posterior fold (em.py:156, log-mean-exp over
    runs) is completed by exchange_fold_blocks -- each rank returns ITS row block
    `rows` = shard_bounds(R, rank, world) of read_mix, like run_em_sharded does.
    A rank without restarts (world > n_multi) builds no plan at all.
    `timing` (dict) receives loop_s / fold_s / combine_s of this rank.
    records: the replicated matrix as a preprocess.CodedMatrix (build_em_records_device); `mat` may then be None.
    Each rank's restarts then run one after another through the records loop (mxm_em_loop_coded), and with
    want_read_mix the posterior blocks are decoded from the records' log tables (em.posterior); without it the
    caller forms posterior rows on demand from "ln_theta_k" (em.RecordsPosterior).
    """
    import math
    import time
    rank, world = _world(group)
    n_multi = int(args.n_multi)
    if records is not None:
        dev = records.rec.device
        n_rows, n_haps = records.n_rows, records.n_haps
    else:
        dev = require_gpu() if not isinstance(mat, torch.Tensor) else mat.device
        n_rows, n_haps = mat.shape
    if inits is None:
        inits = broadcast_inits(n_multi, n_haps, args.init_alpha, dev, group)
    inits = numpy.ascontiguousarray(inits, dtype=numpy.float64)
    mine = list(range(rank, n_multi, world))
    has_fold = [r < n_multi for r in range(world)]             # rank r got run r at least
    ln_sum = torch.zeros(n_haps, dtype=torch.float64, device=dev)
    counts = torch.zeros((n_multi, 2), dtype=torch.int64, device=dev)       # iters, done
    run_props = torch.zeros((n_multi, n_haps), dtype=torch.float64, device=dev)
    theta_k = torch.zeros((n_multi, n_haps), dtype=torch.float64, device=dev)    # each run's log theta_k (exp'ed: additive)
    fold = None

    def sync():
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)

    sync()
    t0 = time.perf_counter()
    ln_k = None
    plan = None
    if mine:
        plan = _em.EmPlan(mat, weights, n_runs=len(mine), records=records)
        ln_cur, ln_new, states = _em.em_loop(plan, inits[mine], args.tolerance, args.max_iter)
        ln_k = ln_cur.cpu().numpy()
        ln_sum += ln_new.sum(dim=0)
        for j, run in enumerate(mine):
            counts[run, 0] = states[j][1]
            counts[run, 1] = states[j][0]
            run_props[run] = torch.exp(ln_new[j])
            theta_k[run] = ln_cur[j]
    sync()
    t1 = time.perf_counter()
    if mine and want_read_mix:
        fold = plan.release_linear()                  # the loop is over: reuse its R x H storage
        for j in range(len(mine)):
            fold = _em.posterior(plan, ln_k[j], out=fold, fold=(j > 0))
    sync()
    t2 = time.perf_counter()
    if _collective(group):
        dist.all_reduce(ln_sum, group=group)
        dist.all_reduce(counts, group=group)
        dist.all_reduce(run_props, group=group)
        dist.all_reduce(theta_k, group=group)          # every run lives on exactly one rank, the others hold zeros
    read_mix = None
    lo, hi = shard_bounds(n_rows, rank, world)
    if want_read_mix:
        delta = -math.log(n_multi) if n_multi > 1 else 0.0
        if _collective(group):
            read_mix = exchange_fold_blocks(fold, has_fold, n_rows, n_haps, delta, dev, group)
        else:
            read_mix = exchange_fold_blocks(fold, [True], n_rows, n_haps, delta, dev, None)
    sync()
    t3 = time.perf_counter()
    if timing is not None:
        timing.update(loop_s=t1 - t0, fold_s=t2 - t1, combine_s=t3 - t2)
    props = torch.exp(ln_sum / n_multi).cpu().numpy() if n_multi > 1 else run_props[0].cpu().numpy()
    counts = counts.cpu().numpy()
    return {"props": props, "read_mix": read_mix, "rows": (lo, hi), "iters": [int(x) for x in counts[:, 0]],
            "done": [int(x) for x in counts[:, 1]], "run_props": run_props.cpu().numpy(), "inits": inits,
            "ln_theta_k": theta_k.cpu().numpy()}
