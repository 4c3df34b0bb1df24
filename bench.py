#!/usr/bin/env python
"""
bench.py -- EM hot-path throughput on MI355X (BASELINE.json metric:
"EM iters/sec + read x hap cells/sec, 1M reads x 5.4k haps, 1/2/4/8 GPUs").

    python bench.py [--gpus N] [--steps K] [--warmup W]
                    [--scaling strong|weak] [--total-rows R] [--rows R_PER_GPU]
                    [--mode rows|restarts] [--restarts B]

A "step" is one full EM iteration (em.py:57-91 + the convergence bookkeeping
of em.py:126-143) over this rank's resident rows of the synthetic (synth-v1)
reads x haplogroups matrix: mxm_em_iter -> [RCCL all-reduce of the H column
sums when N > 1] -> mxm_m_finalize.  The tolerance is 0 inside the timed
region so that exactly K iterations run; nothing is skipped or cached.

Scaling (SURVEY.md section 8 e): rows shard over the ranks.
  strong (default)  the metric's own problem: --total-rows (1 000 000) rows IN TOTAL,
                    rank r holds rows shard_bounds(total, r, N) of the same global read
                    set at every N; --total-rows 10000000 --gpus 8 is BASELINE config 4
  weak              --rows rows PER GPU (per-GPU work fixed as N grows)
At N = 1 both are the same 1 000 000 x 5408 workload.

--gpus N > 1 without a launcher (no RANK in the environment): this process starts N
rank processes itself (python -m torch.distributed.run, 127.0.0.1) BEFORE touching
torch or the GPU, relays their output and exits with their status.  Under the
driver's own torch.distributed.run it is a plain rank.

--mode restarts (BASELINE config 5): the matrix is replicated, --restarts EM
restarts are dealt over the ranks and run to convergence with no per-iteration
traffic; reports restart-iterations/s, each rank's loop time (idle tail) and the
end-of-run combine.  --steps / --warmup do not apply there (a run is a run).

Inputs are resident in HBM before the clock starts (the matrix is BUILT on the
device from CSR observations by mxm_build_em_matrix; build, linearize and
posterior pass are timed separately and reported as extra fields).

One JSON line on stdout (rank 0); progress on stderr.
"""

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_BYTES_PER_S = 8.0e12        # MI355X HBM3E spec (MI355X_MICROARCH.md)
PARITY_PROPS_BAR = 1e-9              # in-run parity bar on proportions (north-star bar: 1e-6)


def log(msg):
    sys.stderr.write("[bench] %s\n" % msg)
    sys.stderr.flush()


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--mode", default="rows", choices=["rows", "restarts"],
                    help="rows: row-sharded EM iteration (configs 2-4); restarts: replicated matrix, "
                         "restarts dealt over the ranks, run to convergence (config 5)")
    ap.add_argument("--scaling", default=None, choices=["strong", "weak"],
                    help="default strong (the metric's 1M-row problem at every N); weak if --rows is given")
    ap.add_argument("--total-rows", type=int, default=1000000, help="reads in total (strong scaling)")
    ap.add_argument("--rows", type=int, default=None, help="reads per GPU (selects weak scaling)")
    ap.add_argument("--cpu-rows", type=int, default=32768)
    ap.add_argument("--cpu-iters", type=int, default=4)
    ap.add_argument("--no-cpu-baseline", action="store_true",
                    help="skip the oracle leg (cpu_baseline and parity_in_run become null)")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--restarts", type=int, default=1,
                    help="mode rows: EM restarts advanced together (config 3 uses 10); a step then is one "
                         "iteration of EVERY restart and value counts cells x restarts.  mode restarts: "
                         "restarts in total (config 5 uses 64)")
    ap.add_argument("--batch-tile", type=int, default=4,
                    help="restarts sharing one pass over the matrix (1 = unbatched schedule)")
    ap.add_argument("--storage", default="f64", choices=["f64", "f32", "coded"],
                    help="form of the streamed matrix; f32 is a labelled opt-in variant (fp64 math on "
                         "float-stored P); coded = lossless row dictionaries (one byte per cell + the row's "
                         "distinct fp64 values).  The default line measures f64 and reports coded beside it")
    ap.add_argument("--records", action="store_true",
                    help="mode rows: the build leaves the matrix as row-dictionary records and NO dense matrix is "
                         "made (implies --storage coded; 10^7 rows then fit one GPU); no posterior pass")
    ap.add_argument("--min-rows-per-wg", type=int, default=0, help="tuning knob (0 = library default)")
    ap.add_argument("--build-kernel", default="auto", choices=["auto", "bytes", "lut", "sparse"])
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise the process group and issue the per-iteration all-reduce even with "
                         "one rank (exercises the RCCL path on a single-GPU box)")
    ap.add_argument("--exchange", default="rccl", choices=["rccl", "oneshot"],
                    help="the per-iteration exchange of a multi-rank step: torch.distributed's all-reduce (default) or the "
                         "library's one-shot exchange (dist.OneShotExchange; unmeasured over xGMI)")
    ap.add_argument("--backend", default="nccl",
                    help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to test the "
                         "multi-process path on a single GPU)")
    return ap.parse_args(argv)


def spawn_ranks(opts, argv):
    """
    `python bench.py --gpus N` with no launcher: start the N rank processes here.  This
    process has not imported torch and never touches the GPU; the ranks are fresh
    interpreters (no exec of a GPU-initialised process anywhere).
    """
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what RCCL needs on this driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(opts.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    log("starting %d ranks: %s" % (opts.gpus, " ".join(cmd[1:])))
    return subprocess.call(cmd, env=env)


def cpu_reference_leg(mat_rows, n_iters, tol=1e-4):
    """
    The oracle's em_step (numpy restatement of em.py:57-91, single thread like the
    reference) on a bounded sample of the same matrix, iterated the way run_em does
    (em.py:126-143: stop when converged(), result = theta_{k+1} with the posterior under
    theta_k).  CHECKER / baseline only -- never part of the measured GPU path.
    -> dict(rate cells/s, seconds, iters, done, l1, init, ln_new, best (row argmax of the posterior))
    """
    import numpy
    from oracle import em_oracle
    n_rows, n_haps = mat_rows.shape
    wts = numpy.ones(n_rows, dtype=numpy.int64)
    numpy.random.seed(7)
    init = numpy.random.dirichlet([1.0] * n_haps)
    ln_props = numpy.log(init)
    buf = numpy.empty_like(mat_rows)
    iters, done, l1 = 0, 2, float("nan")
    t0 = time.perf_counter()
    for _ in range(n_iters):
        buf, ln_next = em_oracle.em_step(mat_rows, wts, ln_props, buf)
        iters += 1
        l1 = float(numpy.sum(numpy.abs(numpy.exp(ln_props) - numpy.exp(ln_next))))    # em.py:53-54
        if l1 < tol:
            done = 1
            ln_props = ln_next
            break
        ln_props = ln_next
    dt = time.perf_counter() - t0
    return {"rate": n_rows * n_haps * iters / dt, "seconds": dt, "iters": iters, "done": done, "l1": l1,
            "init": init, "ln_new": ln_props, "best": buf.argmax(axis=1)}


def cpu_linear_blas_leg(mat_rows, init, n_iters):
    """
    NOT the reference's algorithm: the same change of variables the GPU loop uses (P = exp(M - rowmax) once;
    per iteration two BLAS matrix-vector products, Z = P p and T = P^T (w / Z)) in numpy on the host's
    multi-threaded BLAS -- the fairest CPU line one can put beside the GPU number, reported as an extra.
    -> dict(rate cells/s, seconds, iters, threads, max |dprops| against `check` if given)
    """
    import numpy
    n_rows, n_haps = mat_rows.shape
    rowmax = mat_rows.max(axis=1, keepdims=True)
    lin = numpy.exp(mat_rows - rowmax)
    p = init.copy()
    w = numpy.ones(n_rows)
    lin @ p                                             # warm the BLAS threads
    t0 = time.perf_counter()
    for _ in range(n_iters):
        z = lin @ p
        t = lin.T @ (w / z)
        p = p * t
        p /= p.sum()
    dt = time.perf_counter() - t0
    try:
        import threadpoolctl
        threads = max((info.get("num_threads", 1) for info in threadpoolctl.threadpool_info()), default=1)
    except Exception:
        threads = os.cpu_count()
    return {"rate": n_rows * n_haps * n_iters / dt, "seconds": dt, "iters": n_iters, "threads": int(threads), "props": p}


def parity_in_run(em, torch, slab, cpu, n_iters, tol=1e-4, storage="f64"):
    """
    The HIP path on the same slab, same init, same loop bounds as the oracle leg above
    (outside the timed region): proportions, stopping iteration and haplogroup calls.
    """
    import numpy
    n_rows = slab.shape[0]
    wts = torch.ones(n_rows, dtype=torch.float64, device=slab.device)
    plan = em.EmPlan(slab, wts, n_runs=1, storage=storage)
    ln_cur, ln_new, states = em.em_loop(plan, cpu["init"][None, :], tol, n_iters)
    done, iters, l1 = states[0]
    post = em.posterior(plan, ln_cur[0])
    best = post.argmax(dim=1).cpu().numpy()
    d_props = float(numpy.abs(numpy.exp(ln_new[0].cpu().numpy()) - numpy.exp(cpu["ln_new"])).max())
    return {"rows": int(n_rows), "iters": int(iters), "max_abs_dprops": d_props,
            "iters_equal": bool(iters == cpu["iters"] and done == cpu["done"]),
            "argmax_equal": bool(numpy.array_equal(best, cpu["best"])),
            "l1_gpu": float(l1), "l1_cpu": cpu["l1"],
            "against": "oracle em_step x%d (em.py:57-91, :126-143) on the first %d rows of the same "
                       "matrix, same Dirichlet init" % (cpu["iters"], n_rows)}


def loop_ms_per_iteration(em, torch, plan, props_row, n_iter):
    """
    Wall time per iteration of the product's own loop driver (em.em_loop -> mxm_em_loop / mxm_em_loop_coded: for records
    the whole loop in one persistent launch, em_fused_coded_kernel) over `n_iter` iterations with the tolerance at 0 --
    launch, final state read-back and host synchronisation included.  The second of two runs is reported.
    """
    init = props_row.detach().cpu().numpy().reshape(1, -1)
    best = None
    for _ in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _, _, states = em.em_loop(plan, init, 0.0, n_iter)
        torch.cuda.synchronize()
        best = (time.perf_counter() - t0) * 1e3 / max(states[0][1], 1)
    return best


RESTARTS10 = 10                  # BASELINE config 3: ten restarts on one matrix
RESTARTS10_ITERS = 48            # iterations of every restart in the timed run (whole bursts of the loop driver's 16)


def restarts10_leg(em, torch, lib, cplan, n_rows, n_haps):
    """
    BASELINE config 3 on the default route (records + quad dictionary): ten restarts through the product's own loop driver
    (em.em_loop -> mxm_em_loop_coded), tolerance 0 so that every restart runs exactly RESTARTS10_ITERS iterations.  Full
    tiles of three restarts share a pass over the records (em_iter_quad_batched_kernel, round 6); the same run with one
    restart per pass (round 5's schedule) is timed beside it, and one iteration's per-restart column sums are compared.
    """
    import numpy
    inits = numpy.random.RandomState(7).dirichlet([1.0] * n_haps, size=RESTARTS10)
    out = {"restarts": RESTARTS10, "iterations_each": RESTARTS10_ITERS, "tile": int(cplan.restart_tile())}

    def run(tile):
        lib.mxm_set_coded_batch_tile(tile)
        try:
            em.em_loop(cplan, inits, 0.0, 3)                               # warm
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            _, _, states = em.em_loop(cplan, inits, 0.0, RESTARTS10_ITERS)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        finally:
            lib.mxm_set_coded_batch_tile(3)
        done = sum(s[1] for s in states)
        return dt, done

    def sums(tile):
        lib.mxm_set_coded_batch_tile(tile)
        try:
            p = torch.from_numpy(inits[:3].copy()).to(cplan.dev)
            cs = torch.zeros_like(p)
            cplan.em_iter(p, p.log(), em.new_state(3, cplan.dev), cs)
            torch.cuda.synchronize()
            return cs
        finally:
            lib.mxm_set_coded_batch_tile(3)

    dt3, done3 = run(3)
    dt1, done1 = run(1)
    a, b = sums(3), sums(1)
    out.update({"restart_iterations_per_s": done3 / dt3, "ms_per_restart_iteration": dt3 * 1e3 / max(done3, 1),
                "restart_iterations": int(done3), "value": float(n_rows) * n_haps * done3 / dt3, "unit": "cells/s",
                "one_per_pass_restart_iterations_per_s": done1 / dt1,
                "one_per_pass_ms_per_restart_iteration": dt1 * 1e3 / max(done1, 1),
                "max_rel_dcolsum_vs_one_per_pass": float(((a - b).abs() / b.abs().amax(dim=1, keepdim=True)).max().item()),
                "kernel": "em_iter_quad_batched_kernel<3, %d, 1> + <3, %d, 6>" % (((((n_haps + 7) // 8 * 2 + 255) // 256 + 1) // 2,) * 2),
                "note": "wall time of mxm_em_loop_coded over %d iterations of each of %d restarts (launches, state read-backs "
                        "and the single pass of the tenth restart included); three restarts share the row loads, the tables "
                        "in LDS and the lookups of one pass" % (RESTARTS10_ITERS, RESTARTS10)})
    return out


def coded_leg(em, torch, lib, plan, mat, wts, props, ln_props, steps):
    """
    The EM iteration of the default line over the SAME matrix in row-dictionary storage
    (EmPlan(storage="coded")): same proportions in, column sums compared with the dense pass,
    `steps` iterations timed the same way (HIP events around the whole step).
    """
    import numpy
    n_rows, n_haps = mat.shape
    torch.cuda.synchronize()
    quads_mode, em.QUADS = em.QUADS, False            # first the records alone; the quad dictionary is attached below
    try:
        cplan = em.EmPlan(mat, wts, n_runs=RESTARTS10, storage="coded")     # (workspace for the restarts10 sub-leg below)
        if cplan.coded is None:
            return None
        t0 = time.perf_counter()
        cplan.encode()
        torch.cuda.synchronize()
        encode_ms = (time.perf_counter() - t0) * 1e3
    finally:
        em.QUADS = quads_mode
    state = em.new_state(1, mat.device)
    cs_dense = torch.zeros_like(props)
    cs_coded = torch.zeros_like(props)
    plan.em_iter(props, ln_props, state, cs_dense)
    cplan.em_iter(props, ln_props, state, cs_coded)
    rel = float(((cs_coded - cs_dense).abs() / cs_dense.abs().clamp_min(1e-300)).max().item())
    ln_a, ln_b = ln_props.clone(), ln_props.clone()
    p_cur = props.clone()
    beg, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    # the row-pass kernel of EVERY timed step by its own pair of events (the library records them around the launch): the
    # mean is what a rocprofv3 average of the same run shows; the steps of a run's first iterations are the slow ones
    # (profiles/r05/step_sequence.txt), so the last step alone would flatter the kernel
    kevs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    for a_ev, b_ev in kevs:
        a_ev.record()
        b_ev.record()

    def one(pair=None):
        if pair is not None:
            lib.mxm_set_timing_events(pair[0].cuda_event, pair[1].cuda_event)
        cplan.em_iter(p_cur, ln_a, state, cs_coded)
        if pair is not None:
            lib.mxm_set_timing_events(None, None)
        cplan.finalize(cs_coded, ln_a, ln_b, p_cur, state, 0.0, 1 << 30)

    SETTLE = 30        # iterations of a run before the proportions have concentrated (profiles/r05/step_sequence.txt)

    def timed_steps():
        """-> (ms per step, kernel ms) once a run has settled, and the kernel ms over the run's FIRST `steps` iterations."""
        p_cur.copy_(props)                                 # a fresh run: the same start for every leg
        ln_a.copy_(ln_props)
        state.zero_()
        for i in range(steps):                             # the early phase, kernel by kernel
            one(kevs[i])
        torch.cuda.synchronize()
        early = float(numpy.mean([a_ev.elapsed_time(b_ev) for a_ev, b_ev in kevs]))
        for _ in range(max(0, SETTLE - steps)):
            one()
        torch.cuda.synchronize()
        beg.record()
        for i in range(steps):
            one(kevs[i])
        end.record()
        torch.cuda.synchronize()
        each = [a_ev.elapsed_time(b_ev) for a_ev, b_ev in kevs]
        return beg.elapsed_time(end) / steps, float(numpy.mean(each)), early

    ms, kernel_ms, kernel_ms_first = timed_steps()
    loop_ms = loop_ms_per_iteration(em, torch, cplan, props[0], max(steps, 50))
    rec_kernel_bytes, rec_bytes_per_iteration = float(cplan.coded_record_bytes), float(cplan.coded_bytes)   # (the quads change them)
    # the same step with a quad dictionary beside the records (EmPlan.attach_quads: what "auto" does from 3e5 rows)
    quads = None
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if cplan.attach_quads(True):
        torch.cuda.synchronize()
        quad_build_ms = (time.perf_counter() - t0) * 1e3
        cs_quad = torch.zeros_like(props)
        cplan.em_iter(props, ln_props, state, cs_quad)
        qrel = float(((cs_quad - cs_dense).abs() / cs_dense.abs().clamp_min(1e-300)).max().item())
        qms, qkernel, qkernel_first = timed_steps()
        quads = {"ms_per_step": qms, "value": float(n_rows) * n_haps / (qms * 1e-3), "kernel": "em_iter_quad_coded_kernel",
                 "kernel_ms": qkernel, "kernel_ms_first_steps": qkernel_first, "kernel_bytes": float(cplan.coded_record_bytes),
                 "hbm_frac": cplan.coded_record_bytes / (qkernel * 1e-3) / HBM_PEAK_BYTES_PER_S,
                 "quad_rows": int(cplan.quad_rows_n), "quad_bytes": float(cplan.quad_bytes), "build_ms": quad_build_ms,
                 "max_rel_dcolsum": qrel,
                 "note": "one code byte per FOUR columns for the rows with at most 256 distinct value quadruples, beside the "
                         "records (csrc/quad_kernels.hpp); the rows without quads share the grid; same timing as ms_per_step",
                 "restarts10": restarts10_leg(em, torch, lib, cplan, n_rows, n_haps)}
    # "ms_per_step" keeps ONE meaning across rounds -- HIP events over the per-iteration kernels, rounds 2-3's
    # quantity and what a multi-GPU step runs around its all-reduce; the one-launch loop run_em takes on one GPU has a key
    # of its own (round 4 reported it AS ms_per_step: "schema" tells the two layouts apart)
    return {"schema": 5, "ms_per_step": ms, "value": float(n_rows) * n_haps / (ms * 1e-3), "unit": "cells/s",
            "ms_per_step_is": "the separate-launch path (em_iter_coded_kernel -> column reduce + finalize), HIP events over "
                              "%d steps: rounds 2-3's quantity (round 4 printed it as per_iteration_kernels_ms_per_step)" % steps,
            "one_launch_loop_ms_per_iteration": loop_ms,
            "one_launch_loop_value": float(n_rows) * n_haps / (loop_ms * 1e-3),
            "one_launch_loop_note": "what run_em itself iterates over records on one GPU: mxm_em_loop_coded, the whole loop in "
                                    "one persistent launch (em_fused_coded_kernel), wall time of %d iterations incl. launch "
                                    "and state read-back (round 4 printed it as ms_per_step)" % max(steps, 50),
            "per_iteration_kernels_ms_per_step": ms,
            "rows_with_16bit_codes": int(cplan.coded_wide),
            "kernel": "em_iter_coded_kernel", "kernel_ms": kernel_ms, "kernel_ms_first_steps": kernel_ms_first,
            "kernel_ms_is": "mean over %d timed steps (HIP events around each launch) AFTER the first %d iterations of a run; "
                            "kernel_ms_first_steps is the same mean over the run's first %d iterations, where all proportions are "
                            "of one magnitude and the card runs at lower clocks (profiles/r05/step_sequence.txt) -- a rocprofv3 "
                            "average over all launches of the run lies between the two" % (steps, SETTLE, steps),
            "bytes_per_iteration": rec_bytes_per_iteration,
            "kernel_bytes": rec_kernel_bytes,
            "hbm_frac": rec_kernel_bytes / (kernel_ms * 1e-3) / HBM_PEAK_BYTES_PER_S,
            "rows_left_dense": int(cplan.coded_rest), "encode_ms": encode_ms, "max_rel_dcolsum": rel, "quads": quads,
            "note": "same iteration, matrix stored as one byte per cell + each row's distinct fp64 values "
                    "(decodes to the dense matrix bit for bit); not the headline value"}


def pmc_traffic(n_rows, n_haps, storage="f64", kernel=None, algo_bytes=None, root=None, matrix=None):
    """
    HBM bytes per launch of the streaming kernel from the committed rocprofv3 PMC passes
    (profiles/rNN/pmc_traffic_*.json, written by tools/pmc_summary.py from separate --pmc FETCH_SIZE /
    --pmc WRITE_SIZE runs of this script).  Counters cannot be read from inside the process, so this is
    the latest round's profiled value for the SAME workload (rows per GPU, haplogroups), the SAME storage
    of the matrix and the SAME kernel instance this run launched (`kernel`, e.g.
    "em_iter_wide_kernel<512, 6, 1, 3, 1>" from mxm_describe_stream_kernel) -- a plan in row-dictionary
    storage also launches em_iter_wide_kernel, on its few dense leftover rows, and that figure must never
    stand in for the dense matrix's.  A value outside [0.9, 1.5] x the algorithmic bytes is refused
    (it would say the file is not about this kernel).  matrix ("records" | "encoded"): which records a coded file was
    taken on -- the build's (--records) or the encoder's; files without the field are the encoder's.
    -> (bytes, path relative to the repo) or None.
    """
    import glob
    import re
    root = root or ROOT
    found = []
    for path in glob.glob(os.path.join(root, "profiles", "*", "pmc_traffic_*.json")):
        try:
            with open(path) as fin:
                data = json.load(fin)
        except (OSError, ValueError):
            continue
        meta = data.get("_workload", {})
        if meta.get("rows_per_gpu") != n_rows or meta.get("haps") != n_haps:
            continue
        # files written before the field existed: the coded runs carry it in their name
        have = meta.get("storage") or ("coded" if "coded" in os.path.basename(path) else "f64")
        if have != storage:
            continue
        if matrix is not None and meta.get("matrix", "encoded") != matrix:
            continue
        for name, rec in data.items():
            if name.startswith("_") or (kernel is not None and not kernel.endswith("*") and name != kernel):
                continue
            if kernel is None and not name.startswith("em_iter_wide_kernel"):
                continue
            if kernel is not None and kernel.endswith("*") and not name.startswith(kernel[:-1]):
                continue
            val = float(rec["hbm_bytes_per_launch"])
            if algo_bytes and not (0.9 * algo_bytes <= val <= 1.5 * algo_bytes):
                continue
            rnd = re.search(r"r(\d+)", os.path.basename(os.path.dirname(path)))
            found.append((int(rnd.group(1)) if rnd else -1, val, os.path.relpath(path, root)))
    if not found:
        return None
    found.sort()
    return found[-1][1], found[-1][2]


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    opts = parse_args(argv)
    if opts.gpus < 1:
        raise SystemExit("bench: --gpus must be >= 1")
    if opts.gpus > 1 and "RANK" not in os.environ:
        raise SystemExit(spawn_ranks(opts, argv))          # before torch / the GPU are touched
    # stdout carries ONE line, the JSON: libraries that write to file descriptor 1 themselves (RCCL prints a
    # version banner there when a communicator is made) are pointed at stderr for the life of the process
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import numpy
    import torch
    import torch.distributed as dist
    from mixemt_amd import _lib, em, phylotree, preprocess, synth
    from mixemt_amd import dist as mdist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != opts.gpus:
        raise SystemExit("bench: --gpus %d but the launcher started WORLD_SIZE=%d ranks; they must agree "
                         "(plain `python bench.py --gpus N` starts its own ranks)" % (opts.gpus, world))
    n_dev = torch.cuda.device_count()
    if n_dev < 1:
        raise SystemExit("bench: no ROCm GPU visible (the EM hot path has no CPU fallback)")
    if opts.backend == "nccl" and world > n_dev:
        raise SystemExit("bench: %d ranks but %d GPU(s) visible" % (world, n_dev))
    dev = torch.device("cuda", local_rank % n_dev)
    torch.cuda.set_device(dev)
    use_dist = world > 1 or opts.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if opts.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(opts.backend, rank=rank, world_size=world)
    lib = _lib.load()
    if world > n_dev:
        # ranks sharing a GPU (gloo test set-up only): two persistent one-launch loops on one device could
        # starve each other (include/mixemt_hip.h, mxm_em_loop)
        lib.mxm_set_loop_fused(0, 0)

    # ---- which rows live here ---------------------------------------------------------------
    scaling = opts.scaling or ("weak" if opts.rows is not None else "strong")
    if opts.mode == "restarts":
        scaling = "strong"                      # total work (restarts) fixed, matrix replicated
        total_rows = opts.rows if opts.rows is not None else opts.total_rows
        lo, hi = 0, total_rows
    elif scaling == "weak":
        per_gpu = opts.rows if opts.rows is not None else opts.total_rows
        total_rows = per_gpu * world
        lo, hi = rank * per_gpu, (rank + 1) * per_gpu
    else:
        total_rows = opts.total_rows if opts.rows is None else opts.rows * world
        lo, hi = mdist.shard_bounds(total_rows, rank, world)
    n_rows = hi - lo
    if n_rows < 1:
        raise SystemExit("bench: rank %d of %d has no rows (%d in total)" % (rank, world, total_rows))

    # ---- inputs: Build 17 + RSRS tables, synth-v1 reads, matrix built on the device -------------
    t0 = time.perf_counter()
    refseq = phylotree.load_rsrs()
    phy = phylotree.load_build17(refseq)
    haps = sorted(phy.hap_var)
    tables = preprocess.HapVarTables.build(refseq, phy, haps)
    n_haps = len(haps)
    if opts.records:
        opts.storage = "coded"                        # (round 6: --mode restarts takes records too -- config 5 on the default route)
    need_gb = (2.0 + (1.0 if opts.mode == "rows" else 0.0)) * n_rows * n_haps * 8 / 1e9
    if opts.records:
        need_gb = n_rows * (8.0 + 0.07 * 2 * n_haps * 8 / 1024.0) * 1024 / 1e9       # records + ~7 % dense rows twice
    free_gb = torch.cuda.mem_get_info(dev)[0] / 1e9
    if need_gb > free_gb:
        raise SystemExit("bench: %d rows x %d haplogroups per GPU need about %.0f GB (matrix, linearised copy, "
                         "posterior), %.0f GB free: use more GPUs or fewer rows" % (n_rows, n_haps, need_gb, free_gb))
    row_ptr, site, obs, _ = synth.synth_rows(tables, len(refseq), lo, hi, seed=opts.seed)
    if rank == 0:
        log("tables + rows [%d, %d) of %d synthetic reads (%.1f sites/read) on host: %.1f s"
            % (lo, hi, total_rows, row_ptr[-1] / float(n_rows), time.perf_counter() - t0))
    row_ptr_d = torch.from_numpy(row_ptr).to(dev)
    site_d = torch.from_numpy(site.view(numpy.int16)).to(dev)
    obs_d = torch.from_numpy(obs).to(dev)
    tables.device()
    records = slab = None
    if opts.records:
        # CSR -> records, no dense matrix; built twice, the second call timed (as below)
        records = preprocess.build_em_records_device(tables, row_ptr_d, site_d, obs_d)
        del records
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        records = preprocess.build_em_records_device(tables, row_ptr_d, site_d, obs_d)
        torch.cuda.synchronize()
        build_s = time.perf_counter() - t0
        mat = None
        n_slab = min(n_rows, opts.cpu_rows)           # the oracle leg's rows, dense (the checker needs them)
        slab = preprocess.build_em_matrix_device(tables, row_ptr_d[:n_slab + 1].clone(), site_d[:int(row_ptr[n_slab])],
                                                 obs_d[:int(row_ptr[n_slab])])
    else:
        mat = torch.empty((n_rows, n_haps), dtype=torch.float64, device=dev)
        # built twice, the second call timed: the first one pays one-time library set-up (code object load,
        # the device sort behind the position-ordered row schedule), which is not the kernel's rate
        preprocess.build_em_matrix_device(tables, row_ptr_d, site_d, obs_d, out=mat, kernel=opts.build_kernel)
        torch.cuda.synchronize()
        samples = []
        for _ in range(3):                             # whole calls (marker kernel, fallback rows, host in between): the median
            t0 = time.perf_counter()
            preprocess.build_em_matrix_device(tables, row_ptr_d, site_d, obs_d, out=mat, kernel=opts.build_kernel)
            torch.cuda.synchronize()
            samples.append(time.perf_counter() - t0)
        build_s = sorted(samples)[1]
    wts = torch.ones(n_rows, dtype=torch.float64, device=dev)
    lib.mxm_set_batch_tile(opts.batch_tile)
    if opts.min_rows_per_wg > 0:
        lib.mxm_set_min_rows_per_wg(opts.min_rows_per_wg)

    env = dict(lib=lib, dev=dev, rank=rank, world=world, use_dist=use_dist, mat=mat, wts=wts, n_rows=n_rows,
               n_haps=n_haps, total_rows=total_rows, scaling=scaling, build_s=build_s, records=records, slab=slab)
    census = rank_census(torch, dist, dev, rank, world, use_dist, opts.backend, opts.gpus)
    line = bench_restarts(opts, env) if opts.mode == "restarts" else bench_rows(opts, env)
    if rank == 0:
        # what the collective itself saw: a line that says "n_gpus: 8" must show eight ranks that answered a
        # device-side all-reduce, each on a GPU of its own
        line.update({k: census[k] for k in ("ranks_seen", "devices", "backend")})
        if not census["ok"]:
            line["sanity_ok"] = False
            line["sanity_note"] = census["note"]
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0 and not line.get("sanity_ok", False):
        raise SystemExit(3)


def rank_census(torch, dist, dev, rank, world, use_dist, backend, gpus):
    """
    Who took part, as the collective saw it: `ranks_seen` = an all-reduce (SUM) of a one held on each rank's device (the
    backend's own tensors: device memory under nccl = RCCL), `devices` = every rank's GPU name and PCI address, all-gathered.
    ok is False when ranks_seen differs from --gpus, or when two ranks of an RCCL group sit on one PCI address -- a
    multi-GPU line must not be taken on trust from WORLD_SIZE alone.
    """
    props = torch.cuda.get_device_properties(dev)
    pci = tuple(getattr(props, k, None) for k in ("pci_domain_id", "pci_bus_id", "pci_device_id"))
    mine = {"rank": rank, "device_index": dev.index, "name": props.name,
            "pci": None if any(v is None for v in pci) else "%04x:%02x:%02x" % pci,
            "uuid": str(getattr(props, "uuid", "")) or None}
    if not use_dist:
        return {"ranks_seen": 1, "devices": [mine], "backend": None, "ok": gpus == 1, "note": "--gpus %d without a process group" % gpus}
    one = torch.ones(1, dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
    dist.all_reduce(one, op=dist.ReduceOp.SUM)
    seen = int(round(float(one.item())))
    every = [None] * world
    dist.all_gather_object(every, mine)
    note = None
    ok = seen == gpus
    if not ok:
        note = "the all-reduce saw %d rank(s), --gpus says %d" % (seen, gpus)
    if backend == "nccl":
        addr = [d["pci"] or d["uuid"] for d in every]
        known = [a for a in addr if a]
        if len(set(known)) < len(known):
            ok = False
            note = "two ranks of the RCCL group share a GPU: %s" % (addr,)
    return {"ranks_seen": seen, "devices": every, "backend": backend, "ok": ok, "note": note}


def bench_rows(opts, env):
    """Row-sharded EM iteration: the BASELINE metric (configs 2, 3, 4)."""
    import numpy
    import torch
    import torch.distributed as dist
    from mixemt_amd import _lib, em
    from mixemt_amd import dist as mdist
    (lib, dev, rank, world, use_dist, mat, wts, n_rows, n_haps, total_rows, scaling, build_s) = (
        env[k] for k in ("lib", "dev", "rank", "world", "use_dist", "mat", "wts", "n_rows", "n_haps",
                         "total_rows", "scaling", "build_s"))
    n_runs = opts.restarts
    records, slab = env.get("records"), env.get("slab")
    plan = em.EmPlan(mat, wts, n_runs=n_runs, storage=opts.storage, records=records)   # allocates P and linearises once (untimed: hipMalloc)
    torch.cuda.synchronize()
    t0 = time.perf_counter()                      # timed again on the now-resident buffers
    if records is not None:
        pass                                      # the records came out of the build: nothing to convert
    elif plan.coded is not None:
        plan.encode()                            # mxm_encode_rows + the dense rest, second time
    else:
        lin_fn = lib.mxm_linearize_f32 if opts.storage == "f32" else lib.mxm_linearize
        _lib.check(lin_fn(mat.data_ptr(), mat.stride(0), n_rows, n_haps, plan.lin.data_ptr(),
                          plan.lin.stride(0), plan.rowmax.data_ptr(),
                          torch.cuda.current_stream().cuda_stream), "mxm_linearize")
    if use_dist and plan.coded is not None and em.QUADS == "auto":
        # what dist.sharded_em_loop does on entry: over several ranks the per-iteration kernels are the only loop, and a
        # records shard gets its quad dictionary from SHARD_QUADS_MIN_ROWS rows (the timed step below runs them)
        plan.attach_quads("auto", min_rows=mdist.SHARD_QUADS_MIN_ROWS)
    torch.cuda.synchronize()
    linearize_s = time.perf_counter() - t0
    if rank == 0:
        log("matrix build %.3f s (%.3g cells/s), linearize %.3f s, HBM in use %.1f GB"
            % (build_s, n_rows * n_haps / build_s, linearize_s, torch.cuda.memory_allocated() / 1e9))

    # ---- loop state: init = sequential Dirichlet draws after numpy.random.seed(7) on rank 0 ------
    numpy.random.seed(7)
    init = numpy.stack([em.init_props(n_haps, 1.0) for _ in range(n_runs)])   # sequential draws
    ln0, p0 = em.log_inits(init)
    props_cur = torch.from_numpy(p0).to(dev)
    ln_cur = torch.from_numpy(ln0).to(dev)
    if use_dist:
        dist.broadcast(ln_cur, src=0)
        props_cur = torch.exp(ln_cur)
    ln_new = ln_cur.clone()
    colsum = torch.zeros_like(props_cur)
    state = em.new_state(n_runs, dev)
    total = opts.warmup + opts.steps
    evs = [tuple(torch.cuda.Event(enable_timing=True) for _ in range(4)) for _ in range(opts.steps)]
    for quad in evs:                       # materialise the hipEvent_t handles
        for ev in quad:
            ev.record()

    oneshot = mdist.OneShotExchange(colsum.numel()) if (use_dist and opts.exchange == "oneshot") else None

    def step(quad=None):
        if quad is not None:
            lib.mxm_set_timing_events(quad[0].cuda_event, quad[1].cuda_event)
        plan.em_iter(props_cur, ln_cur, state, colsum)
        if quad is not None:
            lib.mxm_set_timing_events(None, None)
        if use_dist:
            if quad is not None:
                quad[2].record()
            if oneshot is not None:
                oneshot.reduce(colsum, state)
            else:
                dist.all_reduce(colsum, op=dist.ReduceOp.SUM)
            if quad is not None:
                quad[3].record()
        plan.finalize(colsum, ln_cur, ln_new, props_cur, state, 0.0, total + 1)

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def run_loop(n_iters):
        """n_iters iterations of EVERY restart through the product's own loop driver (mxm_em_loop on one
        rank, dist.sharded_em_loop over several): restarts advance in full tiles dealt round-robin, so
        a "step" of B restarts costs B / tile passes over the matrix, not ceil(B / tile)."""
        if use_dist:
            _, _, sts = mdist.sharded_em_loop(plan, init, 0.0, n_iters, check_every=16,
                                              exchange=oneshot if oneshot is not None else "rccl")
        else:
            _, _, sts = em.em_loop(plan, init, 0.0, n_iters)
        return sts

    batched = n_runs > 1
    issue_s = None
    if batched:
        if opts.warmup > 0:
            run_loop(opts.warmup)
        fence()
        t0 = time.perf_counter()
        loop_states = run_loop(opts.steps)
        fence()
        elapsed = time.perf_counter() - t0
    else:
        for _ in range(opts.warmup):
            step()
        fence()
        t0 = time.perf_counter()
        for i in range(opts.steps):
            step(evs[i])
        issue_s = time.perf_counter() - t0           # the host has ENQUEUED every step by now (nothing waited for yet)
        fence()
        elapsed = time.perf_counter() - t0
    if use_dist:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    if batched:
        # one pass of a full tile, measured after the timed region (the loop driver's last launches are
        # no-ops of restarts that have already done their iterations)
        tile = min(n_runs, plan.restart_tile())
        lib.mxm_set_timing_events(evs[0][0].cuda_event, evs[0][1].cuda_event)
        plan.em_iter(props_cur[:tile], ln_cur[:tile], state[:tile], colsum[:tile])
        lib.mxm_set_timing_events(None, None)
        torch.cuda.synchronize()
        kernel_ms = numpy.array([evs[0][0].elapsed_time(evs[0][1])])
        all_reduce_us = None
        st = (loop_states[0][0], loop_states[0][1] + opts.warmup, loop_states[0][2])
        total_ok = all(s[0] == 2 and s[1] == opts.steps for s in loop_states)
    else:
        kernel_ms = numpy.array([q[0].elapsed_time(q[1]) for q in evs])
        # colreduce done -> all-reduced sums visible to this rank's stream (includes waiting for the slowest rank)
        all_reduce_us = float(numpy.mean([q[2].elapsed_time(q[3]) for q in evs]) * 1e3) if use_dist else None
        st = em.read_state(state)[0]
        total_ok = st[1] == total
    kernel_ms_per_rank = [float(kernel_ms.mean())]
    all_reduce_us_per_rank = None
    if use_dist:
        # per-rank view of a step, so that a scaling result explains itself: streaming kernel, the exchange as this
        # rank's stream saw it (colreduce done -> all-reduced sums visible: includes waiting for the slowest rank)
        mine = torch.tensor([float(kernel_ms.mean()), all_reduce_us if all_reduce_us is not None else -1.0],
                            dtype=torch.float64, device=dev)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        kernel_ms_per_rank = [float(x[0].item()) for x in every]
        if all_reduce_us is not None:
            all_reduce_us_per_rank = [float(x[1].item()) for x in every]
    # sanity (untimed): one more E+M pass; the M-step sums  sum_h p_h T_h  must add up to the
    # total weight of all ranks' rows, once per restart
    plan.em_iter(props_cur, ln_cur, state, colsum)
    if use_dist:
        dist.all_reduce(colsum, op=dist.ReduceOp.SUM)
    mass = float((props_cur * colsum).sum().item())
    sane = total_ok and abs(mass - total_rows * n_runs) < 1e-6 * total_rows * n_runs
    if rank == 0:
        log("%d steps in %.4f s; streaming kernel avg %.4f ms (min %.4f, max %.4f); "
            "sum(colsum)=%.6f iters=%d%s" % (opts.steps, elapsed, kernel_ms.mean(), kernel_ms.min(),
                                             kernel_ms.max(), mass, st[1],
                                             "" if all_reduce_us is None else "; all-reduce %.1f us" % all_reduce_us))

    # ---- posterior pass (reported, not part of the step) ----------------------------------------
    posterior_ms = None
    try:
        if mat is None:
            raise ValueError("no dense matrix in this run (--records)")
        out = torch.empty((n_rows, n_haps), dtype=torch.float64, device=dev)
        ln_theta = ln_cur[0].clone()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        em.posterior(plan, ln_theta, out=out)
        torch.cuda.synchronize()
        posterior_ms = (time.perf_counter() - t0) * 1e3
        del out
    except Exception as exc:       # out of memory at very large --rows: report and go on
        log("posterior pass skipped: %s" % exc)

    # ---- CPU leg (rank 0, N = 1): oracle em_step on a bounded sample of the same matrix, timed as the
    # cpu_baseline, and its result compared with the HIP path on the same slab (parity_in_run) -----
    cpu = None
    parity = None
    if rank == 0 and world == 1 and not opts.no_cpu_baseline:
        n_cpu = min(n_rows, opts.cpu_rows)
        dense_rows = mat[:n_cpu] if mat is not None else slab
        sample = dense_rows.cpu().numpy()
        leg = cpu_reference_leg(sample, opts.cpu_iters)
        cpu = {"value": leg["rate"], "unit": "cells/s", "cores": 1, "kind": "port",
               "sample": "oracle em_step (numpy restatement of em.py:57-91) x%d on the first %d rows "
                         "x %d haps of the same matrix, %.1f s; host has %d cores, 1 used like the "
                         "reference" % (leg["iters"], n_cpu, n_haps, leg["seconds"], os.cpu_count())}
        log("cpu baseline: %.3g cells/s (%.1f s)" % (leg["rate"], leg["seconds"]))
        try:
            blas = cpu_linear_blas_leg(sample, leg["init"], max(leg["iters"], 4) * 4)
            cpu["linear_blas_variant"] = {
                "value": blas["rate"], "unit": "cells/s", "cores": blas["threads"],
                "note": "NOT the reference's algorithm: numpy on the host's multi-threaded BLAS with the GPU loop's change of "
                        "variables (P = exp(M - rowmax) once, two matrix-vector products per iteration), %d iterations on the "
                        "same %d rows, %.2f s" % (blas["iters"], n_cpu, blas["seconds"])}
            log("cpu, linear-space BLAS variant (not the reference's algorithm): %.3g cells/s on %d threads"
                % (blas["rate"], blas["threads"]))
        except Exception as exc:
            log("linear-space BLAS variant skipped: %s" % exc)
        parity = parity_in_run(em, torch, dense_rows, leg, opts.cpu_iters, storage=opts.storage)
        log("parity in run: max |dprops| %.2e, iterations equal %s, haplogroup calls equal %s"
            % (parity["max_abs_dprops"], parity["iters_equal"], parity["argmax_equal"]))
        sane = sane and parity["max_abs_dprops"] < PARITY_PROPS_BAR and parity["iters_equal"] \
            and parity["argmax_equal"]

    # ---- the same iteration over row dictionaries (lossless; reported beside the dense fp64 line) ----
    coded_info = None
    if rank == 0 and world == 1 and opts.storage == "f64" and n_runs == 1:
        try:
            coded_info = coded_leg(em, torch, lib, plan, mat, wts, props_cur, ln_cur, opts.steps)
            log("row dictionaries: %.3f ms per iteration in the one-launch loop, %.3f through the per-iteration kernels "
                "(%.2f GB read), encode %.1f ms, column sums within %.1e"
                % (coded_info["one_launch_loop_ms_per_iteration"], coded_info["per_iteration_kernels_ms_per_step"],
                   coded_info["bytes_per_iteration"] / 1e9, coded_info["encode_ms"], coded_info["max_rel_dcolsum"]))
        except Exception as exc:
            log("coded-storage leg skipped: %s" % exc)

    if rank != 0:
        return None
    cells = float(total_rows) * n_haps
    elem = 4.0 if opts.storage == "f32" else 8.0
    algo_bytes = float(n_rows) * n_haps * elem        # this rank's stored matrix is read once per iteration
    if plan.coded is not None:
        algo_bytes = float(plan.coded_record_bytes)   # the records em_iter_coded_kernel reads (the rows that stay
                                                      # dense go through em_iter_wide_kernel afterwards)
    achieved = algo_bytes / (kernel_ms.mean() * 1e-3)
    traffic = None
    kernel_name = {"f64": "em_iter_wide_kernel", "f32": "em_iter_wide_f32_kernel", "coded": "em_iter_coded_kernel"}[plan.storage]
    if plan.storage == "f64":
        import ctypes
        name_buf = ctypes.create_string_buffer(96)
        tile = min(n_runs, plan.restart_tile())
        if lib.mxm_describe_stream_kernel(n_haps, tile, name_buf, len(name_buf)) == 0:
            kernel_name = name_buf.value.decode()
        if n_runs == 1:
            traffic = pmc_traffic(n_rows, n_haps, "f64", kernel_name, algo_bytes)
    has_quads = plan.coded is not None and getattr(plan, "_quad_keep", None) is not None
    rec_kind = "records" if records is not None else "encoded"
    if has_quads:                                     # both row passes in one grid, most rows from the quad records
        kernel_name = "em_iter_quad_coded_kernel" if min(n_runs, plan.restart_tile()) < 3 else "em_iter_quad_batched_kernel (a tile of 3 restarts)"
        if n_runs == 1:                               # calibrated like the records' kernel (pmc_calibrate_coded.py --quads)
            traffic = pmc_traffic(n_rows, n_haps, "coded", "em_iter_quad_coded_kernel*", algo_bytes, matrix=rec_kind)
    elif plan.storage == "coded" and n_runs == 1:
        # the records' kernel: counter bytes calibrated against a bare reader of exactly the same records in the same
        # counter pass (tools/pmc_calibrate_coded.py -> tools/pmc_summary.py); same [0.9, 1.5] x rule as the dense line
        traffic = pmc_traffic(n_rows, n_haps, "coded", "em_iter_coded_kernel*", algo_bytes, matrix=rec_kind)
    return {
        "metric": "read x hap cells/sec through the EM iteration (EM iters/sec reported beside it as "
                  "em_iters_per_s), %d reads x %d haps in total, whole job" % (total_rows, n_haps),
        "value": cells * n_runs * opts.steps / elapsed,
        "unit": "cells/s",
        "em_iters_per_s": n_runs * opts.steps / elapsed,
        "n_gpus": world, "steps": opts.steps, "warmup": opts.warmup,
        "ms_per_step": elapsed / opts.steps * 1e3,
        "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
        "dtype": {"f64": "f64", "f32": "f64 arithmetic on f32-stored matrix (opt-in variant)",
                  "coded": "f64 (matrix stored as lossless row dictionaries: opt-in variant)"}[plan.storage],
        "data": "synthetic (synth-v1 reads in blocks of %d, matrix built on device)" % 125000,
        "config": {"workload": "%d reads x %d haplogroups in total (Phylotree B17 + RSRS), %d per rank, "
                               "%d EM restart(s) advanced together (tile %d; several restarts: full tiles "
                               "dealt round-robin by the loop driver), %s matrix"
                               % (total_rows, n_haps, n_rows, n_runs, opts.batch_tile if plan.storage != "coded" else plan.restart_tile(),
                                  {"f64": "fp64", "f32": "fp32-stored", "coded": "row-dictionary (lossless fp64)"}[plan.storage]
                                  + (", records straight from the build, no dense matrix on the device" if records is not None else "")),
                   "total_rows": total_rows, "rows_per_gpu": n_rows, "haps": n_haps, "restarts": n_runs,
                   "scaling": scaling,
                   "sharding": "rows over %d rank(s), 1 all-reduce of %d fp64 per iteration"
                               % (world, n_haps * n_runs)},
        "roofline": {"bound": "hbm", "achieved": achieved / 1e9, "peak": HBM_PEAK_BYTES_PER_S / 1e9,
                     "unit": "GB/s", "frac": achieved / HBM_PEAK_BYTES_PER_S,
                     "traffic": traffic[0] if traffic else None,
                     "traffic_source": traffic[1] if traffic else None,
                     "kernel": kernel_name,
                     "kernel_ms": float(kernel_ms.mean()),
                     "kernel_ms_min": float(kernel_ms.min()), "kernel_ms_max": float(kernel_ms.max()),
                     "algorithmic_bytes_per_launch": algo_bytes},
        "cpu_baseline": cpu,
        "parity_in_run": parity,
        "coded_storage": coded_info,
        "all_reduce_us": all_reduce_us,
        "exchange": opts.exchange if use_dist else None,
        "all_reduce_us_per_rank": all_reduce_us_per_rank,
        "all_reduce_us_min_mean_max": (None if not all_reduce_us_per_rank else
                                       [min(all_reduce_us_per_rank), sum(all_reduce_us_per_rank) / len(all_reduce_us_per_rank),
                                        max(all_reduce_us_per_rank)]),
        "kernel_ms_per_rank": kernel_ms_per_rank,
        # what a step spends outside the streaming kernel and the exchange on the slowest rank: column reduce, finalize,
        # launch gaps, host (one restart per step only: with several the passes of a step are not one kernel)
        # host time to enqueue one step (library calls + the collective's launch): while it stays well below ms_per_step the
        # host runs ahead of the GPU and is not what a step waits for (no graph capture needed at this step size)
        "host_issue_us_per_step": None if issue_s is None else issue_s / opts.steps * 1e6,
        "step_remainder_us": (None if batched else
                              (elapsed / opts.steps * 1e3 - max(kernel_ms_per_rank)) * 1e3
                              - (max(all_reduce_us_per_rank) if all_reduce_us_per_rank else 0.0)),
        # What this run's own numbers allow at best: with the exchange free, a step costs the slowest rank's kernel plus
        # the fixed remainder, and one GPU holding all N shards would need N kernels plus the same remainder
        # (the streaming kernel's time is linear in the rows).  ceiling = (N k + rem) / (k + rem); the measured
        # speed-up can only be lower (the all-reduce, waiting for the slowest rank).
        "projected_scaling_ceiling": (None if batched else _scaling_ceiling(world, max(kernel_ms_per_rank),
                                                                            elapsed / opts.steps * 1e3, all_reduce_us_per_rank)),
        "matrix_build_cells_per_s": float(n_rows) * n_haps / build_s,
        "linearize_ms": linearize_s * 1e3,
        "posterior_pass_ms": posterior_ms,
        "sanity_ok": bool(sane),
    }


def _scaling_ceiling(world, kernel_ms, step_ms, all_reduce_us_per_rank):
    exchange_ms = (max(all_reduce_us_per_rank) if all_reduce_us_per_rank else 0.0) * 1e-3
    rem_ms = max(step_ms - kernel_ms - exchange_ms, 0.0)
    return {"speedup_at_most": (world * kernel_ms + rem_ms) / (kernel_ms + rem_ms), "n_gpus": world,
            "kernel_ms": kernel_ms, "remainder_ms": rem_ms, "exchange_ms": exchange_ms,
            "formula": "(N * kernel + remainder) / (kernel + remainder): the exchange taken as free"}


def bench_restarts(opts, env):
    """
    BASELINE config 5: the matrix replicated on every rank, --restarts EM restarts dealt over the
    ranks (run i -> rank i % N), each rank keeps one full tile of restarts in flight and refills a
    slot when its restart converges; no per-iteration traffic; at the end one all-reduce of the
    summed log-proportions and the row-block exchange of the folded posterior.
    """
    import argparse as _ap
    import numpy
    import torch
    import torch.distributed as dist
    from mixemt_amd import dist as mdist
    (dev, rank, world, use_dist, mat, wts, n_rows, n_haps, build_s) = (
        env[k] for k in ("dev", "rank", "world", "use_dist", "mat", "wts", "n_rows", "n_haps", "build_s"))
    as_records = env.get("records") is not None
    args = _ap.Namespace(init_alpha=1.0, tolerance=1e-4, max_iter=10000, n_multi=opts.restarts, verbose=False)
    numpy.random.seed(7)
    timing = {}
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = mdist.run_em_restart_parallel(mat, wts, args, timing=timing, records=env.get("records"))
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    stats = torch.tensor([timing["loop_s"], timing["fold_s"], timing["combine_s"], wall],
                         dtype=torch.float64, device=dev)
    if use_dist:
        every = [torch.zeros_like(stats) for _ in range(world)]
        dist.all_gather(every, stats)
    else:
        every = [stats]
    every = torch.stack(every).cpu().numpy()
    loop_s = every[:, 0]
    iters = res["iters"]
    total_iters = int(sum(iters))
    # sanity: every restart stopped by convergence, the proportions are a geometric mean of unit-sum vectors
    sane = all(d == 1 for d in res["done"]) and float(res["run_props"].sum(axis=1).min()) > 0.999999
    lo, hi = res["rows"]
    if res["read_mix"] is not None:
        lse = torch.logsumexp(res["read_mix"][: min(hi - lo, 4096)], dim=1)
        # a log-mean-exp of row-normalised posteriors is itself row-normalised
        sane = sane and float(lse.abs().max().item()) < 1e-9
    if rank != 0:
        return None
    cells = float(n_rows) * n_haps
    return {
        "metric": "EM restart-iterations/sec over %d restarts run to convergence (cells/sec = x R x H), "
                  "%d reads x %d haps replicated on every GPU, whole job" % (opts.restarts, n_rows, n_haps),
        "value": cells * total_iters / float(loop_s.max()),
        "unit": "cells/s",
        "restart_iters_per_s": total_iters / float(loop_s.max()),
        "n_gpus": world, "steps": total_iters, "warmup": 0,
        "ms_per_step": float(loop_s.max()) / max(total_iters, 1) * 1e3,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f64 (matrix stored as lossless row dictionaries: opt-in variant)" if as_records else "f64",
        "data": "synthetic (synth-v1 reads in blocks of %d, matrix built on device)" % 125000,
        "config": {"workload": "%d EM restarts (sequential Dirichlet inits after seed 7) on %d reads x %d "
                               "haplogroups replicated per GPU%s, dealt over %d rank(s), tile %d with slot refill"
                               % (opts.restarts, n_rows, n_haps, " as row-dictionary records + quad dictionary" if as_records else "",
                                  world, 3 if as_records else opts.batch_tile),
                   "rows_per_gpu": n_rows, "haps": n_haps, "restarts": opts.restarts, "mode": "restarts"},
        "iters_per_restart": [int(x) for x in iters],
        "loop_s_per_rank": [float(x) for x in loop_s],
        "idle_tail_s_per_rank": [float(loop_s.max() - x) for x in loop_s],
        "posterior_fold_s_per_rank": [float(x) for x in every[:, 1]],
        "combine_s_per_rank": [float(x) for x in every[:, 2]],
        "wall_s": float(every[:, 3].max()),
        "matrix_build_cells_per_s": cells / build_s,
        "roofline": None, "cpu_baseline": None,
        "sanity_ok": bool(sane),
    }


if __name__ == "__main__":
    main()
