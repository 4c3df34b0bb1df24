#!/usr/bin/env python
"""
bench.py -- EM hot-path throughput on MI355X (BASELINE.json metric:
"EM iters/sec + read x hap cells/sec, 1M reads x 5.4k haps, 1/2/4/8 GPUs").

    python bench.py [--gpus N] [--steps K] [--warmup W] [--rows R_PER_GPU]

A "step" is one full EM iteration (em.py:57-91 + the convergence bookkeeping
of em.py:126-143) over this rank's resident rows of the synthetic (synth-v1)
reads x haplogroups matrix: mxm_em_iter -> [RCCL all-reduce of the H column
sums when N > 1] -> mxm_m_finalize.  The tolerance is 0 inside the timed
region so that exactly K iterations run; nothing is skipped or cached.

Inputs are resident in HBM before the clock starts (the matrix is BUILT on the
device from CSR observations by mxm_build_em_matrix; build and posterior pass
are timed separately and reported as extra fields).  Rows are sharded over
ranks with per-GPU work fixed ("weak"); value = cells of all ranks / time.

One JSON line on stdout (rank 0); progress on stderr.
"""

import argparse
import json
import os
import sys
import time

import numpy

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_BYTES_PER_S = 8.0e12        # MI355X HBM3E spec (MI355X_MICROARCH.md)


def log(msg):
    sys.stderr.write("[bench] %s\n" % msg)
    sys.stderr.flush()


def cpu_baseline(mat_rows, n_iters):
    """
    The oracle's em_step (numpy restatement of em.py:57-91, single thread like
    the reference) on a bounded sample of the same matrix.  CHECKER/baseline
    only -- never part of the measured GPU path.
    """
    from oracle import em_oracle
    n_rows, n_haps = mat_rows.shape
    wts = numpy.ones(n_rows, dtype=numpy.int64)
    numpy.random.seed(7)
    ln_props = numpy.log(numpy.random.dirichlet([1.0] * n_haps))
    buf = numpy.empty_like(mat_rows)
    t0 = time.perf_counter()
    for _ in range(n_iters):
        buf, ln_props = em_oracle.em_step(mat_rows, wts, ln_props, buf)
    dt = time.perf_counter() - t0
    return n_rows * n_haps * n_iters / dt, dt


def pmc_traffic(n_rows, n_haps):
    """
    HBM bytes per launch of the streaming kernel from the committed rocprofv3 PMC
    passes (profiles/*/pmc_traffic_*.json, written by tools/pmc_summary.py from
    separate --pmc FETCH_SIZE / --pmc WRITE_SIZE runs of this script).  Counters
    cannot be read from inside the process, so this is the latest profiled value
    for the same workload shape, or None.
    """
    import glob
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "pmc_traffic_*.json"))):
        try:
            with open(path) as fin:
                data = json.load(fin)
        except (OSError, ValueError):
            continue
        meta = data.get("_workload", {})
        if meta.get("rows_per_gpu") != n_rows or meta.get("haps") != n_haps:
            continue
        for name, rec in data.items():
            if name.startswith("em_iter_wide_kernel"):
                best = (rec["hbm_bytes_per_launch"], os.path.relpath(path, ROOT))
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rows", type=int, default=1000000, help="reads (matrix rows) per GPU")
    ap.add_argument("--cpu-rows", type=int, default=32768)
    ap.add_argument("--cpu-iters", type=int, default=4)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--restarts", type=int, default=1,
                    help="EM restarts advanced together (config 3 uses 10); a step then is one "
                         "iteration of EVERY restart and value counts cells x restarts")
    ap.add_argument("--batch-tile", type=int, default=4,
                    help="restarts sharing one pass over the matrix (1 = unbatched schedule)")
    ap.add_argument("--storage", default="f64", choices=["f64", "f32"],
                    help="element type of the streamed matrix; f32 is a labelled opt-in variant "
                         "(fp64 math on float-stored P), never the default")
    ap.add_argument("--min-rows-per-wg", type=int, default=0, help="tuning knob (0 = library default)")
    ap.add_argument("--build-kernel", default="auto", choices=["auto", "packed", "bytes"])
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise the process group and issue the per-iteration all-reduce even with "
                         "one rank (exercises the RCCL path on a single-GPU box)")
    ap.add_argument("--backend", default="nccl",
                    help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to test the "
                         "multi-process path on a single GPU)")
    opts = ap.parse_args()

    import torch
    import torch.distributed as dist
    from mixemt_amd import _lib, em, phylotree, preprocess, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != opts.gpus:
        log("warning: --gpus %d but WORLD_SIZE=%d; using WORLD_SIZE" % (opts.gpus, world))
    n_dev = torch.cuda.device_count()
    if opts.backend == "nccl" and world > n_dev:
        raise SystemExit("bench: %d ranks but %d GPU(s) visible" % (world, n_dev))
    dev = torch.device("cuda", local_rank % max(n_dev, 1))
    torch.cuda.set_device(dev)
    use_dist = world > 1 or opts.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if opts.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(opts.backend, rank=rank, world_size=world)
    lib = _lib.load()

    # ---- inputs: Build 17 + RSRS tables, synth-v1 reads, matrix built on the device -------------
    t0 = time.perf_counter()
    refseq = phylotree.load_rsrs()
    phy = phylotree.load_build17(refseq)
    haps = sorted(phy.hap_var)
    tables = preprocess.HapVarTables.build(refseq, phy, haps)
    n_rows, n_haps = opts.rows, len(haps)
    row_ptr, site, obs, _ = synth.synth_reads(tables, len(refseq), n_rows, seed=opts.seed + rank)
    if rank == 0:
        log("tables + %d synthetic reads (%.1f sites/read) on host: %.1f s"
            % (n_rows, row_ptr[-1] / float(n_rows), time.perf_counter() - t0))
    row_ptr_d = torch.from_numpy(row_ptr).to(dev)
    site_d = torch.from_numpy(site.view(numpy.int16)).to(dev)
    obs_d = torch.from_numpy(obs).to(dev)
    mat = torch.empty((n_rows, n_haps), dtype=torch.float64, device=dev)
    tables.device()
    if opts.build_kernel == "packed":
        tables.packed_device()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    preprocess.build_em_matrix_device(tables, row_ptr_d, site_d, obs_d, out=mat, kernel=opts.build_kernel)
    torch.cuda.synchronize()
    build_s = time.perf_counter() - t0
    wts = torch.ones(n_rows, dtype=torch.float64, device=dev)
    plan = em.EmPlan(mat, wts, n_runs=opts.restarts, storage=opts.storage)          # allocates P and linearises once (untimed: hipMalloc)
    torch.cuda.synchronize()
    t0 = time.perf_counter()                      # timed again on the now-resident buffers
    lin_fn = lib.mxm_linearize_f32 if opts.storage == "f32" else lib.mxm_linearize
    _lib.check(lin_fn(mat.data_ptr(), mat.stride(0), n_rows, n_haps, plan.lin.data_ptr(),
                      plan.lin.stride(0), plan.rowmax.data_ptr(),
                      torch.cuda.current_stream().cuda_stream), "mxm_linearize")
    torch.cuda.synchronize()
    linearize_s = time.perf_counter() - t0
    if rank == 0:
        log("matrix build %.3f s (%.3g cells/s), linearize %.3f s, HBM in use %.1f GB"
            % (build_s, n_rows * n_haps / build_s, linearize_s,
               torch.cuda.memory_allocated() / 1e9))

    # ---- loop state: one restart, init = first Dirichlet draw after numpy.random.seed(7) --------
    n_runs = opts.restarts
    lib.mxm_set_batch_tile(opts.batch_tile)
    if opts.min_rows_per_wg > 0:
        lib.mxm_set_min_rows_per_wg(opts.min_rows_per_wg)
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
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
           for _ in range(opts.steps)]
    for a, b in evs:                       # materialise the hipEvent_t handles
        a.record()
        b.record()

    def step(pair=None):
        if pair is not None:
            lib.mxm_set_timing_events(pair[0].cuda_event, pair[1].cuda_event)
        plan.em_iter(props_cur, ln_cur, state, colsum)
        if pair is not None:
            lib.mxm_set_timing_events(None, None)
        if use_dist:
            dist.all_reduce(colsum, op=dist.ReduceOp.SUM)
        plan.finalize(colsum, ln_cur, ln_new, props_cur, state, 0.0, total + 1)

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(opts.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for i in range(opts.steps):
        step(evs[i])
    fence()
    elapsed = time.perf_counter() - t0
    if use_dist:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    kernel_ms = numpy.array([a.elapsed_time(b) for a, b in evs])
    st = em.read_state(state)[0]
    # sanity (untimed): one more E+M pass; the M-step sums  sum_h p_h T_h  must add up to the
    # total weight of all ranks' rows, once per restart
    plan.em_iter(props_cur, ln_cur, state, colsum)
    if use_dist:
        dist.all_reduce(colsum, op=dist.ReduceOp.SUM)
    mass = float((props_cur * colsum).sum().item())
    sane = (st[1] == total) and abs(mass - n_rows * world * n_runs) < 1e-6 * n_rows * world * n_runs
    if rank == 0:
        log("%d steps in %.4f s; streaming kernel avg %.4f ms (min %.4f, max %.4f); "
            "sum(colsum)=%.6f iters=%d" % (opts.steps, elapsed, kernel_ms.mean(), kernel_ms.min(),
                                            kernel_ms.max(), mass, st[1]))

    # ---- posterior pass (reported, not part of the step) ----------------------------------------
    posterior_ms = None
    try:
        out = torch.empty((n_rows, n_haps), dtype=torch.float64, device=dev)
        with numpy.errstate(divide="ignore"):
            ln_theta = ln_cur[0].clone()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        em.posterior(plan, ln_theta, out=out)
        torch.cuda.synchronize()
        posterior_ms = (time.perf_counter() - t0) * 1e3
        del out
    except Exception as exc:       # out of memory at very large --rows: report and go on
        log("posterior pass skipped: %s" % exc)

    # ---- CPU baseline: oracle em_step on a bounded sample of the same matrix (rank 0, N = 1) ----
    cpu = None
    if rank == 0 and world == 1 and not opts.no_cpu_baseline:
        n_cpu = min(n_rows, opts.cpu_rows)
        sample = mat[:n_cpu].cpu().numpy()
        rate, dt = cpu_baseline(sample, opts.cpu_iters)
        cpu = {"value": rate, "unit": "cells/s", "cores": 1, "kind": "port",
               "sample": "oracle em_step (numpy restatement of em.py:57-91) x%d on the first %d rows "
                         "x %d haps of the same matrix, %.1f s; host has %d cores, 1 used like the "
                         "reference" % (opts.cpu_iters, n_cpu, n_haps, dt, os.cpu_count())}
        log("cpu baseline: %.3g cells/s (%.1f s)" % (rate, dt))

    if rank == 0:
        cells = float(n_rows) * n_haps
        elem = 4.0 if opts.storage == "f32" else 8.0
        algo_bytes = cells * elem                     # the stored matrix is read once per iteration
        achieved = algo_bytes / (kernel_ms.mean() * 1e-3)
        traffic = pmc_traffic(n_rows, n_haps) if (opts.storage == "f64" and n_runs == 1) else None
        line = {
            "metric": "read x hap cells/sec through the EM iteration (EM iters/sec reported beside it as "
                      "em_iters_per_s), 1M reads x 5.4k haps per GPU, whole job",
            "value": cells * world * n_runs * opts.steps / elapsed,
            "unit": "cells/s",
            "em_iters_per_s": n_runs * opts.steps / elapsed,
            "n_gpus": world, "steps": opts.steps, "warmup": opts.warmup,
            "ms_per_step": elapsed / opts.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64" if opts.storage == "f64" else "f64 arithmetic on f32-stored matrix (opt-in variant)",
            "data": "synthetic (synth-v1 reads, matrix built on device)",
            "config": {"workload": "%d reads x %d haplogroups per GPU (Phylotree B17 + RSRS), "
                                   "%d EM restart(s) advanced together (tile %d), %s matrix"
                                   % (n_rows, n_haps, n_runs, opts.batch_tile,
                                      "fp64" if opts.storage == "f64" else "fp32-stored"),
                       "rows_per_gpu": n_rows, "haps": n_haps, "restarts": n_runs,
                       "sharding": "rows over %d rank(s), 1 all-reduce of %d fp64 per iteration"
                                   % (world, n_haps)},
            "roofline": {"bound": "hbm", "achieved": achieved / 1e9, "peak": HBM_PEAK_BYTES_PER_S / 1e9,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_BYTES_PER_S,
                         "traffic": traffic[0] if traffic else None,
                         "traffic_source": traffic[1] if traffic else None,
                         "kernel": "em_iter_wide_kernel" if opts.storage == "f64" else "em_iter_wide_f32_kernel", "kernel_ms": float(kernel_ms.mean()),
                         "algorithmic_bytes_per_launch": algo_bytes},
            "cpu_baseline": cpu,
            "matrix_build_cells_per_s": cells / build_s,
            "linearize_ms": linearize_s * 1e3,
            "posterior_pass_ms": posterior_ms,
            "sanity_ok": bool(sane),
        }
        print(json.dumps(line))
        sys.stdout.flush()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
