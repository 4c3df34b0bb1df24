"""
Golden g10: the reference's own build_em_matrix + run_em on 20 000 synth-v1 reads x 5408 haplogroups with
de-duplication-style weights (tools/gen_golden.py: 265 s of reference build over 6 processes, 3761 s of reference EM,
534 iterations) -- the size at which the product leaves the one-launch loops (1.08 * 10^8 cells) and at which
storage="auto" takes the row-dictionary branch.  Every way the product can run this input must reproduce the
reference: identical stopping iteration, haplogroup call of every row, votes, proportions within 1e-9
(reference: mixemt/preprocess.py:177-198, mixemt/em.py:94-165).
The fixture stores the generator's seed instead of the 2 MB of CSR observations; its digest pins them.
"""
import hashlib
import os
import socket

import numpy
import pytest

from conftest import em_args, golden

pytestmark = pytest.mark.gpu

PROPS_ATOL = 1e-9


def _sha(arr):
    return hashlib.sha256(numpy.ascontiguousarray(arr).tobytes()).hexdigest()


def _inputs(tables, ref_len, g):
    from mixemt_amd import synth
    row_ptr, site, obs, _ = synth.synth_reads(tables, ref_len, int(g["n_rows"]), seed=int(g["synth_seed"]))
    assert _sha(row_ptr) + _sha(site) + _sha(obs) == str(g["csr_sha256"])
    return row_ptr, site, obs, g["wts"].astype(numpy.int64)


def _check(res, g, best=None, votes=None, mix=None):
    assert numpy.array_equal(res["inits"], g["inits"])
    assert res["iters"] == list(g["iters"]) and res["done"] == [1]
    assert numpy.abs(res["props"] - g["props"]).max() < PROPS_ATOL
    if mix is not None:
        host = mix.cpu().numpy()
        best = host.argmax(axis=1)
        assert numpy.allclose(host[g["mix_pick"]], g["mix_rows"], rtol=0, atol=1e-8)
        assert numpy.allclose(host.max(axis=1), g["mix_rowmax"], rtol=0, atol=1e-8)
    if best is not None:
        assert _sha(best.astype(numpy.int32)) == str(g["mix_argmax_sha256"])      # (the generator hashed int32)
        assert numpy.array_equal(best, g["mix_argmax"])
    if votes is not None:
        assert numpy.array_equal(votes, g["votes"])
        assert numpy.array_equal(numpy.flatnonzero(votes >= 10), g["contributors"])


@pytest.fixture(scope="module")
def g10(b17):
    from mixemt_amd import preprocess
    refseq, phy, haps, tables = b17
    g = golden("g10_run_em_20k")
    row_ptr, site, obs, wts = _inputs(tables, len(refseq), g)
    mat = preprocess.build_em_matrix_device(tables, row_ptr, site, obs)
    return dict(g=g, row_ptr=row_ptr, site=site, obs=obs, wts=wts, mat=mat, tables=tables, haps=haps)


def test_build_equals_the_reference_matrix(g10):
    g, mat = g10["g"], g10["mat"]
    host = mat.cpu().numpy()
    assert _sha(host) == str(g["mat_sha256"])                       # every one of 1.08e8 cells, bit for bit
    assert numpy.array_equal(host.sum(axis=1), g["mat_row_sum"])


def test_dense_per_iteration_kernels_reproduce_the_reference_run(g10):
    """Above 10^8 cells a single restart takes the per-iteration streaming kernels (many rows per workgroup)."""
    from mixemt_amd import assign, em
    g = g10["g"]
    numpy.random.seed(23)
    res = em.run_em_ex(g10["mat"], g10["wts"], em_args(), storage="f64")
    assert res["storage"] == "f64"
    _check(res, g, mix=res["read_mix"])
    best, votes = assign.row_argmax_votes(res["read_mix"], g10["wts"])
    _check(res, g, best=best, votes=votes)


def test_storage_auto_takes_the_row_dictionaries_and_reproduces_it(g10):
    from mixemt_amd import em
    g = g10["g"]
    numpy.random.seed(23)
    res = em.run_em_ex(g10["mat"], g10["wts"], em_args())            # the default
    assert res["storage"] == "coded"                                 # 1.08e8 cells > 1.5e7: the coded branch
    _check(res, g, mix=res["read_mix"])


def test_records_from_the_build_reproduce_it_without_any_dense_matrix(g10):
    from mixemt_amd import assign, em, preprocess
    g = g10["g"]
    cm = preprocess.build_em_records_device(g10["tables"], g10["row_ptr"], g10["site"], g10["obs"])
    numpy.random.seed(23)
    res = em.run_em_ex(None, g10["wts"], em_args(), want_read_mix=False, records=cm)
    best, votes = assign.row_argmax_votes_records(cm, res["ln_theta_k"], g10["wts"])
    _check(res, g, best=best, votes=votes)
    assert int(cm.rest_rows.numel()) == 0 and 0 < int(cm.wide_rows().numel()) < 0.05 * cm.n_rows   # no row stays dense (round 4); some have 16-bit codes


def _worker(rank, world, port, out_dir):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    sys.path.insert(0, os.path.dirname(here))
    import torch
    import torch.distributed as dist
    from conftest import em_args as mk
    from mixemt_amd import _lib, dist as mdist, phylotree, preprocess
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    _lib.load().mxm_set_loop_fused(0, 0)                 # two processes share the GPU
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
        numpy.random.seed(23 if rank == 0 else 999)      # only rank 0's stream may matter
        res = mdist.run_em_sharded(shard, torch.from_numpy(wts[lo:hi]).cuda(), mk(), check_every=8)
        mix = res["read_mix"].cpu().numpy()
        numpy.savez(os.path.join(out_dir, "rank%d.npz" % rank), props=res["props"], iters=numpy.array(res["iters"]),
                    best=mix.argmax(axis=1), lo=lo, hi=hi, inits=res["inits"])
    finally:
        dist.destroy_process_group()


def test_two_ranks_row_sharded_reproduce_it(tmp_path):
    """The same run with the rows split over two ranks (gloo over the one GPU): every rank stops on the reference's
    iteration with the reference's proportions, and owns the calls of its rows."""
    import torch
    import torch.multiprocessing as mp
    torch.cuda.empty_cache()
    g = golden("g10_run_em_20k")
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    res = [numpy.load(str(tmp_path / ("rank%d.npz" % r))) for r in range(2)]
    for r in res:
        assert numpy.array_equal(r["inits"], g["inits"])
        assert list(r["iters"]) == list(g["iters"])
        assert numpy.abs(r["props"] - g["props"]).max() < PROPS_ATOL
        assert numpy.array_equal(r["props"], res[0]["props"])
        assert numpy.array_equal(r["best"], g["mix_argmax"][int(r["lo"]):int(r["hi"])])
    assert int(res[0]["hi"]) == int(res[1]["lo"]) == 10000
