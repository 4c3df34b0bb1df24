#!/usr/bin/env python
"""
Row-dictionary storage against the dense fp64 matrix on one GPU:
    python tools/time_coded.py [rows] [reps] [shapes, e.g. 0,4]
encodes the synth-v1 matrix, checks that the decoded rows equal mxm_linearize's output bit for bit,
and times one EM iteration (streaming kernel + column reduce) in both forms from the same proportions.
"""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy
import torch
from mixemt_amd import _lib, em, phylotree, preprocess, synth

em.QUADS = False        # this tool measures the records' own pass (em_iter_coded_kernel): no quad dictionary beside them

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
only = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else None     # shapes to time
refseq = phylotree.load_rsrs()
phy = phylotree.load_build17(refseq)
haps = sorted(phy.hap_var)
tables = preprocess.HapVarTables.build(refseq, phy, haps)
row_ptr, site, obs, _ = synth.synth_rows(tables, len(refseq), 0, rows, seed=1)
dev = torch.device("cuda")
mat = preprocess.build_em_matrix_device(tables, torch.from_numpy(row_ptr).to(dev),
                                        torch.from_numpy(site.view(numpy.int16)).to(dev), torch.from_numpy(obs).to(dev))
H = len(haps)
wts = torch.ones(rows, dtype=torch.float64, device=dev)
plan = em.EmPlan(mat, wts, n_runs=1)
lib = plan.lib
stream = torch.cuda.current_stream().cuda_stream


def timed(fn, n=reps):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


cap = lib.mxm_coded_bytes(rows, H)
rec = torch.empty(cap, dtype=torch.uint8, device=dev)
rec_off = torch.empty(rows, dtype=torch.int64, device=dev)
ndist = torch.empty(rows, dtype=torch.int32, device=dev)
rowmax = torch.empty(rows, dtype=torch.float64, device=dev)
stats = torch.zeros(2, dtype=torch.int64, device=dev)


def encode():
    _lib.check(lib.mxm_encode_rows(mat.data_ptr(), mat.stride(0), rows, H, rec.data_ptr(), cap, rec_off.data_ptr(),
                                   ndist.data_ptr(), rowmax.data_ptr(), stats.data_ptr(), stream), "mxm_encode_rows")


t_enc = timed(encode, 3)
used, left = [int(v) for v in stats.cpu()]
nd = ndist.cpu().numpy()
print("%d x %d: encode %.2f ms; records %.3f GB (dense P: %.3f GB, x%.2f smaller); %d rows stay dense (%.2f %%); "
      "table entries per coded row: mean %.1f, median %d, max %d"
      % (rows, H, t_enc * 1e3, used / 1e9, rows * H * 8 / 1e9, rows * H * 8.0 / max(used, 1), left, 100.0 * left / rows,
         nd[nd > 0].mean(), numpy.median(nd[nd > 0]), nd.max()))
assert torch.equal(rowmax, plan.rowmax), "rowmax differs from mxm_linearize"

rest_idx = torch.nonzero(ndist == 0).flatten()
p_rest = plan.lin.index_select(0, rest_idx).contiguous() if rest_idx.numel() else None
w_rest = wts.index_select(0, rest_idx).contiguous() if rest_idx.numel() else None
wide_idx = torch.nonzero(ndist > 256).flatten()
coded = _lib.Coded(rec.data_ptr(), rec_off.data_ptr(), ndist.data_ptr(), rows,
                   p_rest.data_ptr() if p_rest is not None else None, p_rest.stride(0) if p_rest is not None else 0,
                   w_rest.data_ptr() if w_rest is not None else None, int(rest_idx.numel()),
                   wide_idx.data_ptr() if wide_idx.numel() else None, int(wide_idx.numel()))

# decode == linearize, bit for bit (coded rows)
dec = torch.full_like(plan.lin, -1.0)
_lib.check(lib.mxm_decode_rows(ctypes.byref(coded), H, dec.data_ptr(), dec.stride(0), stream), "mxm_decode_rows")
same = True
for lo in range(0, rows, 50000):                     # in slabs: a boolean gather of the whole matrix would double it
    hi = min(rows, lo + 50000)
    ok_rows = ndist[lo:hi] > 0
    same = same and torch.equal(dec[lo:hi][ok_rows].view(torch.int64), plan.lin[lo:hi][ok_rows][:, :H].view(torch.int64))
print("decoded rows == mxm_linearize rows bit for bit: %s" % same)
del dec
assert same

numpy.random.seed(7)
p0 = numpy.random.dirichlet([1.0] * H)
props = torch.from_numpy(p0).to(dev).reshape(1, H)
lnp = torch.log(props)
state = em.new_state(1, dev)
cs_dense = torch.zeros(1, H, dtype=torch.float64, device=dev)
cs_coded = torch.zeros(1, H, dtype=torch.float64, device=dev)


def dense_iter():
    _lib.check(lib.mxm_em_iter(mat.data_ptr(), mat.stride(0), plan.lin.data_ptr(), plan.lin.stride(0), wts.data_ptr(),
                               props.data_ptr(), lnp.data_ptr(), rows, H, 1, state.data_ptr(), cs_dense.data_ptr(),
                               plan.ws.data_ptr(), plan.ws_bytes, stream), "mxm_em_iter")


def coded_iter():
    _lib.check(lib.mxm_em_iter_coded(ctypes.byref(coded), wts.data_ptr(), props.data_ptr(), H, 1, state.data_ptr(),
                                     cs_coded.data_ptr(), plan.ws.data_ptr(), plan.ws_bytes, stream), "mxm_em_iter_coded")


t_dense = timed(dense_iter)
t_coded = timed(coded_iter)
rel = ((cs_coded - cs_dense).abs() / cs_dense.abs().clamp_min(1e-300)).max().item()
print("one EM iteration (pass + column reduce): dense fp64 %.3f ms, row dictionaries %.3f ms (x%.2f); "
      "max relative difference of the column sums %.2e" % (t_dense * 1e3, t_coded * 1e3, t_dense / t_coded, rel))
print("  = %.3g cells/s dense, %.3g cells/s coded; coded bytes per iteration %.2f GB -> %.0f GB/s"
      % (rows * H / t_dense, rows * H / t_coded, used / 1e9, used / t_coded / 1e9))
