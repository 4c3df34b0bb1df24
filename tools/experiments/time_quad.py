#!/usr/bin/env python
"""
Round-5 experiment: the records row pass over a QUAD dictionary (quad_experiment.hip; profiles/r05/experiments.md 1d).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -I include -I mixemt_amd/csrc \
          tools/experiments/quad_experiment.hip -o tools/experiments/_build/libquad.so
    python tools/experiments/time_quad.py [rows] [lib]

The quad records are made here from the product's records with torch (a prototype of the encoder, not a product path):
a row's 1352 aligned groups of four code bytes are sorted, the distinct ones numbered, the table filled with the four
values each names.  Rows with more than 256 distinct quads (and wide rows, and rows without a record) get weight 0 in
BOTH kernels, so the column sums of the quad kernel can be compared with the product kernel's on the same rows.
"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy
import torch
from mixemt_amd import _lib, em, phylotree, preprocess, synth

em.QUADS = False        # this tool measures the records' own pass (em_iter_coded_kernel): no quad dictionary beside them

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
path = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "tools", "experiments", "_build", "libquad.so")
x = ctypes.CDLL(path)
P, I, L = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
x.quad_time.restype = ctypes.c_float
x.quad_time.argtypes = [I, P, P, P, P, P, L, I, P, L, I, I, ctypes.POINTER(I)]
refseq = phylotree.load_rsrs(); phy = phylotree.load_build17(refseq); haps = sorted(phy.hap_var)
tables = preprocess.HapVarTables.build(refseq, phy, haps)
H = len(haps)
row_ptr, site, obs, _ = synth.synth_rows(tables, len(refseq), 0, rows, seed=1)
cm = preprocess.build_em_records_device(tables, row_ptr, site, obs)
dev = cm.rec.device
nd = cm.ndist
ldc = (H + 7) & ~7
nquad_cols = ldc // 4
assert nquad_cols <= 1536
byte_rows = (nd > 0) & (nd <= 256)

# ---- quad records ---------------------------------------------------------------------------------------------------
QCODE = 2048
nq_all = torch.zeros(rows, dtype=torch.int32, device=dev)
qoff = torch.zeros(rows, dtype=torch.int64, device=dev)
chunk = 20000
# first pass: sizes; second pass: fill (the records go to one buffer at 32-byte aligned offsets)
pieces = []
ar_code = torch.arange(ldc, device=dev, dtype=torch.int64)
ar_tbl = torch.arange(256, device=dev, dtype=torch.int64)
tpos = torch.arange(256, device=dev, dtype=torch.int64)
slot_of = torch.full((256, 8), -1, dtype=torch.int64, device=dev)       # quad index of byte j of thread t
for j in range(6):
    qidx = tpos + 256 * j
    slot_of[:, j] = torch.where(qidx < nquad_cols, qidx, torch.full_like(qidx, -1))
for a in range(0, rows, chunk):
    b = min(rows, a + chunk)
    n = b - a
    off = cm.rec_off[a:b]
    ok = byte_rows[a:b]
    off_safe = torch.where(ok, off, torch.zeros_like(off))
    codes = cm.rec[(off_safe[:, None] + ar_code[None, :]).reshape(-1)].reshape(n, ldc)            # uint8
    ptab_raw = cm.rec[(off_safe[:, None, None] + ldc + 8 * ar_tbl[None, :, None] +
                       torch.arange(8, device=dev)[None, None, :]).clamp(max=cm.rec.numel() - 1).reshape(-1)]
    ptab = ptab_raw.reshape(n, 256, 8).contiguous().view(torch.float64).reshape(n, 256)
    ptab = torch.where(ar_tbl[None, :] < nd[a:b, None], ptab, torch.zeros_like(ptab))
    quads = codes.reshape(n, nquad_cols, 4).contiguous().view(torch.int32).reshape(n, nquad_cols).to(torch.int64) & 0xffffffff
    sv, si = torch.sort(quads, dim=1)
    new = torch.zeros_like(sv)
    new[:, 1:] = (sv[:, 1:] != sv[:, :-1]).to(torch.int64)
    rank = torch.cumsum(new, dim=1)
    nq = rank[:, -1] + 1
    qcode = torch.empty_like(rank)
    qcode.scatter_(1, si, rank)
    good = ok & (nq <= 256)
    nq = torch.where(good, nq, torch.zeros_like(nq))
    uq = torch.zeros((n, 256), dtype=torch.int64, device=dev)
    uq.scatter_(1, rank.clamp(max=255), sv)
    bytes4 = torch.stack([(uq >> (8 * e)) & 0xff for e in range(4)], dim=2)                       # [n, 256, 4]
    qtab = torch.gather(ptab, 1, bytes4.reshape(n, 1024)).reshape(n, 256, 4)
    # thread-contiguous code bytes
    qc = torch.zeros((n, 256, 8), dtype=torch.uint8, device=dev)
    for j in range(6):
        col = slot_of[:, j]
        live = col >= 0
        qc[:, live, j] = qcode[:, col[live]].clamp(max=255).to(torch.uint8)
    pieces.append((a, b, nq.to(torch.int32), qc.reshape(n, QCODE), qtab))
    nq_all[a:b] = nq.to(torch.int32)
sizes = QCODE + 32 * nq_all.to(torch.int64)
sizes = torch.where(nq_all > 0, sizes, torch.zeros_like(sizes))
qoff = torch.cumsum(sizes, 0) - sizes
total = int(sizes.sum().item())
qrec = torch.zeros(total + 8192, dtype=torch.uint8, device=dev)
for a, b, nq, qc, qtab in pieces:
    n = b - a
    o = qoff[a:b]
    live = nq > 0
    idx = (o[:, None] + torch.arange(QCODE, device=dev)[None, :])[live]
    qrec[idx.reshape(-1)] = qc[live].reshape(-1)
    tb = qtab.reshape(n, 256 * 4).contiguous().view(torch.uint8).reshape(n, 256 * 32)
    ar = torch.arange(256 * 32, device=dev)
    m = live[:, None] & (ar[None, :] < (32 * nq.to(torch.int64))[:, None])
    idx = (o[:, None] + QCODE + ar[None, :])[m]
    qrec[idx] = tb[m]
del pieces
quad_rows = nq_all > 0
wts = quad_rows.to(torch.float64)
print("one MI355X; %d rows x %d haplogroups: %d byte-coded rows, of which %d (%.1f %%) have <= 256 distinct quads "
      "(mean %.1f, median %d); byte records %.3f GB, quad records %.3f GB"
      % (rows, H, int(byte_rows.sum().item()), int(quad_rows.sum().item()), 100.0 * quad_rows.sum().item() / max(1, byte_rows.sum().item()),
         nq_all[quad_rows].double().mean().item(), int(nq_all[quad_rows].median().item()),
         float(((ldc + 8 * nd.to(torch.int64))[quad_rows]).sum().item()) / 1e9, total / 1e9))

plan = em.EmPlan(None, wts, n_runs=1, records=cm)
numpy.random.seed(7)
props = torch.from_numpy(em.init_props(H, 1.0)[None, :]).to(dev)
colsum = torch.zeros((1, H), dtype=torch.float64, device=dev)
plan.em_iter(props, props.log(), em.new_state(1, dev), colsum)
torch.cuda.synchronize()
want = colsum[0].cpu().numpy()
n_cu = torch.cuda.get_device_properties(0).multi_processor_count
ldpart = (H + 7) & ~7
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
state = em.new_state(1, dev)
for rep in range(2):
    ev0.record()
    for _ in range(20):
        plan.em_iter(props, props.log(), state, colsum)
    ev1.record(); torch.cuda.synchronize()
    print("product   mxm_em_iter_coded (row pass + column reduce), the same rows                          %7.3f ms per iteration"
          % (ev0.elapsed_time(ev1) / 20))
partial = torch.zeros((n_cu * 3, ldpart), dtype=torch.float64, device=dev)
names = {0: "quad records, table through registers, 2 rows in flight, 2 workgroups per CU",
         1: "... 3 rows in flight",
         2: "quad records, table global -> LDS directly (compiler's waits: vmcnt(0) per step), 3 rows in flight",
         3: "... 5 rows in flight",
         4: "... 3 rows in flight, 3 workgroups per CU (spills)",
         5: "quad records, table through registers, 3 rows in flight, ONE wait for the row's lookups",
         6: "... 2 rows in flight, one wait",
         7: "quad records, table through registers, 4 rows in flight",
         8: "... 5 rows in flight",
         9: "... 3 rows in flight, the wave's sum on the matrix core (2 v_mfma_f64_16x16x4 + 3 adds instead of the DPP ladder)",
         10: "... 2 rows in flight, the wave's sum on the matrix core"}
for rep in range(2):
    for v in sorted(names):
        g = I(0)
        partial.zero_()
        torch.cuda.synchronize()
        ms = x.quad_time(v, qrec.data_ptr(), qoff.data_ptr(), nq_all.data_ptr(), wts.data_ptr(), props.data_ptr(), rows, H,
                         partial.data_ptr(), ldpart, n_cu, 20, ctypes.byref(g))
        if ms < 0:
            print("variant %d failed: %g" % (v, ms)); continue
        got = partial[: g.value].sum(dim=0)[:H].cpu().numpy()
        rel = numpy.abs(got - want).max() / numpy.abs(want).max()
        print("quad %d %-100s %7.3f ms   column sums within %.1e" % (v, names[v], ms, rel))
