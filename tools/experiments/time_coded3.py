#!/usr/bin/env python
"""
Round-5 experiments on the records row pass, see coded3_experiment.hip:
bare readers of the records (what limits the loads?) and the row pass with the accumulation one row behind.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -I include -I mixemt_amd/csrc \
          tools/experiments/coded3_experiment.hip -o tools/experiments/_build/libcoded3.so
    python tools/experiments/time_coded3.py [rows] [lib]

Column sums of every row-pass variant are checked against the product kernel (mxm_em_iter_coded with the wide rows'
weights set to 0: the experiment's kernels skip those rows); times are HIP events over back-to-back launches.
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
path = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "tools", "experiments", "_build", "libcoded3.so")
x = ctypes.CDLL(path)
P, I, L = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
x.coded3_reader.restype = ctypes.c_float
x.coded3_reader.argtypes = [I, P, P, P, I, L, P, P, I, I]
x.coded3_time.restype = ctypes.c_float
x.coded3_time.argtypes = [I, P, P, P, I, P, P, L, I, P, L, I, I, ctypes.POINTER(I)]
refseq = phylotree.load_rsrs(); phy = phylotree.load_build17(refseq); haps = sorted(phy.hap_var)
tables = preprocess.HapVarTables.build(refseq, phy, haps)
H = len(haps)
row_ptr, site, obs, _ = synth.synth_rows(tables, len(refseq), 0, rows, seed=1)
cm = preprocess.build_em_records_device(tables, row_ptr, site, obs)
dev = cm.rec.device
nd = cm.ndist
byte_rows = (nd > 0) & (nd <= 256)
wts = byte_rows.to(torch.float64)
plan = em.EmPlan(None, wts, n_runs=1, records=cm)
numpy.random.seed(7)
props = torch.from_numpy(em.init_props(H, 1.0)[None, :]).to(dev)
colsum = torch.zeros((1, H), dtype=torch.float64, device=dev)
plan.em_iter(props, props.log(), em.new_state(1, dev), colsum)
torch.cuda.synchronize()
want = colsum[0].cpu().numpy()
n_cu = torch.cuda.get_device_properties(0).multi_processor_count
ldc = (H + 7) & ~7
ldpart = (H + 7) & ~7
ldc_bytes = ldc
bytes_read = float((ldc_bytes + 8 * nd[byte_rows].to(torch.int64)).sum().item())
order = torch.argsort(torch.where(byte_rows, cm.rec_off, torch.full_like(cm.rec_off, 1 << 62)), stable=True)
sink = torch.zeros(4, dtype=torch.int32, device=dev)
print("one MI355X; %d rows x %d haplogroups as records (%.2f GB in use), %d byte-coded rows = %.3f GB of codes + P tables"
      % (rows, H, cm.used / 1e9, int(byte_rows.sum().item()), bytes_read / 1e9))
readers = {0: "4 B/lane x 6, 2 rows in flight, nt, 2 workgroups per CU (the product kernel's loads)",
           1: "4 B/lane x 6, 5 rows in flight",
           2: "16 B/lane, 2 rows in flight",
           3: "16 B/lane, 5 rows in flight",
           4: "4 B/lane x 6, 2 rows in flight, default cache policy",
           5: "4 B/lane x 6, 2 rows in flight, records taken in ADDRESS order",
           6: "16 B/lane, 5 rows in flight, address order",
           7: "16 B/lane, 5 rows in flight, 4 workgroups per CU",
           8: "16 B/lane, 5 rows in flight, default cache policy"}
for rep in range(2):
    for v in sorted(readers):
        ms = x.coded3_reader(v, cm.rec.data_ptr(), cm.rec_off.data_ptr(), cm.ndist.data_ptr(), ldc, rows, order.data_ptr(),
                             sink.data_ptr(), n_cu, 20)
        print("reader %d  %-85s %7.3f ms = %5.2f TB/s" % (v, readers[v], ms, bytes_read / ms / 1e9) if ms > 0 else "reader %d failed %g" % (v, ms))
# the product kernel itself, same weights (byte-coded rows only), for the same-process comparison
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
state = em.new_state(1, dev)
for rep in range(2):
    ev0.record()
    for _ in range(20):
        plan.em_iter(props, props.log(), state, colsum)
    ev1.record(); torch.cuda.synchronize()
    print("product   mxm_em_iter_coded (row pass + column reduce, wide rows at weight 0)                              %7.3f ms per iteration"
          % (ev0.elapsed_time(ev1) / 20))
partial = torch.zeros((n_cu * 2, ldpart), dtype=torch.float64, device=dev)
names = {0: "accumulation one row behind, proportions in LDS, 3 rows in flight, 2 workgroups per CU",
         1: "... 5 rows in flight (48 B scratch)",
         2: "... proportions in registers, 3 rows in flight, 2 workgroups per CU",
         3: "... proportions in registers, 1 workgroup per CU",
         4: "TWO rows per step (one barrier per pair, two independent chains), 2 pairs in flight, 2 workgroups per CU",
         5: "... 3 pairs in flight"}
for rep in range(2):
    for v in sorted(names):
        g = I(0)
        partial.zero_()
        torch.cuda.synchronize()
        ms = x.coded3_time(v, cm.rec.data_ptr(), cm.rec_off.data_ptr(), cm.ndist.data_ptr(), ldc, wts.data_ptr(), props.data_ptr(),
                           rows, H, partial.data_ptr(), ldpart, n_cu, 20, ctypes.byref(g))
        if ms < 0:
            print("variant %d failed: %g" % (v, ms)); continue
        got = partial[: g.value].sum(dim=0)[:H].cpu().numpy()
        rel = numpy.abs(got - want).max() / numpy.abs(want).max()
        print("delayed %d %-92s %7.3f ms   column sums within %.1e" % (v, names[v], ms, rel))
