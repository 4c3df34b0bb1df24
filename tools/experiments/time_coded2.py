#!/usr/bin/env python
"""
Two restarts per pass over records, in one thread: timing experiment, see coded2_experiment.hip.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -I include -I mixemt_amd/csrc \
          tools/experiments/coded2_experiment.hip -o /tmp/libcoded2.so
    python tools/experiments/time_coded2.py [rows] [/tmp/libcoded2.so]

Column sums of every variant are checked against the product kernel (mxm_em_iter_coded with the wide rows' weights
set to 0: the experiment's kernels skip those rows); times are HIP events over `reps` back-to-back launches.
"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy
import torch
from mixemt_amd import _lib, em, phylotree, preprocess, synth

em.QUADS = False        # this tool measures the records' own pass (em_iter_coded_kernel): no quad dictionary beside them

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
path = sys.argv[2] if len(sys.argv) > 2 else "/tmp/libcoded2.so"
x = ctypes.CDLL(path)
x.coded2_time.restype = ctypes.c_float
x.coded2_time.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
                          ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int,
                          ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
refseq = phylotree.load_rsrs(); phy = phylotree.load_build17(refseq); haps = sorted(phy.hap_var)
tables = preprocess.HapVarTables.build(refseq, phy, haps)
H = len(haps)
row_ptr, site, obs, _ = synth.synth_rows(tables, len(refseq), 0, rows, seed=1)
cm = preprocess.build_em_records_device(tables, row_ptr, site, obs)
dev = cm.rec.device
nd = cm.ndist
wts = ((nd > 0) & (nd <= 256)).to(torch.float64)              # byte-coded rows only, for both sides
plan = em.EmPlan(None, wts, n_runs=2, records=cm)
numpy.random.seed(7)
props = torch.from_numpy(numpy.stack([em.init_props(H, 1.0) for _ in range(2)])).to(dev)
colsum = torch.zeros((2, H), dtype=torch.float64, device=dev)
state = em.new_state(2, dev) if hasattr(em, "new_state") else None
plan.em_iter(props, props.log(), state, colsum)
torch.cuda.synchronize()
want = colsum.cpu().numpy()
n_cu = torch.cuda.get_device_properties(0).multi_processor_count
ldc = (H + 7) & ~7
ldpart = (H + 7) & ~7
partial = torch.zeros((3 * n_cu * 2, ldpart), dtype=torch.float64, device=dev)
names = {0: "256 threads x 24 cells, 1 restart, 2 workgroups per CU (the product kernel's shape)",
         1: "512 threads x 12 cells, 2 restarts, 1 workgroup per CU",
         2: "256 threads x 24 cells, 2 restarts, 1 workgroup per CU (1 wave per SIMD, 256 VGPRs + 77 AGPRs)",
         3: "512 threads x 12 cells, 1 restart, 2 workgroups per CU",
         4: "256 threads x 24 cells, 1 restart, 3 workgroups per CU, values looked up twice",
         5: "256 threads x 24 cells, 1 restart, 2 workgroups per CU, values looked up twice",
         6: "as 4 with three rows in flight (NBUF 3)"}
print("one MI355X; %d rows x %d haplogroups as records (%.2f GB), %d of them byte-coded; average of 20 launches"
      % (rows, H, cm.used / 1e9, int(wts.sum().item())))
for rep in range(2):
    for variant in (0, 1, 2, 3, 4, 5, 6):
        g, n = ctypes.c_int(0), ctypes.c_int(0)
        partial.zero_()
        torch.cuda.synchronize()
        ms = x.coded2_time(variant, cm.rec.data_ptr(), cm.rec_off.data_ptr(), cm.ndist.data_ptr(), ldc, wts.data_ptr(),
                           props.data_ptr(), rows, H, partial.data_ptr(), ldpart, n_cu, 20, ctypes.byref(g), ctypes.byref(n))
        if ms < 0:
            print("variant %d failed: %g" % (variant, ms)); continue
        got = partial[: g.value * n.value].view(g.value, n.value, ldpart).sum(dim=0)[:, :H].cpu().numpy()
        rel = numpy.abs(got - want[: n.value]).max() / numpy.abs(want).max()
        print("variant %d  %-95s %7.3f ms per launch = %7.3f ms per restart-iteration   column sums within %.1e"
              % (variant, names[variant], ms, ms / n.value, rel))
