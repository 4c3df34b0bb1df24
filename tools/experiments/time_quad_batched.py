#!/usr/bin/env python
"""
Round-5 experiment (sized for round 6): B restarts through ONE pass over the quad records (quadB_kernel in
quad_experiment.hip) -- BASELINE config 3 runs ten restarts on one matrix, and over records every restart reads them anew
(744 restart-iterations/s against the dense matrix's 552, profiles/r05/bench_1m_coded_10restarts.json).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -I include -I mixemt_amd/csrc \
          tools/experiments/quad_experiment.hip -o tools/experiments/_build/libquad.so
    python tools/experiments/time_quad_batched.py [rows] [lib]

The quad records are the product's own (EmPlan.attach_quads); rows without them get weight 0 in both kernels, so restart b's
column sums are compared with the product kernel's on the same rows with the same proportions.
"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy
import torch
from mixemt_amd import em, phylotree, preprocess, synth

em.QUADS = False
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
path = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "tools", "experiments", "_build", "libquad.so")
x = ctypes.CDLL(path)
P, I, L = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
x.quadB_time.restype = ctypes.c_float
x.quadB_time.argtypes = [I, I, P, P, P, P, P, L, L, I, P, L, I, I, ctypes.POINTER(I)]
refseq = phylotree.load_rsrs(); phy = phylotree.load_build17(refseq); haps = sorted(phy.hap_var)
tables = preprocess.HapVarTables.build(refseq, phy, haps)
H = len(haps)
row_ptr, site, obs, _ = synth.synth_rows(tables, len(refseq), 0, rows, seed=1)
cm = preprocess.build_em_records_device(tables, row_ptr, site, obs)
dev = cm.rec.device
probe = em.EmPlan(None, torch.ones(rows, dtype=torch.float64, device=dev), n_runs=1, records=cm)
assert probe.attach_quads(True)
qrec, qoff, nquad, quad_rows, _ = probe._quad_keep
wts = (nquad > 0).to(torch.float64)
plan = em.EmPlan(None, wts, n_runs=1, records=cm)
assert plan.attach_quads(True)
print("one MI355X; %d rows x %d haplogroups, %d rows with quad records (%.3f GB)" % (rows, H, int(wts.sum().item()), probe.quad_bytes / 1e9))
BMAX = 4
numpy.random.seed(7)
props = torch.from_numpy(numpy.stack([em.init_props(H, 1.0) for _ in range(BMAX)])).to(dev)
want = []
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for b in range(BMAX):
    colsum = torch.zeros((1, H), dtype=torch.float64, device=dev)
    pb = props[b:b + 1].contiguous()
    plan.em_iter(pb, pb.log(), em.new_state(1, dev), colsum)
    torch.cuda.synchronize()
    want.append(colsum[0].cpu().numpy())
state = em.new_state(1, dev)
pb = props[0:1].contiguous()
for rep in range(2):
    ev0.record()
    for _ in range(20):
        plan.em_iter(pb, pb.log(), state, colsum)
    ev1.record(); torch.cuda.synchronize()
    print("product   em_iter_quad_coded_kernel + column reduce, one restart                       %7.3f ms per restart-iteration"
          % (ev0.elapsed_time(ev1) / 20))
n_cu = torch.cuda.get_device_properties(0).multi_processor_count
ldpart = (H + 7) & ~7
partial = torch.zeros((BMAX * n_cu, ldpart), dtype=torch.float64, device=dev)
for rep in range(2):
    for B, nbuf in ((1, 3), (2, 3), (2, 4), (3, 3), (3, 4), (4, 3), (4, 4)):
        g = I(0)
        partial.zero_()
        torch.cuda.synchronize()
        ms = x.quadB_time(B, nbuf, qrec.data_ptr(), qoff.data_ptr(), nquad.data_ptr(), wts.data_ptr(), props.data_ptr(), H, rows, H,
                          partial.data_ptr(), ldpart, n_cu, 10, ctypes.byref(g))
        if ms < 0:
            print("B=%d nbuf=%d failed: %g" % (B, nbuf, ms)); continue
        rel = 0.0
        for b in range(B):
            got = partial[b * g.value:(b + 1) * g.value].sum(dim=0)[:H].cpu().numpy()
            rel = max(rel, numpy.abs(got - want[b]).max() / numpy.abs(want[b]).max())
        print("batched   %d restart(s) per pass, 512 threads per row, %d rows in flight, 1 workgroup per CU   %7.3f ms per pass = %7.3f per restart-iteration   column sums within %.1e"
              % (B, nbuf - 1, ms, ms / B, rel))
