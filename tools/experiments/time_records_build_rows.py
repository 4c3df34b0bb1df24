#!/usr/bin/env python
"""Stage times of build_em_records_device on synth-v1 rows: python tools/experiments/time_records_build_rows.py [rows]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["MXM_PIPELINE_TIMING"] = "1"
import numpy
import torch
from mixemt_amd import _lib, phylotree, preprocess, synth

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 10000000
refseq = phylotree.load_rsrs(); phy = phylotree.load_build17(refseq); haps = sorted(phy.hap_var)
tables = preprocess.HapVarTables.build(refseq, phy, haps)
row_ptr, site, obs, _ = synth.synth_rows(tables, len(refseq), 0, rows, seed=1)
torch.zeros(1, device="cuda"); _lib.load(); torch.cuda.synchronize()
for rep in range(2):
    t0 = time.perf_counter()
    cm = preprocess.build_em_records_device(tables, row_ptr, site, obs)
    torch.cuda.synchronize()
    print("call %d: %.1f ms; handed back %d; capacity %.1f GB, used %.1f GB; stages %s"
          % (rep, (time.perf_counter() - t0) * 1e3, preprocess.build_em_matrix_device.last_fallback, cm.capacity / 1e9, cm.used / 1e9,
             preprocess.build_em_records_device.last_timing), flush=True)
    del cm
