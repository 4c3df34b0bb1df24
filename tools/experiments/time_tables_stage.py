"""Round 6: where the pipeline's "tables" stage (12-16 ms) goes -- host encodes and device uploads timed apart.
python tools/experiments/time_tables_stage.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy, torch
from mixemt_amd import _lib, phylotree, preprocess
refseq = phylotree.load_rsrs(); phy = phylotree.load_build17(refseq); haps = sorted(phy.hap_var)
torch.zeros(1, device="cuda"); _lib.load(); torch.cuda.synchronize()
for rep in range(3):
    tables = preprocess.HapVarTables.build(refseq, phy, haps)
    laps = []
    t = time.perf_counter()
    def lap(name):
        global t
        torch.cuda.synchronize()
        now = time.perf_counter(); laps.append("%s %.2f" % (name, (now - t) * 1e3)); t = now
    tables.sparse(); lap("sparse()")
    tables.lut(); lap("lut()")
    tables._vectors_device(); lap("lhit/lmiss up")
    tables.sparse_device(); lap("sparse_device")
    tables.lut_device(); lap("lut_device")
    print("; ".join(laps))
