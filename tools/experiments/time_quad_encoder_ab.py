import ctypes, os, sys, time
sys.path.insert(0, ".")
import numpy, torch
from mixemt_amd import _lib, em, phylotree, preprocess, synth
from mixemt_amd._dev import current_stream
rows = 1000000
refseq = phylotree.load_rsrs(); phy = phylotree.load_build17(refseq); haps = sorted(phy.hap_var)
tables = preprocess.HapVarTables.build(refseq, phy, haps)
row_ptr, site, obs, _ = synth.synth_rows(tables, len(refseq), 0, rows, seed=1)
cm = preprocess.build_em_records_device(tables, row_ptr, site, obs)
coded = cm.struct()
for path in ("mixemt_amd/lib/libmixemt_hip.so", "build_ab/libmxm_quadfixed.so"):
    lib = ctypes.CDLL(os.path.abspath(path))
    for name, (restype, argtypes) in _lib.SIGNATURES.items():
        fn = getattr(lib, name); fn.restype, fn.argtypes = restype, argtypes
    cap = rows * (2048 + 32 * 256)
    qrec = torch.empty(cap, dtype=torch.uint8, device="cuda")
    qoff = torch.empty(rows, dtype=torch.int64, device="cuda"); nquad = torch.empty(rows, dtype=torch.int32, device="cuda"); stats = torch.empty(2, dtype=torch.int64, device="cuda")
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        lib.mxm_build_quads(ctypes.byref(coded), len(haps), qrec.data_ptr(), cap, qoff.data_ptr(), nquad.data_ptr(), stats.data_ptr(), current_stream())
        torch.cuda.synchronize(); print(os.path.basename(path), "%.2f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
