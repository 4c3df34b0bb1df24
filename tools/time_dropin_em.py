import sys, time, argparse
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy, torch
from mixemt_amd import em, preprocess, phylotree
refseq = phylotree.load_rsrs(); phy = phylotree.load_build17(refseq); haps = sorted(phy.hap_var)
tables = preprocess.HapVarTables.build(refseq, phy, haps)
for name, seed in (("g4_run_em", 7), ("g9_run_em_2400", 7)):
    g = numpy.load("/root/repo/tests/golden/%s.npz" % name)
    mat = preprocess.build_em_matrix_device(tables, g["row_ptr"], g["site"], g["obs"]).cpu().numpy()
    wts = g["wts"]
    args = argparse.Namespace(init_alpha=1.0, tolerance=1e-4, max_iter=10000, n_multi=1, verbose=False)
    for rep in range(3):
        numpy.random.seed(seed)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        res = em.run_em_ex(mat, wts, args)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print("%s %d x %d numpy in -> numpy out: %.1f ms wall, loop %.1f ms (%d iterations, %s), plan %.1f ms"
              % (name, mat.shape[0], mat.shape[1], dt * 1e3, res["loop_s"] * 1e3, sum(res["iters"]), res["storage"], res["plan_s"] * 1e3))
