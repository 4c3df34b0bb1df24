import sys, numpy
sys.path.insert(0, '.')
import torch
from mixemt_amd import phylotree, preprocess, synth
from oracle import c_oracle
refseq = phylotree.load_rsrs(); phy = phylotree.load_build17(refseq)
haps = sorted(phy.hap_var)
t = preprocess.HapVarTables.build(refseq, phy, haps)
m = t.markers()
for n_rows in (1536, 1537, 3000, 5000):
    rp, si, ob, _ = synth.synth_reads(t, len(refseq), n_rows, seed=7)
    got = preprocess.build_em_matrix_device(t, rp, si, ob, kernel="sparse").cpu().numpy()
    want = c_oracle.build_em_matrix(t.expected, t.lhit, t.lmiss, rp, si, ob, len(haps))
    bad = numpy.flatnonzero((got != want).any(axis=1))
    print(n_rows, "rows: mismatching rows", len(bad), "fallback", preprocess.build_em_matrix_device.last_fallback)
    lens = numpy.diff(m["mk_ptr"])
    for r in bad[:12]:
        cols = numpy.flatnonzero(got[r] != want[r])
        s = si[rp[r]:rp[r+1]].astype(int)
        print("  row", r, "n", len(s), "light", int(lens[s].sum()), "heavy", int((m["heavy_id"][s] >= 0).sum()),
              "bad cols", len(cols), cols[:6], "got", got[r, cols[:3]], "want", want[r, cols[:3]],
              "distinct want", len(numpy.unique(want[r])), "r%grid", r % 1536, "r//grid", r // 1536)
