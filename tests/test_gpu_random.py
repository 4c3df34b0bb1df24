"""
Seeded differential tests: many small random problems through the C ABI against
the oracle -- random trees / alphabets / mutation counts for the matrix build,
random shapes, weights, -inf patterns and restart counts for the EM loop.
"""
import collections

import numpy
import pytest

from conftest import em_args
from oracle import build_oracle, em_oracle

pytestmark = pytest.mark.gpu


class RandomPhylo(object):
    """Duck-typed phylo (hap_var + variants) with an arbitrary alphabet, built directly."""

    def __init__(self, rng, n_sites, n_haps, ref_len, alphabet):
        sites = numpy.sort(rng.choice(ref_len, size=n_sites, replace=False))
        self.refseq = "".join(rng.choice(list(alphabet), size=ref_len))
        self.variants = collections.defaultdict(collections.Counter)
        self.hap_var = {}
        for pos in sites:
            for base in rng.choice(list(alphabet), size=rng.integers(1, 4)):
                self.variants[int(pos)][str(base)] += int(rng.integers(1, 40))
        for h in range(n_haps):
            carried = rng.choice(sites, size=min(n_sites, int(rng.integers(0, 12))), replace=False)
            # variant strings 'X<pos+1>Y'; some derive to the reference base again (no marker)
            self.hap_var["hap%03d" % h] = ["%s%d%s" % (self.refseq[int(p)], int(p) + 1,
                                                      str(rng.choice(list(alphabet))))
                                           for p in numpy.sort(carried)]

    def get_variant_pos(self):
        return sorted(self.variants.keys())


@pytest.mark.parametrize("seed", range(12))
def test_random_tables_build_bitwise(seed):
    from mixemt_amd import preprocess
    rng = numpy.random.default_rng(1000 + seed)
    alphabet = ["ACGT", "ACGTN", "ACGTNacgt", "AC"][seed % 4]
    n_sites = int(rng.integers(3, 120))
    n_haps = int(rng.integers(1, 700))
    phy = RandomPhylo(rng, n_sites, n_haps, 400, alphabet)
    haps = sorted(phy.hap_var)
    sites = phy.get_variant_pos()
    reads = []
    for _ in range(int(rng.integers(1, 60))):
        picked = numpy.sort(rng.choice(sites, size=int(rng.integers(1, min(n_sites, 40) + 1)), replace=False))
        reads.append(",".join("%d:%s" % (p, rng.choice(list(alphabet + "N"))) for p in picked))
    want = build_oracle.build_em_matrix_np(phy.refseq, phy, reads, haps)
    got = preprocess.build_em_matrix(phy.refseq, phy, reads, haps, em_args())
    assert numpy.array_equal(got, want)
    tables = preprocess.HapVarTables.build(phy.refseq, phy, haps)
    rp, si, ob = preprocess.encode_signatures(reads, tables)
    for kernel, ok in (("bytes", True), ("lut", tables.lut() is not None), ("sparse", tables.lut() is not None)):
        if ok:
            alt = preprocess.build_em_matrix_device(tables, rp, si, ob, kernel=kernel).cpu().numpy()
            assert numpy.array_equal(alt, want), kernel


@pytest.mark.parametrize("seed", range(16))
def test_random_em_runs_vs_oracle(seed):
    """run_em end to end (init draws, stop iteration, multi-run fold) on random matrices."""
    from mixemt_amd import em
    rng = numpy.random.default_rng(2000 + seed)
    n_rows = int(rng.integers(2, 400))
    n_haps = int(rng.choice([2, 3, 9, 64, 65, 66, 127, 512, 513, 1000, 2048, 2050]))
    n_multi = int(rng.choice([1, 1, 2, 4]))
    mat = rng.normal(-30.0, 12.0, size=(n_rows, n_haps))
    # plant a mixture so that the loop converges in tens of iterations
    truth = rng.choice(n_haps, size=min(3, n_haps), replace=False)
    for r in range(n_rows):
        mat[r, truth[r % len(truth)]] += 25.0
    if seed % 3 == 0:
        mat[rng.random(mat.shape) < 0.01] = -numpy.inf
    wts = rng.integers(1, 6, size=n_rows)
    args = em_args(n_multi=n_multi, max_iter=int(rng.choice([3, 40, 400])), tolerance=float(rng.choice([1e-4, 1e-6])))
    numpy.random.seed(seed)
    res = em.run_em_ex(mat, wts, args)
    trace = []
    numpy.random.seed(seed)
    props, mix = em_oracle.run_em(mat, wts, args, trace=trace)
    assert res["iters"] == [t["iters"] for t in trace]
    assert numpy.abs(res["props"] - props).max() < 1e-10
    got_mix = res["read_mix"].cpu().numpy()
    fin = numpy.isfinite(mix)
    assert numpy.array_equal(numpy.isfinite(got_mix), fin)
    assert numpy.allclose(got_mix[fin], mix[fin], rtol=0, atol=1e-8)


@pytest.mark.parametrize("seed", range(6))
def test_random_consumers_vs_numpy(seed):
    """argmax / votes / assignment kernels against their NumPy definitions (assemble.py:115-123, :267-334)."""
    from mixemt_amd import assign
    rng = numpy.random.default_rng(3000 + seed)
    n_rows, n_haps = int(rng.integers(1, 500)), int(rng.integers(2, 900))
    mix = rng.normal(-10, 5, size=(n_rows, n_haps))
    if seed % 2:
        mix[rng.random(mix.shape) < 0.05] = -numpy.inf
    wts = rng.integers(1, 9, size=n_rows)
    best, votes = assign.row_argmax_votes(mix, wts)
    assert numpy.array_equal(best, mix.argmax(axis=1))
    want_votes = numpy.zeros(n_haps)
    numpy.add.at(want_votes, mix.argmax(axis=1), wts)
    assert numpy.array_equal(votes, want_votes)
    n_con = int(rng.integers(2, min(6, n_haps) + 1))
    cols = sorted(int(c) for c in rng.choice(n_haps, size=n_con, replace=False))
    props = rng.dirichlet([1.0] * n_haps)
    haps = ["h%d" % i for i in range(n_haps)]
    contribs = [["hap%d" % (i + 1), haps[c], props[c]] for i, c in enumerate(cols)]
    table = assign.assign_read_indexes(contribs, (props, mix), haps, [[str(i)] for i in range(n_rows)], 2.0)
    log_props = numpy.log(props)
    for r in range(n_rows):
        vals = mix[r, cols] - log_props[cols]
        order = numpy.argsort(vals)[::-1]
        with numpy.errstate(invalid="ignore"):
            ok = vals[order[0]] - vals[order[1]] >= numpy.log(2.0)
        if not numpy.isfinite(vals[order[1]]) and not numpy.isfinite(vals[order[0]]):
            continue                                           # -inf - -inf: NaN in the reference too
        name = contribs[order[0]][0] if ok else "unassigned"
        assert r in table[name], (r, vals, name)


@pytest.mark.parametrize("n_rows,n_haps", [(1, 1), (600, 1), (5, 1), (1, 2), (1, 70), (2, 64), (3, 65), (1, 5408), (7, 8191)])
def test_degenerate_shapes(n_rows, n_haps):
    """Single rows, ONE column (the refinement EM of an unmixed sample, bin/mixemt:311-320), two columns, the switch-over
    width between the two loop kernels."""
    from mixemt_amd import em
    rng = numpy.random.default_rng(n_rows * 10007 + n_haps)
    mat = rng.normal(-20.0, 5.0, size=(n_rows, n_haps))
    wts = rng.integers(1, 4, size=n_rows)
    args = em_args(max_iter=25, n_multi=2)
    numpy.random.seed(4)
    res = em.run_em_ex(mat, wts, args)
    trace = []
    numpy.random.seed(4)
    props, mix = em_oracle.run_em(mat, wts, args, trace=trace)
    assert res["iters"] == [t["iters"] for t in trace]
    assert numpy.abs(res["props"] - props).max() < 1e-10
    assert numpy.allclose(res["read_mix"].cpu().numpy(), mix, rtol=0, atol=1e-8)
