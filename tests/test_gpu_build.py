"""
GPU parity of build_em_matrix (mxm_build_em_matrix through the drop-in wrapper)
against the oracle and the reference-derived golden matrices.  Bit-exact: the
kernel adds the same fp64 terms in the same order as preprocess.py:86-96.
"""
import hashlib

import numpy
import pytest

from conftest import em_args, golden
from oracle import build_oracle, c_oracle

pytestmark = pytest.mark.gpu


def _sha(arr):
    return hashlib.sha256(numpy.ascontiguousarray(arr).tobytes()).hexdigest()


def test_toy_matrices_bitwise(toy):
    from mixemt_amd import preprocess
    ref, phy, haps = toy
    g = golden("g1_toy")
    for key, mkey in (("reads", "mat"), ("reads_b", "mat_b")):
        reads = str(g[key]).split("\n")
        mat = preprocess.build_em_matrix(ref, phy, reads, haps, em_args())
        assert mat.dtype == numpy.float64 and mat.flags["C_CONTIGUOUS"]
        assert mat.shape == (len(reads), len(haps))
        assert numpy.array_equal(mat, g[mkey])


def test_reference_hand_computed_values(toy):
    """preprocess_test.py:268-284."""
    from mixemt_amd import preprocess
    ref, phy, haps = toy
    reads = ["1:A,2:C", "1:T,2:C", "3:T,4:T", "2:A,4:T"]
    r1 = [(0.01 / 3) * (0.01 / 3)] + [0.99 * (0.01 / 3)] * 8
    r2 = [0.99 * (0.01 / 3)] + [(0.01 / 3) * (0.01 / 3)] * 8
    r3 = ([0.98 * (0.02 / 3)] + [(0.02 / 3) * 0.98] + [(0.02 / 3) * (0.02 / 3)] + [(0.02 / 3) * 0.98]
          + [0.98 * 0.98] + [(0.02 / 3) * 0.98] * 3 + [(0.02 / 3) * (0.02 / 3)])
    r4 = ([0.99 * (0.02 / 3)] + [(0.01 / 3) * 0.98] + [(0.01 / 3) * (0.02 / 3)]
          + [(0.01 / 3) * 0.98] * 5 + [0.99 * (0.02 / 3)])
    mat = preprocess.build_em_matrix(ref, phy, reads, haps, em_args())
    assert numpy.allclose(mat, numpy.log(numpy.array([r1, r2, r3, r4])))


def test_build17_golden_bitwise(b17):
    from mixemt_amd import preprocess
    refseq, phy, haps, tables = b17
    g = golden("g2_build_b17")
    mat = preprocess.build_em_matrix_device(tables, g["row_ptr"], g["site"], g["obs"]).cpu().numpy()
    assert mat.shape == (1032, 5408)
    assert numpy.array_equal(mat[:32], g["mat32"])
    assert _sha(mat) == str(g["mat_sha256"])
    assert numpy.array_equal(mat.sum(axis=1), g["row_sum"])
    assert numpy.array_equal(mat.argmax(axis=1), g["row_argmax"])


def test_drop_in_signature_strings(b17):
    """The reference's own call shape: refseq, phylo, signature strings, hap names, args."""
    from mixemt_amd import preprocess, synth
    refseq, phy, haps, tables = b17
    g = golden("g2_build_b17")
    sigs = synth.signatures(tables, g["row_ptr"][:9], g["site"], g["obs"])
    mat = preprocess.build_em_matrix(refseq, phy, sigs, haps, em_args())
    assert numpy.array_equal(mat, g["mat32"][:8])


@pytest.mark.parametrize("n_rows,seed", [(1, 5), (257, 6), (5000, 7)])
def test_against_c_oracle_bitwise(b17, n_rows, seed):
    from mixemt_amd import preprocess, synth
    refseq, phy, haps, tables = b17
    row_ptr, site, obs, _ = synth.synth_reads(tables, len(refseq), n_rows, seed=seed)
    got = preprocess.build_em_matrix_device(tables, row_ptr, site, obs).cpu().numpy()
    want = c_oracle.build_em_matrix(tables.expected, tables.lhit, tables.lmiss, row_ptr, site, obs,
                                    len(haps))
    assert numpy.array_equal(got, want)


@pytest.mark.parametrize("n_cols", [1, 3, 5, 100, 1023, 1025])
def test_ragged_column_counts(b17, n_cols):
    """H not a multiple of 4 / of the 1024-column tile; H = #contributors sized subsets."""
    from mixemt_amd import preprocess, synth
    refseq, phy, haps, tables = b17
    sub = haps[:n_cols]
    sub_tables = preprocess.HapVarTables.build(refseq, phy, sub)
    row_ptr, site, obs, _ = synth.synth_reads(tables, len(refseq), 40, seed=8)
    got = preprocess.build_em_matrix_device(sub_tables, row_ptr, site, obs).cpu().numpy()
    want = c_oracle.build_em_matrix(sub_tables.expected, sub_tables.lhit, sub_tables.lmiss, row_ptr,
                                    site, obs, n_cols)
    assert got.shape == (40, n_cols) and numpy.array_equal(got, want)


def test_long_reads_more_observations_than_one_lds_pass(b17):
    """A 3 kb fragment covers > 512 variant sites: the kernel stages them in several passes."""
    from mixemt_amd import preprocess, synth
    refseq, phy, haps, tables = b17
    row_ptr, site, obs, _ = synth.synth_reads(tables, len(refseq), 6, seed=9, read_len=3000)
    assert numpy.diff(row_ptr).max() > 512
    got = preprocess.build_em_matrix_device(tables, row_ptr, site, obs).cpu().numpy()
    want = c_oracle.build_em_matrix(tables.expected, tables.lhit, tables.lmiss, row_ptr, site, obs,
                                    len(haps))
    assert numpy.array_equal(got, want)


def test_unusual_observations_and_errors(b17):
    from mixemt_amd import preprocess
    refseq, phy, haps, tables = b17
    s0, s1 = int(tables.sites[0]), int(tables.sites[1])
    reads = ["%d:N,%d:a" % (s0, s1), "%d:%s" % (s0, refseq[s0])]
    got = preprocess.build_em_matrix(refseq, phy, reads, haps, em_args())
    want = build_oracle.build_em_matrix_np(refseq, phy, reads, haps)
    assert numpy.array_equal(got, want)
    with pytest.raises(ValueError):
        preprocess.build_em_matrix(refseq, phy, [""], haps, em_args())
    with pytest.raises(KeyError):
        preprocess.build_em_matrix(refseq, phy, ["0:A"], haps, em_args())
    empty = preprocess.build_em_matrix(refseq, phy, [], haps, em_args())
    assert empty.shape == (0, len(haps))


def test_reference_phylotree_object_is_accepted(toy):
    """Duck-typing: any object with .variants and .hap_var works (the drop-in claim)."""
    from mixemt_amd import preprocess

    class Bare(object):
        pass
    ref, phy, haps = toy
    bare = Bare()
    bare.variants = {p: dict(c) for p, c in phy.variants.items()}
    bare.hap_var = dict(phy.hap_var)
    g = golden("g1_toy")
    reads = str(g["reads"]).split("\n")
    assert numpy.array_equal(preprocess.build_em_matrix(ref, bare, reads, haps, em_args()), g["mat"])


@pytest.mark.parametrize("kernel", ["bytes", "lut", "sparse"])
def test_all_kernels_give_reference_bits(b17, kernel):
    """The marker kernel, the lookup-table kernel and the byte-table kernel are interchangeable."""
    from mixemt_amd import preprocess
    refseq, phy, haps, tables = b17
    g = golden("g2_build_b17")
    mat = preprocess.build_em_matrix_device(tables, g["row_ptr"], g["site"], g["obs"],
                                            kernel=kernel).cpu().numpy()
    assert numpy.array_equal(mat[:32], g["mat32"])
    assert _sha(mat) == str(g["mat_sha256"])


def test_tables_that_do_not_qualify_fall_back(b17, toy):
    from mixemt_amd import preprocess
    ref, phy, haps = toy
    tables = preprocess.HapVarTables.build(ref, phy, haps)
    tables._lut = False                  # as if the alphabet check had failed: "auto" must take the byte-table kernel
    g = golden("g1_toy")
    reads = str(g["reads"]).split("\n")
    rp, si, ob = preprocess.encode_signatures(reads, tables)
    got = preprocess.build_em_matrix_device(tables, rp, si, ob).cpu().numpy()
    assert numpy.array_equal(got, g["mat"])
    with pytest.raises(ValueError):
        preprocess.build_em_matrix_device(tables, rp, si, ob, kernel="lut")


def test_prob_for_vars_closed_forms(toy):
    """preprocess_test.py:76-95: sums of logs against closed-form products, custom mutation weights."""
    import math
    import torch
    from mixemt_amd import preprocess
    ref, phy, haps = toy
    obs_i = ",".join("%d:%s" % (p, b) for p, b in zip(range(9), "GAAAAAAAA"))
    for mut_max, want in ((0.10, {"I": 0.9 ** 9, "C": (0.9 ** 7) * ((0.1 / 3) ** 2),
                                  "D": (0.9 ** 5) * ((0.1 / 3) ** 4)}),
                          (0.50, {"I": (0.9 ** 7) * (0.8 ** 2),
                                  "C": (0.9 ** 5) * (0.8 ** 2) * ((0.1 / 3) ** 2)})):
        tables = preprocess.HapVarTables.build(ref, phy, haps, mut_wt=0.10, mut_max=mut_max)
        rp, si, ob = preprocess.encode_signatures([obs_i], tables)
        row = preprocess.build_em_matrix_device(tables, rp, si, ob).cpu().numpy()[0]
        for hap, prob in want.items():
            assert abs(row[haps.index(hap)] - math.log(prob)) < 1e-7          # assertAlmostEqual (7 places)


@pytest.mark.parametrize("n_cols", [1, 3, 4, 5, 255, 1023, 1024, 1025, 2050])
def test_lut_kernel_ragged_widths_rows_and_order(b17, n_cols):
    """Widths around the 4-column lane and the 1024-column tile; a handful to a few thousand rows;
    position-sorted row order on and off: always the oracle's bits."""
    from mixemt_amd import preprocess, synth
    refseq, phy, haps, tables = b17
    sub = haps[50:50 + n_cols]
    sub_tables = preprocess.HapVarTables.build(refseq, phy, sub)
    assert sub_tables.lut() is not None
    for n_rows, seed in ((1, 1), (7, 2), (300, 3), (2100, 4)):
        row_ptr, site, obs, _ = synth.synth_reads(tables, len(refseq), n_rows, seed=seed)
        want = c_oracle.build_em_matrix(sub_tables.expected, sub_tables.lhit, sub_tables.lmiss, row_ptr,
                                        site, obs, n_cols)
        for sort_rows in (False, True):
            got = preprocess.build_em_matrix_device(sub_tables, row_ptr, site, obs, kernel="lut",
                                                    sort_rows=sort_rows).cpu().numpy()
            assert numpy.array_equal(got, want), (n_cols, n_rows, sort_rows)


def test_lut_kernel_long_reads_unusual_bases_and_strided_output(b17):
    import torch
    from mixemt_amd import preprocess, synth
    refseq, phy, haps, tables = b17
    row_ptr, site, obs, _ = synth.synth_reads(tables, len(refseq), 5, seed=9, read_len=3000)
    assert numpy.diff(row_ptr).max() > 256           # several staging passes per row
    obs = obs.copy()
    obs[::17] = ord("N")                 # never matches an expected base
    obs[5::29] = ord("a")                # lower case is a different string in the reference
    obs[3::31] = 0                       # multi-character observation (encoded as 0)
    want = c_oracle.build_em_matrix(tables.expected, tables.lhit, tables.lmiss, row_ptr, site, obs,
                                    len(haps))
    got = preprocess.build_em_matrix_device(tables, row_ptr, site, obs, kernel="lut").cpu().numpy()
    assert numpy.array_equal(got, want)
    out = torch.full((5, len(haps) + 3), -1.0, dtype=torch.float64, device="cuda")     # odd leading dimension
    preprocess.build_em_matrix_device(tables, row_ptr, site, obs, out=out, kernel="lut")
    host = out.cpu().numpy()
    assert numpy.array_equal(host[:, :len(haps)], want) and (host[:, len(haps):] == -1.0).all()


@pytest.mark.parametrize("n_cols", [1, 2, 3, 255, 1024, 1025, 2050, 5408])
def test_sparse_kernel_ragged_widths_and_rows(b17, n_cols):
    """The marker kernel (one in-order sum per distinct cell value of a row): the oracle's bits at every width,
    from one row to a few thousand, rows longer than its masks (128 observations) included (they take the
    lookup-table kernel through the fallback list)."""
    from mixemt_amd import preprocess, synth
    refseq, phy, haps, tables = b17
    sub = haps[50:50 + n_cols] if n_cols < len(haps) else haps
    sub_tables = tables if n_cols == len(haps) else preprocess.HapVarTables.build(refseq, phy, sub)
    for n_rows, seed, read_len in ((1, 1, 150), (7, 2, 150), (300, 3, 150), (2100, 4, 150), (40, 5, 400)):
        row_ptr, site, obs, _ = synth.synth_reads(tables, len(refseq), n_rows, seed=seed, read_len=read_len)
        want = c_oracle.build_em_matrix(sub_tables.expected, sub_tables.lhit, sub_tables.lmiss, row_ptr,
                                        site, obs, n_cols)
        got = preprocess.build_em_matrix_device(sub_tables, row_ptr, site, obs, kernel="sparse").cpu().numpy()
        assert numpy.array_equal(got, want), (n_cols, n_rows)
        left = preprocess.build_em_matrix_device.last_fallback
        long_rows = int((numpy.diff(row_ptr) > 128).sum())                # (round 6: rows of 65 .. 128 observations stay)
        assert long_rows <= left <= long_rows + max(2, n_rows // 50)      # + the odd row with more than 352 / 704 distinct masks
        if read_len == 400:
            assert 0 < left < n_rows                      # both paths in one call


def test_sparse_kernel_unusual_bases_strided_output_and_golden(b17):
    import hashlib
    import torch
    from mixemt_amd import preprocess, synth
    refseq, phy, haps, tables = b17
    row_ptr, site, obs, _ = synth.synth_reads(tables, len(refseq), 600, seed=9)
    obs = obs.copy()
    obs[::17] = ord("N")                 # never matches an expected base
    obs[5::29] = ord("a")                # lower case is a different string in the reference
    obs[3::31] = 0                       # multi-character observation (encoded as 0)
    want = c_oracle.build_em_matrix(tables.expected, tables.lhit, tables.lmiss, row_ptr, site, obs,
                                    len(haps))
    out = torch.full((600, len(haps) + 3), -1.0, dtype=torch.float64, device="cuda")     # odd leading dimension
    preprocess.build_em_matrix_device(tables, row_ptr, site, obs, out=out, kernel="sparse")
    host = out.cpu().numpy()
    assert numpy.array_equal(host[:, :len(haps)], want) and (host[:, len(haps):] == -1.0).all()
    g = golden("g9_run_em_2400")         # the reference's own 2400 x 5408 matrix
    mat = preprocess.build_em_matrix_device(tables, g["row_ptr"], g["site"], g["obs"], kernel="sparse")
    assert hashlib.sha256(mat.cpu().numpy().tobytes()).hexdigest() == str(g["mat_sha256"])


@pytest.mark.parametrize("n_cols", [3000, 5408])
def test_sparse_kernel_full_table_fallback(b17, n_cols):
    """Rows with more distinct values than the dedup table is allowed to hold (limit lowered to 6 here, so a
    third of ordinary rows qualify) go through the fallback list and come out with the same bits."""
    from mixemt_amd import _lib, preprocess, synth
    refseq, phy, haps, tables = b17
    lib = _lib.load()
    sub_tables = preprocess.HapVarTables.build(refseq, phy, haps[:n_cols]) if n_cols < len(haps) else tables
    row_ptr, site, obs, _ = synth.synth_reads(tables, len(refseq), 700, seed=31)
    want = c_oracle.build_em_matrix(sub_tables.expected, sub_tables.lhit, sub_tables.lmiss, row_ptr, site, obs, n_cols)
    got = preprocess.build_em_matrix_device(sub_tables, row_ptr, site, obs, kernel="sparse").cpu().numpy()
    assert numpy.array_equal(got, want)
    plain = preprocess.build_em_matrix_device.last_fallback
    lib.mxm_set_sparse_max_distinct(6)                   # (conftest resets every knob after the test)
    got = preprocess.build_em_matrix_device(sub_tables, row_ptr, site, obs, kernel="sparse").cpu().numpy()
    assert numpy.array_equal(got, want)
    assert plain + 50 < preprocess.build_em_matrix_device.last_fallback < 700      # both paths, many rows each


def test_dense_tables_expanded_on_the_device_equal_the_hosts(b17):
    """mxm_expand_tables: E and Ecode from the marker form (0.4 MB) on the device, bit for bit the host's 22 MB tables --
    Build 17, a sub-tree with an odd width, and a table whose majority base is NOT the reference base at some sites."""
    import torch
    from mixemt_amd import phylotree, preprocess
    refseq, phy, haps, tables = b17
    cases = [tables, preprocess.HapVarTables.build(refseq, phy, haps[100:1137])]
    few = [h for h in haps if h.startswith("L0")][:40] or haps[:40]          # a clade: many sites where most carry the marker
    cases.append(preprocess.HapVarTables.build(refseq, phy, few))
    for t in cases:
        exp_d, lhit_d, lmiss_d = t.device()
        assert numpy.array_equal(exp_d.cpu().numpy(), t.expected)
        assert numpy.array_equal(lhit_d.cpu().numpy(), t.lhit) and numpy.array_equal(lmiss_d.cpu().numpy(), t.lmiss)
        lut = t.lut()
        if lut is not None:
            assert numpy.array_equal(t.lut_device()["ecode"].cpu().numpy(), lut["ecode"])
