"""
The C ABI from a plain C program (examples/c_abi_demo.c): HIP runtime for memory,
libmixemt_hip.so for the path, no Python objects in between.  Its output is
checked against the reference-derived goldens (toy tree, config 1).
"""
import os
import struct
import subprocess

import numpy
import pytest

from conftest import ROOT, golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def demo(tmp_path_factory):
    from mixemt_amd import build
    lib = build.build()
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    exe = str(tmp_path_factory.mktemp("cdemo") / "c_abi_demo")
    # plain C, plain gcc: the HIP runtime API header for hipMalloc/hipMemcpy, nothing else
    subprocess.run(["gcc", "-std=c99", "-O2", "-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(rocm, "include"),
                    "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "c_abi_demo.c"),
                    "-L" + os.path.dirname(lib), "-lmixemt_hip", "-L" + os.path.join(rocm, "lib"), "-lamdhip64",
                    "-lm", "-Wl,-rpath," + os.path.dirname(lib), "-Wl,-rpath," + os.path.join(rocm, "lib"),
                    "-o", exe], check=True)
    return exe


def _run(exe, tmp_path, tables, row_ptr, site, obs, wts, init, tol, max_iter, records=False):
    n_rows, n_haps, n_sites = len(row_ptr) - 1, tables.n_haps, len(tables.sites)
    prob, res = str(tmp_path / "problem.bin"), str(tmp_path / "result.bin")
    with open(prob, "wb") as f:
        f.write(struct.pack("<6q", n_rows, n_haps, n_sites, len(site), tables.expected.shape[1], max_iter))
        f.write(struct.pack("<d", tol))
        for arr, dt in ((tables.expected, numpy.uint8), (tables.lhit, numpy.float64), (tables.lmiss, numpy.float64),
                        (row_ptr, numpy.int64), (site, numpy.uint16), (obs, numpy.uint8),
                        (wts, numpy.float64), (init, numpy.float64)):
            f.write(numpy.ascontiguousarray(arr, dtype=dt).tobytes())
    proc = subprocess.run([exe, prob, res] + (["records"] if records else []), stdout=subprocess.PIPE,
                          stderr=subprocess.STDOUT, text=True)
    assert proc.returncode == 0, proc.stdout
    raw = open(res, "rb").read()
    iters, done = struct.unpack_from("<2q", raw, 0)
    off = 16
    props = numpy.frombuffer(raw, dtype=numpy.float64, count=n_haps, offset=off)
    off += 8 * n_haps
    mat = numpy.frombuffer(raw, dtype=numpy.float64, count=n_rows * n_haps, offset=off).reshape(n_rows, n_haps)
    off += 8 * n_rows * n_haps
    mix = numpy.frombuffer(raw, dtype=numpy.float64, count=n_rows * n_haps, offset=off).reshape(n_rows, n_haps)
    return iters, done, props, mat, mix


def test_toy_tree_from_c(demo, tmp_path, toy):
    """9 haplogroups: the narrow (log-space) loop, em_test.py:78-116's instance."""
    from mixemt_amd import preprocess
    ref, phy, haps = toy
    g = golden("g1_toy")
    tables = preprocess.HapVarTables.build(ref, phy, haps)
    row_ptr, site, obs = preprocess.encode_signatures(str(g["reads"]).split("\n"), tables)
    iters, done, props, mat, mix = _run(demo, tmp_path, tables, row_ptr, site, obs, numpy.ones(10),
                                        g["m1_s2_inits"][0], 1e-4, 1000)
    assert numpy.array_equal(mat, g["mat"])                       # build: bit-exact
    assert (iters, done) == (int(g["m1_s2_iters"][0]), 1)
    assert numpy.abs(props - g["m1_s2_props"]).max() < 1e-9
    assert numpy.allclose(numpy.exp(mix), numpy.exp(g["m1_s2_mix"]), rtol=0, atol=1e-9)


def test_config1_from_c(demo, tmp_path, b17):
    """BASELINE config 1 (1000 x 100): the streaming kernel."""
    from mixemt_amd import preprocess
    refseq, phy, haps, tables = b17
    g = golden("g7_config1")
    sub = preprocess.HapVarTables.build(refseq, phy, [haps[c] for c in g["cols"]])
    iters, done, props, mat, mix = _run(demo, tmp_path, sub, g["row_ptr"], g["site"], g["obs"],
                                        numpy.ones(1000), g["inits"][0], 1e-4, 10000)
    assert numpy.array_equal(mat, g["mat"])
    assert (iters, done) == (int(g["iters"][0]), 1)
    assert numpy.abs(props - g["props"]).max() < 1e-9
    assert numpy.array_equal(mix.argmax(axis=1), g["mix"].argmax(axis=1))


def test_config1_from_c_over_records(demo, tmp_path, b17):
    """The same problem with the loop and the posterior over row-dictionary records (mxm_encode_rows ->
    mxm_em_loop_coded -> mxm_em_step_coded), descriptor filled in by hand from C."""
    from mixemt_amd import preprocess
    refseq, phy, haps, tables = b17
    g = golden("g7_config1")
    sub = preprocess.HapVarTables.build(refseq, phy, [haps[c] for c in g["cols"]])
    iters, done, props, mat, mix = _run(demo, tmp_path, sub, g["row_ptr"], g["site"], g["obs"],
                                        numpy.ones(1000), g["inits"][0], 1e-4, 10000, records=True)
    assert numpy.array_equal(mat, g["mat"])
    assert (iters, done) == (int(g["iters"][0]), 1)
    assert numpy.abs(props - g["props"]).max() < 1e-9
    assert numpy.array_equal(mix.argmax(axis=1), g["mix"].argmax(axis=1))
    assert numpy.allclose(numpy.exp(mix), numpy.exp(g["mix"]), rtol=0, atol=1e-9)
