import argparse
import os
import sys

import numpy
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a ROCm GPU (run with -m gpu on the MI355X box)")
    config.addinivalue_line("markers", "slow: CPU oracle runs that take minutes (excluded by default)")


def pytest_collection_modifyitems(config, items):
    if "slow" in (config.getoption("-m") or ""):
        return
    skip = pytest.mark.skip(reason="slow oracle run; select with -m slow")
    for item in items:
        if "slow" in item.keywords:
            item.add_marker(skip)


def golden(name):
    path = os.path.join(GOLDEN, name + ".npz")
    if not os.path.exists(path):
        pytest.skip("golden fixture %s missing" % name)
    return numpy.load(path)


def em_args(**kw):
    """The five fields the hot path reads from the CLI namespace (em.py:117-135),
    at the CLI defaults (bin/mixemt:395-414)."""
    args = argparse.Namespace(init_alpha=1.0, tolerance=0.0001, max_iter=10000, n_multi=1,
                              verbose=False)
    for key, val in kw.items():
        setattr(args, key, val)
    return args


@pytest.fixture(scope="session", autouse=True)
def built_library():
    """The in-tree library (hipcc cross-compiles without a GPU; a no-op when it is up to date):
    host-side entry points such as the signature parser live in it, so CPU tests need it too."""
    from mixemt_amd import build
    return build.build()


@pytest.fixture(autouse=True)
def default_tuning(built_library):
    """Every tuning / measurement knob of the library (include/mixemt_hip_tuning.h: process-wide state) is back at
    its default after each test, whatever the test set and however it ended -- so that e.g. config 3 at 10^6 rows
    runs the restart schedule the product ships, not the one an earlier test selected."""
    yield
    from mixemt_amd import _lib
    _lib.load().mxm_reset_tuning()


@pytest.fixture(scope="session")
def b17():
    """(refseq, phylo, sorted haplogroups, HapVarTables) for Build 17 + RSRS."""
    from mixemt_amd import phylotree, preprocess
    refseq = phylotree.load_rsrs()
    phy = phylotree.load_build17(refseq)
    haps = sorted(phy.hap_var)
    return refseq, phy, haps, preprocess.HapVarTables.build(refseq, phy, haps)


@pytest.fixture(scope="session")
def toy():
    from mixemt_amd import phylotree
    return "AAAAAAAAA", phylotree.example(), list("ABCDEFGHI")
