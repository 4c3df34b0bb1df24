"""
ORACLE -- test infrastructure.  ctypes loader for oracle/_build/liboracle.so
(plain-C restatement, see oracle.c); `make -C oracle` builds it.
"""

import ctypes
import os
import subprocess

import numpy

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None


def build():
    subprocess.run(["make", "-C", _HERE], check=True, stdout=subprocess.PIPE,
                   stderr=subprocess.STDOUT)
    return _SO


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
        _lib.orc_build_em_matrix.restype = None
        _lib.orc_em_step.restype = None
    return _lib


def _p(arr):
    return arr.ctypes.data_as(ctypes.c_void_p)


def build_em_matrix(expected, lhit, lmiss, row_ptr, site, obs, n_haps):
    lib = load()
    expected = numpy.ascontiguousarray(expected, dtype=numpy.uint8)
    lhit = numpy.ascontiguousarray(lhit, dtype=numpy.float64)
    lmiss = numpy.ascontiguousarray(lmiss, dtype=numpy.float64)
    row_ptr = numpy.ascontiguousarray(row_ptr, dtype=numpy.int64)
    site = numpy.ascontiguousarray(site, dtype=numpy.uint16)
    obs = numpy.ascontiguousarray(obs, dtype=numpy.uint8)
    n_rows = len(row_ptr) - 1
    mat = numpy.empty((n_rows, n_haps))
    lib.orc_build_em_matrix(_p(expected), ctypes.c_int64(expected.shape[1]), _p(lhit), _p(lmiss),
                            _p(row_ptr), _p(site), _p(obs), ctypes.c_int64(n_rows),
                            ctypes.c_int32(n_haps), _p(mat), ctypes.c_int64(n_haps))
    return mat


def em_step(mat, wts, lnp):
    lib = load()
    mat = numpy.ascontiguousarray(mat, dtype=numpy.float64)
    wts = numpy.ascontiguousarray(wts, dtype=numpy.float64)
    lnp = numpy.ascontiguousarray(lnp, dtype=numpy.float64)
    out = numpy.empty_like(mat)
    new = numpy.empty(mat.shape[1])
    lib.orc_em_step(_p(mat), ctypes.c_int64(mat.shape[1]), _p(wts), _p(lnp),
                    ctypes.c_int64(mat.shape[0]), ctypes.c_int32(mat.shape[1]), _p(out),
                    ctypes.c_int64(mat.shape[1]), _p(new))
    return out, new
