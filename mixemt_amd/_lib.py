"""
ctypes binding of libmixemt_hip.so (the C ABI in include/mixemt_hip.h).

There is no CPU fallback: if the shared object is missing or a symbol cannot
be resolved this module raises, and every product entry point raises with it.
"""

import ctypes
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
# MXM_LIB points at an alternative build of the same ABI (kernel-shape tuning runs)
LIB_PATH = os.environ.get("MXM_LIB") or os.path.join(_PKG, "lib", "libmixemt_hip.so")

c_i32, c_i64, c_f64 = ctypes.c_int32, ctypes.c_int64, ctypes.c_double
c_ptr, c_size = ctypes.c_void_p, ctypes.c_size_t


class EmState(ctypes.Structure):
    """mxm_em_state (include/mixemt_hip.h)."""
    _fields_ = [("done", c_i32), ("iters", c_i32), ("l1", c_f64), ("ticket", ctypes.c_uint32), ("error", ctypes.c_uint32)]


class Coded(ctypes.Structure):
    """mxm_coded (include/mixemt_hip.h): a matrix in row-dictionary storage."""
    _fields_ = [("rec", c_ptr), ("rec_off", c_ptr), ("ndist", c_ptr), ("R", c_i64),
                ("P_rest", c_ptr), ("ldp_rest", c_i64), ("w_rest", c_ptr), ("R_rest", c_i64),
                ("wide_rows", c_ptr), ("n_wide", c_i64),
                # the quad dictionary beside the records (round 5; all NULL / 0 = none)
                ("qrec", c_ptr), ("qoff", c_ptr), ("nquad", c_ptr), ("quad_rows", c_ptr), ("n_quad_rows", c_i64),
                ("byte_rows", c_ptr), ("n_byte_rows", c_i64)]


class AlnColumns(ctypes.Structure):
    """mxm_aln_columns (include/mixemt_hip.h): alignments as columns (host pointers)."""
    _fields_ = [("n_aln", c_i64), ("n_frag", c_i64), ("ref_start", c_ptr), ("mapq", c_ptr), ("frag", c_ptr),
                ("cig_ptr", c_ptr), ("cigar", c_ptr), ("seq_ptr", c_ptr), ("seq", c_ptr), ("qual", c_ptr),
                ("has_qual", c_ptr)]


class AlnSizes(ctypes.Structure):
    """mxm_aln_sizes (include/mixemt_hip.h)."""
    _fields_ = [("n_rows", c_i64), ("nnz", c_i64), ("n_grouped", c_i64), ("n_dropped", c_i64), ("text_bytes", c_i64),
                ("n_frag_seen", c_i64), ("frag_nnz", c_i64)]


class BamSizes(ctypes.Structure):
    """mxm_bam_sizes (include/mixemt_hip.h)."""
    _fields_ = [("n_aln", c_i64), ("n_frag", c_i64), ("n_cigar", c_i64), ("n_bases", c_i64), ("names_bytes", c_i64),
                ("n_ref", c_i64), ("n_records_total", c_i64), ("n_skipped_unplaced", c_i64)]


# name -> (restype, argtypes); must list every symbol the two headers declare
# (include/mixemt_hip.h: the boundary; include/mixemt_hip_tuning.h: measurement / shape knobs)
SIGNATURES = {
    "mxm_version": (ctypes.c_int, []),
    "mxm_last_error": (ctypes.c_char_p, []),
    "mxm_linear_supported": (ctypes.c_int, [c_i32]),
    "mxm_workspace_bytes": (c_size, [c_i64, c_i32, c_i32]),
    "mxm_restart_tile": (ctypes.c_int, [c_i32]),
    "mxm_restart_tile_coded": (ctypes.c_int, [ctypes.POINTER(Coded), c_i32]),
    "mxm_quad_loop_min_rows": (c_i64, [c_i32]),
    "mxm_build_em_matrix": (ctypes.c_int, [c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr,
                                           c_i64, c_i32, c_i32, c_ptr, c_i64, c_ptr]),
    "mxm_build_em_matrix_lut": (ctypes.c_int, [c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr,
                                               c_i64, c_i32, c_i32, c_ptr, c_i64, c_ptr]),
    "mxm_build_em_matrix_sparse": (ctypes.c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr,
                                                  c_i64, c_i32, c_i32, c_ptr, c_i64, c_ptr, c_ptr, c_ptr]),
    "mxm_build_em_matrix_lut_rows": (ctypes.c_int, [c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr,
                                                    c_i64, c_i32, c_i32, c_ptr, c_i64, c_ptr]),
    "mxm_scatter_records": (ctypes.c_int, [c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr]),
    "mxm_preload": (ctypes.c_int, []),
    "mxm_record_bytes": (c_size, [c_i64, c_i32]),
    "mxm_build_em_records": (ctypes.c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr,
                                            c_i64, c_i32, c_i32, c_ptr, c_i64, c_ptr, c_size, c_ptr, c_ptr, c_ptr,
                                            c_ptr, c_ptr, c_ptr, c_ptr]),
    "mxm_linearize": (ctypes.c_int, [c_ptr, c_i64, c_i64, c_i32, c_ptr, c_i64, c_ptr, c_ptr]),
    "mxm_em_iter": (ctypes.c_int, [c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_i64, c_i32, c_i32,
                                   c_ptr, c_ptr, c_ptr, c_size, c_ptr]),
    "mxm_m_finalize": (ctypes.c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_i32, c_i32, c_f64, c_i32, c_ptr,
                                      c_ptr]),
    "mxm_em_loop": (ctypes.c_int, [c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_i32, c_i32,
                                   c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_f64, c_i32, c_i32, c_ptr, c_size,
                                   c_ptr, ctypes.POINTER(EmState)]),
    "mxm_linearize_f32": (ctypes.c_int, [c_ptr, c_i64, c_i64, c_i32, c_ptr, c_i64, c_ptr, c_ptr]),
    "mxm_em_iter_f32": (ctypes.c_int, [c_ptr, c_i64, c_ptr, c_ptr, c_i64, c_i32, c_i32, c_ptr, c_ptr, c_ptr,
                                       c_size, c_ptr]),
    "mxm_em_loop_f32": (ctypes.c_int, [c_ptr, c_i64, c_ptr, c_i64, c_i32, c_i32, c_ptr, c_ptr, c_ptr, c_ptr,
                                       c_ptr, c_f64, c_i32, c_i32, c_ptr, c_size, c_ptr,
                                       ctypes.POINTER(EmState)]),
    "mxm_coded_bytes": (c_size, [c_i64, c_i32]),
    "mxm_exchange_handle_bytes": (c_size, []),
    "mxm_exchange_create": (ctypes.c_int, [c_i32, c_i32, c_i64, ctypes.POINTER(c_ptr), c_ptr]),
    "mxm_exchange_connect": (ctypes.c_int, [c_ptr, c_ptr]),
    "mxm_exchange_push": (ctypes.c_int, [c_ptr, c_ptr, c_i64, c_ptr]),
    "mxm_exchange_pull": (ctypes.c_int, [c_ptr, c_ptr, c_i64, c_ptr, c_i32, c_ptr]),
    "mxm_exchange_reduce": (ctypes.c_int, [c_ptr, c_ptr, c_i64, c_ptr, c_i32, c_ptr]),
    "mxm_exchange_info": (ctypes.c_int, [c_ptr, ctypes.POINTER(c_i32), ctypes.POINTER(c_i64)]),
    "mxm_exchange_destroy": (None, [c_ptr]),
    "mxm_expand_tables": (ctypes.c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i32, c_i32, c_i64, c_ptr, c_ptr]),
    "mxm_quad_bytes": (c_size, [c_i64, c_i32]),
    "mxm_quad_lists_scratch_bytes": (c_size, [c_i64]),
    "mxm_quad_lists": (ctypes.c_int, [c_ptr, c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_size, c_ptr]),
    "mxm_build_quads": (ctypes.c_int, [ctypes.POINTER(Coded), c_i32, c_ptr, c_size, c_ptr, c_ptr, c_ptr, c_ptr]),
    "mxm_encode_rows": (ctypes.c_int, [c_ptr, c_i64, c_i64, c_i32, c_ptr, c_size, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    "mxm_decode_rows": (ctypes.c_int, [ctypes.POINTER(Coded), c_i32, c_ptr, c_i64, c_ptr]),
    "mxm_row_argmax_votes_coded": (ctypes.c_int, [ctypes.POINTER(Coded), c_i32, c_i32, c_ptr, c_ptr, c_ptr, c_ptr, c_i64,
                                                  c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_size, c_ptr]),
    "mxm_em_step_coded": (ctypes.c_int, [ctypes.POINTER(Coded), c_i32, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_ptr, c_i64,
                                         c_ptr, c_i64, c_i32, c_ptr]),
    "mxm_gather_columns_coded": (ctypes.c_int, [ctypes.POINTER(Coded), c_i32, c_ptr, c_i32, c_ptr, c_i64, c_ptr, c_i64,
                                                c_ptr, c_i64, c_ptr]),
    "mxm_em_iter_coded": (ctypes.c_int, [ctypes.POINTER(Coded), c_ptr, c_ptr, c_i32, c_i32, c_ptr, c_ptr, c_ptr,
                                         c_size, c_ptr]),
    "mxm_em_loop_coded": (ctypes.c_int, [ctypes.POINTER(Coded), c_ptr, c_i32, c_i32, c_ptr, c_ptr, c_ptr, c_ptr,
                                         c_ptr, c_f64, c_i32, c_i32, c_ptr, c_size, c_ptr,
                                         ctypes.POINTER(EmState)]),
    "mxm_em_step": (ctypes.c_int, [c_ptr, c_i64, c_ptr, c_ptr, c_i64, c_i32, c_ptr, c_i64, c_i32,
                                   c_ptr, c_ptr, c_size, c_ptr]),
    "mxm_log_normalize": (ctypes.c_int, [c_ptr, c_i32, c_ptr, c_ptr]),
    "mxm_l1_exp_diff": (ctypes.c_int, [c_ptr, c_ptr, c_i32, c_ptr, c_ptr]),
    "mxm_add_scalar": (ctypes.c_int, [c_ptr, c_i64, c_i64, c_i32, c_f64, c_ptr]),
    "mxm_assign_reads": (ctypes.c_int, [c_ptr, c_i64, c_ptr, c_ptr, c_i32, c_i64, c_i32, c_f64, c_ptr, c_ptr]),
    "mxm_diag_stream_read": (ctypes.c_int, [c_ptr, c_size, c_i32, c_i32, c_ptr, c_ptr]),
    "mxm_diag_stream_coded": (ctypes.c_int, [ctypes.POINTER(Coded), c_i32, c_i32, c_ptr, c_ptr]),
    "mxm_diag_stream_quads": (ctypes.c_int, [ctypes.POINTER(Coded), c_i32, c_i32, c_ptr, c_ptr]),
    "mxm_diag_fused_force_abort": (ctypes.c_int, [c_i32]),
    "mxm_set_fused_coded_grid": (ctypes.c_int, [c_i32]),
    "mxm_set_quad_left_grid": (ctypes.c_int, [c_i32]),
    "mxm_set_coded_batch_tile": (ctypes.c_int, [c_i32]),
    "mxm_set_sparse_long_rows": (ctypes.c_int, [c_i32]),
    "mxm_set_quad_encoder": (ctypes.c_int, [c_i32]),
    "mxm_set_sparse_long_entries": (ctypes.c_int, [c_i32]),
    "mxm_diag_fused_stamps": (ctypes.c_int, [c_ptr, ctypes.POINTER(ctypes.c_ulonglong)]),
    "mxm_set_timing_events": (ctypes.c_int, [c_ptr, c_ptr]),
    "mxm_set_batch_tile": (ctypes.c_int, [c_i32]),
    "mxm_encode_signatures": (ctypes.c_int64, [ctypes.c_char_p, c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_ptr, c_ptr,
                                               c_i64]),
    "mxm_aln_encode": (ctypes.c_int, [ctypes.POINTER(AlnColumns), c_ptr, c_i64, c_ptr, c_i32, c_i32, c_i32, c_i32,
                                      ctypes.POINTER(c_ptr)]),
    "mxm_aln_sizes_of": (ctypes.c_int, [c_ptr, ctypes.POINTER(AlnSizes)]),
    "mxm_aln_fetch": (ctypes.c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    "mxm_aln_fetch_fragments": (ctypes.c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    "mxm_aln_free": (None, [c_ptr]),
    "mxm_bam_read": (ctypes.c_int, [ctypes.c_char_p, c_i32, ctypes.POINTER(c_ptr)]),
    "mxm_bam_sizes_of": (ctypes.c_int, [c_ptr, ctypes.POINTER(BamSizes)]),
    "mxm_bam_columns": (ctypes.c_int, [c_ptr, ctypes.POINTER(AlnColumns)]),
    "mxm_bam_fetch_names": (ctypes.c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    "mxm_bam_free": (None, [c_ptr]),
    "mxm_set_compact_restarts": (ctypes.c_int, [c_i32]),
    "mxm_set_loop_graph": (ctypes.c_int, [c_i32]),
    "mxm_set_loop_fused": (ctypes.c_int, [c_i32, c_i32]),
    "mxm_set_progress_callback": (ctypes.c_int, [c_ptr, c_ptr, c_i32]),
    "mxm_set_min_rows_per_wg": (ctypes.c_int, [c_i32]),
    "mxm_reset_tuning": (ctypes.c_int, []),
    "mxm_describe_stream_kernel": (ctypes.c_int, [c_i32, c_i32, ctypes.c_char_p, c_size]),
    "mxm_set_sparse_max_distinct": (ctypes.c_int, [c_i32]),
    "mxm_row_argmax_votes": (ctypes.c_int, [c_ptr, c_i64, c_ptr, c_i64, c_i32, c_ptr, c_ptr,
                                            c_ptr, c_size, c_ptr]),
    "mxm_first_seen": (ctypes.c_int, [c_ptr, c_i64, c_i32, c_ptr, c_ptr]),
    "mxm_gather_columns": (ctypes.c_int, [c_ptr, c_i64, c_i64, c_i32, c_ptr, c_i32, c_ptr, c_i64, c_ptr]),
    "mxm_fold_logaddexp": (ctypes.c_int, [c_ptr, c_i64, ctypes.POINTER(c_ptr), ctypes.POINTER(c_i64), c_i32,
                                          c_i64, c_i32, c_f64, c_ptr]),
}

# the MXM_VERSION of include/mixemt_hip.h these signatures were written for; load() refuses any other
ABI_VERSION = 600

PROGRESS_FN = ctypes.CFUNCTYPE(None, ctypes.POINTER(EmState), c_i32, c_ptr)

_lib = None


class MixemtHipError(RuntimeError):
    pass


def load():
    """dlopen the library and bind every symbol; raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MixemtHipError(
            "libmixemt_hip.so is not built (%s); run `python -m mixemt_amd.build` -- "
            "there is no CPU fallback for the EM hot path" % LIB_PATH)
    # PyTorch-ROCm first: it brings the HIP runtime (soname libamdhip64.so.7) this
    # library binds to, so both share ONE runtime -- streams and pointers are then
    # interchangeable.  Loading order matters; a second runtime must not appear.
    from . import _dev  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    # the version first: a stale library lacks newer symbols, and the useful message is "rebuild", not AttributeError
    try:
        lib.mxm_version.restype = ctypes.c_int
        lib.mxm_version.argtypes = []
        have = lib.mxm_version()
    except AttributeError:
        have = None
    if have != ABI_VERSION:
        raise MixemtHipError("%s reports ABI version %s, this binding was written for %d: rebuild with "
                             "`python -m mixemt_amd.build --force`" % (LIB_PATH, have, ABI_VERSION))
    for name, (restype, argtypes) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            raise MixemtHipError("%s has ABI version %d but lacks %s: rebuild with "
                                 "`python -m mixemt_amd.build --force`" % (LIB_PATH, have, name))
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    # the library's code object is loaded onto the device now rather than by the first kernel launch (18 ms that would land
    # in whatever stage happens to launch first); nothing to do without a GPU (the host functions work there)
    try:
        if _dev.torch is not None and _dev.torch.cuda.is_available():
            lib.mxm_preload()
    except Exception:        # pragma: no cover - a preload that fails only moves the cost back
        pass
    return lib


def check(rc, what):
    """Turn a negative status into the exception the reference's CLI catches
    (ValueError, bin/mixemt:325-327)."""
    if rc != 0:
        msg = load().mxm_last_error().decode("utf-8", "replace")
        raise ValueError("%s failed (%d): %s" % (what, rc, msg))
