"""
mixemt_amd -- MI355X-native EM mixture-deconvolution core for mixemt.

Only what the hot path needs:
    phylotree   host model: Phylotree CSV -> hap_var / variants tables
    preprocess  build_em_matrix (drop-in) + table / signature encoders
    em          init_props, converged, em_step, run_em (drop-in)
    dist        row-sharded multi-GPU EM loop (RCCL all-reduce per iteration)
    synth       synthetic read generator for bench / tests
    csrc/       HIP kernels + the C ABI (include/mixemt_hip.h)

Importing the package never touches the GPU or the shared library; the first
product call does, and raises if either is missing (no CPU fallback).
"""

__version__ = "0.1.0"
