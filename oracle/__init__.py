"""
ORACLE -- test infrastructure only.

CPU restatements of the reference's hot path (preprocess.build_em_matrix and
the em.run_em loop).  Imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py as the CHECKER; never by the product package
(mixemt_amd/), which has no CPU fallback and fails loudly without its HIP
library.
"""
