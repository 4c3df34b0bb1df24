"""
On-disk state of a run -- the `-s PREFIX` / `-l PREFIX` files of the reference
(bin/mixemt:168-245, README.md:251-262), same names and formats so either tool
can resume the other's run:

    PREFIX.haps       one haplogroup id per line (matrix column order)
    PREFIX.reads      'row<TAB>id<TAB>id...' per matrix row
    PREFIX.em.npy     EM input matrix       float64 [R][H]  (numpy.save)
    PREFIX.mat.npy    posterior matrix      float64 [R][H]
    PREFIX.prop.npy   proportions           float64 [H]

Matrices may live on the GPU -- as tensors or as row-dictionary records (preprocess.CodedMatrix): they are streamed
to the .npy files in row slabs
through a memory-mapped array, so a 43 GB matrix never needs a host copy.
"""

import sys

import numpy

try:
    import torch
except ImportError:          # pragma: no cover
    torch = None

SLAB_BYTES = 256 << 20


def save_matrix(path, mat):
    """numpy.save-compatible write of a numpy array or (device) tensor."""
    records = hasattr(mat, "dense") and hasattr(mat, "rec")       # preprocess.CodedMatrix: decoded slab by slab
    if not records and (torch is None or not isinstance(mat, torch.Tensor)):
        numpy.save(path, mat)
        return
    if not path.endswith(".npy"):
        path += ".npy"                      # numpy.save appends it too
    out = numpy.lib.format.open_memmap(path, mode="w+", dtype=numpy.float64, shape=tuple(mat.shape))
    if records:
        step = max(1, SLAB_BYTES // max(1, mat.shape[1] * 8))
        for lo in range(0, mat.shape[0], step):
            out[lo:lo + step] = mat.dense(lo, lo + step).cpu().numpy()
    elif mat.dim() == 1:
        out[:] = mat.cpu().numpy()
    else:
        step = max(1, SLAB_BYTES // max(1, mat.shape[1] * 8))
        for lo in range(0, mat.shape[0], step):
            out[lo:lo + step] = mat[lo:lo + step].cpu().numpy()
    out.flush()
    del out


def dump_all(prefix, haps, reads, em_mat, em_results):
    """bin/mixemt:214-245; a failure is a warning, as there."""
    try:
        with open("%s.haps" % prefix, "w") as fout:
            for hap in haps:
                fout.write("%s\n" % hap)
        with open("%s.reads" % prefix, "w") as fout:
            for i, ids in enumerate(reads):
                fout.write("%d\t%s\n" % (i, "\t".join(ids)))
        props, read_hap_mat = em_results
        save_matrix("%s.em" % prefix, em_mat)
        save_matrix("%s.mat" % prefix, read_hap_mat)
        save_matrix("%s.prop" % prefix, props)
    except (ValueError, IOError) as inst:
        sys.stderr.write("Warning: %s\n" % inst)


def load_prev(prefix, mmap=False):
    """
    bin/mixemt:168-211 -> (haps, reads, wts, init_mat, (props, read_hap_mat)).
    A missing PREFIX.em.npy only disables refinement (init_mat = None, :205-210);
    any other failure raises ValueError (the reference prints and exits).
    mmap=True maps the matrices instead of reading them (for upload in slabs).
    """
    mode = "r" if mmap else None
    try:
        with open("%s.haps" % prefix) as fin:
            haps = [line.rstrip() for line in fin]
        reads, counts = [], []
        with open("%s.reads" % prefix) as fin:
            for line in fin:
                ids = line.rstrip().split("\t")[1:]
                reads.append(ids)
                counts.append(len(ids))
        read_hap_mat = numpy.load("%s.mat.npy" % prefix, mmap_mode=mode)
        props = numpy.load("%s.prop.npy" % prefix)
        wts = numpy.array(counts)
    except (ValueError, IOError) as inst:
        raise ValueError("Error loading previous results:\n%s" % inst)
    try:
        init_mat = numpy.load("%s.em.npy" % prefix, mmap_mode=mode)
    except IOError as inst:
        sys.stderr.write("Error loading previous EM input:\n%s\n" % inst)
        sys.stderr.write("Contribution estimate refinement will be skipped\n")
        init_mat = None
    return haps, reads, wts, init_mat, (props, read_hap_mat)
