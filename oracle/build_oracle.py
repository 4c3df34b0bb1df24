"""
ORACLE -- test infrastructure, not product code.

CPU restatement of the reference's EM-input construction:
    HapVarBaseMatrix.__init__/add_hap_markers  /root/reference/mixemt/preprocess.py:39-67
    HapVarBaseMatrix._prob                     preprocess.py:69-84
    HapVarBaseMatrix.prob_for_vars             preprocess.py:86-96
    pos_obs_from_sig                           preprocess.py:151-160
    build_em_matrix                            preprocess.py:177-198
Three forms of the same arithmetic, each checked against the one above it:
    build_em_matrix      -- per-cell Python loops, dictionary look-ups (small cases)
    build_em_matrix_np   -- one numpy vector over haplogroups per observed site,
                            added in signature order => bit-identical sums
    oracle.c             -- plain C over flat tables (larger cases, cpu baseline)
Pinned by tests/golden (G1/G2: matrices produced by importing the reference).

The variant-string helpers are restated here (phylotree.py:338-365) so the
oracle does not import the product package.
"""

import math

import numpy


def _bare(var):
    if var.startswith("("):
        var = var[1:-1]
    return var.rstrip("!")


def var_pos(var):
    """phylotree.py:338-351: 0-based site."""
    return int(_bare(var)[1:-1]) - 1


def var_der(var):
    """phylotree.py:354-364: derived base."""
    return var.rstrip(")!")[-1].upper()


def parse_sig(read_sig):
    """preprocess.py:151-160: 'pos:base,pos:base' -> [(int pos, str base)]."""
    out = []
    for item in read_sig.split(","):
        pos, obs = item.split(":")
        out.append((int(pos), obs))
    return out


class HapVarBase(object):
    """preprocess.py:23-96 (defaults mut_wt=0.01, mut_max=0.5 from :39)."""

    def __init__(self, refseq, phylo, mut_wt=0.01, mut_max=0.5):
        self.refseq = refseq
        self.mut_prob = {}
        for pos in phylo.variants:
            self.mut_prob[pos] = min(mut_max,
                                     mut_wt * sum(phylo.variants[pos].values()))
        self.markers = {}
        for hap in phylo.hap_var:
            carried = {}
            for var in phylo.hap_var[hap]:
                pos, der = var_pos(var), var_der(var)
                if der != refseq[pos]:
                    carried[pos] = der
            self.markers[hap] = carried

    def prob(self, hap, pos, base):
        """preprocess.py:69-84."""
        carried = self.markers[hap]
        if pos in carried:
            if carried[pos] == base:
                return 1.0 - self.mut_prob[pos]
        elif self.refseq[pos] == base:
            return 1.0 - self.mut_prob[pos]
        return self.mut_prob[pos] / 3.0

    def prob_for_vars(self, hap, pos_obs):
        """preprocess.py:86-96: in-order sum of math.log, starting at int 0."""
        total = 0
        for pos, obs in pos_obs:
            total += math.log(self.prob(hap, pos, obs))
        return total


def build_em_matrix(refseq, phylo, reads, haplogroups):
    """preprocess.py:177-198, cell by cell."""
    hvb = HapVarBase(refseq, phylo)
    mat = numpy.empty((len(reads), len(haplogroups)))
    for i, sig in enumerate(reads):
        pos_obs = parse_sig(sig)
        for j, hap in enumerate(haplogroups):
            mat[i, j] = hvb.prob_for_vars(hap, pos_obs)
    return mat


def flat_tables(refseq, phylo, haplogroups):
    """
    Flatten the dictionaries into arrays (the oracle's own encoding, used by
    the numpy and C forms):
        sites[S]   sorted 0-based variant sites (= phylo.get_variant_pos())
        exp[S][H]  uint8, ord() of the base haplogroup h is expected to show at
                   site s: its marker if it carries one, else refseq[site]
        lhit[S]    math.log(1 - mu_s)      lmiss[S]  math.log(mu_s / 3)
    """
    hvb = HapVarBase(refseq, phylo)
    sites = sorted(phylo.variants.keys())
    where = {pos: k for k, pos in enumerate(sites)}
    exp = numpy.empty((len(sites), len(haplogroups)), dtype=numpy.uint8)
    for k, pos in enumerate(sites):
        exp[k, :] = ord(refseq[pos])
    for j, hap in enumerate(haplogroups):
        for pos, der in hvb.markers[hap].items():
            if pos in where:
                exp[where[pos], j] = ord(der)
    lhit = numpy.array([math.log(1.0 - hvb.mut_prob[p]) for p in sites])
    lmiss = numpy.array([math.log(hvb.mut_prob[p] / 3.0) for p in sites])
    return sites, exp, lhit, lmiss


def build_em_matrix_np(refseq, phylo, reads, haplogroups, tables=None):
    """Same sums as build_em_matrix, one numpy vector per observed site."""
    sites, exp, lhit, lmiss = tables or flat_tables(refseq, phylo, haplogroups)
    where = {pos: k for k, pos in enumerate(sites)}
    mat = numpy.empty((len(reads), len(haplogroups)))
    for i, sig in enumerate(reads):
        row = numpy.zeros(len(haplogroups))
        for pos, obs in parse_sig(sig):
            k = where[pos]
            code = ord(obs) if len(obs) == 1 else 0
            row = row + numpy.where(exp[k] == code, lhit[k], lmiss[k])
        mat[i] = row
    return mat
