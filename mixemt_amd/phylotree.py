"""
Host-side haplogroup model: Phylotree CSV -> flat tables for the device path.

This is the build's own reader for the indented-CSV Phylotree builds that
mixemt ships (reference: mixemt/phylotree.py:118-229 for the object surface,
:461-488 for the line format).  It keeps the attribute surface that the hot
path's boundary reads -- ``hap_var`` (haplogroup id -> SNP variant strings),
``variants`` (0-based site -> {derived allele: mutation count}), ``ignore``,
``refseq``, ``get_variant_pos()``, ``add_custom_hap()``, ``ignore_sites()`` --
so a `Phylotree` from either package can be handed to
`mixemt_amd.preprocess.build_em_matrix`.

Internally the tree is a struct-of-arrays (parent index / name / own variants)
and every node's cumulative variant set is produced in one top-down sweep,
instead of a linked node graph walked leaf-to-root per haplogroup.

Nothing here runs on the GPU; it is one-time host parsing (~1 s for Build 17).
"""

import collections
import gzip
import os

_DATA_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")


# --------------------------------------------------------------------------
# variant-string helpers (reference semantics: phylotree.py:338-434)
# --------------------------------------------------------------------------

def _core(var):
    """'(A95c)' / 'A263G!' / 'C3516a' -> the bare 'A95c' / 'A263G' / 'C3516a'."""
    if var[:1] == "(":
        var = var[1:-1]
    return var.rstrip("!")


def pos_from_var(var):
    """0-based site of a SNP variant string (reference: phylotree.py:338)."""
    return int(_core(var)[1:-1]) - 1


def der_allele(var):
    """Derived base, upper-cased (reference: phylotree.py:354)."""
    return var.rstrip(")!")[-1].upper()


def anc_allele(var):
    """Ancestral base, upper-cased (reference: phylotree.py:367)."""
    return var.lstrip("(")[0].upper()


def is_snp(var):
    """Indels carry '.' or 'd' (reference: phylotree.py:380)."""
    return "." not in var and "d" not in var


def is_unstable(var):
    return var[:1] == "("


def is_backmutation(var):
    return "!" in var


def rm_snp_annot(var):
    """Strip (), ! and case annotations (reference: phylotree.py:421)."""
    return _core(var).upper()


def _split_line(line):
    """
    One CSV line -> (level, hap_id, variant tokens).

    The level is the number of leading empty fields.  A node without an id
    shows its (space padded) variant field where the id would be, one column
    early; that is recognised by the blank in the field
    (reference: phylotree.py:461-488).
    """
    fields = line.rstrip().split(",")
    first = 0
    while fields[first] == "":
        first += 1
    if " " in fields[first]:
        # anonymous node: the id column (first-1) is blank
        return first - 1, "", fields[first].split()
    return first, fields[first], fields[first + 1].split()


class Phylotree(object):
    """
    Flat representation of a Phylotree build.

    Attributes (same meaning as the reference object, phylotree.py:14-34):
        refseq, variants, ignore, hap_var, anon_haps, rm_unstable, rm_backmut
    Extra (this build):
        names[i], parent[i], anon[i], node_vars[i]  -- struct-of-arrays tree
    """

    def __init__(self, phy_in=None, refseq=None, anon_haps=False,
                 rm_unstable=False, rm_backmut=False):
        self.refseq = refseq
        self.anon_haps = anon_haps
        self.rm_unstable = rm_unstable
        self.rm_backmut = rm_backmut
        self.variants = collections.defaultdict(collections.Counter)
        self.ignore = set()
        self.hap_var = None
        self.names = []
        self.parent = []
        self.anon = []
        self.node_vars = []
        if phy_in is not None:
            self._parse(phy_in)
            self.process_variants()
            self.process_haplotypes()

    # -- construction ------------------------------------------------------

    def _parse(self, lines):
        """Indented CSV -> parent/name/variant arrays (cf. phylotree.py:143)."""
        open_at_level = []          # node index currently open at each depth
        anon_children = []          # per node: anonymous children named so far
        for line in lines:
            level, hap_id, toks = _split_line(line)
            del open_at_level[level:]
            par = open_at_level[-1] if open_at_level else -1
            is_anon = hap_id == ""
            if is_anon and par >= 0:
                anon_children[par] += 1
                hap_id = "%s[%d]" % (self.names[par], anon_children[par])
            idx = len(self.names)
            self.names.append(hap_id)
            self.parent.append(par)
            self.anon.append(is_anon)
            self.node_vars.append([v for v in toks if is_snp(v)])
            anon_children.append(0)
            open_at_level.append(idx)

    def process_variants(self):
        """
        Site table: pos -> Counter(derived allele -> #mutation events), after
        the unstable / back-mutation filters (reference: phylotree.py:168-196).

        Like the reference, the counters are NOT cleared when this runs a
        second time (from `ignore_sites`), so counts accumulate -- that is
        part of the reference's observable behaviour for `-e`.
        """
        for own in self.node_vars:
            for var in own:
                pos = pos_from_var(var)
                if self.rm_unstable and is_unstable(var):
                    self.ignore.add(pos)
                elif self.rm_backmut and is_backmutation(var):
                    self.ignore.add(pos)
                else:
                    self.variants[pos][der_allele(var)] += 1
        for pos in self.ignore:
            self.variants.pop(pos, None)
        dropped = self.ignore
        self.node_vars = [[v for v in own if pos_from_var(v) not in dropped]
                          for own in self.node_vars]

    def process_haplotypes(self):
        """
        hap id -> cumulative variant list (sorted by site); a mutation nearer
        the node masks an older one at the same site; nodes with identical
        lists merge into one 'A/B' id (reference: phylotree.py:198-219, :66-88).
        """
        cumulative = []             # per node: {pos: bare variant}, top-down
        groups = collections.OrderedDict()
        for i, own in enumerate(self.node_vars):
            mine = {}
            for var in own:
                mine.setdefault(pos_from_var(var), rm_snp_annot(var))
            if self.parent[i] >= 0:
                merged = dict(cumulative[self.parent[i]])
                merged.update(mine)
            else:
                merged = mine
            cumulative.append(merged)
            if self.anon_haps or not self.anon[i]:
                key = ",".join(merged[p] for p in sorted(merged))
                groups.setdefault(key, []).append(self.names[i])
        table = {}
        for key, members in groups.items():
            table["/".join(members)] = [v for v in key.split(",") if v != ""]
        self.hap_var = table

    # -- queries / edits -----------------------------------------------------

    def get_variant_pos(self):
        """Sorted 0-based variant sites (reference: phylotree.py:221)."""
        return sorted(self.variants.keys())

    def add_custom_hap(self, hap_id, variants):
        """Reference: phylotree.py:231-251 (ValueError on a name clash)."""
        if hap_id in self.hap_var:
            raise ValueError("Custom haplogroup name '%s' already in use."
                             % (hap_id))
        self.hap_var[hap_id] = [v for v in variants
                                if pos_from_var(v) not in self.ignore]

    def ignore_sites(self, sites_str):
        """
        '5,10-12' (1-based, inclusive ranges) -> ignored 0-based sites, then
        rebuild both tables (reference: phylotree.py:253-274).
        """
        for item in sites_str.split(","):
            if "-" in item:
                lo, hi = item.split("-")
                self.ignore.update(range(int(lo) - 1, int(hi)))
            else:
                self.ignore.add(int(item) - 1)
        self.process_variants()
        self.process_haplotypes()


# --------------------------------------------------------------------------
# packaged data (Phylotree Build 17 + RSRS; data files, gzip'd)
# --------------------------------------------------------------------------

def read_fasta_first(path):
    """First record of a (optionally gzip'd) FASTA as one upper-case string."""
    opener = gzip.open if path.endswith(".gz") else open
    seq = []
    with opener(path, "rt") as fin:
        seen = False
        for line in fin:
            if line.startswith(">"):
                if seen:
                    break
                seen = True
            elif seen:
                seq.append(line.strip())
    if not seen:
        raise ValueError('no records in "%s"' % path)
    return "".join(seq).upper()


def load_rsrs():
    """Default reference (reference default: bin/mixemt:155-156)."""
    return read_fasta_first(os.path.join(_DATA_DIR, "RSRS.mtDNA.fa.gz"))


def load_build17(refseq=None, anon_haps=True, rm_unstable=False,
                 rm_backmut=False):
    """
    Default tree with the CLI's default flags (reference: bin/mixemt:60-101,
    anon_haps default True at :376-377) -> H = 5408 haplogroups, S = 4070 sites.
    """
    if refseq is None:
        refseq = load_rsrs()
    path = os.path.join(_DATA_DIR, "mtDNA_tree_Build_17.csv.gz")
    with gzip.open(path, "rt") as fin:
        return Phylotree(fin, refseq=refseq, anon_haps=anon_haps,
                         rm_unstable=rm_unstable, rm_backmut=rm_backmut)


def load_rcrs():
    """The revised Cambridge Reference Sequence shipped beside RSRS (reference: mixemt/ref/)."""
    return read_fasta_first(os.path.join(_DATA_DIR, "rCRS.mtDNA.fa.gz"))


def load_build16(refseq=None, anon_haps=True, rm_unstable=False, rm_backmut=False):
    """Phylotree Build 16, the older tree the reference also ships (mixemt/phylotree/README.md)."""
    if refseq is None:
        refseq = load_rsrs()
    path = os.path.join(_DATA_DIR, "mtDNA_tree_Build_16.csv.gz")
    with gzip.open(path, "rt") as fin:
        return Phylotree(fin, refseq=refseq, anon_haps=anon_haps, rm_unstable=rm_unstable,
                         rm_backmut=rm_backmut)


def example():
    """The 9-haplogroup toy tree used throughout the reference's tests
    (em_test.py:78-88); data, restated here for the parity tests."""
    #            I
    #           / \
    #          /   H
    #         /   / \
    #        A   F   G
    #           / \ / \
    #          B  C D  E
    rows = ["I, A1G ,,",
            ",H, A3T A5T ,,",
            ",,F, A6T ,,",
            ",,,B, A8T ,,",
            ",,,C, T5A ,,",
            ",,G, A7T ,,",
            ",,,D, A9T ,,",
            ",,,E, A4T ,,",
            ",A, A2T A4T ,,"]
    return Phylotree(rows)
