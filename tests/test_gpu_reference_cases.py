"""
The reference's own unit cases for the consumers of the EM result (SURVEY.md section 8 f-1 .. f-3), restated on
the device path with the reference's inputs and expectations:

    assemble_test.py:57-63, 126-145   _find_contribs_from_reads (weights all one, min_reads, a weight that saves one)
    assemble_test.py:226-237          TestAssignReads.setUp (4 x 9 log matrix, two contributors)
    assemble_test.py:251-273          assign_read_indexes (min_fold 2, 1.5, 200, a single contributor)
    assemble_test.py:239-249          _find_best_n_for_read (through a one-row assign: best and runner-up)
"""

import argparse

import numpy
import pytest

from mixemt_amd import assign

pytestmark = pytest.mark.gpu

HAPS = list("ABCDEFGHI")
PROPS = numpy.array([0.40, 0.01, 0.01, 0.01, 0.3, 0.01, 0.01, 0.01, 0.01])


def _args(min_reads=1):
    return argparse.Namespace(min_reads=min_reads, verbose=False)


@pytest.fixture()
def mix_mat():
    # assemble_test.py:59-62 -- linear values; only the row argmax matters
    return numpy.array([[0.91, 0.01, 0.01, 0.01, 0.01, 0.01, 0.01, 0.01, 0.01],
                        [0.91, 0.01, 0.01, 0.01, 0.01, 0.01, 0.01, 0.01, 0.01],
                        [0.01, 0.01, 0.01, 0.01, 0.91, 0.01, 0.01, 0.01, 0.01]])


def test_find_contribs_from_reads_wts_all_one(mix_mat):
    assert assign.find_contribs_from_reads(mix_mat, numpy.array([1, 1, 1]), _args()) == [0, 4]


def test_find_contribs_from_reads_wts_all_one_min_reads(mix_mat):
    assert assign.find_contribs_from_reads(mix_mat, numpy.array([1, 1, 1]), _args(min_reads=2)) == [0]


def test_find_contribs_from_reads_wts_save_min_reads(mix_mat):
    assert assign.find_contribs_from_reads(mix_mat, [1, 1, 2], _args(min_reads=2)) == [0, 4]


@pytest.fixture()
def em_results():
    mat = numpy.log(numpy.array([[0.91, 0.01, 0.01, 0.01, 0.01, 0.01, 0.01, 0.01, 0.01],
                                 [0.91, 0.01, 0.01, 0.01, 0.01, 0.01, 0.01, 0.01, 0.01],
                                 [0.30, 0.01, 0.01, 0.01, 0.40, 0.01, 0.01, 0.01, 0.01],
                                 [0.01, 0.01, 0.01, 0.01, 0.91, 0.01, 0.01, 0.01, 0.01]]))
    return PROPS, mat


CONS = [["hap1", "A", 0.40], ["hap2", "E", 0.3]]
READS = [["A", "B"], ["C"], ["D"], ["E", "F", "G"]]


def test_assign_reads_simple(em_results):
    res = assign.assign_read_indexes(CONS, em_results, HAPS, READS, 2.0)
    assert dict(res) == {"hap1": {0, 1}, "hap2": {3}, "unassigned": {2}}


def test_assign_reads_simple_low_min_fold(em_results):
    res = assign.assign_read_indexes(CONS, em_results, HAPS, READS, 1.5)
    assert dict(res) == {"hap1": {0, 1}, "hap2": {2, 3}}


def test_assign_reads_simple_high_min_fold(em_results):
    res = assign.assign_read_indexes(CONS, em_results, HAPS, READS, 200)
    assert dict(res) == {"unassigned": {0, 1, 2, 3}}


def test_assign_reads_simple_only_one_con(em_results):
    res = assign.assign_read_indexes(CONS[0:1], em_results, HAPS, READS, 2)
    assert dict(res) == {"hap1": {0, 1, 2, 3}}


def test_best_and_runner_up_of_a_row():
    """assemble_test.py:239-249: among columns 1, 3, 5 of [.1 .2 .1 .3 .9 .1] the best two are 3 then 1 (column 4,
    the row maximum, is not a contributor).  With equal proportions the row goes to the contributor at column 3
    when 0.3 / 0.2 reaches min_fold, and to nobody just above it."""
    prob = numpy.log(numpy.array([[0.1, 0.2, 0.1, 0.3, 0.9, 0.1]]))
    haps = list("abcdef")
    props = numpy.full(6, 1.0 / 6.0)
    cons = [["hap1", "b", 0.2], ["hap2", "d", 0.2], ["hap3", "f", 0.2]]
    assert dict(assign.assign_read_indexes(cons, (props, prob), haps, [["r"]], 1.49)) == {"hap2": {0}}
    assert dict(assign.assign_read_indexes(cons, (props, prob), haps, [["r"]], 1.51)) == {"unassigned": {0}}
    # dividing out the proportions (assemble.py:302-305): a contributor three times as abundant loses its lead
    props2 = numpy.array([0.1, 0.1, 0.1, 0.45, 0.15, 0.1])
    assert dict(assign.assign_read_indexes(cons, (props2, prob), haps, [["r"]], 1.2)) == {"hap1": {0}}
