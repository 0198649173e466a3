"""bit_array_impl::set(int index, bool val) of the reference, lib/src/bit-array.hpp:144-149:

    auto mask = (k_one << b_offset(index));
    auto& ref = m_data[b_index(index)];
    ref ^= (underlying_type(val) ^ ref) & mask;

`underlying_type(val)` is 0 or 1, so the only bit `val` can contribute sits at offset 0
of a block: at offset 0 the bit becomes `val`, at every other offset
`(0-or-1 ^ ref) & mask == ref & mask` and the bit is cleared whatever `val` is.  The one
caller is the mutation loop of best_solution_recorder (itm-optimizer-common.hpp:523), so
this is what a mutated variable really gets.  The expected words below are that
expression evaluated by hand / in Python integers, not outputs of the oracle.
"""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle_ffi as O

MASK64 = (1 << 64) - 1


def reference_expression(word, offset, val):
    mask = 1 << offset
    return (word ^ ((int(bool(val)) ^ word) & mask)) & MASK64


@pytest.mark.parametrize("index", [0, 1, 5, 63, 64, 65, 127, 128, 191])
@pytest.mark.parametrize("val", [0, 1])
@pytest.mark.parametrize("fill", [0, MASK64, 0xA5A5A5A5A5A5A5A5])
def test_set_value_is_the_reference_expression(index, val, fill):
    nwords = index // 64 + 1
    words = np.full(nwords, fill, np.uint64)
    O.lib().orc_bitarray_set_value(words.ctypes.data_as(C.POINTER(C.c_uint64)), index, val)
    expect = [fill] * nwords
    expect[index // 64] = reference_expression(fill, index % 64, val)
    assert [int(w) for w in words] == expect


def test_truth_table_written_out():
    # offset 0: the bit takes val; any other offset: the bit is cleared
    assert reference_expression(0b0, 0, 1) == 0b1
    assert reference_expression(0b1, 0, 0) == 0b0
    assert reference_expression(0b1, 0, 1) == 0b1
    assert reference_expression(0b0000, 2, 1) == 0b0000  # val = 1 does NOT set bit 2
    assert reference_expression(0b0100, 2, 1) == 0b0000  # ... it clears it
    assert reference_expression(0b0100, 2, 0) == 0b0000
    assert reference_expression(0b1011, 2, 1) == 0b1011  # other bits untouched
