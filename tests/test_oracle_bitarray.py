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


# ---------------------------------------------------------------------------
# lib/test/lib.cpp:677-783 (check_bit_array), the reference's own assertions,
# against the oracle's BitArray and the host library's bit_array.
ONES = MASK64
MID = 32  # bit_per_block / 2


def _run(side, size, ops):
    ops_a = np.array(ops, np.int32).reshape(-1, 3)
    words = np.zeros(size // 64 + 1, np.uint64)
    gets = np.zeros(max(1, int((ops_a[:, 0] == 5).sum())), np.int32)
    ip = C.POINTER(C.c_int)
    if side == "oracle":
        fn = O.lib().orc_bitarray_script
        fn.argtypes = [C.c_int, ip, C.c_int, C.POINTER(C.c_uint64), ip]
        n = fn(size, ops_a.ctypes.data_as(ip), len(ops_a), words.ctypes.data_as(C.POINTER(C.c_uint64)),
               gets.ctypes.data_as(ip))
    else:
        from baryonyx_amd import host_ffi as H
        fn = H.lib().bxc_bitarray_script
        fn.argtypes = [C.c_int, ip, C.c_int, C.POINTER(C.c_ulonglong), ip]
        n = fn(size, ops_a.ctypes.data_as(ip), len(ops_a), words.ctypes.data_as(C.POINTER(C.c_ulonglong)),
               gets.ctypes.data_as(ip))
    assert n == size // 64 + 1  # :681-682 block_size() == 1 + size / bit_per_block
    return [int(w) for w in words], [int(g) for g in gets]


@pytest.mark.parametrize("side", ["oracle", "host"])
def test_check_bit_array_set_unset_invert(side):
    n = 100
    gets = [(5, i, 0) for i in range(n)]
    _, g = _run(side, n, [(0, 0, 0)] + gets)  # :684-686 zeros
    assert g == [0] * n
    _, g = _run(side, n, [(1, 0, 0)] + gets)  # :688-690 ones
    assert g == [1] * n
    _, g = _run(side, n, [(1, 0, 0)] + [(3, i, 0) for i in range(n)] + gets)  # :692-695 unset all
    assert g == [0] * n
    _, g = _run(side, n, [(2, i, 0) for i in range(n)] + gets)  # :697-700 set all
    assert g == [1] * n
    every_other = [(2, i, 0) for i in range(n)] + [(3, i, 0) for i in range(0, n, 2)]
    _, g = _run(side, n, every_other + gets)  # :702-705
    assert g == [0, 1] * (n // 2)
    _, g = _run(side, n, every_other + [(4, i, 0) for i in range(n)] + gets)  # :707-711 invert
    assert g == [1, 0] * (n // 2)


@pytest.mark.parametrize("side", ["oracle", "host"])
@pytest.mark.parametrize("frm,to,block0,block1", [
    (MID, 3 * MID, (ONES << MID) & ONES, ONES >> MID),  # :718-733
    (0, MID, ONES >> MID, ONES),                        # :735-748
    (MID, 4 * MID, (ONES << MID) & ONES, 0),            # :750-765
    (4 * MID - 1, 4 * MID, ONES, ONES - 1),             # :767-782
])
def test_check_bit_array_assign(side, frm, to, block0, block1):
    words, _ = _run(side, 128, [(1, 0, 0), (6, frm, to)])  # b.ones(); c.zeros(); b.assign(c, from, to)
    assert words[0] == block0 and words[1] == block1


@pytest.mark.parametrize("side", ["oracle", "host"])
def test_set_value_script_matches_the_expression(side):
    for index in (0, 5, 63, 64, 100):
        for val in (0, 1):
            words, _ = _run(side, 128, [(1, 0, 0), (8, index, val)])
            expect = [ONES, ONES, ONES]
            expect[index // 64] = reference_expression(ONES, index % 64, val)
            assert words == expect
