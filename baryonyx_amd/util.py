"""Small host-side helpers shared by the driver, bench.py and the tests."""
import numpy as np


def x_bits(words, n):
    """bit_array blocks -> 0/1 vector (bit j = block j/64, offset j%64;
    lib/src/bit-array.hpp:60-66,317-325)."""
    return np.unpackbits(np.ascontiguousarray(words, np.uint64).view(np.uint8), bitorder="little")[:n]


def pack_bits(bits):
    n = len(bits)
    padded = np.zeros((n // 64 + 1) * 64, np.uint8)
    padded[:n] = bits
    return np.packbits(padded, bitorder="little").view(np.uint64).copy()


def is_feasible(inst, words):
    """is_valid_solution on the merged constraints."""
    x = x_bits(words, inst.n).astype(np.int64)
    v = np.add.reduceat(inst.factor.astype(np.int64) * x[inst.col], inst.row_ptr[:-1])
    return bool(np.all((inst.cmin <= v) & (v <= inst.cmax)))
