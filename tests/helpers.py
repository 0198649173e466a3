"""Shared helpers of the parity tests (oracle side = test infrastructure)."""
import numpy as np


def x_bits(words, n):
    return np.unpackbits(np.ascontiguousarray(words).view(np.uint8), bitorder="little")[:n]


def pack_bits(bits):
    n = len(bits)
    nwords = n // 64 + 1
    padded = np.zeros(nwords * 64, np.uint8)
    padded[:n] = bits
    return np.packbits(padded, bitorder="little").view(np.uint64).copy()


def random_x(n, p, seed):
    rs = np.random.RandomState(seed)
    return pack_bits((rs.random_sample(n) < p).astype(np.uint8))


def is_feasible(inst, words):
    x = x_bits(words, inst.n).astype(np.int64)
    v = np.add.reduceat(inst.factor.astype(np.int64) * x[inst.col], inst.row_ptr[:-1])
    return bool(np.all((inst.cmin <= v) & (v <= inst.cmax)))


def objective(inst, words):
    x = x_bits(words, inst.n).astype(np.float64)
    return float((inst.dense_costs() * x).sum() + inst.obj_constant)


def bits_equal(a, b):
    """Bitwise equality of two float64 arrays (NaN-safe, -0.0 != +0.0)."""
    a = np.ascontiguousarray(a, np.float64).view(np.uint64)
    b = np.ascontiguousarray(b, np.float64).view(np.uint64)
    return bool(np.array_equal(a, b))


def first_diff(a, b):
    a = np.ascontiguousarray(a, np.float64).view(np.uint64)
    b = np.ascontiguousarray(b, np.float64).view(np.uint64)
    d = np.flatnonzero(a != b)
    return None if d.size == 0 else int(d[0])
