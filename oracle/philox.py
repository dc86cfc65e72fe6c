"""Philox4x32-10 restated in numpy (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11).
TEST INFRASTRUCTURE: checks csrc/optim.hip's generator bit for bit.  Pinned by the Random123 known-answer vectors
(tests/test_oracle_golden.py::test_philox_kat)."""
import numpy as np

M0, M1 = 0xD2511F53, 0xCD9E8D57
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = 0xFFFFFFFF


def philox4x32_10(counter, key):
    """counter: 4 python ints (32 bit), key: 2 ints -> 4 ints"""
    c0, c1, c2, c3 = counter
    k0, k1 = key
    for _ in range(10):
        p0 = M0 * c0
        p1 = M1 * c2
        c0, c1, c2, c3 = ((p1 >> 32) ^ c1 ^ k0) & MASK, p1 & MASK, ((p0 >> 32) ^ c3 ^ k1) & MASK, p0 & MASK
        k0 = (k0 + W0) & MASK
        k1 = (k1 + W1) & MASK
    return [c0, c1, c2, c3]


def bits(seed, stream, offset, n4):
    """matches dg_philox_bits: counter = (offset + i, stream) as two 64-bit halves, key = seed"""
    out = np.empty(4 * n4, dtype=np.uint32)
    for i in range(n4):
        lo = (offset + i) & (2**64 - 1)
        r = philox4x32_10([lo & MASK, lo >> 32, stream & MASK, (stream >> 32) & MASK], [seed & MASK, (seed >> 32) & MASK])
        out[4 * i:4 * i + 4] = r
    return out


def uniform24(words):
    """kind 0 of dg_philox_fill: (bits >> 8) * 2^-24"""
    return (words >> 8).astype(np.float32) * np.float32(1.0 / 16777216.0)
