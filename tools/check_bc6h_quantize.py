#!/usr/bin/env python3
"""Exhaustive check (CPU, numpy) that the integer form of the BC6H endpoint quantisation in csrc/bc6h_kernel.hip equals the
reference's binary32 sequence (QuantizeSingleEndpointElementUnsigned / Signed, BC67.cpp:2425-2445: a division rounded up,
a subtraction, a ceiling, a clamp) for every colour-space value the clamp lets through, and that the v_mul_hi_u32 constant
divides exactly.   python tools/check_bc6h_quantize.py"""
import numpy as np


def div_round_up(a, b):
    """a / b in binary32 rounded toward +inf: the round-to-nearest quotient corrected by the sign of the exact residual"""
    a32, b32 = np.float32(a), np.float32(b)
    q = (a32 / b32).astype(np.float32)
    r = a32.astype(np.float64) - q.astype(np.float64) * np.float64(b)
    return np.where(r > 0, np.nextafter(q, np.float32(np.inf)), q).astype(np.float32)


elem = np.arange(0, 31744, dtype=np.int64)
v = np.minimum(div_round_up((elem * 64).astype(np.float32), 31.0), np.float32(65535.0))
i = np.minimum(np.ceil((v - np.float32(32768.0)).astype(np.float32)).astype(np.int64), 32767)
expanded = (i & 0xffff) ^ 0x8000
c = (elem * 64 + 30) // 31
assert np.array_equal(expanded, c), "unsigned"
assert np.array_equal(((elem * 64 + 30) * 2216757579 >> 32) >> 4, c), "mul_hi, unsigned"
v = div_round_up((elem * 32).astype(np.float32), 31.0)
i = np.minimum(np.ceil(v).astype(np.int64), 32767)
c2 = (elem * 32 + 30) // 31
assert np.array_equal(i, np.minimum(c2, 32767)), "signed"
assert np.array_equal(((elem * 32 + 30) * 2216757579 >> 32) >> 4, c2), "mul_hi, signed"
print("OK: 31744 values, unsigned and signed")
