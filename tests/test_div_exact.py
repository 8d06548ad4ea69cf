"""CPU: the three-instruction division by 255 the tile kernels use for UNORM8 texel / framebuffer decoding
(csrc/tile_common.h unorm8f) equals the IEEE binary32 division SPEC.md spells, for EVERY byte; the same holds for the
SNORM8 / SNORM16 divisors (checked here too, though the geometry kernel keeps the plain division: it measured faster).  The fused multiply-adds are evaluated exactly with rationals and rounded once, to nearest
even, like v_fma_f32."""
from fractions import Fraction

import numpy as np


def _rn32(fr: Fraction) -> np.float32:
    if fr == 0:
        return np.float32(0.0)
    x = np.float32(float(fr))
    best = None
    for c in (x, np.nextafter(x, np.float32(np.inf)), np.nextafter(x, np.float32(-np.inf))):
        d = abs(Fraction(float(c)) - fr)
        even = (int(np.float32(c).view(np.uint32)) & 1) == 0
        if best is None or d < best[0] or (d == best[0] and even and not best[2]):
            best = (d, c, even)
    return np.float32(best[1])


def _fma32(a, b, c) -> np.float32:
    return _rn32(Fraction(float(a)) * Fraction(float(b)) + Fraction(float(c)))


def _check(d: float, lo: int, hi: int, r_bits: int):
    dd = np.float32(d)
    r = np.array([r_bits], dtype=np.uint32).view(np.float32)[0]
    assert r == np.float32(1.0) / dd  # the kernels' constant is RN(1 / d)
    plain_wrong = 0
    for x in range(lo, hi + 1):
        xf = np.float32(x)
        want = xf / dd
        q0 = xf * r
        q1 = _fma32(_fma32(-q0, dd, xf), r, q0)
        assert q1.view(np.uint32) == want.view(np.uint32), (d, x, want, q1)
        plain_wrong += int(q0.view(np.uint32) != want.view(np.uint32))
    return plain_wrong


def test_unorm8_and_snorm8_division_sequences_are_exact():
    assert _check(255.0, 0, 255, 0x3B808081) > 0  # the bare product is NOT enough (126 of 256 values differ)
    assert _check(127.0, -128, 127, 0x3C010204) > 0


def test_snorm16_division_sequence_is_exact():
    _check(32767.0, -32768, 32767, 0x38000100)
