"""Identities of correctly rounded binary32 arithmetic that the device code relies on to do less work for the same
bits (cudaraytracer_amd/csrc/crt_device.hpp: normalize3_len, div_nr).  The device checks them over all 2^32 bit
patterns in crt_selftest_rcp (GPU suite); here the same statements on the host, where numpy's float32 division is
the IEEE one."""
import numpy as np


def _all_significands():
    # every binary32 value in [1, 4): two binades cover every significand on both sides of the reciprocal
    return np.arange(0x3F800000, 0x40800000, dtype=np.uint32).view(np.float32)


def test_third_reciprocal_returns_the_first():
    """vec::normalize multiplies by 1/len with len = 1/(1/sqrt(len^2)) (vec.hpp:398-422): 1/(1/(1/s)) == 1/s for
    every s, so one reciprocal of the square root is enough; scaling by a power of two is exact, so [1, 4) stands
    for every s whose chain stays in the normal range."""
    s = _all_significands()
    one = np.float32(1.0)
    r1 = one / s
    r3 = one / (one / r1)
    assert np.array_equal(r1.view(np.uint32), r3.view(np.uint32))
    # ... while the second reciprocal does not return s itself: the length keeps both of its reciprocals
    assert np.count_nonzero((one / r1).view(np.uint32) != s.view(np.uint32)) > 1_000_000


def test_scaling_does_not_matter():
    s = _all_significands()[:: 4099]
    one = np.float32(1.0)
    for e in (-60, -13, 7, 40):
        t = np.ldexp(s, e).astype(np.float32)
        r1 = one / t
        assert np.array_equal(r1.view(np.uint32), (one / (one / r1)).view(np.uint32))
