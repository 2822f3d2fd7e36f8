"""TEST INFRASTRUCTURE (never imported by the product path).  numpy restatement of the library's counter-based Gaussian
noise (include/fdbm_hip.h, "Gaussian noise on the device"; csrc/elementwise.hip: philox4x32_10 / rng_complex_normal):
the stand-in for the reference's `torch.randn_like(complex state)` draws (fdbm/bridge.py:47,108, fdbm/util/predictors.py:46,
fdbm/util/correctors.py:48,76) when no noise tensors are injected.

Philox4x32-10 is the generator of Salmon et al., "Parallel random numbers: as easy as 1, 2, 3" (SC'11), Random123 v1.x -
a third-party algorithm that is not part of /root/reference; it is pinned here by Random123's published known-answer
vectors (KAT, below) and the device kernel is pinned against this file (tests/test_hip_ops.py::test_device_rng_*)."""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
MAGIC = 0x46444D42

# (counter, key) -> output: two of Random123's published known-answer vectors for philox4x32-10 (all-zero input; the
# pi-digits input)
KAT = [
    ((0x00000000, 0x00000000, 0x00000000, 0x00000000), (0x00000000, 0x00000000),
     (0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8)),
    ((0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344), (0xA4093822, 0x299F31D0),
     (0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1)),
]


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Ten rounds on uint32 arrays (broadcast); returns the four output words."""
    c0, c1, c2, c3 = (np.asarray(v, dtype=np.uint64) & np.uint64(0xFFFFFFFF) for v in (c0, c1, c2, c3))
    k0, k1 = int(k0) & 0xFFFFFFFF, int(k1) & 0xFFFFFFFF
    mask = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0, p1 = M0 * c0, M1 * c2                       # 32 x 32 -> 64 bit products
        hi0, lo0, hi1, lo1 = p0 >> np.uint64(32), p0 & mask, p1 >> np.uint64(32), p1 & mask
        c0, c1, c2, c3 = hi1 ^ c1 ^ np.uint64(k0), lo1, hi0 ^ c3 ^ np.uint64(k1), lo0
        k0, k1 = (k0 + W0) & 0xFFFFFFFF, (k1 + W1) & 0xFFFFFFFF
    return tuple(v.astype(np.uint32) for v in (c0, c1, c2, c3))


def complex_normal(n, draw, seed, first=0):
    """Elements first .. first + n - 1 of draw `draw` under `seed` (a 64-bit integer): complex64 [n], Re, Im ~ N(0, 1/2)."""
    e = np.arange(first, first + n, dtype=np.uint64)
    x0, x1, _, _ = philox4x32_10(e & np.uint64(0xFFFFFFFF), e >> np.uint64(32), np.uint64(draw), np.uint64(MAGIC),
                                 seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    two24 = np.float32(5.9604644775390625e-08)
    u1 = ((x0 >> np.uint32(8)).astype(np.float32) + np.float32(0.5)) * two24
    u2 = ((x1 >> np.uint32(8)).astype(np.float32) + np.float32(0.5)) * two24
    r = np.sqrt(np.float32(-2.0) * np.log(u1), dtype=np.float32)
    th = np.float32(6.283185307179586) * u2
    h = np.float32(0.7071067811865476)
    return ((r * np.cos(th, dtype=np.float32)) * h + 1j * ((r * np.sin(th, dtype=np.float32)) * h)).astype(np.complex64)
