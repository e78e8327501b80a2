"""Deterministic synthetic data for fixtures, parity tests and the bench.

Everything here is exact integer arithmetic followed by one int->float32 conversion, so the same
call produces bit-identical arrays in this container, on the GPU box and inside the golden-vector
generator (tests/golden/make_golden.py).  That lets fixtures store only *outputs* of the reference:
the 27-70 MB of plane data, the pre-drawn random numbers and the synthetic RGB-D image are
regenerated on both sides instead of being committed.

The synthetic workload follows SURVEY.md section 8(d): depth image ~ U(0.5, 2.5) m, colour ~ U(0,1),
planes with std 0.01 (reference: src/ESLAM.py:201-210 draws N(0, 0.01^2); we use a uniform with the
same variance because it can be generated exactly), camera at the AABB centre with identity rotation.
"""
import numpy as np

_M1 = np.uint64(0x9E3779B97F4A7C15)
_M2 = np.uint64(0xD6E8FEB86659FD93)
_M3 = np.uint64(0xCA5A826395121157)
_S32 = np.uint64(32)


def hash_u24(n, stream):
    """n 24-bit integers, a fixed function of (index, stream)."""
    with np.errstate(over="ignore"):
        x = np.arange(n, dtype=np.uint64) + (np.uint64(stream) << np.uint64(40))
        x = (x + np.uint64(1)) * _M1
        x ^= x >> _S32
        x *= _M2
        x ^= x >> _S32
        x *= _M3
        x ^= x >> _S32
    return (x >> np.uint64(40)).astype(np.uint32)


def hash_uniform(shape, stream):
    """float32 array in [0, 1) (multiples of 2^-24), deterministic in (shape, stream)."""
    n = int(np.prod(shape)) if len(shape) else 1
    u = hash_u24(n, stream).astype(np.float32) * np.float32(2.0 ** -24)
    return u.reshape(shape)


def hash_randint(high, shape, stream):
    """int64 array in [0, high), deterministic; modulo bias is irrelevant for test data."""
    n = int(np.prod(shape)) if len(shape) else 1
    with np.errstate(over="ignore"):
        a = hash_u24(n, stream).astype(np.uint64)
        b = hash_u24(n, stream + 7919).astype(np.uint64)
        v = ((a << np.uint64(24)) | b) % np.uint64(high)
    return v.astype(np.int64).reshape(shape)


PLANE_STD = 0.01
_PLANE_HALF_WIDTH = np.float32(PLANE_STD * np.sqrt(3.0))


def plane_fill(shape, stream):
    """Feature-plane contents: uniform, zero mean, std 0.01; logical shape [1, C, h, w] (NCHW order)."""
    u = hash_uniform(shape, stream)
    return ((u - np.float32(0.5)) * np.float32(2.0)) * _PLANE_HALF_WIDTH


def depth_image(H, W, stream, zero_fraction=0.0):
    """Depth ~ U(0.5, 2.5) m; a `zero_fraction` of the pixels set to 0 (= missing depth)."""
    d = np.float32(0.5) + np.float32(2.0) * hash_uniform((H, W), stream)
    if zero_fraction > 0:
        m = hash_uniform((H, W), stream + 1) < np.float32(zero_fraction)
        d = np.where(m, np.float32(0), d).astype(np.float32)
    return d


def color_image(H, W, stream):
    return hash_uniform((H, W, 3), stream)
