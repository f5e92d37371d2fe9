import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def same_bits(a, b):
    """bit-for-bit equality of two f32 arrays (NaNs with equal payload compare equal)"""
    return np.array_equal(bits(a), bits(b))


def ulp_diff(a, b):
    """max distance in units in the last place between two finite f32 arrays"""
    ia = bits(a).astype(np.int64)
    ib = bits(b).astype(np.int64)
    ia = np.where(ia & 0x80000000, 0x80000000 - ia, ia)
    ib = np.where(ib & 0x80000000, 0x80000000 - ib, ib)
    return np.abs(ia - ib)


def psnr(a, b, peak=255.0):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    mse = np.mean((a - b) ** 2)
    return float("inf") if mse == 0 else 10.0 * np.log10(peak * peak / mse)


def same_values(a, b):
    """bit-for-bit equality except that any NaN equals any NaN (x86 and gfx950 produce different
    default-NaN payloads for inf*0)"""
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    return bool(((bits(a) == bits(b)) | (np.isnan(a) & np.isnan(b))).all())
