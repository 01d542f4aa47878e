"""ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle/rle_ref.c for the rules and for the
"parity unpinned" statement).

ctypes front-end of the C restatement of the polygon -> RLE step, shaped like the reference's
call sites (core/evaluation/coco_evaluation.py:311-318: `get_polygon_rles(polygons,
image_shape)` -> list of {"size": [h, w], "counts": bytes}).
"""
import ctypes
import os

import numpy as np

from . import extreme_utils_ref as _cref

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "librle_ref.so")
_lib = None


def build(force=False):
    src = os.path.join(_HERE, "rle_ref.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        _cref.build(force=True)          # the Makefile builds both libraries
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        h = ctypes.CDLL(_SO)
        h.ref_rle_fr_poly.restype = ctypes.c_long
        h.ref_rle_fr_poly.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_long, ctypes.c_long,
                                      ctypes.c_void_p, ctypes.c_long]
        h.ref_rle_to_string.restype = ctypes.c_long
        h.ref_rle_to_string.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p]
        h.ref_rle_decode.restype = None
        h.ref_rle_decode.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_long]
        h.ref_rle_area.restype = ctypes.c_uint32
        h.ref_rle_area.argtypes = [ctypes.c_void_p, ctypes.c_long]
        _lib = h
    return _lib


def fr_poly(poly, h, w):
    """poly: (V,2) or (2V,) array of x,y (float32 values are widened exactly, as `.tolist()`
    does in the reference) -> uint32 run lengths (column-major, first run = zeros)."""
    xy = np.ascontiguousarray(np.asarray(poly, dtype=np.float64).reshape(-1))
    k = xy.size // 2
    cap = 64
    while True:
        out = np.zeros((cap,), dtype=np.uint32)
        m = lib().ref_rle_fr_poly(xy.ctypes.data, k, int(h), int(w), out.ctypes.data, cap)
        if m <= cap:
            return out[:m].copy()
        cap = int(m)


def to_string(counts):
    counts = np.ascontiguousarray(counts, dtype=np.uint32)
    buf = ctypes.create_string_buffer(7 * counts.size + 1)
    n = lib().ref_rle_to_string(counts.ctypes.data, counts.size, buf)
    return buf.raw[:n]


def decode(counts, h, w):
    """(h, w) uint8 mask (the library returns it in Fortran order; same values)."""
    counts = np.ascontiguousarray(counts, dtype=np.uint32)
    mask = np.zeros((w, h), dtype=np.uint8)
    lib().ref_rle_decode(counts.ctypes.data, counts.size, mask.ctypes.data, int(h) * int(w))
    return mask.T


def area(counts):
    counts = np.ascontiguousarray(counts, dtype=np.uint32)
    return int(lib().ref_rle_area(counts.ctypes.data, counts.size))


def get_polygon_rles(polygons, image_shape):
    """coco_evaluation.py:311-318 on an (N, 2V) array."""
    h, w = image_shape
    return [{"size": [int(h), int(w)], "counts": to_string(fr_poly(p, h, w))}
            for p in np.asarray(polygons)]
