"""ORACLE — test infrastructure only (never imported by the product path).

CPU restatement of the squash-resize the reference's preprocessing performs on PIL images:
``transforms.get_image_transform(336)`` (``core_system.py:200``) applied at ``:335`` / ``:439`` is
``Resize((S, S), BILINEAR)`` on a PIL RGB image, i.e. Pillow's 8-bit separable resample with a
triangle filter whose support grows with the down-scale factor (antialiasing), fixed-point
coefficients (22 fractional bits) and a uint8 intermediate after the horizontal pass.  The crop
variant is ``img.crop(bbox)`` followed by the same resize (the idea left commented out at
``core_system.py:687-690``).

Pinned: Pillow itself is installed in this image, so ``tests/test_oracle.py`` checks this
restatement bit-for-bit against ``PIL.Image.resize`` (the library the reference calls) on seeded
images; the device kernel is then checked against both.
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def precompute_coeffs(in_size, out_size):
    """Integer taps of one axis: bounds [out,2] (first tap, tap count) and coeffs [out, ksize]."""
    scale = float(np.float32(in_size) - np.float32(0)) / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    coeffs = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        xmin = max(xmin, 0)
        xmax = int(center + support + 0.5)
        xmax = min(xmax, in_size)
        xmax -= xmin
        w = np.zeros(ksize, dtype=np.float64)
        ww = 0.0
        for x in range(xmax):
            a = (x + xmin - center + 0.5) * ss
            a = -a if a < 0.0 else a
            v = 1.0 - a if a < 1.0 else 0.0
            w[x] = v
            ww += v
        if ww != 0.0:
            w[:xmax] = w[:xmax] / ww
        for x in range(ksize):
            k = w[x] * (1 << PRECISION_BITS)
            coeffs[xx, x] = int(-0.5 + k) if w[x] < 0 else int(0.5 + k)
        bounds[xx] = (xmin, xmax)
    return bounds, coeffs


def _pass(img, bounds, coeffs, axis):
    """img uint8 [H, W, C]; resample along `axis` (0 = vertical, 1 = horizontal)."""
    src = img.astype(np.int64)
    out_size = bounds.shape[0]
    shape = list(img.shape)
    shape[axis] = out_size
    out = np.empty(shape, dtype=np.uint8)
    for o in range(out_size):
        lo, n = int(bounds[o, 0]), int(bounds[o, 1])
        k = coeffs[o, :n].astype(np.int64)
        if axis == 1:
            acc = (src[:, lo:lo + n, :] * k[None, :, None]).sum(axis=1)
        else:
            acc = (src[lo:lo + n, :, :] * k[:, None, None]).sum(axis=0)
        acc = (acc + (1 << (PRECISION_BITS - 1))) >> PRECISION_BITS
        val = np.clip(acc, 0, 255).astype(np.uint8)
        if axis == 1:
            out[:, o, :] = val
        else:
            out[o, :, :] = val
    return out


def crop_resize_u8(img, size, box=None):
    """img: uint8 [H, W, 3]; box (x0, y0, x1, y1) half-open or None.  Returns uint8 [size, size, 3]
    equal to ``Image.fromarray(img).crop(box).resize((size, size), Image.BILINEAR)``."""
    if box is not None:
        x0, y0, x1, y1 = box
        img = img[y0:y1, x0:x1]
    h, w = img.shape[:2]
    bh, kh = precompute_coeffs(w, size)
    bv, kv = precompute_coeffs(h, size)
    tmp = _pass(img, bh, kh, axis=1)
    return _pass(tmp, bv, kv, axis=0)
