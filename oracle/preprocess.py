"""CPU ORACLE (test infrastructure, never on the product path) for the demo's image pre-processing,
python/pose/estimate_pose.py:83-103 of the reference:

    replicate the last row / column 64 times            (:89-95)
    scipy.misc.imresize(image, scale, interp='bilinear') (:96)
    astype(float32) - [104, 117, 123]                    (:97)
    paste into zeros(ceil(H*s/8)*8, ceil(W*s/8)*8, 3)    (:85-88, :99-103)

`scipy.misc.imresize` is third-party and gone from SciPy (>= 1.3); what it did for a uint8 HxWx3 array and a
float `size` was `PIL.Image.fromarray(a).resize((int(W*s), int(H*s)), BILINEAR)` (scipy/misc/pilutil.py of
SciPy <= 1.2: `toimage` keeps uint8 data as is, the new size is `(array(im.size) * size).astype(int)`).
The arithmetic is therefore Pillow's `ImagingResample` for 8-bit images, restated here from its published
algorithm (Pillow src/libImaging/Resample.c: `precompute_coeffs`, `normalize_coeffs_8bpc`,
`ImagingResampleHorizontal_8bpc` / `Vertical_8bpc`, bilinear filter with support 1):

  * separable, horizontal pass first, uint8 intermediate;
  * per output coordinate xx: centre = (xx + 0.5) * scale, support = max(scale, 1); taps
    [int(centre - support + 0.5), int(centre + support + 0.5)) clipped to the image; weight of tap x is
    max(0, 1 - |x - centre + 0.5| / max(scale, 1)), normalised to sum 1 in double precision;
  * weights are converted to 22-bit fixed point with round-half-away; a pixel is
    clip8((2^21 + sum(pixel * weight)) >> 22);
  * a pass whose input and output sizes agree is skipped; if both agree the image is returned unchanged.

PARITY PINNING: Pillow itself is the reference for this step and is present in the image (PIL 12.2), so
tests/test_preprocess.py compares this restatement bit-for-bit with `PIL.Image.resize` live and with
tests/golden/preprocess_golden.npz (made by tests/golden/make_preprocess_golden.py with that Pillow).
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2
MEAN_BGR = (104.0, 117.0, 123.0)
PAD = 64
STRIDE = 8


def bilinear_coeffs(in_size, out_size):
    """-> (bounds int32 [out,2] = (first tap, tap count), coeffs int32 [out, ksize]) — Resample.c
    precompute_coeffs + normalize_coeffs_8bpc for the bilinear filter and the full-image box."""
    scale = float(in_size) / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    coeffs = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        ws = []
        ww = 0.0
        for x in range(xmax):
            v = (x + xmin - center + 0.5) * ss
            if v < 0.0:
                v = -v
            w = 1.0 - v if v < 1.0 else 0.0
            ws.append(w)
            ww += w
        for x in range(xmax):
            w = ws[x] / ww if ww != 0.0 else ws[x]
            coeffs[xx, x] = int(-0.5 + w * (1 << PRECISION_BITS)) if w < 0 else int(0.5 + w * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, coeffs


def _resample_axis0(img, out_size):
    """8-bit resample along axis 0 of an (L, ...) uint8 array."""
    bounds, coeffs = bilinear_coeffs(img.shape[0], out_size)
    out = np.empty((out_size,) + img.shape[1:], np.uint8)
    src = img.astype(np.int64)
    for xx in range(out_size):
        lo, n = int(bounds[xx, 0]), int(bounds[xx, 1])
        k = coeffs[xx, :n].astype(np.int64).reshape((n,) + (1,) * (img.ndim - 1))
        acc = (1 << (PRECISION_BITS - 1)) + (src[lo:lo + n] * k).sum(axis=0)
        out[xx] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return out


def resize_bilinear_u8(image, size):
    """PIL.Image.fromarray(image).resize(size=(W', H'), BILINEAR) for a uint8 HxWxC image."""
    out_w, out_h = size
    h, w = image.shape[:2]
    img = image
    if out_w != w:
        img = _resample_axis0(img.transpose(1, 0, 2), out_w).transpose(1, 0, 2)
    if out_h != h:
        img = _resample_axis0(img, out_h)
    return img


def preprocess(image, scale):
    """HxWx3 BGR uint8 -> (out_h, out_w, 3) float32 network input (estimate_pose.py:83-103)."""
    image = np.asarray(image)
    h, w = image.shape[:2]
    out_w = int(np.ceil(float(w) * scale / STRIDE) * STRIDE)
    out_h = int(np.ceil(float(h) * scale / STRIDE) * STRIDE)
    padded = np.pad(image, ((0, PAD), (0, PAD), (0, 0)), mode="edge")
    new_w, new_h = int(padded.shape[1] * scale), int(padded.shape[0] * scale)
    scaled = resize_bilinear_u8(padded, (new_w, new_h))
    scaled = scaled.astype("float32") - np.array(MEAN_BGR)
    canvas = np.zeros((out_h, out_w, 3), np.float32)
    hh, ww = min(out_h, scaled.shape[0]), min(out_w, scaled.shape[1])
    canvas[:hh, :ww] = scaled[:hh, :ww]
    return canvas
