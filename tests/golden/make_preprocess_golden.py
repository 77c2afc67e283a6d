"""Generate tests/golden/preprocess_golden.npz: the reference's image pre-processing
(python/pose/estimate_pose.py:83-103) executed with the real Pillow of the build container.

`scipy.misc.imresize` no longer exists (SciPy >= 1.3); for a uint8 HxWx3 array and a float factor it was
    PIL.Image.fromarray(a).resize(tuple((array(im.size) * factor).astype(int)), BILINEAR)
(scipy/misc/pilutil.py, SciPy <= 1.2), which is what `imresize` below does.  Everything else is the
reference's own sequence of NumPy calls, re-typed here because `estimate_pose()` cannot run without a network.
Stored: the uint8 inputs, the scale, the float32 canvases, and the Pillow version that produced them.

    python tests/golden/make_preprocess_golden.py
"""
import os

import numpy as np
import PIL
from PIL import Image

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "preprocess_golden.npz")
MEAN = np.array([104., 117., 123.])
STRIDE = 8

CASES = [((61, 83), 1.0), ((61, 83), 0.5), ((61, 83), 0.75), ((61, 83), 1.25), ((48, 120), 0.33), ((40, 56), 1.7),
         ((30, 200), 0.05), ((17, 9), 2.5), ((64, 64), 0.9999), ((100, 36), 1.0)]


def imresize(arr, factor):
    im = Image.fromarray(arr)
    size = tuple((np.array(im.size) * factor).astype(int))
    return np.asarray(im.resize(size, Image.BILINEAR))


def reference_preprocess(image, scale_factor):
    im_bg_width = int(np.ceil(float(image.shape[1]) * scale_factor / STRIDE) * STRIDE)
    im_bg_height = int(np.ceil(float(image.shape[0]) * scale_factor / STRIDE) * STRIDE)
    pad_size = 64
    image = np.vstack((image, np.tile(image[-1:, :, :], (pad_size, 1, 1))))
    image = np.hstack((image, np.tile(image[:, -1:, :], (1, pad_size, 1))))
    image = imresize(image, scale_factor)
    image = image.astype('float32') - MEAN
    net_input = np.zeros((im_bg_height, im_bg_width, 3), dtype='float32')
    hh, ww = min(net_input.shape[0], image.shape[0]), min(net_input.shape[1], image.shape[1])
    net_input[:hh, :ww, :] = image[:hh, :ww, :]
    return net_input


def make_image(i, hw):
    rs = np.random.RandomState(900 + i)
    img = rs.randint(0, 256, hw + (3,)).astype(np.uint8)
    if i % 3 == 2:  # saturated blocks: exercises the clip and the rounding at 0 / 255
        img[: hw[0] // 2] = (rs.randint(0, 2, (hw[0] // 2, hw[1], 3)) * 255).astype(np.uint8)
    return img


def main():
    data = {"pillow_version": np.array(PIL.__version__), "n": np.array(len(CASES))}
    for i, (hw, s) in enumerate(CASES):
        img = make_image(i, hw)
        data["image_%d" % i] = img
        data["scale_%d" % i] = np.float64(s)
        data["canvas_%d" % i] = reference_preprocess(img, s)
        print(i, hw, s, "->", data["canvas_%d" % i].shape)
    np.savez_compressed(OUT, **data)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
