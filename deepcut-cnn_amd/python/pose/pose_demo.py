#!/usr/bin/env python
"""Command-line demo with the reference's flags (python/pose/pose_demo.py:45-74): predict the pose of an
image (or of every image in a folder), write `<image>_pose.npz` (key `pose`, 5x14) and optionally a PNG
overlay.  Needs an MI355X: `--use_cpu` is refused, this package has no CPU forward path."""
import glob
import logging
import os
import sys

import click
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(_HERE))  # .../python: `caffe`, `pose`, `deepcut_tools`

from pose.estimate_pose import estimate_pose  # noqa: E402

LOG = logging.getLogger(__name__)
COLORS = [[255, 0, 0], [0, 255, 0], [0, 0, 255], [0, 245, 255], [255, 131, 250], [255, 255, 0]] * 2 + [[0, 0, 0], [255, 255, 255]]


def draw_disc(image, cx, cy, radius, color):
    h, w = image.shape[:2]
    yy, xx = np.ogrid[:h, :w]
    image[(xx - int(cx)) ** 2 + (yy - int(cy)) ** 2 <= radius ** 2] = color


@click.command()
@click.argument("image_name", type=click.Path(exists=True, dir_okay=True, readable=True))
@click.option("--out_name", type=click.Path(dir_okay=True, writable=True), default=None)
@click.option("--scales", type=click.STRING, default="1.")
@click.option("--visualize", type=click.BOOL, default=True)
@click.option("--folder_image_suffix", type=click.STRING, default=".png")
@click.option("--use_cpu", type=click.BOOL, is_flag=True, default=False)
@click.option("--gpu", type=click.INT, default=0)
@click.option("--tiling", type=click.Choice(["none", "exact", "reference"]), default="none",
              help="none: one forward per scale (default); exact / reference: see pose.estimate_pose")
@click.option("--model_def", default=None,
              help="model definition (.prototxt).  Default: the definition deepcut_tools.deepercut_prototxt(152) generates, which is "
                   "layer-for-layer the reference's models/deepercut/ResNet-152.prototxt (that directory is not part of this repository)")
@click.option("--model_bin", default=None, required=True,
              help="trained weights (.caffemodel / .h5): the reference fetches them with models/deepercut/download_models.sh; not shipped here")
def predict_pose_from(image_name, out_name, scales, visualize, folder_image_suffix, use_cpu, gpu, tiling, model_def, model_bin):
    import caffe
    from PIL import Image

    if model_def is None:
        import tempfile

        from deepcut_tools import deepercut_prototxt

        fd, model_def = tempfile.mkstemp(suffix=".prototxt")
        with os.fdopen(fd, "w") as f:
            f.write(deepercut_prototxt(152))

    scales = [float(v) for v in scales.split(",")]
    if os.path.isdir(image_name):
        images = sorted(glob.glob(os.path.join(image_name, "*" + folder_image_suffix)))
        folder = True
    else:
        images, folder = [image_name], False
    if use_cpu:
        caffe.set_mode_cpu()  # forward() will raise: there is no CPU path in libdeepcut_hip
    else:
        caffe.set_mode_gpu()
        caffe.set_device(gpu)
    if folder and out_name is not None and not os.path.exists(out_name):
        os.mkdir(out_name)
    for path in images:
        target = (path + "_pose.npz") if out_name is None else (
            os.path.join(out_name, os.path.basename(path) + "_pose.npz") if folder else out_name)
        img = np.asarray(Image.open(path))
        if img.ndim == 2:
            LOG.warning("The image is grayscale! This may deteriorate performance!")
            img = np.dstack((img, img, img))
        img = img[:, :, :3][:, :, ::-1]  # RGB -> BGR
        pose = estimate_pose(img, model_def, model_bin, scales, tiling=None if tiling == "none" else tiling)
        np.savez_compressed(target, pose=pose)
        if visualize and pose is not None:
            vis = img[:, :, ::-1].copy()
            for j in range(14):
                draw_disc(vis, pose[0, j], pose[1, j], 8, COLORS[j])
            Image.fromarray(vis).save(target + "_vis.png")


if __name__ == "__main__":
    logging.basicConfig(level=logging.INFO)
    predict_pose_from()  # pylint: disable=no-value-for-parameter
