"""-m gpu: the demo command line (pose/pose_demo.py, the counterpart of the reference's python/pose/pose_demo.py) end to
end in a subprocess: image file -> .npz with a 5x14 `pose`, with and without tiling; --use_cpu must fail loudly."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEMO = os.path.join(ROOT, "deepcut-cnn_amd", "python", "pose", "pose_demo.py")


@pytest.fixture(scope="module")
def model(tmp_path_factory, synth152):
    from deepcut_tools import deepercut_prototxt

    d = tmp_path_factory.mktemp("demo")
    proto = d / "net.prototxt"
    proto.write_text(deepercut_prototxt(152, 688, 688))
    return str(proto), synth152[0], d


def _run(args):
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(ROOT, "deepcut-cnn_amd", "python"), ROOT]))
    return subprocess.run([sys.executable, DEMO] + args, env=env, capture_output=True, text=True, timeout=300)


@pytest.mark.parametrize("hw,extra", [((240, 320), ["--scales", "0.75,1.0"]), ((720, 960), ["--tiling", "exact"]),
                                      ((96, 128), ["--default-def"])])
def test_demo_writes_a_pose(model, hw, extra):
    from PIL import Image

    proto, weights, d = model
    img = d / ("img_%dx%d.png" % hw)
    Image.fromarray(np.random.RandomState(hw[0]).randint(0, 256, hw + (3,)).astype(np.uint8)).save(str(img))
    out = str(d / ("pose_%d.npz" % hw[0]))
    use_default_def = "--default-def" in extra  # no --model_def: the generated ResNet-152 definition
    extra = [e for e in extra if e != "--default-def"]
    r = _run([str(img)] + ([] if use_default_def else ["--model_def", proto]) + ["--model_bin", weights, "--out_name", out, "--visualize", "False"] + extra)
    assert r.returncode == 0, r.stderr[-2000:]
    pose = np.load(out, allow_pickle=True)["pose"]
    assert pose.shape == (5, 14) and np.isfinite(pose.astype(np.float64)).all()


def test_use_cpu_is_refused(model):
    from PIL import Image

    proto, weights, d = model
    img = d / "small.png"
    Image.fromarray(np.zeros((64, 64, 3), np.uint8)).save(str(img))
    r = _run([str(img), "--model_def", proto, "--model_bin", weights, "--use_cpu"])
    assert r.returncode != 0 and "MI355X path only" in r.stderr
