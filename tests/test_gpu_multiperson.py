"""-m gpu: dc_net_detect_parts / dc_net_decode_pairwise against oracle/multiperson.py on the maps
of a real forward.  The score threshold sits between values, so the candidate SETS must agree exactly; scores and cell
indices bit-for-bit, refined positions to double-precision rounding.

PARITY UNPINNED BY THE REFERENCE: eldar/deepcut-cnn has no consumer of `next_pred` and no part-candidate extraction (it stops at the maps,
SURVEY F6), so there is no reference output, test or golden vector to hold these kernels to.  The oracle restates the INVERSE of the label
encoding of the reference's training layer (src/caffe/layers/pose_data_layer.cpp:686-802), and that encoding is all it is pinned to
(tests/test_multiperson_oracle.py); what is proven here is that the device kernels compute exactly what that restatement computes."""
import numpy as np
import pytest

from conftest import rand_image
from oracle import multiperson as M

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def net(gpu_caffe, synth152):
    from deepcut_tools import deepercut_prototxt

    path, _ = synth152
    n = gpu_caffe.Net(deepercut_prototxt(152, 200, 264), path, gpu_caffe.TEST, from_text=True)
    n.forward_batch(rand_image(12, 200, 264, n=2), want=())
    return n


@pytest.mark.parametrize("radius,max_det,thr", [(1, 16, 0.5), (2, 8, 0.55), (0, 64, 0.62), (3, 4, 0.0)])
def test_part_candidates_match_oracle(net, radius, max_det, thr):
    prob, loc = net.blobs["prob"].data, net.blobs["loc_pred"].data
    for scale in (1.0, 0.75):
        counts, dets = net.detect_parts(scale, thr, radius, max_det)
        for b in range(prob.shape[0]):
            ref_counts, ref = M.nms_candidates(prob[b], loc[b], scale, thr, radius, max_det)
            assert np.array_equal(counts[b], ref_counts)
            assert np.array_equal(dets[b][:, :, 2:], ref[:, :, 2:])        # score, row, col: exact
            assert np.allclose(dets[b][:, :, :2], ref[:, :, :2], rtol=0, atol=1e-9)
    assert counts.sum() > 0


def test_pairwise_decode_matches_oracle(net):
    nxt = net.blobs["next_pred"].data
    rs = np.random.RandomState(0)
    E = nxt.shape[1] // 2
    h, w = nxt.shape[2:]
    cells = [(int(rs.randint(0, 2)), int(rs.randint(0, h)), int(rs.randint(0, w))) for _ in range(40)]
    mean, std = rs.randn(E, 2) * 15, rs.uniform(4, 30, (E, 2))
    for kw in ({}, {"mean": mean, "std": std}):
        got = net.decode_pairwise(np.array(cells), 1.3, **kw)
        assert got.shape == (40, E, 2)
        for b in (0, 1):
            idx = [i for i, c in enumerate(cells) if c[0] == b]
            ref = M.pairwise_positions(nxt[b], [cells[i][1:] for i in idx], 1.3, kw.get("mean"), kw.get("std"))
            assert np.allclose(got[idx], ref, rtol=0, atol=1e-9)
    out_of_map = net.decode_pairwise(np.array([[0, h, 0], [5, 0, 0]]), 1.0)  # refused cells come back as zeros
    assert not out_of_map.any()


def test_detections_feed_the_pairwise_decode(net):
    counts, dets = net.detect_parts(1.0, 0.5, 1, 4)
    trip = [(b, int(dets[b, j, k, 3]), int(dets[b, j, k, 4])) for b in range(dets.shape[0]) for j in range(dets.shape[1])
            for k in range(counts[b, j])]
    assert trip
    pos = net.decode_pairwise(np.array(trip), 1.0)
    assert np.isfinite(pos).all() and pos.shape == (len(trip), 182, 2)


def test_bad_arguments(net, gpu_caffe):
    for args in [(0.0, 0.5, 1, 4), (1.0, -0.1, 1, 4), (1.0, 0.5, -1, 4), (1.0, 0.5, 1, 0)]:
        with pytest.raises(gpu_caffe.DeepcutError):
            net.detect_parts(*args)


def test_more_local_maxima_than_fit_in_lds_is_still_deterministic(gpu_caffe, synth152):
    """threshold 0 / radius 0 makes every one of the 68 x 92 = 6256 cells of a 544x736 forward a candidate — more than the
    4096 keys the selection kernel sorts in LDS: the spill path must return the same (score desc, cell asc) prefix as the
    oracle, run after run (the round-1 kernel kept whichever 1024 candidates arrived first)."""
    from deepcut_tools import deepercut_prototxt

    path, _ = synth152
    n = gpu_caffe.Net(deepercut_prototxt(152, 544, 736), path, gpu_caffe.TEST, from_text=True)
    n.forward_batch(rand_image(13, 544, 736), want=())
    prob, loc = n.blobs["prob"].data, n.blobs["loc_pred"].data
    runs = [n.detect_parts(1.0, 0.0, 0, 96) for _ in range(3)]
    for counts, dets in runs[1:]:
        assert np.array_equal(counts, runs[0][0]) and np.array_equal(dets, runs[0][1])
    counts, dets = runs[0]
    assert (counts == 96).all()
    ref_counts, ref = M.nms_candidates(prob[0], loc[0], 1.0, 0.0, 0, 96)
    assert np.array_equal(counts[0], ref_counts)
    assert np.array_equal(dets[0][:, :, 2:], ref[:, :, 2:])
    assert np.allclose(dets[0][:, :, :2], ref[:, :, :2], rtol=0, atol=1e-9)
    # radius 1 at the same size: ~700 maxima per map, the LDS path, same contract
    counts1, dets1 = n.detect_parts(1.0, 0.0, 1, 4096)
    ref_counts1, ref1 = M.nms_candidates(prob[0], loc[0], 1.0, 0.0, 1, 4096)
    assert np.array_equal(counts1[0], ref_counts1) and counts1.max() < 4096
    assert np.array_equal(dets1[0][:, :, 2:], ref1[:, :, 2:])
