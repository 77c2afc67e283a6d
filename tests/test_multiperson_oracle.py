"""CPU: the multi-person decoders of oracle/multiperson.py invert the reference's label ENCODING
(src/caffe/layers/pose_data_layer.cpp:686-802) — the only thing the reference defines for these maps."""
import numpy as np

from oracle import multiperson as M


def test_decoders_invert_the_training_encoding():
    rs = np.random.RandomState(0)
    J, E, h, w, scale = 3, 4, 9, 11, 0.85
    mean, std = rs.randn(E, 2) * 20, rs.uniform(5, 40, (E, 2))
    prob = np.zeros((J, h, w), np.float32)
    loc = np.zeros((2 * J, h, w), np.float32)
    nxt = np.zeros((2 * E, h, w), np.float32)
    truth = {}
    for j, cell in enumerate([(2, 3), (5, 9), (7, 1)]):
        joint = np.array([cell[1] * 8 + 4 + rs.uniform(-3, 3), cell[0] * 8 + 4 + rs.uniform(-3, 3)]) / scale
        nexts = rs.uniform(0, 80, (E, 2)) / scale
        prob[j][cell] = 0.9
        for l in range(E):
            lt, nt = M.encode_targets(joint, nexts[l], cell, scale, mean[l], std[l])
            loc[2 * j:2 * j + 2, cell[0], cell[1]] = lt
            nxt[2 * l:2 * l + 2, cell[0], cell[1]] = nt
        truth[j] = (cell, joint, nexts)
        counts, dets = M.nms_candidates(prob, loc, scale, 0.5, 1, 4)
        assert counts[j] == 1 and np.allclose(dets[j, 0, :2], joint, atol=1e-4) and tuple(dets[j, 0, 3:]) == cell
        got = M.pairwise_positions(nxt, [cell], scale, mean, std)[0]
        assert np.allclose(got, nexts, atol=1e-3)


def test_nms_rules():
    prob = np.zeros((1, 6, 6), np.float32)
    loc = np.zeros((2, 6, 6), np.float32)
    prob[0, 1, 1] = 0.8
    prob[0, 1, 2] = 0.8   # tie inside one window: the lower cell index wins
    prob[0, 4, 4] = 0.6
    prob[0, 4, 0] = 0.05  # below the threshold
    counts, dets = M.nms_candidates(prob, loc, 1.0, 0.1, 1, 8)
    assert counts[0] == 2
    assert [tuple(d[3:]) for d in dets[0, :2]] == [(1, 1), (4, 4)] and dets[0, 2, 3] == -1
    counts, _ = M.nms_candidates(prob, loc, 1.0, 0.1, 0, 8)  # radius 0: every cell above the threshold
    assert counts[0] == 3
    counts, dets = M.nms_candidates(prob, loc, 1.0, 0.1, 0, 2)  # max_det cuts the weakest
    assert counts[0] == 2 and dets[0, 1, 2] == np.float32(0.8)
