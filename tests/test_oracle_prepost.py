"""CPU: hand-derived known answers for the pre / post-processing oracle (oracle/inferencer_ref.py).  The third-party
code it restates (cv2, mmcv, mmdet, torchvision) is not installed here -- parity unpinned, see the oracle's header --
so the checks below are derived by hand from the published formulas."""
import numpy as np

import inferencer_ref as R


def test_rescale_size_follows_mmcv_imrescale():
    # 480x640 (h, w) into (1152, 768): factor = min(1152/640, 768/480) = 1.6 -> 768 x 1024
    assert R.rescale_size(480, 640, (1152, 768))[:2] == (768, 1024)
    # portrait 640x480: same factor, the result does not fit the landscape pad (config comment: "assume horizontal")
    assert R.rescale_size(640, 480, (1152, 768))[:2] == (1024, 768)
    # rounding: int(x + 0.5)
    assert R.rescale_size(427, 640, (1333, 800))[:2] == (int(427 * 1.873536299765808 + 0.5), 1199)


def test_bilinear_identity_and_upsample_by_two():
    img = np.arange(2 * 3 * 3, dtype=np.uint8).reshape(2, 3, 3) * 10
    assert np.array_equal(R.resize_bilinear_u8(img, 2, 3), img)  # same size: coefficients are exactly 0
    # 1x2 image [0, 100] -> width 4: source x = (d + .5)/2 - .5 = -0.25, 0.25, 0.75, 1.25 -> clamp, 25, 75, clamp
    row = np.array([[[0, 0, 0], [100, 100, 100]]], np.uint8)
    out = R.resize_bilinear_u8(row, 1, 4)[0, :, 0]
    assert out.tolist() == [0, 25, 75, 100]
    # vertical: the same along y
    col = row.transpose(1, 0, 2)
    assert R.resize_bilinear_u8(col, 4, 1)[:, 0, 0].tolist() == [0, 25, 75, 100]


def test_bilinear_downsample_by_two_averages_pairs():
    # width 4 -> 2: source x = 0.5, 2.5 -> mean of (p0, p1) and (p2, p3); fixed point: (1024*(a+b)*2048 + 2^21) >> 22
    row = np.array([[[10] * 3, [20] * 3, [40] * 3, [90] * 3]], np.uint8)
    assert R.resize_bilinear_u8(row, 1, 2)[0, :, 0].tolist() == [15, 65]
    # rounding to nearest with ties up: (0 + 1) / 2 = 0.5 -> (1024*1*2048 + 2^21) >> 22 = 1
    row = np.array([[[0] * 3, [1] * 3]], np.uint8)
    assert R.resize_bilinear_u8(row, 1, 1)[0, 0, 0] == 1


def test_preprocess_pads_before_normalising():
    img = np.full((4, 8, 3), 200, np.uint8)
    x, m, meta = R.preprocess(img, scale=(8, 4), pad_size=(12, 6), mean=(100, 110, 120), std=(50, 10, 2),
                              pad_val=(114, 114, 114))
    assert x.shape == (3, 6, 12) and m.shape == (6, 12) and meta["img_unpadded_shape"] == (4, 8)
    assert np.allclose(x[:, :4, :8].reshape(3, -1)[:, 0], [(200 - 100) / 50, (200 - 110) / 10, (200 - 120) / 2])
    assert np.allclose(x[:, 5, 11], [(114 - 100) / 50, (114 - 110) / 10, (114 - 120) / 2])  # the pad value is normalised too
    assert m[:4, :8].sum() == 0 and m.sum() == 6 * 12 - 32
    assert meta["scale_factor"] == (1.0, 1.0)


def test_batched_nms_is_per_class_and_greedy():
    boxes = np.array([[0, 0, 10, 10],      # A score .9 class 0
                      [1, 1, 11, 11],      # B .8 class 0, IoU(A,B) = 81/119 = .68
                      [0, 0, 10, 10],      # C .7 class 1 (identical to A, other class -> kept)
                      [2, 2, 12, 12],      # D .6 class 0, IoU(A,D) = 64/136 = .47, IoU(B,D) = .68
                      [50, 50, 60, 60]], np.float32)
    scores = np.array([.9, .8, .7, .6, .5], np.float32)
    labels = np.array([0, 0, 1, 0, 0])
    assert R.batched_nms(boxes, scores, labels, 0.5).tolist() == [0, 2, 3, 4]   # B dies by A; D survives A (.47 < .5), B is dead
    assert R.batched_nms(boxes, scores, labels, 0.7).tolist() == [0, 1, 2, 3, 4]
    assert R.batched_nms(boxes, scores, labels, 0.4).tolist() == [0, 2, 4]
    # ties in score keep the input order (stable sort)
    assert R.batched_nms(boxes[[0, 0]], np.array([.5, .5], np.float32), np.array([3, 3]), 0.5).tolist() == [0]
    # iou == thr does not suppress (strict >): two boxes with IoU exactly 1/3
    b = np.array([[0, 0, 2, 1], [1, 0, 3, 1]], np.float32)
    assert R.batched_nms(b, np.array([.9, .8], np.float32), np.array([0, 0]), 1 / 3).tolist() == [0, 1]


def test_postprocess_threshold_nms_rescale():
    boxes = np.array([[0, 0, 16, 8], [1, 0, 17, 8], [100, 100, 120, 140]], np.float32)
    scores = np.array([.9, .85, .05], np.float32)
    labels = np.array([7, 7, 7])
    b, s, l = R.postprocess(boxes, scores, labels, score_threshold=0.1, iou_threshold=0.8, scale_factor=(1.6, 0.8))
    assert s.tolist() == [np.float32(.9)] and l.tolist() == [7]          # second box: IoU 15/17 = .88 > .8; third below thr
    assert np.allclose(b, [[0, 0, 10, 10]])
