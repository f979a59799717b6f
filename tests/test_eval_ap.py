"""tools/eval_ap.py (COCO-style AP, own implementation) against hand-computed cases."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import eval_ap  # noqa: E402


def test_iou():
    iou = eval_ap.box_iou([[0, 0, 10, 10]], [[0, 0, 10, 10], [5, 0, 15, 10], [20, 20, 30, 30]])
    np.testing.assert_allclose(iou, [[1.0, 50 / 150, 0.0]])


def test_hand_computed_precision_recall():
    # one class, two ground-truth boxes; detections by score: TP, FP, TP
    # recall 0.5 reached at precision 1, recall 1.0 at precision 2/3 (envelope): 51 recall points at 1, 50 at 2/3
    g = [dict(boxes=[[0, 0, 10, 10], [20, 20, 30, 30]], labels=[0, 0])]
    d = [dict(boxes=[[0, 0, 10, 10], [50, 50, 60, 60], [20, 20, 30, 30]], scores=[.9, .8, .7], labels=[0, 0, 0])]
    r = eval_ap.coco_ap(d, g)
    assert abs(r["AP"] - (51 + 50 * 2 / 3) / 101) < 1e-12 and r["AP50"] == r["AP75"] == r["AP"]


def test_iou_thresholds_and_classes():
    # a detection shifted by 1 of 10 pixels: IoU = 9 / 11 = 0.818 -> a hit at thresholds .5 ... .8 (7 of 10)
    g = [dict(boxes=[[0, 0, 10, 10]], labels=[3]), dict(boxes=[[0, 0, 10, 10]], labels=[5])]
    d = [dict(boxes=[[1, 0, 11, 10]], scores=[0.9], labels=[3]), dict(boxes=[[0, 0, 10, 10]], scores=[0.5], labels=[4])]
    r = eval_ap.coco_ap(d, g)
    assert abs(r["per_class"][3] - 0.7) < 1e-12 and r["per_class"][5] == 0.0      # class 5 never detected, class 4 has no GT
    assert abs(r["AP"] - 0.35) < 1e-12 and r["AP50"] == 0.5 and r["AP75"] == 0.5


def test_one_match_per_ground_truth_and_max_dets():
    g = [dict(boxes=[[0, 0, 10, 10]], labels=[0])]
    d = [dict(boxes=[[0, 0, 10, 10], [0, 0, 10, 10]], scores=[0.9, 0.8], labels=[0, 0])]   # the duplicate is a false positive
    assert eval_ap.coco_ap(d, g)["AP"] == 1.0      # ... after full recall: does not lower the envelope
    d2 = [dict(boxes=[[50, 50, 60, 60], [0, 0, 10, 10]], scores=[0.9, 0.8], labels=[0, 0])]
    assert abs(eval_ap.coco_ap(d2, g)["AP"] - 0.5) < 1e-12
    assert eval_ap.coco_ap(d2, g, max_dets=1)["AP"] == 0.0
    # non-finite scores (padded proposals on random weights) are dropped, not ranked
    d3 = [dict(boxes=[[0, 0, 10, 10], [0, 0, 10, 10]], scores=[float("nan"), 0.8], labels=[0, 0])]
    assert eval_ap.coco_ap(d3, g)["AP"] == 1.0


def test_proxy_ground_truth_from_detections():
    ref = [dict(boxes=np.array([[0, 0, 10, 10], [5, 5, 9, 9.]]), scores=np.array([0.9, 0.2]), labels=np.array([1, 2]))]
    gts = eval_ap.detections_as_ground_truth(ref, 0.5)
    assert len(gts[0]["labels"]) == 1 and eval_ap.coco_ap(ref, gts)["AP"] == 1.0
    many = [dict(boxes=np.arange(40.).reshape(10, 4), scores=np.linspace(0.9, 0.5, 10), labels=np.zeros(10, int))]
    top3 = eval_ap.detections_as_ground_truth(many, 0.0, top=3)
    assert len(top3[0]["labels"]) == 3 and np.array_equal(top3[0]["boxes"], many[0]["boxes"][:3])


def test_ignore_regions_absorb_detections_like_coco_crowds():
    """a detection inside a crowd region is neither a true nor a false positive (overlap = intersection over ITS area, any
    number of detections per region); the region itself is not ground truth"""
    import eval_ap

    gt = [dict(boxes=[[0, 0, 10, 10], [100, 100, 300, 300]], labels=[1, 1], ignore=[False, True])]
    det = [dict(boxes=[[110, 110, 130, 130], [0, 0, 10, 10], [500, 500, 510, 510], [150, 150, 170, 170]],
                scores=[0.95, 0.9, 0.5, 0.4], labels=[1, 1, 1, 1])]
    r = eval_ap.coco_ap(det, gt)
    # ranking after the two crowd detections leave it: TP (0.9), FP (0.5) -> precision 1 at recall 1
    assert r["n_gt"] == 1 and r["AP"] == 1.0
    # the same boxes without the flag: the crowd box is a second (missed... matched at IoU 0.01? no) ground truth and the
    # detections on it are false positives ahead of the true one
    r2 = eval_ap.coco_ap(det, [dict(boxes=gt[0]["boxes"], labels=gt[0]["labels"])])
    assert r2["n_gt"] == 2 and r2["AP"] < 0.5
    # a class that only has ignore regions is not evaluated
    only = eval_ap.coco_ap([dict(boxes=[[100, 100, 120, 120]], scores=[0.9], labels=[2])],
                           [dict(boxes=[[100, 100, 300, 300]], labels=[2], ignore=[True])])
    assert only["n_gt"] == 0
