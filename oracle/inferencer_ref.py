"""CPU oracle for the pre / post-processing either side of CoDETR.forward (test infrastructure only: imported by
tests/ and smoke(), never by the product).

Restates what the reference's `Inferencer` runs around the model (reference codetr/inferencer.py:343-402, 439-452;
test pipeline configs/co_dino_5scale_swin_l_16xb1_16e_o365tococo.py:88-96; data_preprocessor configs
co_dino_5scale_r50_lsj_8xb2_1x_coco.py:20-26):

  pre   mmdet `Resize(scale, keep_ratio=True)` -> mmcv.imrescale -> cv2.resize(INTER_LINEAR) on uint8,
        mmdet `Pad(size=(w, h), pad_val)` (right / bottom), DetDataPreprocessor `(x - mean) / std` in fp32,
        img_masks = 1 outside the resized image
  post  score threshold, torchvision `batched_nms` (per-class greedy hard NMS, IoU > thr suppresses), boxes / scale_factor

PARITY UNPINNED: mmcv / mmdet / cv2 / torchvision are third-party for the reference (mmdet v3.3.0, mmcv 2.1, opencv,
torchvision 0.21) and none of them is installed here; the reference tree holds no stored values for these steps.  The
arithmetic below follows their published algorithms: OpenCV's 8-bit bilinear resize with 11-bit fixed-point
coefficients (modules/imgproc/src/resize.cpp: coefficient = cvRound(f * 2048) saturated to short, value =
(b0*(a0*p00 + a1*p01) + b1*(a0*p10 + a1*p11) + 2^21) >> 22, source index (dst + 0.5) * scale - 0.5 with edge
clamping) and torchvision's nms (ops/cpu/nms_kernel.cpp: areas (x2-x1)*(y2-y1), inter = max(0, .)*max(0, .),
suppress when inter / (a_i + a_j - inter) > thr, candidates visited in descending score order) applied per class.
"""
import numpy as np


def rescale_size(h, w, scale):
    """mmcv.imrescale's target size for `scale` = (long-ish, short-ish) bound pair -> (new_h, new_w, factor)"""
    max_long, max_short = max(scale), min(scale)
    f = min(max_long / max(h, w), max_short / min(h, w))
    return int(h * f + 0.5), int(w * f + 0.5), f


def _coeffs(dst, src):
    scale = 1.0 / (float(dst) / float(src))  # cv2: inv_scale = dsize / ssize (double), scale = 1. / inv_scale
    s = np.empty(dst, np.int64)
    a1 = np.empty(dst, np.int64)
    for d in range(dst):
        f = np.float32((d + 0.5) * scale - 0.5)  # double expression narrowed to float, then floor / subtract in float
        i = int(np.floor(f))
        f = np.float32(f - np.float32(i))
        if i < 0:
            i, f = 0, np.float32(0.0)
        if i >= src - 1:
            i, f = src - 1, np.float32(0.0)
        s[d] = i
        a1[d] = int(np.rint(f * np.float32(2048.0)))  # cvRound: round half to even
    return s, a1


def resize_bilinear_u8(img, new_h, new_w):
    """img [H, W, C] uint8 -> [new_h, new_w, C] uint8, OpenCV INTER_LINEAR fixed-point arithmetic"""
    H, W, _ = img.shape
    sy, b1 = _coeffs(new_h, H)
    sx, a1 = _coeffs(new_w, W)
    sy1 = np.minimum(sy + 1, H - 1)
    sx1 = np.minimum(sx + 1, W - 1)
    p = img.astype(np.int64)
    a1 = a1[None, :, None]
    a0 = 2048 - a1
    top = a0 * p[sy][:, sx] + a1 * p[sy][:, sx1]
    bot = a0 * p[sy1][:, sx] + a1 * p[sy1][:, sx1]
    b1 = b1[:, None, None]
    b0 = 2048 - b1
    v = (b0 * top + b1 * bot + (1 << 21)) >> 22
    return np.clip(v, 0, 255).astype(np.uint8)


def preprocess(img, scale, pad_size, mean, std, pad_val=(0, 0, 0), dtype=np.float32, pad_size_divisor=1, pad_value=0.0):
    """img [H, W, 3] uint8 RGB -> (inputs [3, Hp, Wp] dtype, mask [Hp, Wp] dtype (1 = padding), meta dict).
    Two paddings, as mmdet applies them: the pipeline's Pad(size, pad_val) on uint8 pixels BEFORE normalisation, and
    DetDataPreprocessor's padding to a multiple of pad_size_divisor AFTER it, filled with pad_value."""
    H, W, _ = img.shape
    nh, nw, _ = rescale_size(H, W, scale)
    r = resize_bilinear_u8(img, nh, nw)
    Wp, Hp = pad_size
    Hp, Wp = max(Hp, nh), max(Wp, nw)
    canvas = np.empty((Hp, Wp, 3), np.uint8)
    canvas[...] = np.asarray(pad_val, np.uint8)
    canvas[:nh, :nw] = r
    x = (canvas.astype(np.float32) - np.asarray(mean, np.float32)) / np.asarray(std, np.float32)
    mask = np.ones((Hp, Wp), np.float32)
    mask[:nh, :nw] = 0
    d = int(pad_size_divisor)
    if d > 1 and (Hp % d or Wp % d):
        Hd, Wd = -(-Hp // d) * d, -(-Wp // d) * d
        xd = np.full((Hd, Wd, 3), np.float32(pad_value), np.float32)
        xd[:Hp, :Wp] = x
        md = np.ones((Hd, Wd), np.float32)
        md[:Hp, :Wp] = mask
        x, mask, Hp, Wp = xd, md, Hd, Wd
    meta = dict(ori_shape=(H, W), img_shape=(nh, nw), img_unpadded_shape=(nh, nw), pad_shape=(Hp, Wp),
                scale_factor=(nw / W, nh / H))
    return x.transpose(2, 0, 1).astype(dtype), mask.astype(dtype), meta


def batched_nms(boxes, scores, labels, iou_threshold):
    """indices kept by per-class greedy NMS, in descending score order (stable for ties)"""
    boxes = np.asarray(boxes, np.float32)
    order = np.argsort(-np.asarray(scores, np.float32), kind="stable")
    areas = (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])
    dead = np.zeros(len(order), bool)
    keep = []
    thr = np.float32(iou_threshold)
    for a, i in enumerate(order):
        if dead[a]:
            continue
        keep.append(int(i))
        for b in range(a + 1, len(order)):
            j = order[b]
            if dead[b] or labels[j] != labels[i]:
                continue
            w = max(np.float32(0), min(boxes[i, 2], boxes[j, 2]) - max(boxes[i, 0], boxes[j, 0]))
            h = max(np.float32(0), min(boxes[i, 3], boxes[j, 3]) - max(boxes[i, 1], boxes[j, 1]))
            inter = np.float32(w) * np.float32(h)
            iou = inter / (areas[i] + areas[j] - inter)
            if iou > thr:
                dead[b] = True
    return np.asarray(keep, np.int64)


def postprocess(boxes, scores, labels, score_threshold, iou_threshold, scale_factor, with_nms=True):
    """reference Inferencer.postprocess_predictions + the rescale of run_inference, one image"""
    boxes, scores, labels = np.asarray(boxes, np.float32), np.asarray(scores, np.float32), np.asarray(labels)
    if score_threshold > 0:
        m = scores > np.float32(score_threshold)
        boxes, scores, labels = boxes[m], scores[m], labels[m]
    if with_nms:
        k = batched_nms(boxes, scores, labels, iou_threshold)
        boxes, scores, labels = boxes[k], scores[k], labels[k]
    sf = np.asarray([scale_factor[0], scale_factor[1]] * 2, np.float32)
    return boxes / sf, scores, labels
