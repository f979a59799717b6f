#!/usr/bin/env python3
"""COCO-style box AP, own implementation (no pycocotools): AP@[.50:.05:.95], AP50, AP75 over classes.

Why: the north star's accuracy bar is "box AP within 0.1 of fp32" (reference README.md:47 quotes 64.1 AP for the
Co-DINO Swin-L checkpoint; detections leave the model through inferencer.py:380-402).  Two uses:

  * real weights + COCO:    python tools/eval_ap.py coco --config <cfg.py> --weights <ckpt.pth> --coco <root> [--dtype fp16]
      runs codetr.Inferencer over val2017 (PIL decoding) and scores against instances_val2017.json -- needs the
      checkpoint and the dataset, neither of which exists offline; the evaluator itself is what the tests exercise;
  * offline PROXY:          python tools/eval_ap.py proxy [--images 8] [--size 768x512] [--dtype fp16|fp8]
      no weights, no COCO: the fp32 CPU oracle's detections above a score threshold ARE the ground truth, and the
      fp16 / fp8 product's detections on the same seeded images and trained-like weights (tests/proxy_ap_case.py)
      are scored against them.  1.0 = indistinguishable from fp32 at every IoU threshold up to 0.95.

Matching follows the COCO evaluation procedure: per class and IoU threshold, detections of all images sorted by score,
each greedily matched to the still-unmatched ground-truth box of its image with the highest IoU >= threshold;
precision made monotone from the right, sampled at the 101 recall points 0, 0.01, ..., 1; AP = mean.  At most
`max_dets` (100) highest-scoring detections per image.  No crowd / area ranges (the proxy has neither)."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
IOU_THRS = np.round(np.arange(0.5, 0.96, 0.05), 2)
REC_THRS = np.linspace(0.0, 1.0, 101)


def box_iou(a, b):
    """a [N,4], b [M,4] xyxy -> IoU [N,M] (float64; COCO's continuous-coordinate convention, no +1)"""
    a = np.asarray(a, dtype=np.float64).reshape(-1, 4)
    b = np.asarray(b, dtype=np.float64).reshape(-1, 4)
    lt = np.maximum(a[:, None, :2], b[None, :, :2])
    rb = np.minimum(a[:, None, 2:], b[None, :, 2:])
    wh = np.clip(rb - lt, 0, None)
    inter = wh[..., 0] * wh[..., 1]
    area_a = np.clip(a[:, 2] - a[:, 0], 0, None) * np.clip(a[:, 3] - a[:, 1], 0, None)
    area_b = np.clip(b[:, 2] - b[:, 0], 0, None) * np.clip(b[:, 3] - b[:, 1], 0, None)
    union = area_a[:, None] + area_b[None, :] - inter
    return np.where(union > 0, inter / np.maximum(union, 1e-300), 0.0)


def coco_ap(detections, ground_truth, iou_thrs=IOU_THRS, max_dets=100):
    """detections: list (one per image) of dict(boxes [N,4] xyxy, scores [N], labels [N]);
    ground_truth: list of dict(boxes [M,4], labels [M], optional ignore [M] bool -- COCO `iscrowd` regions).
    -> dict(AP, AP50, AP75, per_class {label: AP}, n_gt).
    Ignore regions follow the COCO procedure: a detection is matched to an unmatched regular box first (highest IoU above
    the threshold); failing that, to an ignore region whose overlap -- intersection over the DETECTION's area, a crowd box
    can absorb any number of detections -- reaches the threshold, and is then neither a true nor a false positive;
    ignore regions do not count as ground truth."""
    assert len(detections) == len(ground_truth)
    dets = []
    for d in detections:
        s = np.asarray(d["scores"], dtype=np.float64).reshape(-1)
        ok = np.isfinite(s)
        order = np.argsort(-s[ok], kind="stable")[:max_dets]
        dets.append(dict(boxes=np.asarray(d["boxes"], dtype=np.float64).reshape(-1, 4)[ok][order], scores=s[ok][order],
                         labels=np.asarray(d["labels"]).reshape(-1)[ok][order]))
    gts = []
    for g in ground_truth:
        lab = np.asarray(g["labels"]).reshape(-1)
        ign = np.asarray(g.get("ignore", np.zeros(len(lab), dtype=bool)), dtype=bool).reshape(-1)
        gts.append(dict(boxes=np.asarray(g["boxes"], dtype=np.float64).reshape(-1, 4), labels=lab, ignore=ign))
    classes = sorted(set(int(c) for g in gts for c in g["labels"][~g["ignore"]]))
    T = len(iou_thrs)
    ap = np.full((len(classes), T), np.nan)
    for ci, c in enumerate(classes):
        scores, tp, dropped = [], [[] for _ in range(T)], [[] for _ in range(T)]
        n_gt = 0
        for d, g in zip(dets, gts):
            of_c = g["labels"] == c
            gb, gi = g["boxes"][of_c & ~g["ignore"]], g["boxes"][of_c & g["ignore"]]
            sel = d["labels"] == c
            db, ds = d["boxes"][sel], d["scores"][sel]
            n_gt += len(gb)
            if len(db) == 0:
                continue
            iou = box_iou(db, gb) if len(gb) else np.zeros((len(db), 0))
            if len(gi):   # overlap with an ignore region: intersection over the detection's own area
                lt = np.maximum(db[:, None, :2], gi[None, :, :2])
                rb = np.minimum(db[:, None, 2:], gi[None, :, 2:])
                inter = np.clip(rb - lt, 0, None).prod(-1)
                area = np.clip(db[:, 2:] - db[:, :2], 0, None).prod(-1)
                ioa = (inter / np.maximum(area[:, None], 1e-12)).max(1)
            else:
                ioa = np.zeros(len(db))
            scores.append(ds)
            for ti, thr in enumerate(iou_thrs):
                taken = np.zeros(len(gb), dtype=bool)
                hit = np.zeros(len(db), dtype=bool)
                drop = np.zeros(len(db), dtype=bool)
                for i in range(len(db)):   # detections of one image arrive sorted by score
                    if len(gb):
                        cand = np.where(~taken, iou[i], -1.0)
                        j = int(cand.argmax())
                        if cand[j] >= thr:
                            taken[j] = True
                            hit[i] = True
                            continue
                    drop[i] = ioa[i] >= thr
                tp[ti].append(hit)
                dropped[ti].append(drop)
        if n_gt == 0:
            continue
        if not scores:
            ap[ci] = 0.0
            continue
        s = np.concatenate(scores)
        order = np.argsort(-s, kind="stable")
        for ti in range(T):
            t = np.concatenate(tp[ti])[order]
            t = t[~np.concatenate(dropped[ti])[order]]   # detections absorbed by an ignore region leave the ranking
            if len(t) == 0:
                ap[ci, ti] = 0.0
                continue
            ctp, cfp = np.cumsum(t), np.cumsum(~t)
            rec = ctp / n_gt
            prec = ctp / np.maximum(ctp + cfp, 1)
            for i in range(len(prec) - 1, 0, -1):   # precision envelope
                prec[i - 1] = max(prec[i - 1], prec[i])
            idx = np.searchsorted(rec, REC_THRS, side="left")
            q = np.where(idx < len(prec), prec[np.minimum(idx, len(prec) - 1)], 0.0)
            ap[ci, ti] = q.mean()
    valid = ~np.isnan(ap[:, 0]) if len(classes) else np.zeros(0, dtype=bool)
    if not valid.any():
        return dict(AP=float("nan"), AP50=float("nan"), AP75=float("nan"), per_class={}, n_gt=0)
    thr = list(np.round(iou_thrs, 2))
    out = dict(AP=float(ap[valid].mean()), per_class={classes[i]: float(ap[i].mean()) for i in np.where(valid)[0]},
               n_gt=int(sum(int((~g["ignore"]).sum()) for g in gts)))
    out["AP50"] = float(ap[valid][:, thr.index(0.5)].mean()) if 0.5 in thr else float("nan")
    out["AP75"] = float(ap[valid][:, thr.index(0.75)].mean()) if 0.75 in thr else float("nan")
    return out


def nms_per_class(det, iou_thr=0.8):
    """hard per-class NMS of one image's detections, as the Inferencer applies it before detections leave the pipeline
    (reference inferencer.py:388-401, IoU 0.8 from the config's test_cfg): random-weight models emit many near-identical
    boxes, which the evaluation would otherwise count as duplicates"""
    b = np.asarray(det["boxes"], dtype=np.float64).reshape(-1, 4)
    s = np.asarray(det["scores"], dtype=np.float64).reshape(-1)
    l = np.asarray(det["labels"]).reshape(-1)
    keep = []
    for c in np.unique(l):
        idx = np.where((l == c) & np.isfinite(s))[0]
        idx = idx[np.argsort(-s[idx], kind="stable")]
        iou = box_iou(b[idx], b[idx])
        alive = np.ones(len(idx), dtype=bool)
        for i in range(len(idx)):
            if alive[i]:
                keep.append(idx[i])
                alive[i + 1:] &= iou[i, i + 1:] <= iou_thr
    keep = np.array(sorted(keep, key=lambda i: -s[i]), dtype=np.int64)
    return dict(boxes=b[keep], scores=s[keep], labels=l[keep])


def detections_as_ground_truth(detections, score_thr, top=50):
    """the proxy's ground truth: per image the reference (fp32 oracle) detections scoring at least `score_thr`, at most
    its `top` highest (COCO scores the 100 best detections of an image: the ground truth must fit inside them)"""
    out = []
    for d in detections:
        s = np.asarray(d["scores"], dtype=np.float64)
        s = np.where(np.isfinite(s), s, -np.inf)
        order = np.argsort(-s, kind="stable")[:top]
        order = order[s[order] >= score_thr]
        out.append(dict(boxes=np.asarray(d["boxes"], dtype=np.float64)[order], labels=np.asarray(d["labels"])[order]))
    return out


# ---------------------------------------------------------------------------------------------------------------------
def run_proxy(args):
    for p in (os.path.join(ROOT, "co-detr-tensorrt_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle"), ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch

    import proxy_ap_case as C

    W, H = (int(v) for v in args.size.lower().split("x"))
    ref = [nms_per_class(d) for d in C.load_or_make_reference(args.images, H, W, make=args.make_reference)]
    gts = detections_as_ground_truth(ref, args.gt_score, args.gt_top)
    report = {"images": args.images, "size": [W, H], "gt_score_thr": args.gt_score, "gt_top": args.gt_top,
              "ground_truth_boxes": int(sum(len(g["labels"]) for g in gts)),
              "oracle_vs_itself": coco_ap(ref, gts)["AP"]}
    if torch.cuda.is_available():
        for dt in args.dtype.split(","):
            dets = [nms_per_class(d) for d in C.product_detections(args.images, H, W, dt)]
            r = coco_ap(dets, gts)
            report[dt] = {k: r[k] for k in ("AP", "AP50", "AP75")}
    print(json.dumps(report, indent=1))
    if args.out:
        with open(args.out, "w") as f:
            json.dump(report, f, indent=1)


def run_coco(args):
    for p in (os.path.join(ROOT, "co-detr-tensorrt_amd"),):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    from PIL import Image

    import codetr
    from codetr.inferencer import Inferencer

    ann = json.load(open(os.path.join(args.coco, "annotations", "instances_val2017.json")))
    cat_ids = sorted(c["id"] for c in ann["categories"])   # label l of the model <-> cat_ids[l] (mmdet's COCO order)
    cat_to_label = {c: i for i, c in enumerate(cat_ids)}
    by_img = {}
    for a in ann["annotations"]:
        x, y, w, h = a["bbox"]   # crowd regions stay, flagged: detections on them are neither true nor false positives
        by_img.setdefault(a["image_id"], []).append(([x, y, x + w, y + h], cat_to_label[a["category_id"]], bool(a.get("iscrowd", 0))))
    dtype = {"fp16": torch.float16, "bf16": torch.bfloat16, "fp32": torch.float32}[args.dtype]
    model, meta = codetr.build_CoDETR(args.config, args.weights, "cuda:0")
    model = model.to(dtype)
    inf = Inferencer(model, args.config, meta, score_threshold=0.0)
    dets, gts = [], []
    images = ann["images"][:args.limit] if args.limit else ann["images"]
    for im in images:
        rgb = np.asarray(Image.open(os.path.join(args.coco, "val2017", im["file_name"])).convert("RGB"))
        p = inf([rgb], device="cuda:0", dtype=dtype)["predictions"][0]
        dets.append(dict(boxes=p["bboxes"], scores=p["scores"], labels=p["labels"]))
        g = by_img.get(im["id"], [])
        gts.append(dict(boxes=[b for b, _, _ in g], labels=[c for _, c, _ in g], ignore=[i for _, _, i in g]))
    r = coco_ap(dets, gts)
    print(json.dumps({k: r[k] for k in ("AP", "AP50", "AP75", "n_gt")}, indent=1))


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    sub = ap.add_subparsers(dest="mode", required=True)
    p = sub.add_parser("proxy")
    p.add_argument("--images", type=int, default=8)
    p.add_argument("--size", default="768x512")
    p.add_argument("--dtype", default="fp16")
    p.add_argument("--gt-score", type=float, default=0.65)
    p.add_argument("--gt-top", type=int, default=100)
    p.add_argument("--make-reference", action="store_true", help="(re)run the fp32 CPU oracle instead of the fixture")
    p.add_argument("--out", default=None)
    c = sub.add_parser("coco")
    c.add_argument("--config", required=True)
    c.add_argument("--weights", required=True)
    c.add_argument("--coco", required=True)
    c.add_argument("--dtype", default="fp16")
    c.add_argument("--limit", type=int, default=0)
    a = ap.parse_args()
    (run_proxy if a.mode == "proxy" else run_coco)(a)


if __name__ == "__main__":
    main()
