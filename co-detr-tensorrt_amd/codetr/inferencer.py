"""``Inferencer`` -- the reference's image-in / detections-out wrapper (reference codetr/inferencer.py:27-499), with
its pre- and post-processing on the GPU.

Same constructor and call signature for the parts that do not need mmengine's visualiser:
``Inferencer(model, model_file, dataset_meta, score_threshold=None, iou_threshold=None)`` reads ``score_thr`` /
``nms.iou_threshold`` from ``cfg.model.test_cfg[0]`` (reference :60-70), the mean / std of
``cfg.model.data_preprocessor`` (:72-76) and the ``Resize`` / ``Pad`` steps of the test pipeline (:95-101);
``__call__(images, ..., device, dtype)`` takes RGB ``np.ndarray`` images and returns
``{"predictions": [{"labels", "scores", "bboxes"}, ...], "visualization": []}`` (reference :402-485, ``pred2dict``
:303-341).  Visualisation (mmengine ``Visualizer``, cv2) is outside the scope of this build: ``return_vis`` /
``show`` raise.

Per image (reference :441-452, :343-378): upload the uint8 image, one kernel for resize + pad + normalise + mask
(``hip_ops.preprocess_image``), ``model(batch_inputs, img_masks)``, score threshold, per-class NMS
(``hip_ops.batched_nms``), boxes / scale_factor.
"""
from typing import Dict, List, Optional

import numpy as np
import torch

from . import hip_ops
from .config import Config


def rescale_size(h, w, scale):
    """mmcv.imrescale: (new_h, new_w) for a keep-ratio resize into the (long, short) bound pair `scale`"""
    f = min(max(scale) / max(h, w), min(scale) / min(h, w))
    return int(h * f + 0.5), int(w * f + 0.5)


class Inferencer:
    def __init__(self, model, model_file: str, dataset_meta, score_threshold: Optional[float] = None,
                 iou_threshold: Optional[float] = None):
        self.model = model
        self.dataset_meta = dataset_meta
        self.cfg = Config.fromfile(model_file)
        test_cfg = self.cfg.model.test_cfg[0]  # the 0th test_cfg is for the query_head
        self.score_threshold = test_cfg.get("score_thr", 0)
        if score_threshold is not None:
            self.score_threshold = score_threshold
        self.with_nms = False
        if "nms" in test_cfg:
            self.with_nms = True
            self.iou_threshold = test_cfg["nms"].get("iou_threshold", 0.8)
            if iou_threshold is not None:
                self.iou_threshold = iou_threshold
        pre = dict(self.cfg.model.data_preprocessor)
        if pre.pop("type") != "DetDataPreprocessor":
            raise AssertionError("data_preprocessor must be DetDataPreprocessor")
        self.mean = tuple(float(v) for v in pre.get("mean", (0.0, 0.0, 0.0)))
        self.std = tuple(float(v) for v in pre.get("std", (1.0, 1.0, 1.0)))
        self.pad_size_divisor = int(pre.get("pad_size_divisor", 1))
        self.pad_value = float(pre.get("pad_value", 0))   # DetDataPreprocessor: fills the divisor padding, AFTER normalisation
        # test pipeline: Resize(scale, keep_ratio) [+ Pad(size, pad_val)]
        self.scale, self.pad_size, self.pad_val = None, None, (0, 0, 0)
        for step in self.cfg.test_dataloader.dataset.pipeline:
            if step["type"] == "Resize":
                if not step.get("keep_ratio", False):
                    raise NotImplementedError("only keep_ratio=True resizing is on the reference's inference path")
                self.scale = tuple(step["scale"])
            elif step["type"] == "Pad":
                if step.get("size") is not None:
                    self.pad_size = tuple(step["size"])  # (width, height)
                pv = step.get("pad_val", dict(img=0))
                pv = pv.get("img", 0) if isinstance(pv, dict) else pv
                self.pad_val = tuple(pv) if isinstance(pv, (tuple, list)) else (pv,) * 3
        if self.scale is None:
            raise ValueError("Resize is not found in the test pipeline")
        self.num_predicted_imgs = 0

    # ---- pre ------------------------------------------------------------------------------------------
    def preprocess(self, image: np.ndarray, device="cuda:0", dtype=torch.float32):
        """one RGB uint8 image [H, W, 3] -> (batch_inputs [1,3,Hp,Wp], img_masks [1,Hp,Wp], meta)"""
        if image.dtype != np.uint8 or image.ndim != 3 or image.shape[2] != 3:
            raise ValueError("expected an RGB uint8 image of shape (H, W, 3)")
        H, W = image.shape[:2]
        nh, nw = rescale_size(H, W, self.scale)
        Hp, Wp = nh, nw
        if self.pad_size is not None:
            Wp, Hp = max(self.pad_size[0], nw), max(self.pad_size[1], nh)
        src = torch.from_numpy(np.ascontiguousarray(image)).to(device, non_blocking=True)
        # the pipeline's Pad: pad_val pixels, normalised with the image (mmdet Pad runs before DetDataPreprocessor)
        x, m = hip_ops.preprocess_image(src, (nh, nw), (Hp, Wp), self.mean, self.std, self.pad_val, dtype)
        d = self.pad_size_divisor
        if d > 1 and (Hp % d or Wp % d):
            # DetDataPreprocessor's own padding to a multiple of pad_size_divisor: applied to the NORMALISED tensor
            # and filled with pad_value (default 0), not with normalised pad_val pixels; the mask marks it as padding
            Hd, Wd = -(-Hp // d) * d, -(-Wp // d) * d
            x = torch.nn.functional.pad(x, (0, Wd - Wp, 0, Hd - Hp), value=self.pad_value)
            m = torch.nn.functional.pad(m, (0, Wd - Wp, 0, Hd - Hp), value=1.0)
            Hp, Wp = Hd, Wd
        meta = dict(ori_shape=(H, W), img_shape=(nh, nw), img_unpadded_shape=(nh, nw), pad_shape=(Hp, Wp),
                    scale_factor=(nw / W, nh / H))
        return x[None], m[None], meta

    # ---- post -----------------------------------------------------------------------------------------
    def postprocess_predictions(self, batch_boxes, batch_scores, batch_labels):
        """score threshold + per-class NMS per image (reference :380-400)"""
        out = []
        for boxes, scores, labels in zip(batch_boxes, batch_scores, batch_labels):
            if self.score_threshold > 0:
                valid = scores > self.score_threshold
                scores, boxes, labels = scores[valid], boxes[valid], labels[valid]
            if self.with_nms:
                keep = hip_ops.batched_nms(boxes, scores, labels, self.iou_threshold)
                boxes, scores, labels = boxes[keep], scores[keep], labels[keep]
            out.append((boxes, scores, labels))
        return out

    def run_inference(self, batch_inputs, img_masks, metas):
        """model + post-processing + rescale to the original image (reference :343-378)"""
        predictions = self.model(batch_inputs, img_masks)
        results = []
        for i, (boxes, scores, labels) in enumerate(self.postprocess_predictions(*predictions)):
            sf = metas[i]["scale_factor"]
            boxes = boxes / boxes.new_tensor([sf[0], sf[1], sf[0], sf[1]])
            results.append(dict(bboxes=boxes, scores=scores, labels=labels))
        return results

    def __call__(self, images: List[np.ndarray], return_vis: bool = False, show: bool = False, wait_time: int = 0,
                 no_save_vis: bool = False, draw_pred: bool = True, pred_score_thr: float = 0.3,
                 return_datasamples: bool = False, print_result: bool = False, no_save_pred: bool = True,
                 out_dir: str = "", device: str = "cuda:0", dtype: torch.dtype = torch.float32) -> Dict:
        if return_vis or show or not no_save_pred or return_datasamples:
            raise NotImplementedError("visualisation / DetDataSample / file output need mmengine + cv2: not part of this build")
        results_dict = {"predictions": [], "visualization": []}
        for image in images:
            with torch.no_grad():
                x, m, meta = self.preprocess(image, device, dtype)
                res = self.run_inference(x, m, [meta])[0]
            pred = {"labels": res["labels"].tolist(), "scores": res["scores"].float().tolist(),
                    "bboxes": res["bboxes"].float().tolist()}
            if print_result:
                print(pred)
            self.num_predicted_imgs += 1
            results_dict["predictions"].append(pred)
        return results_dict
