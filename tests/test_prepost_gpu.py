"""GPU parity of the pre / post-processing kernels (csrc/prepost.hip, C ABI codetr_preprocess_u8_*,
codetr_batched_nms_f32) and of the Inferencer host class against the CPU oracle (oracle/inferencer_ref.py).
Integer work (resized uint8 image, masks, kept indices): bit-exact.  Normalised values: the same fp32 subtract and
IEEE divide on both sides -> bit-exact in fp32, one rounding in fp16."""
import os

import numpy as np
import pytest
import torch

import inferencer_ref as R

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
MEAN, STD = (123.675, 116.28, 103.53), (58.395, 57.12, 57.375)
CFG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "co-detr-tensorrt_amd", "configs",
                   "co_dino_5scale_swin_l_16xb1_16e_o365tococo.py")


@pytest.mark.parametrize("H,W,scale,pad,pad_val", [
    (480, 640, (1152, 768), (1152, 768), (0, 0, 0)),        # up-sampling 1.6x, right padding
    (1333, 2000, (1152, 768), (1152, 768), (114, 114, 114)),  # down-sampling by a non-integer factor
    (37, 53, (64, 48), (80, 48), (7, 8, 9)),                # tiny, odd sizes
    (768, 1152, (1152, 768), (1152, 768), (0, 0, 0)),       # identity resize
])
def test_preprocess_matches_oracle_bit_for_bit(H, W, scale, pad, pad_val):
    from codetr import _cabi, hip_ops

    rng = np.random.default_rng(H * 7 + W)
    img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    x0, m0, meta = R.preprocess(img, scale, pad, MEAN, STD, pad_val)
    nh, nw = meta["img_shape"]
    Hp, Wp = meta["pad_shape"]
    before = _cabi.CALLS["preprocess"]
    x, m = hip_ops.preprocess_image(torch.from_numpy(img).to(DEV), (nh, nw), (Hp, Wp), MEAN, STD, pad_val, torch.float32)
    assert _cabi.CALLS["preprocess"] == before + 1
    assert torch.equal(x.cpu(), torch.from_numpy(x0))
    assert torch.equal(m.cpu(), torch.from_numpy(m0))
    xh, mh = hip_ops.preprocess_image(torch.from_numpy(img).to(DEV), (nh, nw), (Hp, Wp), MEAN, STD, pad_val, torch.float16)
    assert torch.equal(xh.cpu(), torch.from_numpy(x0).half()) and torch.equal(mh.cpu(), torch.from_numpy(m0).half())


@pytest.mark.parametrize("N,classes,thr,seed", [(300, 80, 0.8, 0), (300, 3, 0.5, 1), (1, 1, 0.5, 2), (1500, 5, 0.6, 3)])
def test_batched_nms_matches_oracle(N, classes, thr, seed):
    from codetr import hip_ops

    rng = np.random.default_rng(seed)
    c = rng.uniform(0, 200, (N, 2)).astype(np.float32)
    wh = rng.uniform(5, 80, (N, 2)).astype(np.float32)
    boxes = np.concatenate((c, c + wh), 1)
    boxes[N // 2:] = boxes[:N - N // 2] + rng.uniform(-3, 3, (N - N // 2, 4)).astype(np.float32)  # near duplicates
    scores = rng.uniform(0, 1, N).astype(np.float32)
    scores[::17] = scores[0]  # ties
    labels = rng.integers(0, classes, N)
    keep0 = R.batched_nms(boxes, scores, labels, thr)
    keep = hip_ops.batched_nms(torch.from_numpy(boxes).to(DEV), torch.from_numpy(scores).to(DEV),
                               torch.from_numpy(labels).to(DEV), thr)
    assert keep.cpu().tolist() == keep0.tolist()
    assert hip_ops.batched_nms(torch.zeros(0, 4, device=DEV), torch.zeros(0, device=DEV),
                               torch.zeros(0, dtype=torch.long, device=DEV), thr).numel() == 0


def test_inferencer_pre_and_post_around_a_stub_model():
    """Inferencer reads thresholds / mean / std / Resize / Pad from the Swin-L config; with a model that returns fixed
    detections the whole wrapper equals the oracle's preprocess + postprocess."""
    from codetr.inferencer import Inferencer

    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (600, 900, 3), dtype=np.uint8)
    seen = {}
    boxes = torch.tensor([[[10., 20., 200., 220.], [12., 22., 202., 222.], [400., 100., 700., 500.],
                           [0., 0., 5., 5.]]], device=DEV)
    scores = torch.tensor([[0.9, 0.8, 0.7, 0.01]], device=DEV)
    labels = torch.tensor([[3, 3, 5, 1]], device=DEV)

    def model(batch_inputs, img_masks):
        seen["x"], seen["m"] = batch_inputs, img_masks
        return boxes, scores, labels

    inf = Inferencer(model, CFG, dataset_meta=None, score_threshold=0.05)
    assert inf.with_nms and abs(inf.iou_threshold - 0.8) < 1e-9 and inf.scale == (1152, 768) and inf.pad_size == (1152, 768)
    out = inf([img], device=DEV, dtype=torch.float32)
    x0, m0, meta = R.preprocess(img, (1152, 768), (1152, 768), MEAN, STD, (0, 0, 0))
    assert seen["x"].shape == (1, 3, 768, 1152) and torch.equal(seen["x"][0].cpu(), torch.from_numpy(x0))
    assert torch.equal(seen["m"][0].cpu(), torch.from_numpy(m0))
    b0, s0, l0 = R.postprocess(boxes[0].cpu().numpy(), scores[0].cpu().numpy(), labels[0].cpu().numpy(), 0.05, 0.8,
                               meta["scale_factor"])
    pred = out["predictions"][0]
    assert pred["labels"] == l0.tolist() and np.allclose(pred["scores"], s0) and np.allclose(pred["bboxes"], b0, rtol=1e-6)
    assert len(pred["labels"]) == 2  # the near-duplicate of class 3 (IoU .96) and the 0.01 box are gone
    with pytest.raises(NotImplementedError):
        inf([img], return_vis=True)


def test_inferencer_pad_size_divisor_pads_in_normalised_space():
    """DetDataPreprocessor(pad_size_divisor=32, pad_value=0): the divisor padding is added AFTER normalisation and holds
    pad_value, while the pipeline's Pad holds normalised pad_val pixels; masks mark both as padding."""
    from codetr.inferencer import Inferencer

    rng = np.random.default_rng(6)
    img = rng.integers(0, 256, (333, 517, 3), dtype=np.uint8)
    seen = {}

    def model(batch_inputs, img_masks):
        seen["x"], seen["m"] = batch_inputs, img_masks
        z = torch.zeros(1, 0, 4, device=DEV)
        return z, torch.zeros(1, 0, device=DEV), torch.zeros(1, 0, dtype=torch.long, device=DEV)

    inf = Inferencer(model, CFG, dataset_meta=None)
    inf.pad_size_divisor, inf.pad_value = 32, 0.0
    inf.scale, inf.pad_size, inf.pad_val = (500, 300), (500, 300), (114, 114, 114)
    inf([img], device=DEV, dtype=torch.float32)
    x0, m0, meta = R.preprocess(img, (500, 300), (500, 300), MEAN, STD, (114, 114, 114), pad_size_divisor=32, pad_value=0.0)
    assert meta["pad_shape"] == (320, 512) and seen["x"].shape == (1, 3, 320, 512)
    assert torch.equal(seen["x"][0].cpu(), torch.from_numpy(x0)) and torch.equal(seen["m"][0].cpu(), torch.from_numpy(m0))
    # the two paddings differ: Pad's region holds (114 - mean) / std, the divisor region holds 0
    assert float(seen["x"][0, 0, 299, 499].abs()) > 0.05 and float(seen["x"][0, :, 310, :].abs().max()) == 0.0
    assert float(seen["m"][0, 310, 0]) == 1.0 and float(seen["m"][0, 0, 505]) == 1.0
