"""smoke(): one small invocation of the hot path on cuda:0, checked against the oracle.

(1) the MSDA HIP op on a model-shaped golden case vs the C oracle (decoder shape: the general kernel; encoder shape: the
    windowed kernel of round 6);
(2) a tiny Swin-backbone CoDETR (every module kind, 2 images, one padded) in fp16 on the GPU vs
    the functional fp32 CPU oracle on the same weights, with the proposal top-k forced equal."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (os.path.join(ROOT, "co-detr-tensorrt_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)


def run():
    import codetr
    import codetr_fp32 as M
    import msda_oracle as O
    from helpers_model import assert_close_lowp, seeded_params, valid_topk

    assert torch.cuda.is_available(), "smoke() needs cuda:0"
    dev = "cuda:0"
    g = np.load(os.path.join(ROOT, "tests", "golden", "msda_g3_dec.npz"))
    t = lambda k, dt: torch.as_tensor(g[k]).to(dev).to(dt)  # noqa: E731
    out = torch.ops.codetr.multi_scale_deformable_attention(
        t("value", torch.float16), t("spatial_shapes", torch.int64), t("level_start_index", torch.int64),
        t("sampling_loc", torch.float16), t("attn_weight", torch.float16), 64)
    ref = O.msda_forward_c(g["value"], g["spatial_shapes"], g["level_start_index"], g["sampling_loc"], g["attn_weight"],
                           dtype=np.float64)
    np.testing.assert_allclose(out.float().cpu().numpy(), ref, rtol=2e-3, atol=1e-3)
    print("[smoke] MSDA HIP kernel (fp16, M=8 D=32 L=5 P=4) matches the oracle:",
          float(np.abs(out.float().cpu().numpy() - ref).max()))

    # (1b) an encoder-shaped call (Nq == S >= 4096): the windowed kernel of csrc/msda_op4.hip behind the same op
    from codetr import _cabi

    shapes = np.asarray([(64, 96), (32, 48), (16, 24), (8, 12), (4, 6)], dtype=np.int64)
    ls = np.concatenate(([0], np.cumsum(shapes[:, 0] * shapes[:, 1])[:-1])).astype(np.int64)
    S = int((shapes[:, 0] * shapes[:, 1]).sum())
    rng = np.random.default_rng(3)
    h16 = lambda a: a.astype(np.float16).astype(np.float64)  # noqa: E731
    value = h16(rng.standard_normal((1, S, 8, 32)))
    cen = np.concatenate([np.stack(np.meshgrid((np.arange(w_) + 0.5) / w_, (np.arange(h_) + 0.5) / h_), -1).reshape(-1, 2)
                          for h_, w_ in shapes], 0)
    loc = h16(cen[None, :, None, None, None, :] + rng.standard_normal((1, S, 8, 5, 4, 2)) * 3.0 / shapes[None, None, None, :, None, ::-1])
    wgt = rng.random((1, S, 8, 5, 4))
    wgt = h16(wgt / wgt.sum((-1, -2), keepdims=True))
    assert _cabi.load().codetr_msda_op4_supported(2, 1, S, 8, 32, 5, S, 4) == 1
    out2 = torch.ops.codetr.multi_scale_deformable_attention(
        torch.as_tensor(value).to(dev).half(), torch.as_tensor(shapes).to(dev), torch.as_tensor(ls).to(dev),
        torch.as_tensor(loc).to(dev).half(), torch.as_tensor(wgt).to(dev).half(), 64)
    ref2 = O.msda_forward_c(value, shapes, ls, loc, wgt, dtype=np.float64)
    np.testing.assert_allclose(out2.float().cpu().numpy(), ref2, rtol=1.1 * 2.0 ** -10, atol=2e-6)
    print("[smoke] MSDA windowed kernel (encoder shape, Nq = S = %d) matches the oracle to one fp16 ulp:" % S,
          float(np.abs(out2.float().cpu().numpy() - ref2).max()))

    from test_model_gpu import _tiny_codetr_cfg

    torch.manual_seed(0)
    model = codetr.CoDETR(**_tiny_codetr_cfg("swin"))
    spec = [(k, tuple(v.shape)) for k, v in model.named_parameters()]
    full = dict(model.state_dict())
    full.update(seeded_params(spec, 77, scale=1.0))
    model.load_state_dict(full)
    model = model.to(dev).half().eval()
    gen = torch.Generator().manual_seed(1)
    img = torch.randn(2, 3, 76, 100, generator=gen)
    mask = torch.zeros(2, 76, 100)
    mask[1, :, 80:] = 1
    cap_o = {}
    kw = dict(backbone="swin", num_heads=(1, 2, 4, 8), window_size=4, num_query=50, max_per_img=20)
    M.codetr_forward(full, img, mask, capture=cap_o, **kw)
    picks = valid_topk(cap_o["enc_outputs_class"], cap_o["enc_outputs_coord_unact"], 50)
    M.codetr_forward(full, img, mask, forced_topk=picks, capture=cap_o, **kw)
    cap = {}
    with torch.no_grad():
        boxes, scores, labels = model(img.to(dev).half(), mask.to(dev).half(),
                                      forced_topk_indices=cap_o["topk_indices"].to(dev), capture=cap)
    torch.cuda.synchronize()
    err = float((cap["memory"].float().cpu() - cap_o["memory"]).abs().max())
    for i, (a, b) in enumerate(zip(cap["neck_feats"], cap_o["neck_feats"])):
        e = assert_close_lowp(a.float().cpu().numpy(), b.numpy(), 1e-2, None, f"neck level {i}")
        print(f"[smoke]   neck level {i}: rel-L2 {e:.2e}")
    rl2 = assert_close_lowp(cap["memory"].float().cpu().numpy(), cap_o["memory"].numpy(), 1.5e-2, 0.25, "encoder memory")
    e = assert_close_lowp(cap["outputs_coords"].float().cpu().numpy(), cap_o["outputs_coords"].numpy(), 3e-2, 0.2, "box coords")
    print(f"[smoke]   box coords (after 2 decoder layers): rel-L2 {e:.2e}")
    assert boxes.shape == (2, 20, 4) and scores.shape == (2, 20) and labels.dtype == torch.int64
    print(f"[smoke] tiny CoDETR fp16 on {torch.cuda.get_device_name(0)} vs fp32 CPU oracle: encoder memory rel-L2 err {rl2:.2e}, max abs {err:.4f}")
