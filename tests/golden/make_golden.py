"""Generate the golden fixtures in tests/golden/ by running the REFERENCE's own Python.

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden.py            # all groups
    python tests/golden/make_golden.py msda       # one group

Every fixture is data: seeded inputs + the outputs the reference produced for them.  The
reference never travels to the GPU box; these .npz files do.  Recipes follow the reference's
tests:

  msda_g1  tests/test_multi_scale_deformable_attention.py:246-364  (seed 3, N=1,M=2,D=2,Lq=2,L=2,P=2)
  msda_g2  tests/test_multi_scale_deformable_attention.py:14-62    (B=2,M=4,Nq=8,D=16,L=3,P=4, 32/16/8 squares)
  msda_g3  model-shaped slice: M=8,D=32,L=5,P=4 on the pyramid of a 64x96 image, Nq=64 and
           Nq=S, locations in [-0.1,1.1) incl. exact borders
  msda_g4  perf-test shape tests/...:417-501 (N=1,M=8,D=64,Lq=100,L=4,P=4, seed 42; squares 16..2 instead of 64..8)
  model_*  module-level captures (see make_model_goldens)
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_import as R  # noqa: E402


def _lsi(shapes):
    return torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))


def _save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = v
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.1f} KiB)")


def make_msda_goldens():
    ops = R.ref("ops")
    f = ops.multi_scale_deformable_attention_pytorch

    # ---- G1: reference tests :246-364 ----
    N, M, D, Lq, L, P = 1, 2, 2, 2, 2, 2
    shapes = torch.as_tensor([(6, 4), (3, 2)], dtype=torch.long)
    S = int(shapes.prod(1).sum())
    torch.manual_seed(3)
    value = torch.rand(N, S, M, D) * 0.01
    loc = torch.rand(N, Lq, M, L, P, 2)
    w = torch.rand(N, Lq, M, L, P) + 1e-5
    w /= w.sum(-1, keepdim=True).sum(-2, keepdim=True)
    _save(
        "msda_g1",
        value=value, spatial_shapes=shapes, level_start_index=_lsi(shapes), sampling_loc=loc, attn_weight=w,
        out_f64=f(value.double(), shapes, loc.double(), w.double()),
        out_f32=f(value, shapes, loc, w),
        out_f16=f(value.half(), shapes, loc.half(), w.half()).float(),
    )

    # ---- G2: reference tests :14-62 (fixed seed added) ----
    torch.manual_seed(1234)
    B, M, Nq, D, L, P = 2, 4, 8, 16, 3, 4
    shapes = torch.tensor([[32, 32], [16, 16], [8, 8]], dtype=torch.long)
    S = int(shapes.prod(1).sum())
    value = torch.rand(B, S, M, D)
    loc = torch.rand(B, Nq, M, L, P, 2)
    w = torch.rand(B, Nq, M, L, P)
    _save(
        "msda_g2",
        value=value.half(), spatial_shapes=shapes, level_start_index=_lsi(shapes), sampling_loc=loc.half(),
        attn_weight=w.half(),
        # fp16-representable inputs; outputs from the reference in fp64 / fp32 / fp16
        out_f64=f(value.half().double(), shapes, loc.half().double(), w.half().double()),
        out_f32=f(value.half().float(), shapes, loc.half().float(), w.half().float()),
        out_f16=f(value.half(), shapes, loc.half(), w.half()).float(),
    )

    # ---- G3: model-shaped slice with border stress ----
    torch.manual_seed(42)
    M, D, L, P = 8, 32, 5, 4
    shapes = torch.tensor([[16, 24], [8, 12], [4, 6], [2, 3], [1, 2]], dtype=torch.long)  # 64x96 image pyramid
    S = int(shapes.prod(1).sum())
    for tag, B, Nq in (("dec", 2, 64), ("enc", 1, S)):
        value = torch.rand(B, S, M, D).half()
        loc = (torch.rand(B, Nq, M, L, P, 2) * 1.2 - 0.1)
        # exact-border and exact-pixel-centre locations
        loc[:, 0] = 0.0
        loc[:, 1] = 1.0
        loc[:, 2] = 0.5
        loc[:, 3, :, :, :, 0] = 0.0
        loc[:, 4, :, :, :, 1] = 1.0
        loc[:, 5] = -0.1
        loc[:, 6] = 1.1
        for l in range(L):  # pixel centres of level l: (i + 0.5) / W
            loc[:, 7, :, l, :, 0] = (torch.arange(P) % shapes[l, 1] + 0.5) / shapes[l, 1]
            loc[:, 7, :, l, :, 1] = (torch.arange(P) % shapes[l, 0] + 0.5) / shapes[l, 0]
        loc = loc.half()
        w = torch.rand(B, Nq, M, L, P)
        w = (w / w.sum((-1, -2), keepdim=True)).half()
        _save(
            f"msda_g3_{tag}",
            value=value, spatial_shapes=shapes, level_start_index=_lsi(shapes), sampling_loc=loc, attn_weight=w,
            out_f64=f(value.double(), shapes, loc.double(), w.double()).float(),  # stored as f32 to stay small
            out_f32=f(value.float(), shapes, loc.float(), w.float()),
        )

    # ---- G4: reference perf-test shape :417-501 ----
    torch.manual_seed(42)
    N, M, D, Lq, L, P = 1, 8, 64, 100, 4, 4
    shapes = torch.tensor([[16, 16], [8, 8], [4, 4], [2, 2]], dtype=torch.long)  # reference uses 64..8; scaled /4 to keep the fixture small
    S = int(shapes.prod(1).sum())
    value = torch.rand(N, S, M, D).half()
    loc = torch.rand(N, Lq, M, L, P, 2).half()
    w = torch.rand(N, Lq, M, L, P)
    w = (w / w.sum((-1, -2), keepdim=True)).half()
    _save(
        "msda_g4",
        value=value, spatial_shapes=shapes, level_start_index=_lsi(shapes), sampling_loc=loc, attn_weight=w,
        out_f32=f(value.float(), shapes, loc.float(), w.float()),
    )


def _sd_np(module, prefix=""):
    return {prefix + k: v.detach().cpu().numpy() for k, v in module.state_dict().items()}


sys.path.insert(0, os.path.join(HERE, ".."))
from helpers_model import pack_param_spec  # noqa: E402


def make_model_goldens():
    """Module-level captures from the reference's own Python classes with seeded parameters.

    Weights are NOT stored: every fixture stores the seed, and tests rebuild the identical
    state_dict with tests/golden/_ref_import.randomize_ semantics re-stated in
    tests/helpers_model.py (pure torch.Generator arithmetic, no reference code) -- except for the
    small modules, whose weights are stored directly."""
    import copy

    tr = R.ref("transformer")
    pe = R.ref("positional_encoding")
    sw = R.ref("swin")
    msda_mod = R.ref("multi_scale_deformable_attention")

    # ---- G5: SinePositionalEncoding on a padded mask (cfg lsj:102-106) ----
    enc = pe.SinePositionalEncoding(num_feats=128, temperature=20, normalize=True)
    mask = torch.zeros(2, 7, 9, dtype=torch.bool)
    mask[0, :, 7:] = True
    mask[1, 5:, :] = True
    mask[1, :, 8:] = True
    _save("model_posenc", mask=mask, out=enc(mask, dtype=torch.float32))

    # ---- G4: MultiScaleDeformableAttention.forward, 2-d and 4-d reference branches ----
    torch.manual_seed(0)
    m = msda_mod.MultiScaleDeformableAttention(embed_dims=256, num_levels=5, dropout=0.0).eval()
    spec = R.randomize_(m, 101)
    shapes = torch.tensor([[8, 12], [4, 6], [2, 3], [1, 2], [1, 1]])
    S = int(shapes.prod(1).sum())
    lsi = _lsi(shapes)
    g = torch.Generator().manual_seed(5)
    B = 2
    value = torch.randn(S, B, 256, generator=g)
    qpos = torch.randn(S, B, 256, generator=g)
    kpm = torch.zeros(B, S, dtype=torch.bool)
    kpm[1, -7:] = True
    ref2 = torch.rand(B, S, 5, 2, generator=g)
    out2 = m(value, value=None, query_pos=qpos, key_padding_mask=kpm, reference_points=ref2, spatial_shapes=shapes,
             level_start_index=lsi)
    Nq = 13
    q = torch.randn(Nq, B, 256, generator=g)
    qp = torch.randn(Nq, B, 256, generator=g)
    ref4 = torch.rand(B, Nq, 5, 4, generator=g) * 0.5 + 0.1
    out4 = m(q, value=value, query_pos=qp, key_padding_mask=kpm, reference_points=ref4, spatial_shapes=shapes,
             level_start_index=lsi)
    _save("model_msda_module", spatial_shapes=shapes, level_start_index=lsi, value=value, query_pos=qpos,
          key_padding_mask=kpm, ref2=ref2, out2=out2, query4=q, query_pos4=qp, ref4=ref4, out4=out4,
          seed=np.int64(101), **pack_param_spec(spec))

    # ---- G6: CoDinoTransformer (2 enc + 2 dec layers, 40 queries, FFN 64) on a 48x64-image pyramid ----
    torch.manual_seed(0)
    cfg = R.transformer_cfg(num_levels=5, num_layers=(2, 2), num_query=40, ffn=64)
    t = tr.CoDinoTransformer(**copy.deepcopy(cfg)).eval()
    t.level_embeds.data.zero_()
    cls_b, reg_b = R.make_branches(num_pred=3, num_classes=80)
    spec_t = R.randomize_(t, 202, prefix="query_head.transformer.")
    spec_c = R.randomize_(cls_b, 203, prefix="query_head.cls_branches.")
    spec_r = R.randomize_(reg_b, 204, prefix="query_head.reg_branches.")
    g = torch.Generator().manual_seed(6)
    shapes_l = [(12, 16), (6, 8), (3, 4), (2, 2), (1, 1)]
    B = 2
    feats = [torch.randn(B, 256, h, w, generator=g) for h, w in shapes_l]
    img_mask = torch.zeros(B, 48, 64)
    img_mask[1, :, 52:] = 1  # right padding on image 1
    img_mask[1, 40:, :] = 1  # bottom padding
    masks = [torch.nn.functional.interpolate(img_mask[:, None], size=f.shape[-2:]).to(torch.bool).squeeze(1) for f in feats]
    pos = [enc(mk, dtype=torch.float32) for mk in masks]
    cap = {}
    # capture encoder memory through a forward hook, everything else from the outputs
    hk = t.encoder.register_forward_hook(lambda mod, a, out: cap.__setitem__("memory", out.permute(1, 0, 2)))
    state, refs = t(feats, masks, pos, reg_branches=reg_b, cls_branches=cls_b)
    hk.remove()
    packed = {}
    for tag, sp, seed in (("t", spec_t, 202), ("c", spec_c, 203), ("r", spec_r, 204)):
        packed.update({f"{tag}.{k}": v for k, v in pack_param_spec(sp).items()})
        packed[f"{tag}.seed"] = np.int64(seed)
    _save("model_transformer", img_mask=img_mask, final_state=state, final_refs_unact=refs, memory=cap["memory"],
          feat_seed=np.int64(6), **packed)

    # ---- G7: Swin: tiny 2-stage backbone (C=32, heads 2/4, window 4), H/W not multiples of the window ----
    torch.manual_seed(0)
    s = sw.SwinTransformer(pretrain_img_size=64, embed_dims=32, depths=(2, 2), num_heads=(2, 4), window_size=4,
                           strides=(4, 2), out_indices=(0, 1), drop_path_rate=0.0, patch_norm=True)
    s.eval()
    spec = R.randomize_(s, 303, scale=2.0, prefix="backbone.")
    g = torch.Generator().manual_seed(7)
    img = torch.randn(2, 3, 44, 58, generator=g)  # -> 11x15 tokens (pads to 12x16), then 6x8 (pads to 8x8)
    outs = s(img)
    _save("model_swin_tiny", img=img, out0=outs[0], out1=outs[1], seed=np.int64(303), **pack_param_spec(spec))


def make_head_goldens():
    """G8: the reference's own ``CoDINOHead.forward`` (co_dino_head.py:120-210: mask pyramid, positional encodings,
    transformer, last-layer class / box branches, sigmoid, top-k over (query, class), label / box decode, scaling and
    clamping) on a tiny pyramid, with the mmdet ``DINOHead`` constructor chain and ``bbox_cxcywh_to_xyxy`` supplied by
    the stand-ins of _ref_import.py.  torch.topk is wrapped during the run to record the proposal selection, so the
    GPU side can force it equal."""
    import copy

    hd = R.ref("co_dino_head")
    cfg = R.transformer_cfg(num_levels=5, num_layers=(2, 2), num_query=40, ffn=64)
    cfg["type"] = "CoDinoTransformer"
    cfg.pop("two_stage_num_proposals")
    torch.manual_seed(0)
    head = hd.CoDINOHead(num_query=40, transformer=R._ConfigDict(copy.deepcopy(cfg)), num_classes=80, as_two_stage=True,
                         positional_encoding=R._ConfigDict(type="SinePositionalEncoding", num_feats=128, temperature=20, normalize=True),
                         loss_cls=dict(type="QualityFocalLoss", use_sigmoid=True, beta=2.0, loss_weight=1.0),
                         test_cfg=dict(max_per_img=25)).eval()
    g = torch.Generator().manual_seed(8)
    shapes_l = [(12, 16), (6, 8), (3, 4), (2, 2), (1, 1)]
    B = 2
    feats = [torch.randn(B, 256, h, w, generator=g) for h, w in shapes_l]
    img_mask = torch.zeros(B, 48, 64)
    img_mask[1, :, 52:] = 1
    img_mask[1, 40:, :] = 1
    real_topk = torch.topk
    # with random weights a padded position can win the proposal top-k and its NaN box (reference transformer.py:338)
    # poisons the image: take the first weight seed whose detections are all finite (padding stays in the case)
    for seed in range(505, 540):
        spec = R.randomize_(head, seed, prefix="query_head.")
        calls = []

        def spy(*a, **k):
            out = real_topk(*a, **k)
            calls.append(out[1].clone())
            return out

        torch.topk = spy
        try:
            boxes, scores, labels = head(feats, img_mask)
        finally:
            torch.topk = real_topk
        assert len(calls) == 2 and calls[0].shape == (B, 40) and calls[1].shape == (B, 25)
        if torch.isfinite(boxes).all() and torch.isfinite(scores).all():
            break
    else:
        raise SystemExit("no seed with finite detections")
    _save("model_head", img_mask=img_mask, boxes=boxes, scores=scores, labels=labels, proposal_topk=calls[0],
          detection_topk=calls[1], feat_seed=np.int64(8), seed=np.int64(seed), **pack_param_spec(spec))


def make_grad_goldens():
    """Gradients of the reference's differentiable PyTorch formulation (ops.py:129-186) by torch.autograd in fp64, on
    the geometry of the reference's own gradient test (tests/test_multi_scale_deformable_attention.py:367-414:
    N=1, M=2, Lq=2, L=2, P=2, shapes (3,2),(2,1), channels 4 / 30 / 32 / 64 / 71 / 1025) plus one model-shaped case
    (M=8, D=32, L=5, P=4 with samples outside the maps)."""
    ops = R.ref("ops")
    f = ops.multi_scale_deformable_attention_pytorch
    out = {}

    def case(tag, N, M, D, Lq, shapes, P, seed, spread=False):
        shapes_t = torch.as_tensor(shapes, dtype=torch.long)
        S = int(shapes_t.prod(1).sum())
        L = len(shapes)
        g = torch.Generator().manual_seed(seed)
        value = (torch.rand(N, S, M, D, generator=g, dtype=torch.float64) * (1.0 if spread else 0.01))
        loc = torch.rand(N, Lq, M, L, P, 2, generator=g, dtype=torch.float64)
        if spread:
            loc = loc * 1.3 - 0.15
        w = torch.rand(N, Lq, M, L, P, generator=g, dtype=torch.float64) + 1e-5
        w = w / w.sum(-1, keepdim=True).sum(-2, keepdim=True)
        go = torch.randn(N, Lq, M * D, generator=g, dtype=torch.float64)
        # inputs representable in fp16 so that lower-precision runs differentiate the same function
        value, loc, w, go = (t.half().double() for t in (value, loc, w, go))
        v, l_, w_ = (t.clone().requires_grad_(True) for t in (value, loc, w))
        with torch.enable_grad():
            y = f(v, shapes_t, l_, w_)
            gv, gl, gw = torch.autograd.grad(y, (v, l_, w_), go)
        out.update({f"{tag}.shapes": shapes_t, f"{tag}.value": value, f"{tag}.loc": loc, f"{tag}.w": w, f"{tag}.go": go,
                    f"{tag}.out": y.detach(), f"{tag}.grad_value": gv, f"{tag}.grad_loc": gl, f"{tag}.grad_w": gw})

    for D in (4, 30, 32, 64, 71, 1025):
        case(f"c{D}", 1, 2, D, 2, [(3, 2), (2, 1)], 2, seed=100 + D)
    case("model", 2, 8, 32, 19, [(12, 18), (6, 9), (3, 5), (2, 3), (1, 2)], 4, seed=77, spread=True)
    _save("msda_grad", **out)


def make_key_goldens():
    """state_dict key names + shapes of the reference's Swin-L backbone and Co-DINO transformer built
    from the config values (swin:10-27, lsj:58-101): the checkpoint-compatibility contract."""
    import copy

    sw = R.ref("swin")
    tr = R.ref("transformer")
    s = sw.SwinTransformer(pretrain_img_size=384, embed_dims=192, depths=[2, 2, 18, 2], num_heads=[6, 12, 24, 48],
                           window_size=12, mlp_ratio=4, qkv_bias=True, qk_scale=None, drop_rate=0.0, attn_drop_rate=0.0,
                           drop_path_rate=0.3, patch_norm=True, out_indices=(0, 1, 2, 3), with_cp=True,
                           convert_weights=True)
    t = tr.CoDinoTransformer(**copy.deepcopy(R.transformer_cfg()))
    spec = [("backbone." + k, tuple(v.shape)) for k, v in s.state_dict().items()]
    spec += [("query_head.transformer." + k, tuple(v.shape)) for k, v in t.state_dict().items()]
    # the head's own parameters as the reference's CoDINOHead._init_layers builds them (co_dino_head.py:95-118:
    # 7 class / 7 box branches, the unused `downsample`); the mmdet DINOHead constructor chain is a stand-in
    hd = R.ref("co_dino_head")
    hcfg = copy.deepcopy(R.transformer_cfg())
    hcfg["type"] = "CoDinoTransformer"
    hcfg.pop("two_stage_num_proposals")
    head = hd.CoDINOHead(num_query=900, transformer=R._ConfigDict(hcfg), num_classes=80, as_two_stage=True,
                         positional_encoding=R._ConfigDict(type="SinePositionalEncoding", num_feats=128, temperature=20, normalize=True),
                         loss_cls=dict(type="QualityFocalLoss", use_sigmoid=True), test_cfg=dict(max_per_img=300))
    spec += [("query_head." + k, tuple(v.shape)) for k, v in head.state_dict().items() if not k.startswith("transformer.")]
    # relative_position_index VALUES of one window (the buffer is part of the checkpoint contract)
    _save("state_dict_keys", rel_index=s.stages[0].blocks[0].attn.w_msa.relative_position_index, **pack_param_spec(spec))


GROUPS = {"msda": make_msda_goldens, "model": make_model_goldens, "keys": make_key_goldens, "head": make_head_goldens,
          "grad": make_grad_goldens}


if __name__ == "__main__":
    if not R.reference_available():
        sys.exit("reference tree not found at " + R.REFERENCE_ROOT)
    torch.set_grad_enabled(False)   # (make_grad_goldens re-enables it locally)
    torch.set_num_threads(8)
    which = sys.argv[1:] or list(GROUPS)
    for g in which:
        GROUPS[g]()
