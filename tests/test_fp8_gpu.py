"""FP8 (OCP e4m3) path -- BASELINE config 5.  No reference counterpart (reference dtypes stop at half), so:

  * kernels against exact restatements: the fp8 GEMM against a float64 product of the DEQUANTISED operands (what the
    instruction computes, up to fp32 accumulation order: 1e-3 relative) and against an exact small-integer case that
    pins the operand / accumulator lane maps; the e4m3 conversion against torch's float8_e4m3fn cast bit for bit;
  * model level against the fp32 CPU oracle at a wider tolerance than fp16 (e4m3 carries 3 mantissa bits: a few per
    cent of relative noise on every GEMM output, accumulating over 22 blocks): relative L2 <= 5e-2 after Swin stage 1,
    <= 1.5e-1 after stages 2-3 and on the encoder memory, mean decoded-box error <= 2 % of the image width, with the
    proposal top-k forced equal; the measured values go to gpurun_out/parity_report.json."""
import os
from functools import partial

import pytest
import torch

import codetr_fp32 as M
import fullsize_cases as F
from helpers_model import assert_close_lowp, detection_agreement, seeded_params, valid_topk

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
FP8 = torch.float8_e4m3fn


def _deq(t8):
    return t8.float().double()


def test_cast_matches_torch_e4m3_bit_for_bit():
    from codetr import hip_ops

    g = torch.Generator(device=DEV).manual_seed(0)
    x = (torch.randn(1 << 16, device=DEV, generator=g) * 50).half()
    x[:8] = torch.tensor([0.0, -0.0, 448.0, 449.0, 1e4, -1e4, 2.0 ** -9, 2.0 ** -10], device=DEV).half()
    for scale in (1.0, 0.37, 3.0):
        got = hip_ops.cast_fp8(x, scale)
        ref = (x.float() / scale).clamp(-448, 448).to(FP8)
        assert torch.equal(got.view(torch.uint8), ref.view(torch.uint8)), scale


def test_layernorm_fp8_vs_torch():
    from codetr import hip_ops

    g = torch.Generator(device=DEV).manual_seed(1)
    for C in (384, 768, 1536, 192):
        x = (torch.randn(1000, C, device=DEV, generator=g) * 3 + 0.5).half()
        w = (1 + 0.1 * torch.randn(C, device=DEV, generator=g)).half()
        b = (0.1 * torch.randn(C, device=DEV, generator=g)).half()
        scale = 0.02
        got = hip_ops.layer_norm_fp8(x, w, b, 1e-5, scale)
        ln = torch.nn.functional.layer_norm(x.float(), (C,), w.float(), b.float(), 1e-5).half()
        ref = (ln.float() / scale).clamp(-448, 448).to(FP8)
        diff = (got.view(torch.uint8) != ref.view(torch.uint8))
        assert diff.float().mean() < 2e-3          # an fp16 rounding boundary of the norm's output now and then
        assert (_deq(got) - _deq(ref)).abs().max() <= 32.0   # ... and never more than one e4m3 step (top binade: 32)


@pytest.mark.parametrize("H,W,C,heads,ws,shift", [(24, 36, 384, 12, 12, 6), (30, 40, 192, 6, 12, 0), (14, 14, 96, 3, 7, 3)])
def test_window_attention_e4m3_output_equals_the_cast_of_its_fp16_output(H, W, C, heads, ws, shift):
    from codetr import hip_ops

    g = torch.Generator(device=DEV).manual_seed(5)
    qkv = torch.randn(2, H * W, 3 * C, device=DEV, generator=g).half()
    bias = (0.1 * torch.randn(3 * C, device=DEV, generator=g)).half()
    rel = torch.randn(heads, ws * ws, ws * ws, device=DEV, generator=g).half()
    o16 = hip_ops.swin_window_attention(qkv, bias, rel, (H, W), heads, ws, shift)
    for scale in (o16.float().abs().max().item() / 448, 0.003):     # the calibrated scale, and one that saturates
        o8 = hip_ops.swin_window_attention(qkv, bias, rel, (H, W), heads, ws, shift, out_scale=scale)
        assert o8.dtype == FP8 and o8.shape == o16.shape
        ref = hip_ops.cast_fp8(o16, scale)
        # the two instantiations of the kernel agree on the fp16 result except for an ulp in about one element per
        # 100 000 (measured 1 of 663 552 in round 3, 1 of 37 632 / 1 of 460 800 / 0 of 663 552 in round 5), which can then
        # fall on the other side of an e4m3 rounding boundary
        diff = o8.view(torch.uint8) != ref.view(torch.uint8)
        assert diff.float().mean().item() <= max(3e-5, 1.5 / diff.numel())
        assert ((_deq(o8) - _deq(ref)).abs() <= 0.126 * _deq(ref).abs().clamp_min(2.0 ** -6)).all()


def test_fp8_gemm_exact_small_integers_pin_the_lane_maps():
    """integers |v| <= 4 are exact in e4m3 and their K = 256 dot products exact in fp32 / fp16: any error in the operand
    or accumulator lane mapping shows as a wrong integer (asymmetric operands, M and N not multiples of the tile)"""
    from codetr import _cabi

    g = torch.Generator(device=DEV).manual_seed(2)
    Mr, N, K = 300, 264, 256
    x = torch.randint(-4, 5, (Mr, K), device=DEV, generator=g).float()
    w = torch.randint(-3, 4, (N, K), device=DEV, generator=g).float()
    x[:, 0] += 0  # keep
    ws = torch.ones(N, device=DEV)
    out = torch.empty(Mr, N, dtype=torch.float16, device=DEV)
    _cabi.linear_fp8(x.to(FP8), w.to(FP8), ws, 1.0, None, None, None, out)
    ref = x.double() @ w.double().T
    assert ref.abs().max() < 2048
    assert torch.equal(out.double(), ref)


@pytest.mark.parametrize("Mr,N,K,act,res,out8", [
    (76800 // 8, 2304, 768, None, False, False),    # qkv-like
    (5000, 768, 768, None, True, False),            # proj + residual, ragged M
    (4096, 3072, 768, "gelu", False, True),         # fc1: GELU, e4m3 out
    (4100, 768, 3072, None, True, False),           # fc2
    (1000, 1152, 384, "relu", False, False),
])
def test_fp8_gemm_vs_dequantised_float64(Mr, N, K, act, res, out8):
    from codetr import _cabi

    g = torch.Generator(device=DEV).manual_seed(3)
    x8 = (torch.randn(Mr, K, device=DEV, generator=g) * 40).to(FP8)
    w8 = (torch.randn(N, K, device=DEV, generator=g) * 60).to(FP8)
    ws = torch.rand(N, device=DEV, generator=g) * 1e-3 + 1e-4
    xs = 0.013
    bias = torch.randn(N, device=DEV, generator=g).half()
    r = torch.randn(Mr, N, device=DEV, generator=g).half() if res else None
    out_scale = 0.05 if out8 else 0.0
    out = torch.empty(Mr, N, dtype=FP8 if out8 else torch.float16, device=DEV)
    _cabi.linear_fp8(x8, w8, ws, xs, bias, r, act, out, out_scale)
    ref = (_deq(x8) @ _deq(w8).T) * xs * ws.double()[None] + bias.double()[None]
    if act == "gelu":
        ref = torch.nn.functional.gelu(ref)
    elif act == "relu":
        ref = ref.relu()
    if out8:
        ref8 = (ref.float().half().float() / out_scale).clamp(-448, 448).to(FP8)
        d = (_deq(out) - _deq(ref8)).abs()
        assert (d > 0).float().mean() < 5e-3 and d.max() <= 32.0
        return
    if res:
        ref = ref.float().half().double() + r.double()
    err = (out.double() - ref).abs().max().item()
    assert err <= 2e-3 * ref.abs().max().item() + 1e-3, err


def test_ffn_fp8_exact_small_integers_pin_the_lane_maps_and_the_w2_permutation():
    """everything an exact small integer: sparse +-1 W1 rows keep the hidden units <= 15 (exact in e4m3), so the whole
    FFN is exact in fp32 / fp16 -- a wrong lane map, LDS swizzle or W2 column order shows as a wrong integer"""
    from codetr import _cabi, hip_ops

    g = torch.Generator(device=DEV).manual_seed(11)
    for Mr, Hd in ((300, 256), (1000, 2048)):
        x = torch.randint(-2, 3, (Mr, 256), device=DEV, generator=g).half()
        w1 = torch.zeros(Hd, 256, device=DEV)
        cols = torch.randint(0, 256, (Hd, 3), device=DEV, generator=g)
        w1.scatter_(1, cols, (torch.randint(0, 2, (Hd, 3), device=DEV, generator=g) * 2 - 1).float())
        b1 = torch.randint(-2, 3, (Hd,), device=DEV, generator=g).half()
        w2 = torch.randint(-2, 3, (256, Hd), device=DEV, generator=g).float()
        b2 = torch.randint(-3, 4, (256,), device=DEV, generator=g).half()
        ones1, ones2 = torch.ones(Hd, device=DEV), torch.ones(256, device=DEV)
        p = torch.arange(128, device=DEV)
        src = 16 * ((p % 32) // 4) + 4 * (p // 32) + p % 4
        w2p = w2.view(256, -1, 128)[:, :, src].reshape(256, Hd).contiguous()
        out = torch.empty(Mr, 256, dtype=torch.float16, device=DEV)
        _cabi.ffn_fp8(x, w1.to(FP8), ones1, b1, w2p.to(FP8), ones2, b2, out, 1.0, 1.0)
        h = (x.double() @ w1.double().T + b1.double()).relu()
        assert h.max() <= 15
        ref = h @ w2.double().T + b2.double() + x.double()
        assert ref.abs().max() < 2048
        assert torch.equal(out.double(), ref), (Mr, Hd, (out.double() - ref).abs().max().item())


@pytest.mark.parametrize("Mr,Hd,ln_in,ln_out,with_pos", [
    (4096, 2048, True, True, True),      # the encoder layer's (norm, ffn, norm) + pos
    (1000, 2048, False, True, False),    # ragged M
    (777, 1024, False, False, False),
    (130, 128, True, False, False),      # a single chunk
    (130, 256, True, False, False),
])
def test_ffn_fp8_vs_dequantised_float64(Mr, Hd, ln_in, ln_out, with_pos):
    from codetr import hip_ops

    g = torch.Generator(device=DEV).manual_seed(12)
    r = lambda *s, k=1.0: torch.randn(*s, device=DEV, generator=g) * k  # noqa: E731
    x = (r(Mr, 256, k=1.5) + 0.2).half()
    w1, b1 = r(Hd, 256, k=0.08).half(), r(Hd, k=0.1).half()
    w2, b2 = r(256, Hd, k=0.03).half(), r(256, k=0.1).half()
    gi, bi = (1 + r(256, k=0.1)).half(), r(256, k=0.1).half()
    go, bo = (1 + r(256, k=0.1)).half(), r(256, k=0.1).half()
    pos = r(Mr, 256).half() if with_pos else None
    LN = torch.nn.functional.layer_norm
    x1 = LN(x.float(), (256,), gi.float(), bi.float(), 1e-5).half() if ln_in else x
    h16 = (x1.float() @ w1.float().T + b1.float()).relu()
    sx, sh = x1.float().abs().max().item() / 448, h16.max().item() / 448
    got = hip_ops.ffn_fp8(x, w1, b1, w2, b2, sx, sh, ln=(go, bo, 1e-5) if ln_out else None, pos=pos,
                          ln_in=(gi, bi, 1e-5) if ln_in else None)
    got2 = None
    if with_pos:
        got, got2 = got
    # the same arithmetic in float64 on the dequantised operands
    s1 = (w1.float().abs().amax(1) / 448).clamp_min(1e-12)
    s2 = (w2.float().abs().amax(1) / 448).clamp_min(1e-12)
    w1q, w2q = (w1.float() / s1[:, None]).to(FP8), (w2.float() / s2[:, None]).to(FP8)
    xq = (x1.float() / sx).clamp(-448, 448).to(FP8)
    h = ((_deq(xq) @ _deq(w1q).T) * (s1.double() * sx)[None] + b1.double()[None]).relu()
    hq = (h.float() / sh).clamp(-448, 448).to(FP8)
    y = (_deq(hq) @ _deq(w2q).T) * (s2.double() * sh)[None] + b2.double()[None]
    y = ypre = (y.float().half().float() + x1.float()).half()
    if ln_out:
        y = LN(y.float(), (256,), go.float(), bo.float(), 1e-5).half()
    # an hq value on an e4m3 rounding boundary may fall the other way under fp32 accumulation order: one e4m3 step (up to
    # 32 * sh in the top binade) of one hidden unit times one W2 entry lands on that row's 256 outputs -- rare, bounded
    d = (got.double() - y.double()).abs()
    tol = 4e-3 * y.double().abs().max().item() + 2e-3
    gain = (go.float().abs().max() / ypre.float().std(1).min()).item() if ln_out else 1.0
    assert (d > tol).any(1).float().mean().item() <= 0.02, (d > tol).any(1).float().mean().item()
    assert d.max().item() <= tol + 2 * 32 * sh * w2.float().abs().max().item() * max(gain, 1.0), d.max().item()
    if with_pos:
        assert torch.equal(got2, (got.float() + pos.float()).half())
    # and the quantised FFN is close to the fp16 one (what the model-level tolerance budget is made of)
    ref16 = hip_ops.ffn_fused(x, w1, b1, w2, b2, ln=(go, bo, 1e-5) if ln_out else None,
                              ln_in=(gi, bi, 1e-5) if ln_in else None) if Hd % 64 == 0 else None
    rel = ((got.float() - ref16.float()).norm() / ref16.float().norm()).item()
    assert rel <= 6e-2, rel


def test_midsize_model_fp8_vs_fp32_oracle():
    import codetr
    from codetr import _cabi, fp8

    cfg = os.path.join(F.CFG_DIR, "co_dino_5scale_swin_l_16xb1_16e_o365tococo.py")
    torch.manual_seed(0)
    model = codetr.build_CoDETR(cfg, None, "cpu")
    spec = [(k, tuple(v.shape)) for k, v in model.named_parameters()]
    full = dict(model.state_dict())
    full.update(seeded_params(spec, 21, scale=1.0))
    model.load_state_dict(full)
    H, W = 512, 768
    g = torch.Generator().manual_seed(9)
    img = torch.randn(4, 3, H, W, generator=g)
    mask = torch.zeros(4, H, W)
    mask[1, :, int(W * 0.8):] = 1
    cap_o = {}
    with torch.no_grad():
        M.codetr_forward(full, img[:2], mask[:2], forced_topk=partial(valid_topk, bound=50.0), capture=cap_o)
    model = model.to(DEV).half().eval()
    x, m = img.to(DEV).half(), mask.to(DEV).half()
    old = hip_ops_min_tiles(8)      # 2 x 512x768: the stage-3 GEMMs have 18-54 output tiles; engage fp8 on stages 1-3
    from codetr import hip_ops
    old_rows, hip_ops.FFN_FUSED_MIN_ROWS = hip_ops.FFN_FUSED_MIN_ROWS, 8192   # 2 x 8184 encoder rows: fused (fp8) FFN
    try:
        assert fp8.calibrate(model, x[2:], m[2:]) == 24      # calibration on images 2, 3; evaluation on 0, 1
        assert sum(hasattr(f, "_fp8_scales") for f in fp8._ffns(model)) == 6    # the encoder's; the decoder's run unfused
        fp8.enable(model)
        before = dict(_cabi.CALLS)
        cap = {}
        with torch.no_grad():
            model(x[:2], m[:2], forced_topk_indices=cap_o["topk_indices"].to(DEV), capture=cap)
        torch.cuda.synchronize()
        n8 = _cabi.CALLS["linear_fp8"] - before["linear_fp8"]
        assert n8 >= 4 * 20 and _cabi.CALLS["layernorm_fp8"] - before["layernorm_fp8"] == n8 // 2, n8
        assert _cabi.CALLS["ffn_fp8"] - before["ffn_fp8"] == 6 and _cabi.CALLS["ffn_fused"] == before["ffn_fused"]
        errs = {}
        for i, (a, b) in enumerate(zip(cap["backbone_feats"], cap_o["backbone_feats"])):
            errs[f"backbone{i}"] = assert_close_lowp(a.float().cpu().numpy(), b.numpy(), 1.0, None, f"fp8 backbone {i}")
        for k in ("memory", "enc_outputs_class", "final_state", "outputs_coords"):
            errs[k] = assert_close_lowp(cap[k].float().cpu().numpy(), cap_o[k].numpy(), 1.0, None, "fp8 " + k)
        errs.update(detection_agreement(cap, cap_o, H, W))
        print("fp8 model errors", {k: round(v, 4) for k, v in errs.items()})
        from test_timed_route_gpu import _report
        _report("midsize_fp8", errs)
        # e4m3 carries 3 mantissa bits: every GEMM output comes out with a few per cent of relative noise, and 22 blocks
        # of it accumulate on the residual stream (measured: 2e-2 after stage 1, 8e-2 after the 18 blocks of stage 2)
        assert errs["backbone0"] <= 1e-2 and errs["backbone1"] <= 5e-2, errs     # stage 0 is fp16; stage 1: 2 fp8 blocks
        assert max(errs["backbone2"], errs["backbone3"], errs["memory"]) <= 1.5e-1, errs
        assert errs["box_err_px_mean"] <= 0.02 * W, errs
    finally:
        hip_ops_min_tiles(old)
        hip_ops.FFN_FUSED_MIN_ROWS = old_rows
        fp8.enable(model, False)


def hip_ops_min_tiles(n):
    from codetr import hip_ops

    old = hip_ops.FP8_MIN_TILES
    hip_ops.FP8_MIN_TILES = n
    return old
