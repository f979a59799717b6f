"""MX block-scaled e4m3 path (BASELINE config 5 with the activation scales the hardware applies itself: one e8m0 exponent
per 32 K-elements, v_mfma_scale_f32_16x16x128_f8f6f4).  No reference counterpart; checked against exact restatements:

  * the scale layout and the instruction's lane / byte semantics by an EXACT case: small-integer e4m3 operands with a
    different power-of-two scale on every (row, block) -- any slip in the layout, the op_sel byte or the lane map gives a
    wrong integer;
  * the GEMM against a float64 product of the dequantised operands (fp32 accumulation order: 2e-3 of the range), ragged M;
  * the producers (cast, LayerNorm, window attention, the GELU epilogue with block scales along N): the block exponent is
    the smallest one that brings the block's maximum into e4m3 range, the payload is the e4m3 rounding of value / 2^e --
    i.e. dequantised values within half an e4m3 step (2^-4 relative) of the fp16 tensor the fp16 path would have written;
  * model level (2 x 512x768, forced proposals) against the fp32 oracle, next to the static-scale scheme of round 2:
    bounds and the measured finding (block scales = static scales in accuracy: mantissa-limited) in
    test_midsize_model_fp8mx_vs_fp32_oracle."""
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


def _scale_index(Mr, K):
    m = torch.arange(Mr, device=DEV)[:, None]
    kb = torch.arange(K // 32, device=DEV)[None, :]
    MB = -(-Mr // 128)
    return ((((kb >> 2) * MB + (m >> 7)) * 64 + (kb & 3) * 16 + (m & 15)) * 8 + ((m >> 4) & 7)).long()


def _pack_scales(exps, Mr, K):
    """exps [Mr, K / 32] integer exponents -> the uint8 scale tensor in the kernels' layout"""
    from codetr import _cabi

    s = torch.full((_cabi.mx_scale_bytes(Mr, K),), 0x7F, dtype=torch.uint8, device=DEV)
    s[_scale_index(Mr, K).reshape(-1)] = (exps.reshape(-1) + 127).to(torch.uint8)
    return s


def _check_block_quantisation(x_ref, x8, scales, what, payload_mismatch=0.0):
    """x_ref: the fp16-valued tensor [rows, C] the producer quantised; (x8, scales) its MX form.  payload_mismatch: share of
    the e4m3 bytes allowed to differ from the quantisation of x_ref (0: none) -- for a producer whose x_ref comes from ANOTHER
    instantiation of the same kernel, which agrees on the 16-bit value except for an ulp in about one element per 100 000"""
    from codetr import hip_ops

    rows, C = x_ref.shape
    e = scales.long()[_scale_index(rows, C)] - 127                                     # [rows, C / 32]
    amax = x_ref.float().abs().view(rows, C // 32, 32).amax(-1)
    # smallest exponent with amax * 2^-e <= 448 (blocks of zeros: the floor, byte 1)
    want = torch.where(amax > 0, torch.ceil(torch.log2(amax.double() / 448.0)).long(), torch.full_like(e, -126)).clamp(-126, 126)
    if payload_mismatch == 0.0:
        assert torch.equal(e, want), f"{what}: block exponents differ in {(e != want).sum().item()} blocks"
    else:   # (an ulp more or less on a block's maximum can move its exponent)
        assert (e != want).float().mean().item() <= payload_mismatch and (e - want).abs().max().item() <= 1, what
    deq = hip_ops.mx_dequant(x8, scales).double().view(rows, C)
    ref8 = (x_ref.double().view(rows, C // 32, 32) / torch.exp2(e.double())[:, :, None]).float().clamp(-448, 448).to(FP8)
    same = x8.view(torch.uint8).view(rows, C) == ref8.view(torch.uint8).view(rows, C)
    assert (~same).float().mean().item() <= payload_mismatch, f"{what}: payload differs in {(~same).sum().item()} elements"
    err = (deq - x_ref.double()).abs()
    bound = 2.0 ** -4 * x_ref.double().abs() + torch.exp2(e.double() - 10).repeat_interleave(32, 1)   # + the block's subnormal step
    if payload_mismatch > 0.0:
        bound = torch.where(same, bound, 3 * bound)     # a differing element is one e4m3 step away, not half a step
    assert (err <= bound).all(), f"{what}: {(err > bound).sum().item()} elements off by more than half an e4m3 step"


@pytest.mark.parametrize("rows,C", [(1000, 384), (257, 768), (64, 128)])
def test_cast_and_layernorm_produce_block_scales(rows, C):
    from codetr import hip_ops

    g = torch.Generator(device=DEV).manual_seed(rows + C)
    x = (torch.randn(rows, C, device=DEV, generator=g) * torch.exp2(torch.randint(-6, 7, (rows, 1), device=DEV, generator=g).float())).half()
    x[3, :64] = 0          # two all-zero blocks
    x8, sx = hip_ops.cast_fp8mx(x)
    _check_block_quantisation(x, x8, sx, "cast")
    w = (1 + 0.1 * torch.randn(C, device=DEV, generator=g)).half()
    b = (0.1 * torch.randn(C, device=DEV, generator=g)).half()
    y8, sy = hip_ops.layer_norm_fp8mx(x, w, b, 1e-5)
    ln = hip_ops.layer_norm(x, w, b, 1e-5)
    _check_block_quantisation(ln, y8, sy, "layernorm")


def test_gemm_exact_integers_with_a_different_scale_on_every_block():
    """integers |v| <= 4 (exact in e4m3) x 2^e with e in [-3, 3] drawn per (row, 32-block): the exact products stay below
    2^24 * 2^-3 and are multiples of 2^-3 -> exact in the fp32 accumulator; the fp16 output is their one rounding"""
    from codetr import _cabi

    g = torch.Generator(device=DEV).manual_seed(5)
    Mr, N, K = 300, 264, 384
    xi = torch.randint(-4, 5, (Mr, K), device=DEV, generator=g).float()
    w = torch.randint(-3, 4, (N, K), device=DEV, generator=g).float()
    e = torch.randint(-3, 4, (Mr, K // 32), device=DEV, generator=g)
    sx = _pack_scales(e, Mr, K)
    ws = torch.full((N,), 2.0 ** -8, device=DEV)            # keeps |y| < 2048 / fp16-exact
    out = torch.empty(Mr, N, dtype=torch.float16, device=DEV)
    _cabi.linear_fp8mx(xi.to(FP8), sx, w.to(FP8), ws, None, None, None, out)
    x_true = (xi.view(Mr, K // 32, 32) * torch.exp2(e.float())[:, :, None]).view(Mr, K)
    ref = (x_true.double() @ w.double().T) * 2.0 ** -8      # exact in float64 -- and in the kernel's fp32 accumulator
    assert ref.abs().max() < 60000
    want = ref.to(torch.float16)                              # ... so the only rounding is the output's (both RNE)
    assert torch.equal(out, want), f"{(out != want).sum().item()} of {ref.numel()} outputs differ"


@pytest.mark.parametrize("Mr,N,K,act,res", [
    (76800 // 8, 2304, 768, None, False),    # qkv-like
    (5000, 768, 768, None, True),            # proj + residual, ragged M
    (4100, 768, 3072, None, True),           # fc2
    (1000, 1152, 384, "relu", False),
])
def test_gemm_vs_dequantised_float64(Mr, N, K, act, res):
    from codetr import _cabi, hip_ops

    g = torch.Generator(device=DEV).manual_seed(3)
    x = (torch.randn(Mr, K, device=DEV, generator=g) * torch.exp2(torch.randn(Mr, 1, device=DEV, generator=g) * 2)).half()
    x8, sx = hip_ops.cast_fp8mx(x)
    w8 = (torch.randn(N, K, device=DEV, generator=g) * 60).to(FP8)
    ws = torch.rand(N, device=DEV, generator=g) * 1e-3 + 1e-4
    bias = torch.randn(N, device=DEV, generator=g).half()
    r = torch.randn(Mr, N, device=DEV, generator=g).half() if res else None
    out = torch.empty(Mr, N, dtype=torch.float16, device=DEV)
    _cabi.linear_fp8mx(x8, sx, w8, ws, bias, r, act, out)
    ref = (hip_ops.mx_dequant(x8, sx).double() @ w8.float().double().T) * ws.double()[None] + bias.double()[None]
    if act == "relu":
        ref = ref.relu()
    if res:
        ref = ref.float().half().double() + r.double()
    err = (out.double() - ref).abs().max().item()
    assert err <= 2e-3 * ref.abs().max().item() + 1e-3, err


def test_gelu_epilogue_writes_block_scales_along_n_for_the_next_gemm():
    """fc1 (GELU, e4m3 + scales out) -> fc2 consumes them: the chained result equals fc2 applied to the dequantised hidden"""
    from codetr import _cabi, hip_ops

    g = torch.Generator(device=DEV).manual_seed(8)
    Mr, C, Hd = 4100, 768, 3072
    x8, sx = hip_ops.cast_fp8mx((torch.randn(Mr, C, device=DEV, generator=g) * 2).half())
    w1 = (torch.randn(Hd, C, device=DEV, generator=g) * 60).to(FP8)
    s1 = torch.rand(Hd, device=DEV, generator=g) * 4e-4 + 1e-4
    b1 = torch.randn(Hd, device=DEV, generator=g).half()
    h8 = torch.empty(Mr, Hd, dtype=FP8, device=DEV)
    sh = torch.empty(_cabi.mx_scale_bytes(Mr, Hd), dtype=torch.uint8, device=DEV)
    _cabi.linear_fp8mx(x8, sx, w1, s1, b1, None, "gelu", h8, sh)
    hid = torch.nn.functional.gelu((hip_ops.mx_dequant(x8, sx).double() @ w1.float().double().T) * s1.double()[None] + b1.double()[None])
    hid16 = hid.float().half()
    # the hidden tensor as MX: exponents / payload as the cast kernel would produce from the fp16 hidden (an fp16 rounding
    # boundary of the GELU output now and then: compare dequantised values)
    deq = hip_ops.mx_dequant(h8, sh).double()
    err = (deq - hid16.double()).abs()
    e = sh.long()[_scale_index(Mr, Hd)] - 127
    bound = 2.0 ** -3 * hid16.double().abs() + torch.exp2(e.double() - 9).repeat_interleave(32, 1)
    assert (err <= bound).float().mean() > 0.9999 and torch.isfinite(deq).all()
    w2 = (torch.randn(C, Hd, device=DEV, generator=g) * 60).to(FP8)
    s2 = torch.rand(C, device=DEV, generator=g) * 1e-4 + 1e-5
    out = torch.empty(Mr, C, dtype=torch.float16, device=DEV)
    _cabi.linear_fp8mx(h8, sh, w2, s2, None, None, None, out)
    ref = (deq @ w2.float().double().T) * s2.double()[None]
    assert (out.double() - ref).abs().max().item() <= 2e-3 * ref.abs().max().item() + 1e-3


@pytest.mark.parametrize("H,W,C,heads,ws,shift", [(24, 36, 384, 12, 12, 6), (30, 40, 128, 4, 12, 0)])
def test_window_attention_block_scaled_output(H, W, C, heads, ws, shift):
    from codetr import hip_ops

    g = torch.Generator(device=DEV).manual_seed(H * W)
    B = 2
    qkv = (torch.randn(B, H * W, 3 * C, device=DEV, generator=g) * torch.exp2(torch.randn(B, H * W, 1, device=DEV, generator=g))).half()
    bias = (0.1 * torch.randn(3 * C, device=DEV, generator=g)).half()
    rel = (0.5 * torch.randn(heads, ws * ws, ws * ws, device=DEV, generator=g)).half()
    o16 = hip_ops.swin_window_attention(qkv, bias, rel, (H, W), heads, ws, shift)
    o8, so = hip_ops.swin_window_attention(qkv, bias, rel, (H, W), heads, ws, shift, out_mx=True)
    # (o16 comes from the 16-bit instantiation of the kernel, the block-scaled output from another one)
    _check_block_quantisation(o16.reshape(-1, C), o8.reshape(-1, C), so, "window attention", payload_mismatch=3e-5)


def test_midsize_model_fp8mx_vs_fp32_oracle():
    """the Swin stage 1-3 linears on MX-scaled e4m3 (no calibration) + the encoder FFN on its calibrated static scales,
    against the fp32 oracle; the same with the round-2 static Swin scales alongside.  Calibration (FFN only / static mode)
    sees two OTHER images than the evaluation."""
    import codetr
    from codetr import _cabi, fp8, hip_ops

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
    old_tiles, hip_ops.FP8_MIN_TILES = hip_ops.FP8_MIN_TILES, 8
    old_rows, hip_ops.FFN_FUSED_MIN_ROWS = hip_ops.FFN_FUSED_MIN_ROWS, 8192
    report = {}
    try:
        fp8.calibrate(model, x[2:], m[2:])          # images 2, 3; evaluation on 0, 1
        for mode, ffn in (("mx", False), ("mx", True), ("static", True)):
            fp8.enable(model, mode=mode)
            if not ffn:
                for f in fp8._ffns(model):
                    f.fp8_mode = None
            before = dict(_cabi.CALLS)
            cap = {}
            with torch.no_grad():
                model(x[:2], m[:2], forced_topk_indices=cap_o["topk_indices"].to(DEV), capture=cap)
            torch.cuda.synchronize()
            assert _cabi.CALLS["linear_fp8"] - before["linear_fp8"] >= 80
            assert (_cabi.CALLS["ffn_fp8"] - before["ffn_fp8"] == 6) == ffn
            errs = {}
            for i, (a, b) in enumerate(zip(cap["backbone_feats"], cap_o["backbone_feats"])):
                errs[f"backbone{i}"] = assert_close_lowp(a.float().cpu().numpy(), b.numpy(), 1.0, None, f"fp8 backbone {i}")
            for k in ("memory", "enc_outputs_class", "final_state", "outputs_coords"):
                errs[k] = assert_close_lowp(cap[k].float().cpu().numpy(), cap_o[k].numpy(), 1.0, None, "fp8 " + k)
            errs.update(detection_agreement(cap, cap_o, H, W))
            report[f"{mode}{'+ffn8' if ffn else ''}"] = {k: float(f"{v:.3e}") for k, v in errs.items()}
        print("fp8 model errors", report)
        from test_timed_route_gpu import _report
        for k, v in report.items():
            _report("midsize_fp8_" + k, v)
        # Measured (profiles/r03_parity_report.json): block scales give the SAME errors as calibrated static scales
        # (memory 7.6e-2 vs 7.7e-2 with the e4m3 FFN, 4.3e-2 with the FFN in fp16): on these weights the error is the 3-bit
        # mantissa of e4m3 on both operands -- ~5 % of relative noise on every incoherent dot product, whatever its scale --
        # not the dynamic range.  What block scales buy is robustness: no calibration pass, no saturation on inputs the
        # calibration did not see.
        mx = report["mx"]
        assert mx["backbone1"] <= 6e-2 and max(mx["backbone2"], mx["backbone3"]) <= 1.2e-1 and mx["memory"] <= 6e-2, mx
        assert report["mx+ffn8"]["memory"] <= 1e-1 and report["mx+ffn8"]["box_err_px_mean"] <= 0.02 * W, report
        assert report["mx+ffn8"]["memory"] <= 1.05 * report["static+ffn8"]["memory"], report   # not worse than static scales
    finally:
        hip_ops.FP8_MIN_TILES = old_tiles
        hip_ops.FFN_FUSED_MIN_ROWS = old_rows
        fp8.enable(model, False)


def test_per_gemm_selection_and_the_accurate_preset():
    """codetr/fp8.py `select`: the e4m3 GEMMs of a Swin block can be chosen one by one (the others stay on the fp16
    kernels, each fed by the right producer), and the `accurate` preset -- the largest selection of the sensitivity map
    (profiles/r04_fp8_sensitivity.json) under an encoder-memory error of 2e-2 against the fp16 product -- meets that bound
    on the proxy's case; the full selection does not (measured 7.8e-2: why config 5 is a FAST mode, not the accurate one)."""
    import proxy_ap_case as C
    from codetr import _cabi, fp8, hip_ops

    H, W = 512, 768
    model, _ = C.build()
    model = model.to(device=DEV, dtype=torch.float16)
    img, mask = C.images(2, H, W)
    img, mask = img.to(DEV, torch.float16), mask.to(DEV, torch.float16)
    enc = model.query_head.transformer.encoder
    orig, mems = enc.forward_bf, []
    enc.forward_bf = lambda *a, **k: (mems.append(orig(*a, **k)), mems[-1])[1]
    old_min = hip_ops.FP8_MIN_TILES
    hip_ops.FP8_MIN_TILES = 0      # the proxy's GEMMs are smaller than the production threshold: this is an accuracy test
    try:
        with torch.no_grad():
            def memory(select, ffn=False):
                fp8.enable(model, select is not None, "mx", select=select if select is not None else "all", ffn=ffn)
                mems.clear()
                before = _cabi.CALLS["linear_fp8"]
                model(img, mask)
                fp8.enable(model, False)
                return mems[0].float(), _cabi.CALLS["linear_fp8"] - before

            ref, n0 = memory(None)
            assert n0 == 0
            rel = lambda m: float(((m - ref).flatten(1).norm(dim=1) / ref.flatten(1).norm(dim=1)).max())  # noqa: E731
            # one GEMM of the two stage-3 blocks: exactly two e4m3 launches, a small error
            m1, n1 = memory({3: ("fc2",)})
            assert n1 == 2 and 0 < rel(m1) < 2e-2
            # fc2 without fc1 takes its e4m3 input from the cast kernel, fc1 without fc2 writes fp16: both orders work
            m2, n2 = memory({3: ("fc1",)})
            assert n2 == 2 and 0 < rel(m2) < 2e-2
            macc, nacc = memory("accurate")
            assert nacc == 2 * 3 + 2 * 1 and rel(macc) <= 2e-2, rel(macc)
            mall, nall = memory("all")
            assert nall == 4 * 22 and rel(mall) > 3e-2, rel(mall)
    finally:
        hip_ops.FP8_MIN_TILES = old_min
        enc.forward_bf = orig
