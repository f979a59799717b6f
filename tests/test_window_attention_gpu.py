"""GPU parity of the fused Swin window-attention kernel (C ABI codetr_window_attention_f16) against the
reference's tensor program for the same step -- pad (zeros -> qkv bias), roll, window partition,
q*scale @ k^T + relative-position bias + (-100) shift mask, softmax, @ v, window reverse, roll back, crop
(reference codetr/swin.py:191-252, 92-112) -- written out in plain PyTorch fp32 below.

Tolerance: inputs are fp16-exact; scores/softmax/accumulation are fp32 on both sides; the kernel
rounds the probabilities to fp16 before P.V (as the reference's fp16 path does), so outputs agree to
~2^-10 relative of the value scale: atol 3e-3 on O(1) values, rtol 5e-3."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def reference_window_attention(qkv, qkv_bias, table, rel_index, H, W, nH, ws, shift):
    """fp32 restatement on [B, H*W, 3C] -> [B, H*W, C]"""
    B, L, C3 = qkv.shape
    C = C3 // 3
    hd = C // nH
    x = qkv.float().view(B, H, W, C3)
    pad_b, pad_r = (-H) % ws, (-W) % ws
    Hp, Wp = H + pad_b, W + pad_r
    full = qkv_bias.float().view(1, 1, 1, C3).expand(B, Hp, Wp, C3).clone()  # pad tokens = qkv(0) = bias
    full[:, :H, :W] = x
    mask = None
    if shift > 0:
        full = torch.roll(full, (-shift, -shift), (1, 2))
        region = torch.zeros(Hp, Wp, device=qkv.device)
        cnt = 0
        for hs in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
            for wsl in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
                region[hs, wsl] = cnt
                cnt += 1
        r = region.view(Hp // ws, ws, Wp // ws, ws).permute(0, 2, 1, 3).reshape(-1, ws * ws)
        mask = (r[:, None, :] != r[:, :, None]).float() * -100.0
    win = full.view(B, Hp // ws, ws, Wp // ws, ws, C3).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws * ws, C3)
    N = ws * ws
    q, k, v = win.view(-1, N, 3, nH, hd).permute(2, 0, 3, 1, 4)
    attn = (q * hd ** -0.5) @ k.transpose(-2, -1)
    bias = table.float()[rel_index.view(-1)].view(N, N, nH).permute(2, 0, 1)
    attn = attn + bias[None]
    if mask is not None:
        nW = mask.shape[0]
        attn = (attn.view(B, nW, nH, N, N) + mask[None, :, None]).view(-1, nH, N, N)
    o = (attn.softmax(-1) @ v).transpose(1, 2).reshape(-1, N, C)
    o = o.view(B, Hp // ws, Wp // ws, ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(B, Hp, Wp, C)
    if shift > 0:
        o = torch.roll(o, (shift, shift), (1, 2))
    return o[:, :H, :W].reshape(B, H * W, C)


def _case(B, H, W, nH, ws, shift, seed=0, bias_scale=1.0, dtype=torch.float16):
    from codetr import hip_ops
    from codetr.swin import WindowMSA

    C = nH * 32
    g = torch.Generator(device=DEV).manual_seed(seed)
    qkv = torch.randn(B, H * W, 3 * C, device=DEV, generator=g).to(dtype)
    qkv_bias = (0.5 * torch.randn(3 * C, device=DEV, generator=g)).to(dtype)
    m = WindowMSA(C, nH, (ws, ws)).to(DEV).to(dtype)
    with torch.no_grad():
        m.relative_position_bias_table.copy_(bias_scale * torch.randn(m.relative_position_bias_table.shape, device=DEV,
                                                                      generator=g))
    out = hip_ops.swin_window_attention(qkv, qkv_bias, m.relative_position_bias(), (H, W), nH, ws, shift)
    torch.cuda.synchronize()
    ref = reference_window_attention(qkv, qkv_bias, m.relative_position_bias_table, m.relative_position_index, H, W,
                                     nH, ws, shift)
    assert out.dtype == dtype
    if dtype == torch.float16:
        torch.testing.assert_close(out.float(), ref, rtol=5e-3, atol=3e-3)
    else:   # bf16: probabilities and the output carry 8 significant bits -> 2^-8 of the value scale
        torch.testing.assert_close(out.float(), ref, rtol=2e-2, atol=2e-2)


@pytest.mark.parametrize("shift", [0, 6])
@pytest.mark.parametrize("B,H,W,nH", [
    (1, 12, 12, 1),      # exactly one window
    (2, 24, 36, 2),      # no padding
    (1, 20, 31, 6),      # pads both ways (24 x 36), stage-0 head count
    (3, 13, 12, 3),      # pad rows only, odd head count (wave tail)
    (1, 40, 60, 4),      # 1920x1280 stage-3 map: pads H 40 -> 48
])
def test_window12(B, H, W, nH, shift):
    _case(B, H, W, nH, 12, shift, seed=H + W + shift)


@pytest.mark.parametrize("ws,shift,H,W", [(7, 0, 14, 14), (7, 3, 16, 20), (8, 4, 17, 9), (4, 2, 11, 15), (4, 0, 8, 8)])
def test_other_window_sizes(ws, shift, H, W):
    _case(2, H, W, 2, ws, shift, seed=ws)


@pytest.mark.parametrize("ws,shift,B,H,W,nH", [(12, 0, 2, 24, 36, 2), (12, 6, 1, 20, 31, 6), (7, 3, 2, 16, 20, 2),
                                               (12, 6, 1, 40, 60, 4)])
def test_bf16(ws, shift, B, H, W, nH):
    """the bf16 instantiation (codetr_window_attention_bf16) against the same fp32 restatement"""
    _case(B, H, W, nH, ws, shift, seed=ws + H, dtype=torch.bfloat16)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["f16", "bf16"])
@pytest.mark.parametrize("ws,shift,H,W,nH", [(12, 0, 24, 36, 2), (12, 6, 20, 31, 6), (8, 4, 17, 9, 2), (4, 2, 11, 15, 3), (4, 0, 8, 8, 2)])
def test_lane_order_bias_is_bit_identical(ws, shift, H, W, nH, dtype):
    """bias_layout 1 of codetr_window_attention_ex (the table permuted into the kernel's lane order, what the model
    launches) against layout 0 (the reference's [nH, N, N]): same values through different loads -> the same bits; and
    the index the permutation comes from"""
    from codetr import _cabi, hip_ops
    from codetr.swin import WindowMSA

    idx = _cabi.window_attention_bias_index(ws)
    NT = ws * ws // 16
    assert idx == [16 * kt + 4 * g + r for g in range(4) for kt in range(NT) for r in range(4)]
    assert _cabi.window_attention_bias_index(7) is None
    C = nH * 32
    g = torch.Generator(device=DEV).manual_seed(ws + H)
    qkv = torch.randn(2, H * W, 3 * C, device=DEV, generator=g).to(dtype)
    qkv_bias = (0.5 * torch.randn(3 * C, device=DEV, generator=g)).to(dtype)
    m = WindowMSA(C, nH, (ws, ws)).to(DEV).to(dtype)
    with torch.no_grad():
        m.relative_position_bias_table.copy_(torch.randn(m.relative_position_bias_table.shape, device=DEV, generator=g))
    outs = []
    for lane in (True, False):
        hip_ops.WINDOW_BIAS_LANE = lane
        try:
            outs.append(hip_ops.swin_window_attention(qkv, qkv_bias, m.relative_position_bias(), (H, W), nH, ws, shift))
        finally:
            hip_ops.WINDOW_BIAS_LANE = True
    assert torch.equal(outs[0].view(torch.int16), outs[1].view(torch.int16))


def test_large_logits_and_mask_dominance():
    """big relative-position biases: softmax must stay finite and the -100 mask must still win."""
    _case(1, 30, 30, 2, 12, 6, seed=9, bias_scale=8.0)


def test_swin_block_fused_matches_oracle():
    """ShiftWindowMSA through the fused path inside a 2-stage Swin (window 12, head_dim 32, padding at both
    stages) against the CPU oracle on the same weights."""
    import numpy as np

    import codetr_fp32 as M
    from codetr.swin import SwinTransformer
    from helpers_model import assert_close_lowp, seeded_params

    s = SwinTransformer(pretrain_img_size=64, embed_dims=64, depths=(2, 2), num_heads=(2, 4), window_size=12,
                        strides=(4, 2), out_indices=(0, 1), drop_path_rate=0.0, patch_norm=True)
    spec = [(k, tuple(v.shape)) for k, v in s.named_parameters()]
    sd = seeded_params(spec, 11, scale=1.5)
    s.load_state_dict(sd, strict=False)
    s = s.to(DEV).half().eval()
    img = torch.randn(2, 3, 100, 150, generator=torch.Generator().manual_seed(3))  # 25 x 38 tokens -> pads to 36 x 48
    from codetr import _cabi

    before = dict(_cabi.CALLS)
    with torch.no_grad():
        outs = s(img.to(DEV).half())
    assert _cabi.CALLS["window_attention"] - before["window_attention"] == 4  # every block took the fused kernel
    # qkv, proj, fc1, fc2 per block + 1 merge + the stem (patch gather + GEMM)
    assert _cabi.CALLS["linear"] - before["linear"] == 4 * 4 + 1 + 1
    assert _cabi.CALLS["patch_im2col"] - before["patch_im2col"] == 1
    assert _cabi.CALLS["layernorm"] > before["layernorm"]
    ref = M.swin_forward({"backbone." + k: v for k, v in sd.items()}, img, num_heads=(2, 4), window_size=12,
                         out_indices=(0, 1))
    for i in range(2):
        assert_close_lowp(outs[i].float().cpu().numpy(), ref[i].numpy(), 1e-2, 0.1, f"swin out {i}")
