"""GPU parity of the MSDA backward kernels (C ABI codetr_msda_backward_*, torch.ops.codetr.
multi_scale_deformable_attention_backward + the autograd registration):

  * against gradients the IMPORTED reference produced (torch.autograd through its differentiable PyTorch formulation,
    ops.py:129-186; tests/golden/msda_grad.npz) on the reference's own gradient-test geometry for every channel count it
    checks -- 4, 30, 32, 64, 71, 1025 (tests/test_multi_scale_deformable_attention.py:367-414) -- and a model-shaped case;
  * by torch.autograd.gradcheck in fp64 with the reference's settings (eps 1e-6, atol 1e-2), same channel counts;
  * against the closed-form CPU oracle (oracle/msda_backward_oracle.py) and an autograd restatement on further shapes.

fp64: 1e-10 relative; fp32: 1e-4; fp16 (fp32 arithmetic, fp16 atomics / stores): 2e-2 of the gradient scale."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _golden_case(tag):
    g = np.load(os.path.join(GOLDEN, "msda_grad.npz"))
    d = {k[len(tag) + 1:]: torch.from_numpy(g[k]) for k in g.files if k.startswith(tag + ".")}
    ss = d["shapes"]
    d["level_start"] = torch.cat((ss.new_zeros(1), ss.prod(1).cumsum(0)[:-1]))
    return d


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-10), (torch.float32, 1e-4), (torch.float16, 2e-2)])
@pytest.mark.parametrize("tag", ["c4", "c30", "c32", "c64", "c71", "c1025", "model"])
def test_msda_backward_vs_reference_autograd_golden(tag, dtype, tol):
    """inputs of the fixture are fp16-representable, so every dtype differentiates the same function"""
    from codetr import _cabi

    d = _golden_case(tag)
    ss, ls = d["shapes"].to(DEV), d["level_start"].to(DEV)
    v, l_, w_ = (d[k].to(dtype).to(DEV).requires_grad_(True) for k in ("value", "loc", "w"))
    before = _cabi.CALLS["msda_backward"]
    out = torch.ops.codetr.multi_scale_deformable_attention(v, ss, ls, l_, w_, 64)
    ftol = {torch.float64: 1e-12, torch.float32: 1e-5, torch.float16: 2e-3}[dtype]
    assert (out.double().cpu() - d["out"]).abs().max() <= ftol * d["out"].abs().max() + (1e-7 if dtype != torch.float64 else 0)
    out.backward(d["go"].to(dtype).to(DEV))
    assert _cabi.CALLS["msda_backward"] == before + 1
    for name, got, ref in (("value", v.grad, d["grad_value"]), ("loc", l_.grad, d["grad_loc"]), ("w", w_.grad, d["grad_w"])):
        scale = ref.abs().max().item() + 1e-300
        err = (got.double().cpu() - ref).abs().max().item()
        assert err <= tol * scale, f"{tag} grad_{name}: max err {err:.3e} vs scale {scale:.3e}"


@pytest.mark.parametrize("channels", [4, 30, 32, 64, 71, 1025])
def test_gradient_numerical_as_the_reference_does(channels):
    """the reference's own test (tests/test_multi_scale_deformable_attention.py:367-414): gradcheck in double"""
    from torch.autograd import gradcheck

    torch.manual_seed(channels)
    N, M, Lq, L, P = 1, 2, 2, 2, 2
    shapes = torch.as_tensor([(3, 2), (2, 1)], dtype=torch.long, device=DEV)
    lsi = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
    S = int(shapes.prod(1).sum())
    value = (torch.rand(N, S, M, channels, device=DEV) * 0.01).double().requires_grad_(True)
    loc = torch.rand(N, Lq, M, L, P, 2, device=DEV).double().requires_grad_(True)
    aw = torch.rand(N, Lq, M, L, P, device=DEV) + 1e-5
    aw = (aw / aw.sum(-1, keepdim=True).sum(-2, keepdim=True)).double().requires_grad_(True)
    assert gradcheck(torch.ops.codetr.multi_scale_deformable_attention, (value, shapes, lsi, loc, aw, 2), eps=1e-6, atol=1e-2)


@pytest.mark.parametrize("D", [30, 71, 96])
def test_generic_channel_kernel_vs_backward_oracle(D):
    """the wave-per-pair kernel on a multi-image case with out-of-range samples, against the closed-form CPU oracle"""
    import msda_backward_oracle as BO

    shapes = [(5, 7), (3, 4)]
    value, loc, w, go, S, L = _case(2, 3, D, 5, shapes, 3, torch.float64, seed=11, border=True)
    ss = torch.tensor(shapes, dtype=torch.int64)
    ls = torch.cat((ss.new_zeros(1), ss.prod(1).cumsum(0)[:-1]))
    gv, gl, gw = BO.msda_backward(value.numpy(), ss.numpy(), ls.numpy(), loc.numpy(), w.numpy(), go.numpy())
    for dtype, tol in ((torch.float64, 1e-10), (torch.float32, 1e-4)):
        vd, ld, wd = (t.to(dtype).to(DEV).requires_grad_(True) for t in (value, loc, w))
        torch.ops.codetr.multi_scale_deformable_attention(vd, ss.to(DEV), ls.to(DEV), ld, wd, 64).backward(go.to(dtype).to(DEV))
        for name, got, ref in (("value", vd.grad, gv), ("loc", ld.grad, gl), ("w", wd.grad, gw)):
            ref = torch.from_numpy(ref)
            assert (got.double().cpu() - ref).abs().max() <= tol * ref.abs().max(), (D, dtype, name)


def test_f16_odd_row_is_reported_unsupported():
    """documented limit of the generic kernel: fp16 value-gradient atomics are packed pairs, M * D must be even"""
    value, loc, w, go, S, L = _case(1, 1, 71, 2, [(3, 2)], 2, torch.float16, seed=2)
    ss = torch.tensor([(3, 2)], dtype=torch.int64, device=DEV)
    ls = torch.zeros(1, dtype=torch.int64, device=DEV)
    v, l_, w_, g = (t.half().to(DEV) for t in (value, loc, w, go))
    with pytest.raises(RuntimeError, match="outside what the kernel family implements|unsupported"):
        torch.ops.codetr.multi_scale_deformable_attention_backward(v, ss, ls, l_, w_, g, torch.zeros_like(v),
                                                                   torch.zeros_like(l_), torch.zeros_like(w_), 64)


def msda_autograd_reference(value, shapes, loc, w):
    """value [B,S,M,D], loc [B,Nq,M,L,P,2] (x,y in [0,1]), w [B,Nq,M,L,P] -> [B,Nq,M*D]; differentiable, any device"""
    B, S, M, D = value.shape
    Nq, L, P = loc.shape[1], loc.shape[3], loc.shape[4]
    out = value.new_zeros(B, Nq, M, D)
    start = 0
    bi = torch.arange(B, device=value.device).view(B, 1, 1, 1).expand(B, Nq, M, P)
    mi = torch.arange(M, device=value.device).view(1, 1, M, 1).expand(B, Nq, M, P)
    for l, (H, W) in enumerate(shapes):
        v = value[:, start:start + H * W]            # [B, HW, M, D]
        x = loc[:, :, :, l, :, 0] * W - 0.5          # [B,Nq,M,P]
        y = loc[:, :, :, l, :, 1] * H - 0.5
        gate = (y > -1) & (x > -1) & (y < H) & (x < W)
        x0, y0 = torch.floor(x.detach()), torch.floor(y.detach())
        lx, ly = x - x0, y - y0
        acc = 0
        for dy, wy in ((0, 1 - ly), (1, ly)):
            for dx, wx in ((0, 1 - lx), (1, lx)):
                xi, yi = (x0 + dx).long(), (y0 + dy).long()
                ok = gate & (xi >= 0) & (xi <= W - 1) & (yi >= 0) & (yi <= H - 1)
                pix = (yi.clamp(0, H - 1) * W + xi.clamp(0, W - 1))
                g = v[bi, pix, mi]                   # [B,Nq,M,P,D]
                acc = acc + (wy * wx * ok.to(value.dtype)).unsqueeze(-1) * g
        out = out + (acc * w[:, :, :, l, :].unsqueeze(-1)).sum(3)
        start += H * W
    return out.reshape(B, Nq, M * D)


def _case(B, M, D, Nq, shapes, P, dtype, seed, border=False):
    g = torch.Generator().manual_seed(seed)
    S = sum(h * w for h, w in shapes)
    L = len(shapes)
    value = torch.rand(B, S, M, D, generator=g, dtype=torch.float64)
    loc = torch.rand(B, Nq, M, L, P, 2, generator=g, dtype=torch.float64)
    if border:
        loc = loc * 1.3 - 0.15          # some samples outside the map / on the border rows
    w = torch.rand(B, Nq, M, L, P, generator=g, dtype=torch.float64)
    w = w / w.sum((-1, -2), keepdim=True)
    go = torch.randn(B, Nq, M * D, generator=g, dtype=torch.float64)
    return value, loc, w, go, S, L


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-10), (torch.float32, 1e-4), (torch.float16, 2e-2)])
@pytest.mark.parametrize("B,M,D,Nq,shapes,P,border", [
    (1, 2, 2, 2, [(6, 4), (3, 2)], 2, False),                  # the reference's gradcheck geometry
    (2, 8, 32, 37, [(12, 18), (6, 9), (3, 5)], 4, True),       # model-shaped heads, out-of-range samples
    (2, 4, 16, 8, [(32, 32), (16, 16), (8, 8)], 4, False),
])
def test_msda_backward_matches_autograd(B, M, D, Nq, shapes, P, border, dtype, tol):
    from codetr import _cabi

    value, loc, w, go, S, L = _case(B, M, D, Nq, shapes, P, dtype, seed=3, border=border)
    # quantise the inputs to the run dtype first so both sides differentiate the same function
    value, loc, w, go = (t.to(dtype).double() for t in (value, loc, w, go))
    vr, lr, wr = (t.clone().requires_grad_(True) for t in (value, loc, w))
    msda_autograd_reference(vr, shapes, lr, wr).backward(go)
    ss = torch.tensor(shapes, dtype=torch.int64, device=DEV)
    ls = torch.cat((ss.new_zeros(1), ss.prod(1).cumsum(0)[:-1]))
    vd, ld, wd = (t.to(dtype).to(DEV).requires_grad_(True) for t in (value, loc, w))
    before = _cabi.CALLS["msda_backward"]
    out = torch.ops.codetr.multi_scale_deformable_attention(vd, ss, ls, ld, wd, 64)
    out.backward(go.to(dtype).to(DEV))
    assert _cabi.CALLS["msda_backward"] == before + 1
    for name, got, ref in (("value", vd.grad, vr.grad), ("sampling_loc", ld.grad, lr.grad), ("attn_weight", wd.grad, wr.grad)):
        assert got is not None and got.dtype == dtype and got.shape == ref.shape
        scale = ref.abs().max().item() + 1e-30
        err = (got.double().cpu() - ref).abs().max().item()
        assert err <= tol * scale, f"grad_{name}: max err {err:.3e} vs scale {scale:.3e}"


def test_msda_backward_contract():
    """pre-zeroed outputs are accumulated into (reference contract), im2col_step and dtype errors are loud"""
    value, loc, w, go, S, L = _case(4, 2, 4, 5, [(4, 4), (2, 2)], 2, torch.float32, seed=5)
    ss = torch.tensor([(4, 4), (2, 2)], dtype=torch.int64, device=DEV)
    ls = torch.tensor([0, 16], dtype=torch.int64, device=DEV)
    v, l_, w_, g = (t.float().to(DEV) for t in (value, loc, w, go))
    gv, gl, gw = torch.zeros_like(v), torch.zeros_like(l_), torch.zeros_like(w_)
    torch.ops.codetr.multi_scale_deformable_attention_backward(v, ss, ls, l_, w_, g, gv, gl, gw, 64)
    gv2 = gv.clone()
    torch.ops.codetr.multi_scale_deformable_attention_backward(v, ss, ls, l_, w_, g, gv2, gl, gw, 2)
    torch.testing.assert_close(gv2, 2 * gv, rtol=1e-5, atol=1e-6)  # value gradient accumulates (atomics)
    with pytest.raises(RuntimeError, match="must divide im2col_step"):
        torch.ops.codetr.multi_scale_deformable_attention_backward(v, ss, ls, l_, w_, g, gv, gl, gw, 3)
    with pytest.raises(RuntimeError):
        torch.ops.codetr.multi_scale_deformable_attention_backward(v, ss, ls, l_, w_, g.half(), gv, gl, gw, 64)
    vb = v.bfloat16()
    with pytest.raises(RuntimeError, match="unsupported dtype"):
        torch.ops.codetr.multi_scale_deformable_attention_backward(vb, ss, ls, l_.bfloat16(), w_.bfloat16(), g.bfloat16(),
                                                                   torch.zeros_like(vb), gl.bfloat16(), gw.bfloat16(), 64)
