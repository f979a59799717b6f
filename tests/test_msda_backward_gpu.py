"""GPU parity of the MSDA backward kernel (C ABI codetr_msda_backward_*, torch.ops.codetr.
multi_scale_deformable_attention_backward + the autograd registration) against PyTorch autograd through a
differentiable fp64 restatement of the forward (the bilinear formulas of reference ms_deform_attn.cu:31-77 written with
tensor ops; the reference's own check is `gradcheck` in fp64, tests/test_multi_scale_deformable_attention.py:367-414).
fp64: 1e-10 relative; fp32: 1e-4; fp16 (fp32 arithmetic, fp16 atomics / stores): 2e-2 of the gradient scale."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


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
