"""Encoder self-attention form of the fused MSDA (csrc/msda_encoder.hip, codetr_msda_encoder_forward_*) against the
general fused kernel (codetr_msda_fused_forward_*, itself checked against the oracle and the reference's golden vectors
in test_msda_gpu.py) and against the float64 CPU oracle directly.

Two kernels serve the entry point:
  * fp16 at the model's shape (5 levels x 4 points): the packed-half blend (v2) -- fp16 corner weights, 8-term fp16
    partial sums added in fp32.  Tolerance, stated here and in include/codetr_hip.h: against the float64 oracle
    rtol 1e-2 / atol 1e-3 element-wise (the reference's own half tolerance, for a kernel that accumulates all 80
    terms in half: tests/test_multi_scale_deformable_attention.py:62, 363-364) and relative L2
    <= 1e-3; against the general kernel (fp32 blend, one rounding) relative L2 <= 1e-3 and |d| <= 3e-3 + 3e-3 |x|
    element-wise.  Measured: relative L2 4-6e-4 (the final fp16 rounding alone is 2e-4).
  * every other shape / bf16: same formulas and accumulation order as the general kernel -- BIT-IDENTICAL.
Cases: pyramids that the 16x8 regions divide and ones they do not, offsets inside the staged window, beyond it (fix-up
queue, global reads; more than one queue pass), far outside the image, padded images (reference points skewed by valid
ratios), arbitrary reference points, ragged L*P, bf16, per-head windows, and launches at BASELINE's full size."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _pyramid_inputs(shapes, B, M, P, off_scale, dtype, seed, ref_mode="centres", valid=(1.0, 1.0)):
    L, D = len(shapes), 32
    S = sum(h * w for h, w in shapes)
    g = torch.Generator(device="cpu").manual_seed(seed)
    value = torch.randn(B, S, M, D, generator=g)
    off = torch.randn(B, S, M, L, P, 2, generator=g) * off_scale
    logits = torch.randn(B, S, M, L * P, generator=g) * 2
    proj = torch.cat((off.reshape(B, S, -1), logits.reshape(B, S, -1), torch.randn(B, S, 24, generator=g)), -1)
    if ref_mode == "centres":
        # get_reference_points (reference transformer.py:280-305): pixel centres / (valid ratio * size), then
        # scaled by every level's valid ratio (here: a per-level jitter)
        refs = []
        for h, w in shapes:
            ys, xs = torch.meshgrid(torch.arange(h) + 0.5, torch.arange(w) + 0.5, indexing="ij")
            refs.append(torch.stack((xs.reshape(-1) / (valid[0] * w), ys.reshape(-1) / (valid[1] * h)), -1))
        ref = torch.cat(refs, 0)[None, :, None, :].expand(B, S, L, 2).clone()   # padded area: refs beyond 1
        ref = ref * (1 + 0.002 * torch.randn(1, 1, L, 2, generator=g))   # per-level valid-ratio jitter
    else:
        ref = torch.rand(B, S, L, 2, generator=g) * 1.2 - 0.1
    to = lambda t: t.to(DEV).to(dtype).contiguous()  # noqa: E731
    return to(value), to(proj), to(ref), S


def _run_both(shapes, B=2, M=8, P=4, off_scale=1.5, dtype=torch.float16, seed=0, windows=None, passes=1, **kw):
    from codetr import _cabi, hip_ops

    L = len(shapes)
    value, proj, ref, S = _pyramid_inputs(shapes, B, M, P, off_scale, dtype, seed, **kw)
    ss = torch.tensor(shapes, dtype=torch.int64, device=DEV)
    ls = torch.cat((ss.new_zeros(1), ss.prod(1).cumsum(0)[:-1]))
    before = _cabi.CALLS["msda_encoder"]
    out = hip_ops.msda_encoder(value, shapes, proj, 0, M * L * P * 2, ref, P, windows, passes)
    assert out is not None and _cabi.CALLS["msda_encoder"] == before + 1, "the encoder kernel did not take the shape"
    want = hip_ops.msda_fused(value, ss, ls, proj, 0, M * L * P * 2, ref, L, P)
    torch.cuda.synchronize()
    return out, want, (value, proj, ref, ss, ls)


def _packed(out_or_dtype, L, P):
    dt = out_or_dtype if isinstance(out_or_dtype, torch.dtype) else out_or_dtype.dtype
    return dt == torch.float16 and L == 5 and P == 4


def _assert_close_packed(out, want, what):
    """packed-half blend vs the fp32-blend kernel (module docstring)"""
    a, b = out.double(), want.double()
    assert torch.equal(torch.isfinite(a), torch.isfinite(b)), f"{what}: non-finite pattern differs"
    fin = torch.isfinite(b)
    d = (a - b)[fin]
    rel = (d.norm() / b[fin].norm().clamp_min(1e-30)).item()
    worst = (d.abs() / (3e-3 + 3e-3 * b[fin].abs())).max().item()   # 3e-3 absolute + 3e-3 relative (1.5 fp16 ulp at |x| = 4)
    assert rel <= 1e-3 and worst <= 1.0, f"{what}: rel L2 {rel:.2e}, max |d| {d.abs().max().item():.2e} ({worst:.2f} x tol)"


def _assert_identical(out, want, what, L=None, P=4):
    if L is not None and _packed(out, L, P):
        return _assert_close_packed(out, want, what)
    a, b = out.view(torch.int16), want.view(torch.int16)
    if not torch.equal(a, b):
        d = (out.float() - want.float()).abs()
        bad = (a != b).nonzero()
        raise AssertionError(f"{what}: {bad.shape[0]} elements differ, max |d| = {d.max().item():.3e}, "
                             f"first at {bad[0].tolist()}")


PYR_DIV = [(40, 64), (20, 32), (10, 16), (5, 8), (3, 4)]          # regions divide levels 0-3
PYR_ODD = [(38, 38), (19, 19), (10, 10), (5, 5), (3, 3)]          # 152-px-like: nothing divides
PYR_TINY = [(5, 7), (3, 4), (2, 2)]


@pytest.mark.parametrize("shapes", [PYR_DIV, PYR_ODD, PYR_TINY], ids=["divisible", "odd", "tiny"])
@pytest.mark.parametrize("off_scale", [1.5, 6.0, 60.0], ids=["inside_halo", "beyond_halo", "outside_image"])
def test_identical_to_general_fused_kernel(shapes, off_scale):
    out, want, _ = _run_both(shapes, off_scale=off_scale, seed=int(off_scale * 10) + len(shapes))
    _assert_identical(out, want, f"{shapes} off_scale {off_scale}", len(shapes))


THREE_PASS_CASES = {
    "divisible_inside": dict(shapes=PYR_DIV, off_scale=1.5, seed=15),
    "odd_beyond_window": dict(shapes=PYR_ODD, off_scale=6.0, seed=65),
    "odd_outside_image": dict(shapes=PYR_ODD, off_scale=60.0, seed=605),
    "padded": dict(shapes=PYR_DIV, off_scale=2.0, seed=5, valid=(0.8, 0.65)),
    "random_refs": dict(shapes=PYR_ODD, off_scale=3.0, seed=6, ref_mode="random"),
    "batch_of_three": dict(shapes=PYR_DIV, B=3, off_scale=2.0, seed=23),
    "narrow_windows": dict(shapes=PYR_ODD, off_scale=2.5, seed=77, windows=[[(0, 1, 0, 1)] * 5] * 8),
    "per_head_windows": dict(shapes=PYR_ODD, off_scale=2.5, seed=78, windows=[[(-m, 8 - m, m - 7, 2)] * 5 for m in range(8)]),
    "full_size_1920x1280": dict(shapes=[(320, 480), (160, 240), (80, 120), (40, 60), (20, 30)], B=1, off_scale=2.0, seed=41),
}


@pytest.mark.parametrize("case", list(THREE_PASS_CASES))
def test_three_pass_kernel_matches_general_kernel(case):
    """passes = 3 (levels staged {0}, {1, 2}, {3, 4}; four workgroups per CU): same results as the single-pass packed
    kernel up to which samples take the fix-up queue -- within the packed blend's tolerance of the general kernel"""
    out, want, _ = _run_both(passes=3, **THREE_PASS_CASES[case])
    _assert_close_packed(out, want, case)


def test_three_pass_fp32_reference_points_against_float64_oracle():
    """valid_counts given: the kernel computes ((x + 0.5) / (vr_q W_q)) vr_l itself in fp32 (get_reference_points + the
    per-level scaling, reference transformer.py:280-305, 530) -- checked against the float64 oracle fed with the same
    formula in float64; the fp16 reference-point tensor it would otherwise read is deliberately garbage here."""
    from codetr import hip_ops
    from oracle import msda_oracle

    shapes, B, M, P, L = PYR_DIV, 2, 8, 4, 5
    value, proj, ref, S = _pyramid_inputs(shapes, B, M, P, 2.0, torch.float16, 91)
    counts = torch.tensor([[[w - (3 * b + l) % 4, h - (2 * b + l) % 3] for l, (h, w) in enumerate(shapes)]
                           for b in range(B)], dtype=torch.float32)
    vr = counts.double() / torch.tensor([[w, h] for h, w in shapes], dtype=torch.float64)     # [B, L, 2]
    refs = []
    for l, (h, w) in enumerate(shapes):
        ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float64) + 0.5, torch.arange(w, dtype=torch.float64) + 0.5,
                                indexing="ij")
        base = torch.stack((xs.reshape(-1)[None] / (vr[:, l, 0, None] * w), ys.reshape(-1)[None] / (vr[:, l, 1, None] * h)), -1)
        refs.append(base[:, :, None, :] * vr[:, None, :, :])          # [B, hw, L, 2]
    ref64 = torch.cat(refs, 1)
    out = hip_ops.msda_encoder(value, shapes, proj, 0, M * L * P * 2, torch.full_like(ref, 0.37), P, None, 3,
                               counts.to(DEV))
    assert out is not None
    off = proj[..., :M * L * P * 2].double().cpu().view(B, S, M, L, P, 2)
    w = torch.softmax(proj[..., M * L * P * 2:M * L * P * 3].double().cpu().view(B, S, M, L * P), -1).view(B, S, M, L, P)
    norm = torch.tensor([[w_, h_] for h_, w_ in shapes], dtype=torch.float64)
    loc = ref64[:, :, None, :, None, :] + off / norm[None, None, None, :, None, :]
    ssn = np.asarray(shapes, dtype=np.int64)
    expect = msda_oracle.msda_forward_numpy(value.double().cpu().numpy(), ssn,
                                            msda_oracle.level_start_index_from_shapes(ssn), loc.numpy(), w.numpy())
    got = out.float().cpu().numpy()
    np.testing.assert_allclose(got, expect, rtol=1e-2, atol=1e-3)
    assert np.linalg.norm(got - expect) / np.linalg.norm(expect) <= 1e-3


def test_padded_image_reference_points():
    out, want, _ = _run_both(PYR_DIV, off_scale=2.0, seed=5, valid=(0.8, 0.65))
    _assert_identical(out, want, "valid ratios (0.8, 0.65)", 5)


def test_arbitrary_reference_points_take_the_checked_loop():
    out, want, _ = _run_both(PYR_ODD, off_scale=3.0, seed=6, ref_mode="random")
    _assert_identical(out, want, "random reference points", 5)


@pytest.mark.parametrize("M,P,L", [(4, 4, 4), (8, 3, 3), (2, 6, 5), (8, 1, 5)])
def test_other_head_point_level_counts(M, P, L):
    out, want, _ = _run_both(PYR_DIV[:L], M=M, P=P, off_scale=2.5, seed=M * 100 + P * 10 + L)
    _assert_identical(out, want, f"M {M} P {P} L {L}", L, P)


def test_bf16():
    out, want, _ = _run_both(PYR_ODD, dtype=torch.bfloat16, off_scale=2.5, seed=11)
    _assert_identical(out, want, "bf16")


def test_single_image_and_batch_of_three():
    for B in (1, 3):
        out, want, _ = _run_both(PYR_DIV, B=B, off_scale=2.0, seed=20 + B)
        _assert_identical(out, want, f"B {B}", 5)


@pytest.mark.parametrize("shapes,off_scale", [(PYR_TINY, 1.5), (PYR_ODD, 1.5), (PYR_ODD, 5.0)],
                         ids=["3_levels_fp32_blend", "5_levels_packed", "5_levels_packed_beyond_window"])
def test_against_float64_oracle(shapes, off_scale):
    """Straight against the CPU oracle (the same check test_msda_gpu.py applies to the general kernel)."""
    from oracle import msda_oracle

    B, M, P = 2, 8, 4
    L = len(shapes)
    out, _, (value, proj, ref, ss, ls) = _run_both(shapes, B=B, M=M, P=P, off_scale=off_scale, seed=31)
    S = value.shape[1]
    off = proj[..., :M * L * P * 2].double().cpu().view(B, S, M, L, P, 2)
    logits = proj[..., M * L * P * 2:M * L * P * 3].double().cpu().view(B, S, M, L * P)
    w = torch.softmax(logits, -1).view(B, S, M, L, P)
    norm = torch.tensor([[w_, h_] for h_, w_ in shapes], dtype=torch.float64)
    loc = ref.double().cpu()[:, :, None, :, None, :] + off / norm[None, None, None, :, None, :]
    ssn = np.asarray(shapes, dtype=np.int64)
    expect = msda_oracle.msda_forward_numpy(value.double().cpu().numpy(), ssn,
                                            msda_oracle.level_start_index_from_shapes(ssn), loc.numpy(), w.numpy())
    got = out.float().cpu().numpy()
    np.testing.assert_allclose(got, expect, rtol=1e-2, atol=1e-3)
    rel = np.linalg.norm(got - expect) / np.linalg.norm(expect)
    assert rel <= (1e-3 if _packed(out, L, P) else 4e-4), rel


def test_windows_change_speed_not_results():
    """The same inputs with a narrow halo, a wide one, per-head windows and bias-derived windows: which samples are
    served from LDS and which from the fix-up queue changes (incl. pairs with more out-of-window samples than one queue
    pass holds), the result stays within the packed blend's rounding of the general kernel's."""
    from codetr import hip_ops

    M, L, P = 8, 5, 4
    wins = {"halo 1": [[(-1, 1, -1, 1)] * L] * M, "halo 6": [[(-6, 6, -6, 6)] * L] * M,
            "per head": [[(-m, 8 - m, m - 7, 2)] * L for m in range(M)],
            "one pixel": [[(0, 1, 0, 1)] * L] * M}
    outs = {}
    for name, w in wins.items():
        out, want, _ = _run_both(PYR_ODD, off_scale=2.5, seed=77, windows=w)
        _assert_close_packed(out, want, name)
        outs[name] = out
    bias = torch.randn(M * L * P * 2) * 2
    w = hip_ops.msda_encoder_windows(bias, PYR_ODD, M, L, P)
    assert len(w) == M and len(w[0]) == L and all(a <= b and c <= d for h in w for (a, b, c, d) in h)
    out, want, _ = _run_both(PYR_ODD, off_scale=2.5, seed=77, windows=w)
    _assert_close_packed(out, want, "bias windows")


def test_full_size_launch_identical():
    """BASELINE's pyramid (1920x1280), one image: every region shape of the real launch."""
    shapes = [(320, 480), (160, 240), (80, 120), (40, 60), (20, 30)]
    out, want, _ = _run_both(shapes, B=1, off_scale=2.0, seed=41)
    _assert_identical(out, want, "1920x1280", 5)


def test_full_size_sampled_queries_against_float64_oracle():
    """BASELINE's pyramid, one image: 768 queries drawn from all levels (incl. the image border and the level
    boundaries) against the float64 CPU oracle -- parity at the full size, not only identity with the general kernel."""
    from oracle import msda_oracle

    shapes = [(320, 480), (160, 240), (80, 120), (40, 60), (20, 30)]
    M, P, L = 8, 4, 5
    out, _, (value, proj, ref, ss, ls) = _run_both(shapes, B=1, off_scale=2.5, seed=43)
    S = value.shape[1]
    g = torch.Generator().manual_seed(1)
    starts = [0]
    for h, w in shapes:
        starts.append(starts[-1] + h * w)
    idx = torch.cat([torch.randint(starts[l], starts[l + 1], (150,), generator=g) for l in range(L)]
                    + [torch.tensor([0, 479, 480 * 319, starts[1] - 1, starts[1], starts[4], S - 1])]).unique()
    sub = proj[0, idx.to(DEV)].double().cpu()
    off = sub[:, :M * L * P * 2].view(1, -1, M, L, P, 2)
    w = torch.softmax(sub[:, M * L * P * 2:M * L * P * 3].view(1, -1, M, L * P), -1).view(1, -1, M, L, P)
    norm = torch.tensor([[w_, h_] for h_, w_ in shapes], dtype=torch.float64)
    loc = ref[0, idx.to(DEV)].double().cpu()[None, :, None, :, None, :] + off / norm[None, None, None, :, None, :]
    ssn = np.asarray(shapes, dtype=np.int64)
    expect = msda_oracle.msda_forward_numpy(value.double().cpu().numpy(), ssn,
                                            msda_oracle.level_start_index_from_shapes(ssn), loc.numpy(), w.numpy())
    got = out[0, idx.to(DEV)].float().cpu().numpy()
    np.testing.assert_allclose(got, expect[0], rtol=1e-2, atol=1e-3)
    assert np.linalg.norm(got - expect[0]) / np.linalg.norm(expect[0]) <= 1e-3


def test_unsupported_shapes_fall_back():
    """A halo whose neighbourhoods do not fit LDS -> the library declines, hip_ops returns None (the module then calls
    the general kernel)."""
    from codetr import _cabi

    value, proj, ref, S = _pyramid_inputs(PYR_DIV, 1, 8, 4, 1.0, torch.float16, 1)
    out = torch.empty(1, S, 256, dtype=torch.float16, device=DEV)
    assert _cabi.msda_encoder(value, PYR_DIV, proj, 0, 8 * 5 * 4 * 2, ref, 4, 40, out) is False
    with pytest.raises(RuntimeError):      # level shapes that do not add up to S
        _cabi.msda_encoder(value, PYR_DIV[:4], proj, 0, 8 * 5 * 4 * 2, ref, 4, 4, out)
