"""Encoder self-attention form of the fused MSDA (csrc/msda_encoder.hip, codetr_msda_encoder_forward_*): the
LDS-staged gather must give BIT-IDENTICAL results to the general fused kernel (codetr_msda_fused_forward_*, itself
checked against the oracle and the reference's golden vectors in test_msda_gpu.py) -- same formulas, same
accumulation order, only the data movement differs.  Cases: pyramids that the 16x8 regions divide and ones they do
not, offsets inside the staged halo (LDS-only loop), beyond it (checked loop, global reads), far outside the image,
padded images (reference points skewed by valid ratios), arbitrary reference points, ragged L*P, bf16, and one
launch at BASELINE's full size.  A float64 oracle comparison on one case guards against a shared mistake."""
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


def _run_both(shapes, B=2, M=8, P=4, off_scale=1.5, dtype=torch.float16, seed=0, **kw):
    from codetr import _cabi, hip_ops

    L = len(shapes)
    value, proj, ref, S = _pyramid_inputs(shapes, B, M, P, off_scale, dtype, seed, **kw)
    ss = torch.tensor(shapes, dtype=torch.int64, device=DEV)
    ls = torch.cat((ss.new_zeros(1), ss.prod(1).cumsum(0)[:-1]))
    before = _cabi.CALLS["msda_encoder"]
    out = hip_ops.msda_encoder(value, shapes, proj, 0, M * L * P * 2, ref, P)
    assert out is not None and _cabi.CALLS["msda_encoder"] == before + 1, "the encoder kernel did not take the shape"
    want = hip_ops.msda_fused(value, ss, ls, proj, 0, M * L * P * 2, ref, L, P)
    torch.cuda.synchronize()
    return out, want, (value, proj, ref, ss, ls)


def _assert_identical(out, want, what):
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
    _assert_identical(out, want, f"{shapes} off_scale {off_scale}")


def test_padded_image_reference_points():
    out, want, _ = _run_both(PYR_DIV, off_scale=2.0, seed=5, valid=(0.8, 0.65))
    _assert_identical(out, want, "valid ratios (0.8, 0.65)")


def test_arbitrary_reference_points_take_the_checked_loop():
    out, want, _ = _run_both(PYR_ODD, off_scale=3.0, seed=6, ref_mode="random")
    _assert_identical(out, want, "random reference points")


@pytest.mark.parametrize("M,P,L", [(4, 4, 4), (8, 3, 3), (2, 6, 5), (8, 1, 5)])
def test_other_head_point_level_counts(M, P, L):
    out, want, _ = _run_both(PYR_DIV[:L], M=M, P=P, off_scale=2.5, seed=M * 100 + P * 10 + L)
    _assert_identical(out, want, f"M {M} P {P} L {L}")


def test_bf16():
    out, want, _ = _run_both(PYR_ODD, dtype=torch.bfloat16, off_scale=2.5, seed=11)
    _assert_identical(out, want, "bf16")


def test_single_image_and_batch_of_three():
    for B in (1, 3):
        out, want, _ = _run_both(PYR_DIV, B=B, off_scale=2.0, seed=20 + B)
        _assert_identical(out, want, f"B {B}")


def test_against_float64_oracle():
    """One case straight against the CPU oracle (the same check test_msda_gpu.py applies to the general kernel)."""
    from oracle import msda_oracle

    shapes, B, M, P = PYR_TINY, 2, 8, 4
    L = len(shapes)
    out, _, (value, proj, ref, ss, ls) = _run_both(shapes, B=B, M=M, P=P, off_scale=1.5, seed=31)
    S = value.shape[1]
    off = proj[..., :M * L * P * 2].double().cpu().view(B, S, M, L, P, 2)
    logits = proj[..., M * L * P * 2:M * L * P * 3].double().cpu().view(B, S, M, L * P)
    w = torch.softmax(logits, -1).view(B, S, M, L, P)
    norm = torch.tensor([[w_, h_] for h_, w_ in shapes], dtype=torch.float64)
    loc = ref.double().cpu()[:, :, None, :, None, :] + off / norm[None, None, None, :, None, :]
    ssn = np.asarray(shapes, dtype=np.int64)
    expect = msda_oracle.msda_forward_numpy(value.double().cpu().numpy(), ssn,
                                            msda_oracle.level_start_index_from_shapes(ssn), loc.numpy(), w.numpy())
    np.testing.assert_allclose(out.float().cpu().numpy(), expect, rtol=4e-3, atol=4e-3)


def test_full_size_launch_identical():
    """BASELINE's pyramid (1920x1280), one image: every region shape of the real launch."""
    shapes = [(320, 480), (160, 240), (80, 120), (40, 60), (20, 30)]
    out, want, _ = _run_both(shapes, B=1, off_scale=2.0, seed=41)
    _assert_identical(out, want, "1920x1280")


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
    np.testing.assert_allclose(got, expect[0], rtol=4e-3, atol=4e-3)


def test_unsupported_shapes_fall_back():
    """A halo whose neighbourhoods do not fit LDS -> the library declines, hip_ops returns None (the module then calls
    the general kernel)."""
    from codetr import _cabi

    value, proj, ref, S = _pyramid_inputs(PYR_DIV, 1, 8, 4, 1.0, torch.float16, 1)
    out = torch.empty(1, S, 256, dtype=torch.float16, device=DEV)
    assert _cabi.msda_encoder(value, PYR_DIV, proj, 0, 8 * 5 * 4 * 2, ref, 4, 40, out) is False
    with pytest.raises(RuntimeError):      # level shapes that do not add up to S
        _cabi.msda_encoder(value, PYR_DIV[:4], proj, 0, 8 * 5 * 4 * 2, ref, 4, 4, out)
