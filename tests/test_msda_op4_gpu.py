"""The windowed kernel behind the public op for encoder-shaped calls (csrc/msda_op4.hip, round 6) against the C / fp64
oracle (oracle/msda_ref.c: reference ms_deform_attn.cu:31-77, 211-261), THROUGH torch.ops.codetr.multi_scale_deformable_
attention and through its own C-ABI entry.

Tolerance: the op's own fp16 criterion (tests/test_msda_gpu.py::test_golden_fp16): within one fp16 ulp (2^-10 relative) of the
exactly rounded result plus fp32 accumulation noise -- the kernel blends in fp32 like the general kernel.

Cases: locations inside the staged windows (query pixel + a few pixels: the encoder's regime), beyond them (the fix-up queue,
more records than a round holds), uniformly random, outside the image (the reference's gate), NaN / inf; pyramids the regions
divide and ones they do not; batch > 1; pyramids the device-side plan turns down (the general kernel must serve them, the
windowed one must leave `out` alone); the routing itself."""
import numpy as np
import pytest
import torch

import msda_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
M, L, P, D = 8, 5, 4, 32
PYR_608 = [(76, 76), (38, 38), (19, 19), (10, 10), (5, 5)]
PYR_DIV = [(64, 96), (32, 48), (16, 24), (8, 12), (4, 6)]
PYR_ODD = [(77, 51), (39, 26), (20, 13), (10, 7), (5, 4)]


def _tensors(shapes):
    ss = np.asarray(shapes, dtype=np.int64)
    ls = np.concatenate(([0], np.cumsum(ss[:, 0] * ss[:, 1])[:-1])).astype(np.int64)
    return ss, ls, int((ss[:, 0] * ss[:, 1]).sum())


def _case(shapes, B, spread_px, seed, mode="pixel"):
    """value, loc, w as fp16-representable float64 arrays.  mode 'pixel': query i sits at pixel i of the pyramid and samples
    around its own position (spread in pixels of each level); 'uniform': locations anywhere in [-0.1, 1.1]"""
    ss, ls, S = _tensors(shapes)
    rng = np.random.default_rng(seed)
    value = rng.standard_normal((B, S, M, D))
    w = rng.random((B, S, M, L, P))
    w = w / w.sum((-1, -2), keepdims=True) * rng.uniform(0.5, 2.0, (B, S, M, 1, 1))   # not normalised: the op must not care
    if mode == "uniform":
        loc = rng.uniform(-0.1, 1.1, (B, S, M, L, P, 2))
    else:
        centres = []
        for (h, w_) in shapes:
            ys, xs = np.meshgrid((np.arange(h) + 0.5) / h, (np.arange(w_) + 0.5) / w_, indexing="ij")
            centres.append(np.stack((xs.ravel(), ys.ravel()), -1))
        c = np.concatenate(centres, 0)                                   # [S, 2] normalised (x, y)
        size = np.asarray([[w_, h] for h, w_ in shapes], dtype=np.float64)  # [L, 2] (W, H)
        off = rng.standard_normal((B, S, M, L, P, 2)) * spread_px
        loc = c[None, :, None, None, None, :] + off / size[None, None, None, :, None, :]
    h = lambda a: a.astype(np.float16).astype(np.float64)  # noqa: E731
    return h(value), ss, ls, h(loc), h(w), S


def _expect(value, ss, ls, loc, w):
    return O.msda_forward_c(value, ss, ls, loc, w, dtype=np.float64)


def _op(value, ss, ls, loc, w):
    import codetr  # noqa: F401

    t = lambda a, dt: torch.as_tensor(np.asarray(a)).to(DEV).to(dt).contiguous()  # noqa: E731
    out = torch.ops.codetr.multi_scale_deformable_attention(t(value, torch.float16), t(ss, torch.int64), t(ls, torch.int64),
                                                            t(loc, torch.float16), t(w, torch.float16), 64)
    torch.cuda.synchronize()
    return out.float().cpu().numpy()


def _direct(value, ss, ls, loc, w):
    """the windowed kernel alone on a NaN-filled output: (rc, output)"""
    from codetr import _cabi

    lib = _cabi.load()
    t = lambda a, dt: torch.as_tensor(np.asarray(a)).to(DEV).to(dt).contiguous()  # noqa: E731
    v, s_, l_, lo, we = t(value, torch.float16), t(ss, torch.int64), t(ls, torch.int64), t(loc, torch.float16), t(w, torch.float16)
    B, S = v.shape[:2]
    out = torch.full((B, S, M * D), float("nan"), dtype=torch.float16, device=DEV)
    rc = lib.codetr_msda_op4_forward_f16(_cabi.current_stream_ptr(v.device), v.data_ptr(), s_.data_ptr(), l_.data_ptr(),
                                         lo.data_ptr(), we.data_ptr(), B, S, M, D, L, S, P, out.data_ptr())
    torch.cuda.synchronize()
    return rc, out.float().cpu().numpy()


def _check(got, ref, what):
    assert np.isfinite(got).all(), f"{what}: {int((~np.isfinite(got)).sum())} outputs not written / not finite"
    np.testing.assert_allclose(got, ref, rtol=1.1 * 2.0 ** -10, atol=2e-6, err_msg=what)


@pytest.mark.parametrize("shapes", [PYR_608, PYR_DIV, PYR_ODD], ids=["608", "divisible", "odd"])
@pytest.mark.parametrize("spread", [1.5, 6.0, 40.0], ids=["inside", "beyond_window", "outside_image"])
def test_windowed_kernel_alone_vs_oracle(shapes, spread):
    value, ss, ls, loc, w, S = _case(shapes, 2, spread, seed=int(spread * 10) + len(shapes[0]))
    rc, got = _direct(value, ss, ls, loc, w)
    assert rc == 0
    _check(got, _expect(value, ss, ls, loc, w), f"spread {spread}")


def test_uniform_locations_and_non_finite_through_the_op():
    value, ss, ls, loc, w, S = _case(PYR_608, 1, 0.0, seed=3, mode="uniform")
    loc[0, 5, 1, 2, 3, 0] = np.nan
    loc[0, 77, 0, 0, 0, 1] = np.inf
    loc[0, 1234, 7, 4, 1, :] = -np.inf
    got = _op(value, ss, ls, loc, w)
    ref = _expect(value, ss, ls, np.nan_to_num(loc, nan=-1e4, posinf=1e4, neginf=-1e4), w)   # all dropped by the reference's gate
    _check(got, ref, "uniform + non-finite")
    rc, alone = _direct(value, ss, ls, loc, w)
    assert rc == 0
    np.testing.assert_array_equal(alone, got)          # the op's result IS the windowed kernel's


def test_plan_turns_down_odd_pyramids_and_the_general_kernel_serves_them():
    # level 0 is not the largest level: the windowed kernel's workgroups must return without touching `out`
    shapes = [(19, 19), (76, 76), (38, 38), (10, 10), (5, 5)]
    value, ss, ls, loc, w, S = _case(shapes, 1, 0.0, seed=4, mode="uniform")
    rc, alone = _direct(value, ss, ls, loc, w)
    assert rc == 0 and np.isnan(alone).all()
    got = _op(value, ss, ls, loc, w)
    np.testing.assert_allclose(got, _expect(value, ss, ls, loc, w), rtol=1.1 * 2.0 ** -10, atol=2e-6)
    # level starts that are not the prefix sums (a padded layout): same
    value, ss, ls, loc, w, S = _case(PYR_608, 1, 2.0, seed=5)
    ls2 = ls.copy()
    ls2[1:] = ls[1:][::-1].copy()
    rc, alone = _direct(value, ss, ls2, loc, w)
    assert rc == 0 and np.isnan(alone).all()


def test_host_side_routing_test():
    from codetr import _cabi

    lib = _cabi.load()
    S = 7725
    assert lib.codetr_msda_op4_supported(2, 2, S, 8, 32, 5, S, 4) == 1
    assert lib.codetr_msda_op4_supported(2, 2, S, 8, 32, 5, 900, 4) == 0     # decoder-shaped
    assert lib.codetr_msda_op4_supported(2, 2, S, 8, 64, 5, S, 4) == 0       # 64-channel heads
    assert lib.codetr_msda_op4_supported(2, 2, S, 8, 32, 4, S, 4) == 0       # 4 levels
    assert lib.codetr_msda_op4_supported(4, 2, S, 8, 32, 5, S, 4) == 0       # fp32
    assert lib.codetr_msda_op4_supported(2, 2, 512, 8, 32, 5, 512, 4) == 0   # launch-bound sizes stay on the general kernel


def test_full_size_matches_the_general_kernel():
    """BASELINE's pyramid (S = 204 600), model-like locations: the windowed kernel against the general kernel of the same
    library (both blend in fp32; sums are taken in a different order: a few fp32 ulps before the final rounding)"""
    shapes = [(320, 480), (160, 240), (80, 120), (40, 60), (20, 30)]
    ss, ls, S = _tensors(shapes)
    g = torch.Generator(device=DEV).manual_seed(1)
    value = torch.randn(1, S, M, D, device=DEV, generator=g).half()
    cs = []
    for (h, w_) in shapes:
        ys, xs = torch.meshgrid((torch.arange(h, device=DEV) + 0.5) / h, (torch.arange(w_, device=DEV) + 0.5) / w_, indexing="ij")
        cs.append(torch.stack((xs.reshape(-1), ys.reshape(-1)), -1))
    c = torch.cat(cs, 0)
    size = torch.tensor([[w_, h] for h, w_ in shapes], device=DEV, dtype=torch.float32)
    off = torch.randn(1, S, M, L, P, 2, device=DEV, generator=g) * 3.0
    loc = (c[None, :, None, None, None, :] + off / size[None, None, None, :, None, :]).half()
    w = torch.softmax(torch.randn(1, S, M, L * P, device=DEV, generator=g), -1).view(1, S, M, L, P).half()
    sst, lst = torch.as_tensor(ss).to(DEV), torch.as_tensor(ls).to(DEV)
    import codetr  # noqa: F401
    from codetr import _cabi

    got = torch.ops.codetr.multi_scale_deformable_attention(value, sst, lst, loc, w, 64)
    # the general kernel alone: a decoder-shaped view of the same problem (Nq != S routes past the windowed kernel)
    half = S // 2
    ref_a = torch.ops.codetr.multi_scale_deformable_attention(value, sst, lst, loc[:, :half].contiguous(), w[:, :half].contiguous(), 64)
    ref_b = torch.ops.codetr.multi_scale_deformable_attention(value, sst, lst, loc[:, half:].contiguous(), w[:, half:].contiguous(), 64)
    ref = torch.cat((ref_a, ref_b), 1)
    torch.cuda.synchronize()
    assert torch.isfinite(got.float()).all()
    torch.testing.assert_close(got.float(), ref.float(), rtol=2.0 ** -10, atol=1e-5)
    assert _cabi.load().codetr_msda_op4_supported(2, 1, S, M, D, L, S, P) == 1
