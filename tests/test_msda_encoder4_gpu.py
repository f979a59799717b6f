"""Round-5 encoder MSDA kernel (csrc/msda_encoder4.hip, C ABI codetr_msda_encoder_forward_packed_f16) against the float64
CPU oracle (oracle/msda_oracle.py: ms_deform_attn.cu:31-77, 211-261 restated), fed with the reference's own prologue in
float64: softmax over the 20 logits, reference points centre / (valid ratio * size) scaled by every level's valid ratio
(transformer.py:280-305, 530), loc = ref + offset / (W, H) (multi_scale_deformable_attention.py:180-196).

Tolerance (the reference's own half tolerance, tests/test_multi_scale_deformable_attention.py:62, 363-364): rtol 1e-2 /
atol 1e-3 element-wise on the fp16 outputs for all but 1e-5 of the elements, rtol 1e-2 / atol 2e-3 for every element, and
relative L2 <= 1e-3.  The kernel blends on packed halves -- fp16 corner weights, 8-term fp16 chains added in fp32 (the
reference's own half instantiation accumulates all 80 terms in half) -- so an element whose terms are large and cancel
carries the rounding of its chains' partial sums (half an fp16 ulp of |partial sum| <= 2 per term = up to 1e-3): with
logits of spread 2 and unit-variance values one element in a million lands between 1e-3 and 2e-3.  Measured relative L2
4-6e-4 (the final fp16 rounding alone is 2e-4).

Cases: pyramids the regions divide and ones they do not, both workgroup shapes (256 threads / 16x8 regions, 512 / 32x8),
offsets inside the windows, beyond them (fix-up queue: global reads, more records than one round holds), far outside the
image (the reference's gate), NaN / inf offsets, padded images (valid counts below the level size), narrow / per-head /
whole-level windows (zero border), the lane-major packing itself, BASELINE's full-size pyramid on sampled queries."""
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

M, L, P, D = 8, 5, 4, 32
PYR_DIV = [(40, 64), (20, 32), (10, 16), (5, 8), (3, 4)]
PYR_ODD = [(38, 38), (19, 19), (10, 10), (5, 5), (3, 3)]
PYR_FULL = [(320, 480), (160, 240), (80, 120), (40, 60), (20, 30)]
# "shipped" is what hip_ops launches by default (MSDA_V4_THREADS / _REGION / _LDS_BUDGET: 512 threads, 16 x 16 regions,
# 64 KiB): the region shape changes the slots per wave, the iteration count, the band walk and the window sizes, so it is
# pinned to the oracle at the op's own tolerance like the two other shapes
CFGS = {"t256": (256, (16, 8), 40 * 1024), "t512": (512, (32, 8), 80 * 1024), "shipped": (512, (16, 16), 64 * 1024)}


def test_shipped_cfg_is_what_hip_ops_launches():
    from codetr import hip_ops

    assert CFGS["shipped"] == (hip_ops.MSDA_V4_THREADS, tuple(hip_ops.MSDA_V4_REGION), hip_ops.MSDA_V4_LDS_BUDGET)


def _inputs(shapes, B, off_scale, seed, counts=None):
    S = sum(h * w for h, w in shapes)
    g = torch.Generator(device="cpu").manual_seed(seed)
    value = torch.randn(B, S, M, D, generator=g).half()
    off = (torch.randn(B, S, M, L, P, 2, generator=g) * off_scale).half()
    logits = (torch.randn(B, S, M, L * P, generator=g) * 2).half()
    if counts is None:
        counts = torch.tensor([[[w, h] for h, w in shapes]] * B, dtype=torch.float32)
    return value, off, logits, counts, S


def _pack(off, logits):
    """lane-major packed projection from the reference layout, through the library's own index table"""
    from codetr import _cabi

    B, S = off.shape[:2]
    cat = torch.cat((off.reshape(B, S, -1), logits.reshape(B, S, -1)), -1)
    idx = torch.tensor(_cabi.msda_pack_projection_index(M, L, P))
    packed = cat[..., idx.clamp_min(0)].clone()
    packed[..., idx < 0] = float("nan")      # the pad columns are never read: poison them
    return packed.contiguous()


def _expect(value, off, logits, counts, shapes, rows=None):
    """float64 oracle on the reference's prologue; rows = flattened query indices (all when None)"""
    from oracle import msda_oracle

    B, S = off.shape[:2]
    size = torch.tensor([[w, h] for h, w in shapes], dtype=torch.float64)
    vr = counts.double() / size                                       # get_valid_ratio (transformer.py:384-400)
    refs = []
    for l, (h, w) in enumerate(shapes):
        ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float64) + 0.5, torch.arange(w, dtype=torch.float64) + 0.5,
                                indexing="ij")
        base = torch.stack((xs.reshape(-1)[None] / (vr[:, l, 0, None] * w), ys.reshape(-1)[None] / (vr[:, l, 1, None] * h)), -1)
        refs.append(base[:, :, None, :] * vr[:, None, :, :])          # [B, hw, L, 2]
    ref = torch.cat(refs, 1)
    o, lg = off.double(), logits.double()
    if rows is not None:
        ref, o, lg = ref[:, rows], o[:, rows], lg[:, rows]
    w = torch.softmax(lg, -1).view(B, -1, M, L, P)
    loc = ref[:, :, None, :, None, :] + o / size[None, None, None, :, None, :]
    ssn = np.asarray(shapes, dtype=np.int64)
    return msda_oracle.msda_forward_numpy(value.double().numpy(), ssn, msda_oracle.level_start_index_from_shapes(ssn),
                                          loc.numpy(), w.numpy())


def _run(shapes, value, off, logits, counts, cfg, windows, head_major=False):
    from codetr import _cabi

    threads, region, _ = CFGS[cfg]
    B, S = off.shape[:2]
    out = torch.full((B, S, M * D), float("nan"), dtype=torch.float16, device=DEV)
    before = _cabi.CALLS["msda_encoder_packed"]
    v = value.to(DEV)
    if head_major:
        v = v.permute(0, 2, 1, 3).contiguous()
    ok = _cabi.msda_encoder_packed(v, shapes, _pack(off, logits).to(DEV), P, windows, counts.to(DEV), region, threads, out,
                                   head_major)
    torch.cuda.synchronize()
    assert ok and _cabi.CALLS["msda_encoder_packed"] == before + 1, "the packed encoder kernel did not take the shape"
    return out.float().cpu().numpy()


def _check(got, expect, what):
    np.testing.assert_allclose(got, expect, rtol=1e-2, atol=2e-3, err_msg=what)
    beyond = np.abs(got - expect) > 1e-3 + 1e-2 * np.abs(expect)
    assert beyond.mean() <= 1e-5, f"{what}: {int(beyond.sum())} of {beyond.size} elements beyond rtol 1e-2 / atol 1e-3"
    rel = np.linalg.norm(got - expect) / max(np.linalg.norm(expect), 1e-30)
    assert rel <= 1e-3, f"{what}: rel L2 {rel:.2e}"


def _halo(h):
    return [[(-h, h, -h, h)] * L] * M


@pytest.mark.parametrize("cfg", list(CFGS))
@pytest.mark.parametrize("shapes", [PYR_DIV, PYR_ODD], ids=["divisible", "odd"])
@pytest.mark.parametrize("off_scale,halo", [(1.0, 4), (6.0, 3), (60.0, 4), (3.0, None)],
                         ids=["inside", "beyond_window", "outside_image", "whole_levels"])
def test_against_float64_oracle(cfg, shapes, off_scale, halo):
    value, off, logits, counts, S = _inputs(shapes, 2, off_scale, seed=int(off_scale * 10) + len(cfg))
    # halo None: levels 2-4 wholly resident (windows clamped to the level + its zero border), levels 0-1 narrow
    win = _halo(halo) if halo is not None else [[(-3, 3, -3, 3)] * 2 + [(-40, 40, -40, 40)] * 3] * M
    got = _run(shapes, value, off, logits, counts, cfg, win)
    _check(got, _expect(value, off, logits, counts, shapes), f"{cfg} off {off_scale} halo {halo}")


@pytest.mark.parametrize("cfg", list(CFGS))
def test_padded_images(cfg):
    """valid pixel counts below the level size: reference points beyond 1 inside the padding, skewed per level"""
    shapes, B = PYR_DIV, 3
    counts = torch.tensor([[[w - (3 * b + l) % 4 - (w // 5 if b == 1 else 0), h - (2 * b + l) % 3 - (h // 3 if b == 2 else 0)]
                            for l, (h, w) in enumerate(shapes)] for b in range(B)], dtype=torch.float32)
    value, off, logits, _, S = _inputs(shapes, B, 2.0, seed=5)
    got = _run(shapes, value, off, logits, counts, cfg, _halo(4))
    _check(got, _expect(value, off, logits, counts, shapes), "padded")


@pytest.mark.parametrize("cfg", list(CFGS))
def test_windows_change_speed_not_results(cfg):
    value, off, logits, counts, S = _inputs(PYR_ODD, 2, 2.5, seed=77)
    expect = _expect(value, off, logits, counts, PYR_ODD)
    wins = {"one pixel": [[(0, 1, 0, 1)] * L] * M, "halo 1": _halo(1), "halo 6": _halo(6),
            "per head": [[(-m, 8 - m, m - 7, 2)] * L for m in range(M)],
            "mixed": [[(-1, 1, -1, 1), (-9, 9, -9, 9), (0, 0, 0, 0), (-127, 127, -127, 127), (-2, 30, -30, 2)]] * M}
    for name, w in wins.items():
        _check(_run(PYR_ODD, value, off, logits, counts, cfg, w), expect, name)


@pytest.mark.parametrize("head_major", [False, True], ids=["op_layout", "head_major"])
@pytest.mark.parametrize("seed", [300, 301, 302])
def test_both_value_layouts(seed, head_major):
    """the op's [B, S, M, D] value map and the head-major [B, M, S, D] one the product's value projection writes: same
    results; offsets partly beyond the windows so that the fix-up path runs"""
    value, off, logits, counts, S = _inputs(PYR_ODD, 2, 3.0, seed=seed)
    expect = _expect(value, off, logits, counts, PYR_ODD)
    for cfg in CFGS:
        _check(_run(PYR_ODD, value, off, logits, counts, cfg, _halo(3), head_major), expect, f"{cfg} head_major {head_major}")


@pytest.mark.parametrize("cfg", list(CFGS))
@pytest.mark.parametrize("head_major", [False, True], ids=["op_layout", "head_major"])
def test_bf16_model_packed_projection_and_output_fp16_value_map(cfg, head_major):
    """codetr_msda_encoder_forward_packed_bf16: offsets / logits arrive as bf16 and the result leaves as bf16; the value
    map is fp16 (the header says why).  Oracle on the bf16-rounded offsets / logits and the fp16 value map; tolerance =
    the output's own bf16 rounding (half an ulp = 2^-9 relative) on top of the packed blend's: rtol 1e-2 / atol 2e-3 and
    relative L2 <= 3e-3."""
    from codetr import _cabi

    value, off, logits, counts, S = _inputs(PYR_ODD, 2, 3.0, seed=555)
    off, logits = off.bfloat16(), logits.bfloat16()
    expect = _expect(value, off.float(), logits.float(), counts, PYR_ODD)
    threads, region, _ = CFGS[cfg]
    B = 2
    out = torch.full((B, S, M * D), float("nan"), dtype=torch.bfloat16, device=DEV)
    v = value.to(DEV)
    if head_major:
        v = v.permute(0, 2, 1, 3).contiguous()
    assert _cabi.msda_encoder_packed(v, PYR_ODD, _pack(off, logits).to(DEV), P, _halo(3), counts.to(DEV), region, threads, out,
                                     head_major)
    torch.cuda.synchronize()
    got = out.float().cpu().numpy()
    np.testing.assert_allclose(got, expect, rtol=1e-2, atol=2e-3)
    assert np.linalg.norm(got - expect) / np.linalg.norm(expect) <= 3e-3


def test_non_finite_offsets_drop_the_sample_like_the_reference_gate():
    """NaN / inf sampling offsets fail cu:249's comparisons: the sample contributes nothing, the other 19 do"""
    value, off, logits, counts, S = _inputs(PYR_ODD, 1, 1.5, seed=9)
    g = torch.Generator().manual_seed(3)
    hit = torch.rand(off.shape[:-1], generator=g) < 0.02
    kind = torch.randint(0, 3, off.shape[:-1], generator=g)
    vals = torch.tensor([float("nan"), float("inf"), float("-inf")], dtype=torch.float16)[kind]
    axis = torch.randint(0, 2, off.shape[:-1], generator=g)
    off[..., 0] = torch.where(hit & (axis == 0), vals, off[..., 0])
    off[..., 1] = torch.where(hit & (axis == 1), vals, off[..., 1])
    off_ref = torch.where(torch.isfinite(off), off, torch.full_like(off, 3e4))   # far outside: gated out by the oracle too
    got = _run(PYR_ODD, value, off, logits, counts, "t256", _halo(4))
    assert np.isfinite(got).all()
    _check(got, _expect(value, off_ref, logits, counts, PYR_ODD), "non-finite offsets")


def test_full_size_sampled_queries():
    """BASELINE's pyramid (1920x1280), the product's window policy, both workgroup shapes: ~800 queries of all levels
    incl. image corners and level boundaries against the oracle; and the two shapes against each other everywhere"""
    from codetr import hip_ops

    shapes = PYR_FULL
    value, off, logits, counts, S = _inputs(shapes, 1, 2.5, seed=43)
    counts[0, :, 0] -= torch.tensor([7.0, 3.0, 2.0, 1.0, 0.0])       # a slightly padded image
    g = torch.Generator().manual_seed(1)
    starts = [0]
    for h, w in shapes:
        starts.append(starts[-1] + h * w)
    idx = torch.cat([torch.randint(starts[l], starts[l + 1], (160,), generator=g) for l in range(L)]
                    + [torch.tensor([0, 479, 480 * 319, starts[1] - 1, starts[1], starts[4], S - 1])]).unique()
    expect = _expect(value, off, logits, counts, shapes, rows=idx)
    outs = []
    saved = (hip_ops.MSDA_V4_THREADS, hip_ops.MSDA_V4_REGION, hip_ops.MSDA_V4_LDS_BUDGET)
    try:
        for cfg, (threads, region, budget) in CFGS.items():
            hip_ops.MSDA_V4_THREADS, hip_ops.MSDA_V4_REGION, hip_ops.MSDA_V4_LDS_BUDGET = threads, region, budget
            bias = torch.randn(M * L * P * 2, generator=g) * 2
            win = hip_ops.msda_encoder_windows_packed(bias, shapes, M, L, P)
            assert len(win) == M and all(a <= b and c <= d for h in win for (a, b, c, d) in h)
            got = _run(shapes, value, off, logits, counts, cfg, win)
            _check(got[:, idx.numpy()], expect, f"1920x1280 {cfg}")
            outs.append(got)
    finally:
        hip_ops.MSDA_V4_THREADS, hip_ops.MSDA_V4_REGION, hip_ops.MSDA_V4_LDS_BUDGET = saved
    d = np.abs(outs[0] - outs[1])
    assert (d <= 3e-3 + 3e-3 * np.abs(outs[0])).all(), d.max()


def test_packed_projection_is_a_row_permutation():
    """hip_ops.msda_packed_projection: the packed Linear's output equals the two reference Linears' outputs, permuted"""
    from codetr import _cabi, hip_ops

    C = 256
    g = torch.Generator().manual_seed(0)
    w_off, b_off = torch.randn(M * L * P * 2, C, generator=g), torch.randn(M * L * P * 2, generator=g)
    w_aw, b_aw = torch.randn(M * L * P, C, generator=g), torch.randn(M * L * P, generator=g)
    wp, bp = hip_ops.msda_packed_projection(w_off, b_off, w_aw, b_aw, M, L, P)
    assert wp.shape == (64 * M, C) and bp.shape == (64 * M,)
    x = torch.randn(7, C, generator=g)
    y = x @ wp.t() + bp
    off = (x @ w_off.t() + b_off).view(7, M, L, P, 2)
    aw = (x @ w_aw.t() + b_aw).view(7, M, L, P)
    for m in (0, 3, 7):
        for p in range(P):
            seg = y[:, m * 64 + p * 16:m * 64 + p * 16 + 16]
            assert torch.allclose(seg[:, 0:10].reshape(7, L, 2), off[:, m, :, p, :], atol=1e-5)
            assert torch.allclose(seg[:, 10:15], aw[:, m, :, p], atol=1e-5)
            assert (seg[:, 15] == 0).all()
    assert _cabi.msda_pack_projection_index(M, 4, P) is None      # only the model's 5 levels x 4 points


def test_contract():
    from codetr import _cabi

    value, off, logits, counts, S = _inputs(PYR_DIV, 1, 1.0, seed=1)
    out = torch.empty(1, S, M * D, dtype=torch.float16, device=DEV)
    args = (value.to(DEV), PYR_DIV, _pack(off, logits).to(DEV), P)
    # windows that do not fit LDS, or a region with more queries than the waves hold: declined, nothing enqueued
    assert _cabi.msda_encoder_packed(*args, _halo(60), counts.to(DEV), (64, 64), 256, out) is False
    assert _cabi.msda_encoder_packed(*args, _halo(2), counts.to(DEV), (32, 16), 256, out) is False
    with pytest.raises(RuntimeError):      # level shapes that do not add up to S
        _cabi.msda_encoder_packed(args[0], PYR_DIV[:4] + [(1, 1)], args[2], P, _halo(2), counts.to(DEV), (16, 8), 256, out)
