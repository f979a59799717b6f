"""GPU parity: the HIP MSDA op (through torch.ops.codetr -> ctypes -> C ABI) against the golden
vectors captured from the reference and against the oracle on seeded inputs.

Tolerances (written here because the bar is "within the reference's own test tolerances"):
  fp64  rtol 1e-12                                   (reference: abs<1e-18, rel<1e-15 on g1)
  fp32  rtol 1e-5, atol 1e-6                         (reference tests :492-493; g1: abs<1e-9, rel<1e-6)
  fp16  rtol 1e-2, atol 1e-3 vs the fp32 oracle      (reference tests :62, :363-364) and
        <= 1 fp16 ulp of the exactly-rounded fp32 result on in-range outputs
  bf16  rtol 2e-2, atol 1e-2                         (no reference counterpart; 8-bit mantissa)
"""
import os

import numpy as np
import pytest
import torch

import msda_oracle as O
from conftest import GOLDEN

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
CASES = ["msda_g1", "msda_g2", "msda_g3_dec", "msda_g3_enc", "msda_g4"]


def _run(g_or_tuple, dtype, im2col_step=64):
    import codetr  # noqa: F401

    if isinstance(g_or_tuple, tuple):
        v, ss, ls, loc, w = g_or_tuple
    else:
        g = g_or_tuple
        v, ss, ls, loc, w = g["value"], g["spatial_shapes"], g["level_start_index"], g["sampling_loc"], g["attn_weight"]
    t = lambda a, dt: torch.as_tensor(np.asarray(a)).to(DEV).to(dt).contiguous()  # noqa: E731
    out = torch.ops.codetr.multi_scale_deformable_attention(
        t(v, dtype), t(ss, torch.int64), t(ls, torch.int64), t(loc, dtype), t(w, dtype), im2col_step)
    torch.cuda.synchronize()
    return out


@pytest.mark.parametrize("case", CASES)
def test_golden_fp64(case):
    g = np.load(os.path.join(GOLDEN, case + ".npz"))
    out = _run(g, torch.float64).cpu().numpy()
    ref = O.msda_forward_c(g["value"], g["spatial_shapes"], g["level_start_index"], g["sampling_loc"],
                           g["attn_weight"], dtype=np.float64)
    np.testing.assert_allclose(out, ref, rtol=1e-12, atol=1e-15)
    if "out_f64" in g.files and g["out_f64"].dtype == np.float64:
        np.testing.assert_allclose(out, g["out_f64"], rtol=1e-12, atol=1e-15)


@pytest.mark.parametrize("case", CASES)
def test_golden_fp32(case):
    g = np.load(os.path.join(GOLDEN, case + ".npz"))
    out = _run(g, torch.float32).cpu().numpy()
    np.testing.assert_allclose(out, g["out_f32"], rtol=1e-5, atol=1e-6)
    if case == "msda_g1":
        err = np.abs(out.astype(np.float64) - g["out_f32"])
        assert err.max() < 1e-9 and (err / np.abs(g["out_f32"])).max() < 1e-6


@pytest.mark.parametrize("case", CASES)
def test_golden_fp16(case):
    g = np.load(os.path.join(GOLDEN, case + ".npz"))
    out = _run(g, torch.float16).float().cpu().numpy()
    # fp16 tensors are first rounded to fp16; the oracle sees exactly those values
    h = lambda a: np.asarray(a).astype(np.float16).astype(np.float64)  # noqa: E731
    ref = O.msda_forward_c(h(g["value"]), g["spatial_shapes"], g["level_start_index"], h(g["sampling_loc"]),
                           h(g["attn_weight"]), dtype=np.float64)
    np.testing.assert_allclose(out, ref, rtol=1e-2, atol=1e-3)
    # single rounding: within 1 fp16 ulp (2^-10 relative) + fp32 accumulation noise
    np.testing.assert_allclose(out, ref, rtol=1.1 * 2.0 ** -10, atol=1e-6 + 6e-8)
    if "out_f16" in g.files:  # and the reference's own (lossier) fp16 path, at its tolerance
        np.testing.assert_allclose(out, g["out_f16"], rtol=1e-2, atol=1e-3)


@pytest.mark.parametrize("case", ["msda_g2", "msda_g3_dec", "msda_g4"])
def test_golden_bf16(case):
    g = np.load(os.path.join(GOLDEN, case + ".npz"))
    out = _run(g, torch.bfloat16).float().cpu().numpy()
    b = lambda a: torch.as_tensor(np.asarray(a)).to(torch.bfloat16).double().numpy()  # noqa: E731
    ref = O.msda_forward_c(b(g["value"]), g["spatial_shapes"], g["level_start_index"], b(g["sampling_loc"]),
                           b(g["attn_weight"]), dtype=np.float64)
    np.testing.assert_allclose(out, ref, rtol=2e-2, atol=1e-2)


def _random_case(seed, B, Nq, M, D, shapes, P, lo=-0.15, hi=1.15):
    rng = np.random.default_rng(seed)
    ss = np.array(shapes, dtype=np.int64)
    ls = O.level_start_index_from_shapes(ss)
    S = int((ss[:, 0] * ss[:, 1]).sum())
    L = len(shapes)
    v = rng.standard_normal((B, S, M, D)).astype(np.float32)
    loc = (rng.random((B, Nq, M, L, P, 2)) * (hi - lo) + lo).astype(np.float32)
    w = rng.random((B, Nq, M, L, P)).astype(np.float32)
    w /= w.sum((-1, -2), keepdims=True)
    return v, ss, ls, loc, w


@pytest.mark.parametrize("dtype,rtol,atol", [(torch.float32, 1e-5, 2e-6), (torch.float16, 2e-3, 2e-3)])
@pytest.mark.parametrize("shape", [
    # (B, Nq, M, D, shapes, P): every tiled lane count + scalar fallbacks + ragged tails
    (1, 1, 1, 8, [(1, 1)], 1),
    (3, 37, 8, 32, [(9, 13), (5, 7), (3, 4), (2, 2), (1, 1)], 4),   # model M/D/L/P, tail block
    (2, 11, 4, 16, [(8, 8), (4, 4)], 2),
    (1, 130, 2, 64, [(12, 7)], 3),
    (2, 5, 3, 128, [(4, 6), (2, 3)], 4),
    (1, 9, 5, 24, [(5, 5)], 2),     # D=24: 3 lanes (fp16) -> scalar; 6 lanes (fp32) -> scalar
    (2, 7, 2, 2, [(6, 4), (3, 2)], 2),
    (1, 3, 1, 256, [(3, 3)], 1),
])
def test_random_shapes_vs_oracle(shape, dtype, rtol, atol):
    B, Nq, M, D, shapes, P = shape
    v, ss, ls, loc, w = _random_case(11, B, Nq, M, D, shapes, P)
    if dtype == torch.float16:
        v, loc, w = (a.astype(np.float16).astype(np.float32) for a in (v, loc, w))
    out = _run((v, ss, ls, loc, w), dtype, im2col_step=B).float().cpu().numpy()
    ref = O.msda_forward_c(v, ss, ls, loc, w, dtype=np.float64, im2col_step=B)
    np.testing.assert_allclose(out, ref, rtol=rtol, atol=atol)


def test_encoder_shape_608_vs_oracle():
    """Config-2 pyramid (608x608), encoder-shaped call Nq = S = 30785, fp16: full tensor vs the C oracle."""
    shapes = [(152, 152), (76, 76), (38, 38), (19, 19), (10, 10)]
    v, ss, ls, loc, w = _random_case(5, 1, sum(h * w_ for h, w_ in shapes), 8, 32, shapes, 4, lo=-0.02, hi=1.02)
    v, loc, w = (a.astype(np.float16).astype(np.float32) for a in (v, loc, w))
    out = _run((v, ss, ls, loc, w), torch.float16).float().cpu().numpy()
    ref = O.msda_forward_c(v, ss, ls, loc, w, dtype=np.float32)
    np.testing.assert_allclose(out, ref, rtol=2e-3, atol=2e-3)


def test_full_size_properties_1920x1280():
    """BASELINE full size (S = Nq = 204600, batch 2): size-independent properties instead of a CPU oracle run.
    (1) constant value map + in-image points + normalised weights -> the constant;
    (2) linearity in value;  (3) batch independence (image 1 alone == image 1 inside the batch)."""
    import codetr  # noqa: F401

    shapes = [(320, 480), (160, 240), (80, 120), (40, 60), (20, 30)]
    ss = torch.tensor(shapes, dtype=torch.int64, device=DEV)
    ls = torch.cat((ss.new_zeros(1), ss.prod(1).cumsum(0)[:-1]))
    S = int(ss.prod(1).sum())
    B, M, D, L, P = 2, 8, 32, 5, 4
    g = torch.Generator(device=DEV).manual_seed(0)
    loc = (torch.rand(B, S, M, L, P, 2, device=DEV, generator=g) * 0.9 + 0.05).half()
    w = torch.rand(B, S, M, L, P, device=DEV, generator=g)
    w = (w / w.sum((-1, -2), keepdim=True)).half()
    op = torch.ops.codetr.multi_scale_deformable_attention
    const = torch.full((B, S, M, D), 0.5, device=DEV, dtype=torch.float16)
    out = op(const, ss, ls, loc, w, 64)
    wsum = w.float().sum((-1, -2))  # fp16-rounded weights do not sum to exactly 1
    expect = (0.5 * wsum)[..., None].expand(B, S, M, D).reshape(B, S, M * D)
    torch.testing.assert_close(out.float(), expect, rtol=2e-3, atol=1e-3)
    v1 = torch.randn(B, S, M, D, device=DEV, generator=g).half()
    v2 = torch.randn(B, S, M, D, device=DEV, generator=g).half()
    o1, o2, o12 = op(v1, ss, ls, loc, w, 64), op(v2, ss, ls, loc, w, 64), op((v1 + v2), ss, ls, loc, w, 64)
    torch.testing.assert_close(o12.float(), o1.float() + o2.float(), rtol=1e-2, atol=1e-2)
    solo = op(v1[1:].contiguous(), ss, ls, loc[1:].contiguous(), w[1:].contiguous(), 64)
    assert torch.equal(solo, o1[1:])


def test_contract_errors():
    import codetr  # noqa: F401

    g = np.load(os.path.join(GOLDEN, "msda_g2.npz"))
    op = torch.ops.codetr.multi_scale_deformable_attention
    t = lambda k, dt: torch.as_tensor(g[k]).to(DEV).to(dt)  # noqa: E731
    v, ss, ls = t("value", torch.float16), t("spatial_shapes", torch.int64), t("level_start_index", torch.int64)
    loc, w = t("sampling_loc", torch.float16), t("attn_weight", torch.float16)
    v3, loc3, w3 = (torch.cat([a, a[:1]]) for a in (v, loc, w))
    with pytest.raises(RuntimeError, match="must divide im2col_step"):
        op(v3, ss, ls, loc3, w3, 2)
    with pytest.raises(RuntimeError, match="contiguous"):
        op(v.transpose(2, 3), ss, ls, loc, w, 64)
    with pytest.raises(RuntimeError, match="dtype"):
        op(v, ss, ls, loc.float(), w, 64)
    assert op(v[:0], ss, ls, loc[:0], w[:0], 64).shape == (0, 8, 64)  # empty batch


def test_opcheck_and_stream_semantics():
    """reference tests :44 (opcheck) + the op enqueues on torch's CURRENT stream."""
    import codetr  # noqa: F401

    g = np.load(os.path.join(GOLDEN, "msda_g2.npz"))
    t = lambda k, dt: torch.as_tensor(g[k]).to(DEV).to(dt)  # noqa: E731
    args = (t("value", torch.float32), t("spatial_shapes", torch.int64), t("level_start_index", torch.int64),
            t("sampling_loc", torch.float32), t("attn_weight", torch.float32), 2)
    torch.library.opcheck(torch.ops.codetr.multi_scale_deformable_attention.default, args,
                          test_utils=("test_schema", "test_faketensor"))
    base = torch.ops.codetr.multi_scale_deformable_attention(*args)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        other = torch.ops.codetr.multi_scale_deformable_attention(*args)
    s.synchronize()
    assert torch.equal(base, other)
    # graph capture: nothing in the launch path syncs or allocates outside torch's allocator
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        captured = torch.ops.codetr.multi_scale_deformable_attention(*args)
    gr.replay()
    torch.cuda.synchronize()
    assert torch.equal(base, captured)


# ---------------------------------------------------------------------------------------------------
# fused variant: softmax + sampling-location arithmetic inside the kernel
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("head_major", [False, True])
@pytest.mark.parametrize("ref_dim", [2, 4])
@pytest.mark.parametrize("extra_cols", [0, 24])
def test_fused_prologue_vs_oracle(ref_dim, extra_cols, head_major):
    """fused kernel == oracle( softmax(logits), ref + normalised offsets ) computed in fp64 on the host."""
    from codetr import _cabi, hip_ops

    rng = np.random.default_rng(3)
    B, Nq, M, D, P = 2, 77, 8, 32, 4
    shapes = [(9, 13), (5, 7), (3, 4), (2, 1), (1, 1)]  # incl. W == 1 levels: x1 never exists there
    L = len(shapes)
    ss = np.array(shapes, dtype=np.int64)
    ls = O.level_start_index_from_shapes(ss)
    S = int((ss[:, 0] * ss[:, 1]).sum())
    h = lambda a: a.astype(np.float16).astype(np.float64)  # noqa: E731
    value = h(rng.standard_normal((B, S, M, D)))
    off = h(rng.standard_normal((B, Nq, M, L, P, 2)) * 2.0)
    logits = h(rng.standard_normal((B, Nq, M, L * P)) * 2.0)
    ref = h(rng.random((B, Nq, L, ref_dim)) * (0.6 if ref_dim == 4 else 1.0) + (0.1 if ref_dim == 4 else 0.0))
    w = np.exp(logits - logits.max(-1, keepdims=True))
    w = (w / w.sum(-1, keepdims=True)).reshape(B, Nq, M, L, P)
    if ref_dim == 2:
        norm = np.stack((ss[:, 1], ss[:, 0]), -1).astype(np.float64)
        loc = ref[:, :, None, :, None, :] + off / norm[None, None, None, :, None, :]
    else:
        loc = ref[:, :, None, :, None, :2] + off / P * ref[:, :, None, :, None, 2:] * 0.5
    expect = O.msda_forward_c(value, ss, ls, loc, w, dtype=np.float64, im2col_step=B)
    # device side: one projection matrix holding (offsets | logits | unrelated columns)
    proj = np.concatenate((off.reshape(B, Nq, -1), logits.reshape(B, Nq, -1),
                           rng.standard_normal((B, Nq, extra_cols))), -1)
    t = lambda a, dt: torch.as_tensor(np.asarray(a)).to(DEV).to(dt).contiguous()  # noqa: E731
    before = _cabi.CALLS["msda_fused"]
    vdev = t(value, torch.float16)
    if head_major:
        vdev = vdev.permute(0, 2, 1, 3).contiguous()  # [B, M, S, D]
    out = hip_ops.msda_fused(vdev, t(ss, torch.int64), t(ls, torch.int64), t(proj, torch.float16), 0,
                             M * L * P * 2, t(ref, torch.float16), L, P, head_major=head_major)
    torch.cuda.synchronize()
    assert _cabi.CALLS["msda_fused"] == before + 1
    np.testing.assert_allclose(out.float().cpu().numpy(), expect, rtol=2e-3, atol=2e-3)


def test_module_takes_fused_path_and_masks_value_rows():
    """MultiScaleDeformableAttention (fp16) = fused projection GEMM + fused MSDA; key_padding_mask rows of the value
    map are zeroed inside the value_proj GEMM.  Compared with the reference's module output (golden)."""
    from codetr import _cabi
    from codetr.multi_scale_deformable_attention import MultiScaleDeformableAttention
    from helpers_model import assert_close_lowp, seeded_params, unpack_param_spec

    g = np.load(os.path.join(GOLDEN, "model_msda_module.npz"))
    m = MultiScaleDeformableAttention(embed_dims=256, num_levels=5, dropout=0.0)
    m.load_state_dict(seeded_params(unpack_param_spec(g), int(g["seed"])))
    m = m.to(DEV).half().eval()
    t = lambda k, dt=torch.float16: torch.as_tensor(g[k]).to(DEV).to(dt)  # noqa: E731
    before = dict(_cabi.CALLS)
    with torch.no_grad():
        out2 = m(t("value"), value=None, query_pos=t("query_pos"), key_padding_mask=t("key_padding_mask", torch.bool),
                 reference_points=t("ref2"), spatial_shapes=t("spatial_shapes", torch.int64),
                 level_start_index=t("level_start_index", torch.int64))
        out4 = m(t("query4"), value=t("value"), query_pos=t("query_pos4"), key_padding_mask=t("key_padding_mask", torch.bool),
                 reference_points=t("ref4"), spatial_shapes=t("spatial_shapes", torch.int64),
                 level_start_index=t("level_start_index", torch.int64))
    assert _cabi.CALLS["msda_fused"] - before["msda_fused"] == 2 and _cabi.CALLS["msda"] == before["msda"]
    assert _cabi.CALLS["linear"] - before["linear"] == 6  # value_proj, (offsets|logits), output_proj  x 2 calls
    assert_close_lowp(out2.float().cpu().numpy(), g["out2"], 5e-3, 5e-2, "module out, 2-d refs")
    assert_close_lowp(out4.float().cpu().numpy(), g["out4"], 5e-3, 5e-2, "module out, 4-d refs")
