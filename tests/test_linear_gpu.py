"""GPU parity of the hand-written MFMA linear kernel (C ABI codetr_linear_{f16,bf16}) against a
plain PyTorch fp32 reference of the same op:  y = act(x @ w.T + b) (+ r).

Tolerance: inputs are exactly representable (fp16/bf16), accumulation is fp32 on both sides, so the
only differences are summation order and the output rounding: |err| <= 2^-10 |y| + K * 2^-22
scale for fp16 (1 ulp of the rounded result + fp32 accumulation noise), 2^-7 for bf16; with a residual
the linear output is rounded before the add (as in the reference's fp16 path), one more ulp of it."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _ref(x, w, b, r, act):
    y = x.float() @ w.float().t()
    if b is not None:
        y = y + b.float()
    if act == "relu":
        y = torch.relu(y)
    elif act == "gelu":
        y = torch.nn.functional.gelu(y)
    if r is not None:
        y = y + r.float()
    return y


def _check(M, N, K, dtype, bias, act, res, seed=0):
    from codetr import hip_ops

    g = torch.Generator(device=DEV).manual_seed(seed)
    x = (torch.randn(M, K, device=DEV, generator=g)).to(dtype)
    w = (torch.randn(N, K, device=DEV, generator=g) / K ** 0.5).to(dtype)
    b = torch.randn(N, device=DEV, generator=g).to(dtype) if bias else None
    r = torch.randn(M, N, device=DEV, generator=g).to(dtype) if res else None
    y = hip_ops.linear(x, w, b, act=act, residual=r)
    torch.cuda.synchronize()
    ref = _ref(x, w, b, r, act)
    assert y.shape == (M, N) and y.dtype == dtype
    ulp = 2.0 ** -10 if dtype == torch.float16 else 2.0 ** -7
    tol = ulp * ref.abs() + 1e-3 * ulp * 64
    if r is not None:  # residual is added to the ROUNDED linear output (two roundings, as `identity + linear(x)`)
        tol = tol + ulp * (ref - r.float()).abs()
    bad = (y.float() - ref).abs() > tol + K * 2.0 ** -22
    assert not bad.any(), f"{int(bad.sum())} / {bad.numel()} outside 1 ulp; max err {(y.float() - ref).abs().max().item()}"


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M,N,K", [
    (128, 128, 64),        # exactly one tile, one K step
    (256, 256, 256),
    (1, 1, 64),            # degenerate edges
    (130, 4, 256),         # box-regression head: N = 4
    (900, 80, 256),        # class head: N = 80 (not a multiple of 16)
    (333, 161, 192),       # ragged M and N, N % 4 != 0 -> scalar store path
    (1000, 320, 256),      # sampling offsets
    (4097, 576, 192),      # Swin stage-0 qkv shape class
    (2880, 1536, 6144),    # Swin stage-3 fc2: long K
])
def test_linear_shapes(M, N, K, dtype):
    _check(M, N, K, dtype, bias=True, act=None, res=False)


@pytest.mark.parametrize("bias,act,res", [(False, None, False), (True, "relu", False), (True, "gelu", False),
                                          (True, None, True), (False, "relu", True), (True, "gelu", True)])
def test_linear_epilogues(bias, act, res):
    _check(517, 384, 384, torch.float16, bias, act, res, seed=3)


def test_linear_model_shape_encoder_ffn():
    """encoder FFN at the 608x608 pyramid: M = 30785, 256 -> 2048 (ReLU) -> 256 (+ identity)."""
    _check(30785, 2048, 256, torch.float16, True, "relu", False, seed=1)
    _check(30785, 256, 2048, torch.float16, True, None, True, seed=2)


def test_linear_batched_view_and_noncontiguous_input():
    from codetr import hip_ops

    g = torch.Generator(device=DEV).manual_seed(5)
    x = torch.randn(3, 50, 128, device=DEV, generator=g).half()
    w = torch.randn(64, 128, device=DEV, generator=g).half() * 0.1
    b = torch.randn(64, device=DEV, generator=g).half()
    y = hip_ops.linear(x, w, b)
    assert y.shape == (3, 50, 64)
    torch.testing.assert_close(y.float(), _ref(x.reshape(-1, 128), w, b, None, None).view(3, 50, 64), rtol=2e-3, atol=2e-3)
    xt = x.transpose(0, 1)  # non-contiguous view
    yt = hip_ops.linear(xt, w, b)
    torch.testing.assert_close(yt, y.transpose(0, 1))
    # sliced weight (the fused q|k in-projection of nn.MultiheadAttention is passed as a row slice)
    w2 = torch.randn(192, 128, device=DEV, generator=g).half() * 0.1
    y2 = hip_ops.linear(x, w2[64:128], None)
    torch.testing.assert_close(y2.float(), _ref(x.reshape(-1, 128), w2[64:128], None, None, None).view(3, 50, 64),
                               rtol=2e-3, atol=2e-3)


def test_linear_unsupported_k_is_loud():
    from codetr import _cabi

    x = torch.zeros(8, 48, device=DEV, dtype=torch.float16)
    w = torch.zeros(8, 48, device=DEV, dtype=torch.float16)
    out = torch.empty(8, 8, device=DEV, dtype=torch.float16)
    assert not _cabi.linear_supported(x, w)
    with pytest.raises(RuntimeError, match="outside what the kernel family implements"):
        _cabi.linear(x, w, None, None, None, out)


def test_linear_row_mask():
    from codetr import hip_ops

    g = torch.Generator(device=DEV).manual_seed(8)
    x = torch.randn(3, 100, 256, device=DEV, generator=g).half()
    w = (torch.randn(256, 256, device=DEV, generator=g) / 16).half()
    b = torch.randn(256, device=DEV, generator=g).half()
    r = torch.randn(3, 100, 256, device=DEV, generator=g).half()
    mask = torch.rand(3, 100, device=DEV, generator=g) < 0.3
    y = hip_ops.linear(x, w, b, row_mask=mask)
    ref = _ref(x.reshape(-1, 256), w, b, None, None).view(3, 100, 256).masked_fill(mask[..., None], 0.0)
    torch.testing.assert_close(y.float(), ref, rtol=2e-3, atol=2e-3)
    assert (y[mask] == 0).all()
    y2 = hip_ops.linear(x, w, b, residual=r, row_mask=mask)  # mask applies to the linear output, residual still added
    torch.testing.assert_close(y2.float(), ref + r.float(), rtol=2e-3, atol=4e-3)
    w5 = (torch.randn(5, 256, device=DEV, generator=g) / 16).half()  # ragged-N path
    y3 = hip_ops.linear(x, w5, None, row_mask=mask)
    assert (y3[mask] == 0).all() and (y3[~mask] != 0).any()


def test_linear_head_major_output():
    """y[b][head][pos][ch] layout of the value projection (+ row mask) == permuted row-major result."""
    from codetr import hip_ops

    g = torch.Generator(device=DEV).manual_seed(4)
    B, S, K, Hh, hd = 2, 333, 256, 8, 32
    x = torch.randn(B, S, K, device=DEV, generator=g).half()
    w = (torch.randn(Hh * hd, K, device=DEV, generator=g) / 16).half()
    b = torch.randn(Hh * hd, device=DEV, generator=g).half()
    mask = torch.rand(B, S, device=DEV, generator=g) < 0.2
    y_rm = hip_ops.linear(x, w, b, row_mask=mask)                     # [B,S,N]
    y_hm = hip_ops.linear(x, w, b, row_mask=mask, head_major=hd)      # [B,H,S,hd]
    assert y_hm.shape == (B, Hh, S, hd)
    assert torch.equal(y_hm, y_rm.view(B, S, Hh, hd).permute(0, 2, 1, 3))


def test_linear_head_major_output_large_short_k():
    """the same at the encoder's value-projection shape class (>= 32 k rows, K = 256: the X-stationary kernel, which since
    round 5 writes 32-wide column blocks -- the encoder MSDA's head-major value map [B, M, S, 32]) incl. a row mask"""
    from codetr import _cabi, hip_ops

    g = torch.Generator(device=DEV).manual_seed(5)
    B, S, K, Hh, hd = 2, 20000, 256, 8, 32
    x = torch.randn(B, S, K, device=DEV, generator=g).half()
    w = (torch.randn(Hh * hd, K, device=DEV, generator=g) / 16).half()
    b = torch.randn(Hh * hd, device=DEV, generator=g).half()
    mask = torch.rand(B, S, device=DEV, generator=g) < 0.2
    assert _cabi.linear_variant(B * S, Hh * hd, K, None, False, hd) == "xs"
    y_rm = hip_ops.linear(x, w, b, row_mask=mask)
    y_hm = hip_ops.linear(x, w, b, row_mask=mask, head_major=hd)
    assert y_hm.shape == (B, Hh, S, hd)
    assert torch.equal(y_hm, y_rm.view(B, S, Hh, hd).permute(0, 2, 1, 3))


def test_value_projection_bf16_operands_fp16_head_major_output():
    """codetr_linear_bf16_f16out (the bf16 model's value projection in front of the packed encoder MSDA kernel): bf16
    x / w / bias, fp32 accumulation, ONE rounding to fp16, head-major [B, M, S, 32] destination, masked rows zero"""
    from codetr import hip_ops

    g = torch.Generator(device=DEV).manual_seed(6)
    B, S, K, Hh, hd = 2, 20000, 256, 8, 32
    x = torch.randn(B, S, K, device=DEV, generator=g).bfloat16()
    w = (torch.randn(Hh * hd, K, device=DEV, generator=g) / 16).bfloat16()
    b = torch.randn(Hh * hd, device=DEV, generator=g).bfloat16()
    mask = torch.rand(B, S, device=DEV, generator=g) < 0.2
    y = hip_ops.value_projection_f16(x, w, b, mask, hd)
    assert y is not None and y.dtype == torch.float16 and y.shape == (B, Hh, S, hd)
    ref = (x.float() @ w.float().t() + b.float()).masked_fill(mask[..., None], 0.0).view(B, S, Hh, hd).permute(0, 2, 1, 3)
    err = (y.float() - ref).abs()
    assert bool((err <= 2.0 ** -10 * ref.abs() + 1e-3 * 2.0 ** -10 * 64 + K * 2.0 ** -22).all()), float(err.max())   # 1 fp16 ulp
    assert hip_ops.value_projection_f16(x[:, :100], w, b, None, hd) is None      # too few rows for the short-K kernel


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M,N,K,bias,act,res", [
    (600, 256, 13824, False, None, False),   # the neck's extra 3x3/s2 level over unfolded patches (1920x1280 input)
    (900, 256, 2048, True, None, True),      # decoder FFN down-projection + identity
    (77, 130, 4096, True, "relu", False),    # ragged M and N (scalar partial stores), K ranges of unequal length
    (128, 8, 2112, True, "gelu", True),      # 33 K tiles
    (1444, 768, 3072, True, None, True),     # 65 ... 128 tiles: Swin stage-2 fc2 of one 608x608 image (72 tiles, 3 passes)
    (864, 1536, 6144, True, None, True),     # Swin stage-3 fc2 of one 1152x768 image (84 tiles, 3 passes)
])
def test_linear_splitk(M, N, K, bias, act, res, dtype):
    """Few output tiles + long K go through the two-pass split-K path (plan > 1) and match the fp32 reference."""
    from codetr import _cabi

    splits, nbytes = _cabi.linear_splitk_plan(M, N, K)
    assert splits > 1 and nbytes == splits * M * N * 4
    before = _cabi.CALLS["linear_splitk"]
    _check(M, N, K, dtype, bias, act, res, seed=11)
    assert _cabi.CALLS["linear_splitk"] == before + 1


def test_linear_splitk_plan_and_row_mask():
    from codetr import _cabi, hip_ops

    assert _cabi.linear_splitk_plan(204600, 256, 2048) == (1, 0)   # plenty of tiles: single pass
    assert _cabi.linear_splitk_plan(900, 256, 256) == (1, 0)       # short K
    assert _cabi.linear_splitk_plan(1444, 768, 3072)[0] == 3       # 72 tiles: 3 x 72 = 216 blocks, one round of the chip
    assert _cabi.linear_splitk_plan(2048, 1024, 4096)[0] == 2      # 128 tiles
    assert _cabi.linear_splitk_plan(2176, 1024, 4096) == (1, 0)    # 136 tiles: single pass
    g = torch.Generator(device=DEV).manual_seed(2)
    x = torch.randn(300, 2048, device=DEV, generator=g).half()
    w = (torch.randn(64, 2048, device=DEV, generator=g) / 45).half()
    b = torch.randn(64, device=DEV, generator=g).half()
    mask = torch.rand(300, device=DEV, generator=g) < 0.3
    before = _cabi.CALLS["linear_splitk"]
    y = hip_ops.linear(x, w, b, row_mask=mask)
    assert _cabi.CALLS["linear_splitk"] == before + 1
    ref = _ref(x, w, b, None, None).masked_fill(mask[:, None], 0.0)
    torch.testing.assert_close(y.float(), ref, rtol=2e-3, atol=2e-3)
    assert (y[mask] == 0).all()
    # a workspace smaller than the plan is refused, not overrun
    out = torch.empty(300, 64, device=DEV, dtype=torch.float16)
    splits, nbytes = _cabi.linear_splitk_plan(300, 64, 2048)
    with pytest.raises(RuntimeError):
        _cabi.linear_splitk(x, w, b, None, None, out, splits, torch.empty(nbytes - 16, dtype=torch.uint8, device=DEV))


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M,N,K,bias,act,res", [
    (32768, 256, 256, True, None, False),      # exactly 256 workgroups
    (40000, 480, 256, True, None, False),      # offsets | logits: N % 32 != 0 (ragged last W chunk)
    (33001, 256, 256, True, None, True),       # out_proj + identity, ragged M
    (35000, 136, 256, True, None, False),      # odd number of 32-column chunks + a partial one
    (34000, 128, 256, False, "relu", True),    # no bias
    (36000, 576, 192, True, None, False),      # Swin stage-0 qkv (K = 192: 24 chunks per row)
    (36000, 768, 192, True, "gelu", False),    # Swin stage-0 fc1
    (33000, 192, 192, True, None, True),       # Swin stage-0 proj + identity
    (33000, 1152, 384, True, None, False),     # Swin stage-1 qkv (K = 384: stays on the tiled kernel)
])
def test_linear_short_k_x_stationary_kernel(M, N, K, bias, act, res, dtype):
    """M >= 32768 rows with K in {192, 256} and 128 <= N <= 992 run the X-stationary kernel (linear_xs_kernel); same
    tolerance."""
    _check(M, N, K, dtype, bias, act, res, seed=21)


def test_linear_short_k_row_masks():
    from codetr import hip_ops

    g = torch.Generator(device=DEV).manual_seed(9)
    M = 33333
    x = torch.randn(M, 256, device=DEV, generator=g).half()
    w = (torch.randn(256, 256, device=DEV, generator=g) / 16).half()
    b = torch.randn(256, device=DEV, generator=g).half()
    r = torch.randn(M, 256, device=DEV, generator=g).half()
    state = torch.zeros(M, dtype=torch.uint8, device=DEV)
    state[::3] = 1
    state[1::7] = 2
    y = hip_ops.linear(x, w, b, act="relu", residual=r, row_mask=state)
    ref = torch.relu(x.float() @ w.float().t() + b.float()).half()
    ref[state == 1] = 0
    ref[state == 2] = torch.relu(b.float()).half()
    ref = (ref.float() + r.float())
    torch.testing.assert_close(y.float(), ref, rtol=2e-3, atol=4e-3)
    # the two kernels agree bit for bit on the masked rows and to rounding elsewhere
    import os
    y_small = hip_ops.linear(x[:1000], w, b, act="relu", residual=r[:1000], row_mask=state[:1000])  # tiled kernel (M < 32768)
    torch.testing.assert_close(y[:1000].float(), y_small.float(), rtol=2e-3, atol=2e-3)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M,N,K,bias,act,res", [
    (32768, 1024, 512, True, None, False),      # exactly 512 tiles of 256 x 256
    (33000, 1000, 768, True, "gelu", False),    # ragged M and N (N % 8 == 0), GELU epilogue
    (40000, 768, 1024, True, None, True),       # + identity
    (36000, 1024, 576, False, "relu", True),    # no bias, K = 9 steps of 64
])
def test_linear_256_tile_kernel(M, N, K, bias, act, res, dtype):
    """>= 200 tiles of 256 x 256 with K >= 384 run linear_256_kernel (8 waves, 128 x 64 per wave); same tolerance."""
    _check(M, N, K, dtype, bias, act, res, seed=31)


def test_linear_256_tile_row_states():
    from codetr import hip_ops

    g = torch.Generator(device=DEV).manual_seed(12)
    M, N, K = 33333, 1024, 512
    x = torch.randn(M, K, device=DEV, generator=g).half()
    w = (torch.randn(N, K, device=DEV, generator=g) / K ** 0.5).half()
    b = torch.randn(N, device=DEV, generator=g).half()
    state = torch.zeros(M, dtype=torch.uint8, device=DEV)
    state[::5] = 1
    state[2::11] = 2
    y = hip_ops.linear(x, w, b, row_mask=state)
    ref = (x.float() @ w.float().t() + b.float()).half()
    ref[state == 1] = 0
    ref[state == 2] = b
    torch.testing.assert_close(y.float(), ref.float(), rtol=2e-3, atol=4e-3)
    assert torch.equal(y[state == 1], torch.zeros_like(y[state == 1])) and torch.equal(y[state == 2], ref[state == 2])


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M,N,K", [(40000, 480, 256), (32768 + 77, 576, 192), (33000, 256, 256)])
def test_linear_xadd_matches_add_then_linear(M, N, K, dtype):
    """codetr_linear_xadd_*: (x + x_add) rounded to the storage type inside the operand load, then the GEMM: against
    the fp32 product of the SAME rounded sum (the separate add kernel's output), and against add + codetr_linear_*
    (which may run a different tile kernel: equal up to the accumulation order)."""
    from codetr import _cabi, hip_ops

    hip_ops.XADD_MIN_ROWS, saved = 0, hip_ops.XADD_MIN_ROWS     # (the host's row threshold is a tuning choice)
    try:
        with torch.no_grad():       # (inference path: the fused form is not offered while autograd records)
            _xadd_case(M, N, K, dtype, _cabi, hip_ops)
    finally:
        hip_ops.XADD_MIN_ROWS = saved


def _xadd_case(M, N, K, dtype, _cabi, hip_ops):
    g = torch.Generator(device=DEV).manual_seed(M + N)
    x = torch.randn(M, K, device=DEV, generator=g).to(dtype)
    a = torch.randn(M, K, device=DEV, generator=g).to(dtype)
    w = (torch.randn(N, K, device=DEV, generator=g) / K ** 0.5).to(dtype)
    b = torch.randn(N, device=DEV, generator=g).to(dtype)
    assert hip_ops.linear_xadd_supported(x, a, w)
    y = hip_ops.linear_xadd(x, a, w, b)
    ref = hip_ops.linear(x + a, w, b)
    exact = (x + a).float() @ w.float().t() + b.float()
    # one rounding of the result: half an ulp at the largest magnitude (2^-11 / 2^-8 relative), with margin
    tol = (1e-3 if dtype == torch.float16 else 8e-3) * exact.abs().max().item()
    assert (y.float() - exact).abs().max().item() < tol
    assert (y.float() - ref.float()).abs().max().item() < tol
    # shapes outside the short-K kernel: the wrapper adds and calls linear (still correct)
    xs, as_ = x[:500], a[:500]
    assert not hip_ops.linear_xadd_supported(xs, as_, w)
    assert torch.equal(hip_ops.linear_xadd(xs, as_, w, b), hip_ops.linear(xs + as_, w, b))
    out = torch.empty(500, N, dtype=dtype, device=DEV)
    assert _cabi.linear_xadd(xs.contiguous(), as_.contiguous(), w, b, out) is False


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("B,S,masked", [(2, 20000, True), (1, 32768 + 77, False)])
def test_encoder_projections_one_launch_matches_the_two_gemms(B, S, masked, dtype):
    """codetr_encoder_projections_*: value = x W_v^T + b_v (row mask, head-major [B, 8, S, 32], FP16) and packed =
    (x + pos) W_p^T + b_p from ONE launch of the X-stationary kernel.  Same kernel, same accumulation order as the two
    launches it replaces -> bit-equal to them (value: codetr_linear_* head-major / codetr_linear_bf16_f16out; packed:
    codetr_linear_xadd_*); and within one ulp of the fp32 products."""
    from codetr import _cabi, hip_ops

    g = torch.Generator(device=DEV).manual_seed(B * S)
    K, Nv, Np, hd = 256, 256, 512, 32
    x = torch.randn(B, S, K, device=DEV, generator=g).to(dtype)
    pos = torch.randn(B, S, K, device=DEV, generator=g).to(dtype)
    wv = (torch.randn(Nv, K, device=DEV, generator=g) / 16).to(dtype)
    bv = torch.randn(Nv, device=DEV, generator=g).to(dtype)
    wp = (torch.randn(Np, K, device=DEV, generator=g) / 16).to(dtype)
    bp = torch.randn(Np, device=DEV, generator=g).to(dtype)
    mask = (torch.rand(B, S, device=DEV, generator=g) < 0.2) if masked else None
    wc, bc = torch.cat((wv, wp), 0).contiguous(), torch.cat((bv, bp), 0).contiguous()
    hip_ops.XADD_MIN_ROWS, saved = 0, hip_ops.XADD_MIN_ROWS
    try:
        with torch.no_grad():
            before = _cabi.CALLS["encoder_projections"]
            both = hip_ops.encoder_projections(x, pos, wc, bc, mask, Nv, hd)
            assert both is not None and _cabi.CALLS["encoder_projections"] == before + 1
            value, packed = both
            assert value.shape == (B, Nv // hd, S, hd) and value.dtype == torch.float16
            assert packed.shape == (B, S, Np) and packed.dtype == dtype
            if dtype == torch.float16:
                v2 = hip_ops.linear(x, wv, bv, row_mask=mask, head_major=hd)
            else:
                v2 = hip_ops.value_projection_f16(x, wv, bv, mask, hd)
            p2 = hip_ops.linear_xadd(x, pos, wp, bp)
            assert torch.equal(value, v2)
            assert torch.equal(packed, p2)
            # fp32 products of the same operands
            ref_v = x.float() @ wv.float().t() + bv.float()
            if mask is not None:
                ref_v = ref_v.masked_fill(mask[..., None], 0.0)
            ref_v = ref_v.view(B, S, Nv // hd, hd).permute(0, 2, 1, 3)
            err = (value.float() - ref_v).abs()
            assert bool((err <= 2.0 ** -10 * ref_v.abs() + 1e-3 * 2.0 ** -10 * 64 + K * 2.0 ** -22).all()), float(err.max())
            ref_p = (x + pos).float() @ wp.float().t() + bp.float()
            ulp = 2.0 ** -10 if dtype == torch.float16 else 2.0 ** -7
            err = (packed.float() - ref_p).abs()
            assert bool((err <= ulp * ref_p.abs() + 1e-3 * ulp * 64 + K * 2.0 ** -22).all()), float(err.max())
            # declined shapes: too few rows, a value width that is not a multiple of 64 columns
            assert hip_ops.encoder_projections(x[:, :100], pos[:, :100], wc, bc, None, Nv, hd) is None
            assert hip_ops.encoder_projections(x, pos, wc[32:].contiguous(), bc[32:].contiguous(), None, Nv - 32, hd) is None
    finally:
        hip_ops.XADD_MIN_ROWS = saved


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M,N,K,act", [(40000, 576, 192, None), (32768 + 33, 768, 192, "gelu"), (33000, 256, 256, "relu")])
def test_linear_ln_matches_layernorm_then_linear(M, N, K, act, dtype):
    """codetr_linear_ln_*: LayerNorm of the rows inside the short-K GEMM's operand load (Swin norm1 -> qkv, norm2 -> fc1)
    against the fp32 formula and against codetr_layernorm_* + codetr_linear_* (statistics are summed in a different
    lane order, so the normalised operand may differ in its last bit on a few elements)."""
    import torch.nn.functional as F
    from codetr import _cabi, hip_ops

    g = torch.Generator(device=DEV).manual_seed(M + N + K)
    x = (torch.randn(M, K, device=DEV, generator=g) * 2 + 0.3).to(dtype)
    gam = (1 + 0.1 * torch.randn(K, device=DEV, generator=g)).to(dtype)
    bet = (0.1 * torch.randn(K, device=DEV, generator=g)).to(dtype)
    w = (torch.randn(N, K, device=DEV, generator=g) / K ** 0.5).to(dtype)
    b = torch.randn(N, device=DEV, generator=g).to(dtype)
    with torch.no_grad():
        assert hip_ops.linear_ln_supported(x, gam, w)
        before = _cabi.CALLS["layernorm"]
        y = hip_ops.linear_ln(x, gam, bet, 1e-5, w, b, act=act)
        assert _cabi.CALLS["layernorm"] == before, "the LayerNorm kernel ran: not fused"
        two = hip_ops.linear(hip_ops.layer_norm(x, gam, bet, 1e-5), w, b, act=act)
        h = F.layer_norm(x.float(), (K,), gam.float(), bet.float(), 1e-5).to(dtype).float()
        exact = h @ w.float().t() + b.float()
        exact = F.gelu(exact) if act == "gelu" else (F.relu(exact) if act == "relu" else exact)
    tol = (2e-3 if dtype == torch.float16 else 1.6e-2) * max(exact.abs().max().item(), 1.0)
    assert (y.float() - exact).abs().max().item() < tol
    assert (y.float() - two.float()).abs().max().item() < tol
    assert (y != two).float().mean().item() < 0.05      # the vast majority of elements are bit-identical
    # shapes outside the short-K kernel: LayerNorm + linear
    with torch.no_grad():
        assert not hip_ops.linear_ln_supported(x[:500], gam, w)
        # (the same two launches on the same rows -- `two` above may come from another GEMM kernel at its row count)
        assert torch.equal(hip_ops.linear_ln(x[:500], gam, bet, 1e-5, w, b, act=act),
                           hip_ops.linear(hip_ops.layer_norm(x[:500], gam, bet, 1e-5), w, b, act=act))
