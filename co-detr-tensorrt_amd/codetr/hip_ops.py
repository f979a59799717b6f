"""Functional compute layer of the host package: every tensor op the model modules need goes
through one of these functions, so that "which kernel runs" is decided in exactly one place.

GPU-only by design.  MSDA always runs the hand-written HIP kernel (C ABI).  The dense ops
(linear / layer-norm / window attention / ...) run the hand-written kernels of libcodetr_hip.so
where they exist (``NATIVE`` lists them) and PyTorch-ROCm library kernels otherwise -- PyTorch
here is plumbing (rocBLAS / hipBLASLt / MIOpen behind ATen), used until the native kernel for
that op lands, never as a fallback for a native kernel that exists.

There is no CPU path: a CPU tensor reaching any of these functions is an error (the CPU
formulation of the model lives in oracle/ and is test infrastructure).
"""

import torch
import torch.nn.functional as F

from . import _cabi

# ops served by hand-written HIP kernels in this build (kept in sync with include/codetr_hip.h)
NATIVE = {"msda", "linear(f16/bf16, K%64==0)", "layer_norm(f16/bf16)", "swin_window_attention(f16, head_dim 32)",
          "msda_fused(softmax + sampling locations in-kernel)", "groupnorm_tokens(f16, 8 ch/group)",
          "sine_pos_tokens(f16, + level_embed)", "ffn_fused(f16, 256 -> hidden -> 256, ReLU, + identity)",
          "mask_pyramid(level masks + running valid counts + valid ratios)",
          "query_sine_embed(f16: sigmoid x valid ratios + sine embedding of the decoder reference boxes)",
          "encoder_geometry(f16: reference points, proposals, keep/drop state)", "row_max(f16)",
          "preprocess_image(u8 -> f16/f32, cv2-exact resize + pad + normalise + mask)", "batched_nms(f32)",
          "patch_merge_layernorm(f16: Swin 2x2 gather + LayerNorm)",
          "msda_encoder_packed(f16/bf16: LDS-staged gather for the encoder's self-attention, csrc/msda_encoder4.hip)",
          "patch_embed(f16/bf16: 4x4 patch gather + GEMM)",
          "mha_attention(f16/bf16: dense softmax attention, head_dim 32, <= 1024 keys)",
          "topk(f16/bf16 rows, k <= 1024: radix select + bitonic sort)",
          "im2col_tokens(16-bit token-major maps)",
          "small_ops(f16: add, sigmoid, gather_rows, decode_boxes, valid_ratios)",
          "linear_fp8(e4m3 x e4m3, K%128==0) + layer_norm_fp8 + cast_fp8", "ffn_fp8(fused FFN, both products e4m3)"}


# bench.py sets this to a list to time every native linear launch with HIP events on its launch stream
# (entries: (start_event, end_event, flops, M, N, K)); None in normal operation
LINEAR_PROFILE = None
# same hook for the other two heavy kernels: dict name -> list of (start_event, end_event, meta dict); None normally
KERNEL_PROFILE = None


def _timed(name, meta, launch, device):
    """run `launch()`; when bench.py has armed KERNEL_PROFILE, bracket it with HIP events on the launch stream"""
    if KERNEL_PROFILE is None:
        return launch()
    st = torch.cuda.current_stream(device)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    r = launch()
    e1.record(st)
    KERNEL_PROFILE.setdefault(name, []).append((e0, e1, meta))
    return r


def _gpu(x, what):
    if not x.is_cuda:
        raise RuntimeError(
            f"codetr.hip_ops.{what}: got a {x.device.type} tensor; this package computes on MI355X only "
            "(the CPU formulation is the oracle in oracle/, not a fallback)"
        )


LINEAR_PP = True     # route switch (A/B): False keeps the long-K layers on the round-4 kernels


def _persistent_ok(mk, head_major, out, x2, w, r2, bias):
    """what both persistent GEMMs ask of their operands: no row mask / head-major output, 16-byte aligned bases"""
    return (mk is None and not head_major and out.data_ptr() % 16 == 0 and x2.data_ptr() % 16 == 0
            and w.data_ptr() % 16 == 0 and (r2 is None or r2.data_ptr() % 16 == 0)
            and (bias is None or bias.data_ptr() % 16 == 0))


def linear(x, weight, bias=None, act=None, residual=None, row_mask=None, head_major=None):
    """y = act(x @ weight.T + bias) (+ residual);  act in {None, 'relu', 'gelu'}.
    row_mask (bool, x.shape[:-1]): rows where it is True come out as zeros (before the residual).
    head_major = head_dim: x must be [B, S, K]; the result is returned as [B, N/head_dim, S, head_dim] (each head's
    map contiguous) instead of [B, S, N] -- the value-map layout of the head-major MSDA kernel (native path only)."""
    _gpu(x, "linear")
    if _cabi.linear_supported(x, weight):
        # hand-written MFMA GEMM with the bias / activation / residual folded into its epilogue
        K = x.shape[-1]
        N = weight.shape[0]
        x2 = x.reshape(-1, K)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        w = weight if weight.is_contiguous() else weight.contiguous()
        r2 = None
        if residual is not None:
            r2 = residual.reshape(-1, N)
            if not r2.is_contiguous():
                r2 = r2.contiguous()
        mk = None
        if row_mask is not None:
            mk = row_mask.reshape(-1)
            if mk.dtype != torch.bool and mk.dtype != torch.uint8:
                mk = mk != 0
            mk = mk.contiguous()
        hm_rows = hm_hd = 0
        if head_major:
            if x.dim() != 3 or residual is not None:
                raise ValueError("head_major needs a [B, S, K] input and no residual")
            hm_rows, hm_hd = x.shape[1], int(head_major)
        out = torch.empty((x2.shape[0], N), dtype=x.dtype, device=x.device)
        if x2.shape[0] > 0:
            with torch.cuda.device(x.device):
                splits, ws_bytes = (1, 0) if head_major else _cabi.linear_splitk_plan(x2.shape[0], N, K)
                if splits > 1:
                    # few output tiles, long K (the neck's extra level as a GEMM): two-pass split-K, fp32 partials
                    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device)
                    launch = lambda: _cabi.linear_splitk(x2, w, bias, r2, act, out, splits, ws, mk)  # noqa: E731
                elif (LINEAR_PP and _persistent_ok(mk, head_major, out, x2, w, r2, bias)
                      and _cabi.linear_pp_preferred(x2.shape[0], N, K, act, r2 is not None)):
                    # the long-K layers with a tile per CU (Swin stages 2-3, stage 1's fc2): ping-pong persistent GEMM
                    launch = lambda: _cabi.linear_pp(x2, w, bias, r2, act, out)  # noqa: E731
                elif (_persistent_ok(mk, head_major, out, x2, w, r2, bias)
                      and _cabi.linear_sk_preferred(x2.shape[0], N, K, act, r2 is not None)):
                    # the large short-K layers (Swin stages 0-2, the encoder's output projections): persistent GEMM
                    launch = lambda: _cabi.linear_sk(x2, w, bias, r2, act, out)  # noqa: E731
                else:
                    launch = lambda: _cabi.linear(x2, w, bias, r2, act, out, mk, hm_rows, hm_hd)  # noqa: E731
                if LINEAR_PROFILE is None:
                    launch()
                else:
                    st = torch.cuda.current_stream(x.device)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(st)
                    launch()
                    e1.record(st)
                    LINEAR_PROFILE.append((e0, e1, 2.0 * x2.shape[0] * N * K, x2.shape[0], N, K))
        if head_major:
            return out.view(x.shape[0], N // hm_hd, hm_rows, hm_hd)
        return out.view(*x.shape[:-1], N)
    if head_major:
        raise RuntimeError("head_major output exists only on the native linear (f16/bf16, K % 64 == 0)")
    # fp32 / odd-K layers (patch-embed is a conv; fp32 runs are parity runs): ATen library GEMM
    y = F.linear(x, weight, bias)
    if act == "relu":
        y = F.relu(y, inplace=True)
    elif act == "gelu":
        y = F.gelu(y)
    elif act is not None:
        raise ValueError(act)
    if row_mask is not None:
        y = y.masked_fill(row_mask[..., None], 0.0)
    if residual is not None:
        y = y + residual
    return y


LN_GEMM = True   # route switch (tools/ab_host_routes.py patches it): False = LayerNorm as its own kernel in front of the GEMM


def linear_ln_supported(x, norm_weight, weight):
    """True when LayerNorm(x) @ weight.T runs as ONE kernel (the norm applied in the short-K GEMM's operand load)"""
    return (LN_GEMM and x.is_cuda and norm_weight is not None and norm_weight.dtype == x.dtype and weight.dtype == x.dtype
            and not torch.is_grad_enabled() and norm_weight.shape[0] == x.shape[-1]
            and _cabi.linear_xadd_supported(x.numel() // max(x.shape[-1], 1), weight.shape[0], x.shape[-1], x.dtype))


def linear_ln(x, norm_weight, norm_bias, eps, weight, bias=None, act=None):
    """act(LayerNorm(x) @ weight.T + bias): Swin's norm1 -> qkv and norm2 -> fc1 (reference swin.py:345-386), where only
    the GEMM reads the normalised rows.  Falls back to layer_norm + linear where the fused form does not apply."""
    _gpu(x, "linear_ln")
    if linear_ln_supported(x, norm_weight, weight):
        K, N = x.shape[-1], weight.shape[0]
        x2 = x.reshape(-1, K)
        x2 = x2 if x2.is_contiguous() else x2.contiguous()
        w = weight if weight.is_contiguous() else weight.contiguous()
        out = torch.empty((x2.shape[0], N), dtype=x.dtype, device=x.device)
        ok = [False]

        def launch():
            ok[0] = _cabi.linear_ln(x2, norm_weight, norm_bias, eps, w, bias, act, out)

        with torch.cuda.device(x.device):
            if LINEAR_PROFILE is None:
                launch()
            else:
                st = torch.cuda.current_stream(x.device)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(st)
                launch()
                e1.record(st)
                if ok[0]:
                    LINEAR_PROFILE.append((e0, e1, 2.0 * x2.shape[0] * N * K, x2.shape[0], N, K))
        if ok[0]:
            return out.view(*x.shape[:-1], N)
    return linear(layer_norm(x, norm_weight, norm_bias, eps), weight, bias, act=act)


XADD = True   # route switch: False = `query + query_pos` as its own kernel / FFN output
# Rows from which `query + query_pos` is added inside the X-stationary kernel's operand load.  Until the encoder's two
# projections became one launch (encoder_projections) the 256-tile GEMM on a stored sum won below 400 000 rows (one
# 1920x1280 image: +0.09 ms per forward with the add folded in); with the one-launch form the fold wins at every size the
# kernel takes (one 1920x1280 image 13.14 -> 13.07 ms, one 1152x768 image 7.18 -> 7.15 ms), so the threshold is 0.
XADD_MIN_ROWS = 0


def linear_xadd_supported(x, x_add, weight):
    """True when (x + x_add) @ weight.T runs as ONE kernel (the add folded into the X-stationary GEMM's operand load)"""
    return (XADD and x.is_cuda and x_add is not None and x_add.shape == x.shape and x_add.dtype == x.dtype
            and weight.dtype == x.dtype and not torch.is_grad_enabled()
            and x.numel() // max(x.shape[-1], 1) >= XADD_MIN_ROWS
            and _cabi.linear_xadd_supported(x.numel() // max(x.shape[-1], 1), weight.shape[0], x.shape[-1], x.dtype))


def linear_xadd(x, x_add, weight, bias=None):
    """(x + x_add) @ weight.T + bias -- `query + query_pos` in front of the (offsets | logits) projection (reference
    multi_scale_deformable_attention.py:161-162, 177-179); the sum is rounded to the storage type first, exactly as
    the separate add.  Falls back to add + linear where the fused form does not apply."""
    _gpu(x, "linear_xadd")
    if linear_xadd_supported(x, x_add, weight):
        K, N = x.shape[-1], weight.shape[0]
        x2, a2 = x.reshape(-1, K), x_add.reshape(-1, K)
        x2 = x2 if x2.is_contiguous() else x2.contiguous()
        a2 = a2 if a2.is_contiguous() else a2.contiguous()
        w = weight if weight.is_contiguous() else weight.contiguous()
        out = torch.empty((x2.shape[0], N), dtype=x.dtype, device=x.device)
        ok = [False]

        def launch():
            ok[0] = _cabi.linear_xadd(x2, a2, w, bias, out)

        with torch.cuda.device(x.device):
            if LINEAR_PROFILE is None:
                launch()
            else:
                st = torch.cuda.current_stream(x.device)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(st)
                launch()
                e1.record(st)
                if ok[0]:
                    LINEAR_PROFILE.append((e0, e1, 2.0 * x2.shape[0] * N * K, x2.shape[0], N, K))
        if ok[0]:
            return out.view(*x.shape[:-1], N)
    return linear(x + x_add, weight, bias)


# One 128-row block of the fused FFN kernel runs ~100 us (it walks the whole hidden dimension alone), and the chip
# holds 256 of them: below ~24k rows the grid is under one wave and the two plain GEMMs (which split N over blocks)
# finish sooner -- measured 98.6 us fused vs 27 us as two GEMMs at the decoder's 900 rows.
FFN_FUSED_MIN_ROWS = 24576


def ffn_fused_supported(x, w1, w2, act):
    return (x.is_cuda and x.numel() // max(x.shape[-1], 1) >= FFN_FUSED_MIN_ROWS
            and _cabi.ffn_fused_supported(x, w1, w2, act))


def derived(params, name, build):
    """A tensor derived from parameter tensors (fused / permuted / gathered weights), built once by `build()` and kept
    ON the first parameter object under attribute `name` -- so it dies with that parameter instead of sitting in a table
    keyed by an address the allocator can hand to a different tensor later.  Rebuilt when any source changed: object
    identity, in-place version counter, storage address (``param.data = ...``, ``.to()``), dtype or device."""
    first = params[0]
    key = tuple((id(p), p.data_ptr(), p._version, p.dtype, p.device) for p in params)
    hit = getattr(first, name, None)
    if hit is not None and hit[0] == key:
        return hit[1]
    with torch.no_grad():
        val = build()
    setattr(first, name, (key, val))
    return val


def _packed_w2(w2):
    """the kernel's pre-packed copy of a second-Linear weight.  The copy hangs on the weight tensor object itself
    (so it dies with it: a table keyed by id() / data_ptr() can hand a NEW tensor that reuses a freed tensor's address
    and id the old tensor's packed weights) and is rebuilt when the tensor is modified in place."""
    hit = getattr(w2, "_codetr_packed_w2", None)
    if hit is None or hit[0] != w2._version or hit[1].device != w2.device or hit[1].dtype != w2.dtype:
        with torch.no_grad(), torch.cuda.device(w2.device):
            hit = (w2._version, _cabi.ffn_pack_w2(w2.detach().contiguous()))
        w2._codetr_packed_w2 = hit
    return hit[1]


SWIN_MLP = True            # route switch (A/B): False = norm2 | fc1 + GELU | fc2 + identity as separate launches
SWIN_MLP_MIN_ROWS = 32768  # below this a stage's MLP is a few dozen tiles: the separate GEMMs fill the chip better


def swin_mlp_supported(x, ln_weight, w1, w2, act):
    """the fused Swin MLP (csrc/swin_mlp.hip) serves x [.., C] with C in {192, 384}, hidden = 4 C, GELU, 16-bit storage"""
    C = x.shape[-1]
    return (SWIN_MLP and x.is_cuda and x.dtype in (torch.float16, torch.bfloat16) and act == "gelu" and C in (192, 384)
            and w1.shape == (4 * C, C) and w2.shape == (C, 4 * C) and ln_weight is not None
            and x.numel() // C >= SWIN_MLP_MIN_ROWS and w1.dtype == x.dtype and w2.dtype == x.dtype)


def swin_mlp(x, ln_weight, ln_bias, eps, w1, b1, w2, b2):
    """y = x + fc2(gelu(fc1(layer_norm(x)))): the second half of a SwinBlock in one launch (reference swin.py:331-352)"""
    _gpu(x, "swin_mlp")
    C = x.shape[-1]
    x2 = x.reshape(-1, C)
    if not x2.is_contiguous():
        x2 = x2.contiguous()
    out = torch.empty_like(x2)
    w1c = w1 if w1.is_contiguous() else w1.contiguous()
    with torch.cuda.device(x.device):
        w2p = _packed_w2(w2)
        _timed("swin_mlp", {"M": x2.shape[0], "C": C},
               lambda: _cabi.swin_mlp(x2, ln_weight, ln_bias, eps, w1c, b1, w2p, b2, out), x.device)
    return out.view(x.shape)


def ffn_fused(x, w1, b1, w2, b2, ln=None, pos=None, ln_in=None):
    """y = x + relu(x @ w1.T + b1) @ w2.T + b2 in one kernel (hidden activation stays on-chip); x [..., 256].
    w2 is the plain nn.Linear weight; its packed form is cached.
    ln = (weight, bias, eps): y = LayerNorm(y) in the same epilogue.  pos (same shape as x): also returns y + pos,
    i.e. the result is (y, y + pos).  ln_in = (weight, bias, eps): the input rows are LayerNorm'ed first (in
    registers; that norm's output is both the FFN input and the identity and is never written)."""
    _gpu(x, "ffn_fused")
    x2 = x.reshape(-1, x.shape[-1])
    if not x2.is_contiguous():
        x2 = x2.contiguous()
    out = torch.empty_like(x2)
    p2 = out2 = None
    if pos is not None:
        p2 = pos.reshape(-1, pos.shape[-1])
        if p2.shape != x2.shape or p2.dtype != x2.dtype:
            raise ValueError("pos must match x in shape and dtype")
        if not p2.is_contiguous():
            p2 = p2.contiguous()
        out2 = torch.empty_like(x2)
    if x2.shape[0] > 0:
        with torch.cuda.device(x.device):
            _timed("ffn_fused", {"M": x2.shape[0], "C": x2.shape[1], "hidden": w1.shape[0]},
                   lambda: _cabi.ffn_fused(x2, w1.contiguous(), b1, _packed_w2(w2), b2, out, ln, p2, out2, ln_in), x.device)
    if pos is not None:
        return out.view(x.shape), out2.view(x.shape)
    return out.view(x.shape)


FFN_OPROJ = True   # route switch: the encoder layer's attention output projection inside the fused FFN launch


def ffn_oproj_fused(attn, wo, bo, identity, w1, b1, w2, b2, ln, pos=None, ln_in=None):
    """The post-norm encoder layer from the attention output on, ONE launch (include/codetr_hip.h
    codetr_ffn_oproj_relu_ln2_*):  x1 = LN_in(identity + (attn @ wo.T + bo));  y = LN(x1 + relu(x1 @ w1.T + b1) @ w2.T + b2);
    with `pos` also y + pos (returns (y, y + pos)).  w1 / w2 are the plain nn.Linear weights (their kernel layouts are
    cached on the tensors)."""
    _gpu(attn, "ffn_oproj_fused")
    C = attn.shape[-1]
    a2, i2 = attn.reshape(-1, C), identity.reshape(-1, C)
    a2 = a2 if a2.is_contiguous() else a2.contiguous()
    i2 = i2 if i2.is_contiguous() else i2.contiguous()
    if i2.shape != a2.shape or i2.dtype != a2.dtype:
        raise ValueError("identity must match the attention output in shape and dtype")
    out = torch.empty_like(a2)
    p2 = out2 = None
    if pos is not None:
        p2 = pos.reshape(-1, C)
        if p2.shape != a2.shape or p2.dtype != a2.dtype:
            raise ValueError("pos must match x in shape and dtype")
        p2 = p2 if p2.is_contiguous() else p2.contiguous()
        out2 = torch.empty_like(a2)
    w1p = derived((w1,), "_codetr_w1_oproj", lambda: w1.detach()[:, torch.tensor(
        _cabi.ffn_oproj_w1_index(C), dtype=torch.long, device=w1.device)].contiguous())
    woc = wo if wo.is_contiguous() else wo.contiguous()
    if a2.shape[0] > 0:
        with torch.cuda.device(attn.device):
            _timed("ffn_fused", {"M": a2.shape[0], "C": C, "hidden": w1.shape[0], "oproj": True},
                   lambda: _cabi.ffn_oproj_fused(a2, woc, bo, i2, w1p, b1, _packed_w2(w2), b2, out, ln_in, ln, p2, out2),
                   attn.device)
    if pos is not None:
        return out.view(attn.shape), out2.view(attn.shape)
    return out.view(attn.shape)


def layer_norm(x, weight, bias, eps=1e-5):
    _gpu(x, "layer_norm")
    if _cabi.layernorm_supported(x, weight):
        C = x.shape[-1]
        x2 = x.reshape(-1, C)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        out = torch.empty_like(x2)
        if x2.shape[0] > 0:
            with torch.cuda.device(x.device):
                _cabi.layernorm(x2, weight, bias, eps, out)
        return out.view(x.shape)
    return F.layer_norm(x, (x.shape[-1],), weight, bias, eps)  # fp32 parity runs


MERGE_LN = True   # route switch: False = separate patch-merge gather + LayerNorm


def patch_merge_layernorm_supported(x, C):
    return (x.is_cuda and x.dtype in (torch.float16, torch.bfloat16) and C % 8 == 0 and 4 * C <= 4096
            and MERGE_LN)


def patch_merge_layernorm(x, hw, weight_kkc, bias_kkc, eps=1e-5):
    """Swin PatchMerging's 2x2 gather + LayerNorm(4C) in one pass: x [B, H*W, C] tokens -> [B, H2*W2, 4C] with the 4C
    axis ordered (ky, kx, c) (weight / bias given in that order)."""
    _gpu(x, "patch_merge_layernorm")
    B, L, C = x.shape
    H, W = hw
    x4 = x.view(B, H, W, C)
    if not x4.is_contiguous():
        x4 = x4.contiguous()
    with torch.cuda.device(x.device):
        return _cabi.patch_merge_layernorm(x4, weight_kkc, bias_kkc, eps)


def group_norm(x, groups, weight, bias, eps=1e-5):
    _gpu(x, "group_norm")
    return F.group_norm(x, groups, weight, bias, eps)


def groupnorm_tokens_supported(x, groups):
    return x.is_cuda and _cabi.groupnorm_tokens_supported(x, groups)


def groupnorm_tokens_into(x, gamma, beta, groups, eps, dest, row_start):
    """GroupNorm of token-major x [B,HW,C] written into dest[:, row_start:row_start+HW, :] (dest [B,S,C] contiguous)."""
    _gpu(x, "groupnorm_tokens_into")
    B, HW, C = x.shape
    if not x.is_contiguous():
        x = x.contiguous()
    if not dest.is_contiguous() or dest.shape[0] != B or dest.shape[2] != C:
        raise AssertionError("destination must be a contiguous [B, S, C] tensor")
    with torch.cuda.device(x.device):
        _cabi.groupnorm_tokens(x, gamma, beta, groups, eps, dest[0, row_start:], dest.shape[1] * C)
    return dest


def query_sine_embed_supported(ref, valid_ratios, pos_feat):
    return (ref.is_cuda and ref.dtype in (torch.float16, torch.bfloat16) and valid_ratios.dtype == ref.dtype
            and ref.shape[-1] in (2, 4) and pos_feat % 8 == 0 and valid_ratios.shape[1] <= ref.shape[-1] * pos_feat // 8)


def query_sine_embed(ref, valid_ratios, pos_feat, temperature=10000.0):
    """Decoder layer head in one launch: (sigmoid(ref)[:, :, None] * valid_ratios (tiled to ref_dim)[:, None],
    gen_sineembed_for_position of its level-0 row) -- see include/codetr_hip.h.  When `valid_ratios` carries its fp32
    twin (``_codetr_f32``, hip_ops.valid_ratios), the returned ref_in carries ``_codetr_ref32``: the same reference points
    with sigmoid and scaling kept in fp32, which msda_fused then samples at."""
    _gpu(ref, "query_sine_embed")
    vr32 = getattr(valid_ratios, "_codetr_f32", None) if MSDA_FP32_REF else None
    with torch.cuda.device(ref.device):
        ref_in, embed, ref32 = _cabi.query_sine_embed(ref.contiguous(), valid_ratios.contiguous(), pos_feat, temperature,
                                                      valid_ratios32=vr32)
    if ref32 is not None:
        ref_in._codetr_ref32 = ref32
    return ref_in, embed


def encoder_geometry(valid_ratios, mask_flat, shapes):
    """Reference points, their per-level scaling, two-stage proposals (already masked) and the keep / drop state of
    every encoder token in one launch (csrc/encoder_geometry.hip); f16 valid_ratios [B,L,2], mask_flat [B,S] bool."""
    _gpu(mask_flat, "encoder_geometry")
    if valid_ratios.dtype not in (torch.float16, torch.bfloat16):
        raise RuntimeError("encoder_geometry is the 16-bit inference path; fp32 parity runs use the ATen formulation")
    with torch.cuda.device(mask_flat.device):
        return _cabi.encoder_geometry(valid_ratios.contiguous(), mask_flat.contiguous(),
                                      [tuple(int(v) for v in s) for s in shapes])


def row_max(x):
    """x.max(-1)[0] for a dense f16 tensor (NaN propagates)"""
    _gpu(x, "row_max")
    if x.dtype not in (torch.float16, torch.bfloat16) or not x.is_contiguous():
        return x.max(-1)[0]
    with torch.cuda.device(x.device):
        return _cabi.row_max(x.view(-1, x.shape[-1])).view(x.shape[:-1])


def preprocess_image(src_u8, resized_hw, pad_hw, mean, std, pad_value=(0, 0, 0), dtype=torch.float16):
    """uint8 HWC RGB image on the GPU -> (normalised [3, Hp, Wp], mask [Hp, Wp]): cv2-exact bilinear resize to
    `resized_hw`, right / bottom padding with `pad_value` to `pad_hw`, (x - mean) / std (csrc/prepost.hip)."""
    _gpu(src_u8, "preprocess_image")
    if src_u8.dtype != torch.uint8 or src_u8.dim() != 3 or src_u8.shape[2] != 3:
        raise ValueError("expected a uint8 [H, W, 3] image")
    with torch.cuda.device(src_u8.device):
        return _cabi.preprocess_u8(src_u8.contiguous(), resized_hw, pad_hw, mean, std, pad_value, dtype)


def batched_nms(boxes, scores, labels, iou_threshold):
    """torchvision.ops.batched_nms semantics (per-class greedy hard NMS): indices of the kept boxes in descending
    score order.  IoU arithmetic in fp32 on the given boxes."""
    _gpu(boxes, "batched_nms")
    if boxes.shape[0] == 0:
        return torch.empty((0,), dtype=torch.int64, device=boxes.device)
    order = torch.sort(scores.float(), descending=True, stable=True)[1]
    with torch.cuda.device(boxes.device):
        keep = _cabi.batched_nms_sorted(boxes.float()[order].contiguous(), labels.to(torch.int64)[order].contiguous(),
                                        iou_threshold)
    return order[keep]


def mask_pyramid(img_masks, shapes):
    """img_masks [B,H,W] (float 0/1, bool or uint8; non-zero = padding) + level shapes [(H_l, W_l)] ->
    (mask_flat [B,S] bool, ycum, xcum, valid_counts [B,L,2] fp32): the level masks (nearest resize), their running
    valid counts and first-row / first-column valid counts, one launch (csrc/mask_pyramid.hip).  ycum / xcum are flat
    fp32 buffers; `level_cums` slices out one level as [B,H_l,W_l]."""
    _gpu(img_masks, "mask_pyramid")
    m = img_masks
    if m.dtype not in (torch.bool, torch.uint8, torch.float16, torch.bfloat16, torch.float32):
        m = m != 0     # (other dtypes: one comparison kernel; the kernel tests the three element widths itself)
    with torch.cuda.device(m.device):
        return _cabi.mask_pyramid(m.contiguous(), [tuple(int(v) for v in s) for s in shapes])


def level_cums(ycum, xcum, B, start, hw):
    """level views [B,H_l,W_l] of mask_pyramid's running-count buffers (start = rows of the levels before)"""
    n = hw[0] * hw[1]
    return (ycum[B * start:B * (start + n)].view(B, hw[0], hw[1]), xcum[B * start:B * (start + n)].view(B, hw[0], hw[1]))


def sine_pos_tokens_into(mask, dest, row_start, level_embed, num_feats, temperature, scale, eps, offset, normalize,
                         cums=None):
    """Sine positional encoding of one level (mask [B,H,W] bool, True = padding) + level_embed, written into
    dest[:, row_start:row_start+H*W, :] (dest [B,S,2*num_feats] f16 contiguous).  cums = (ycum, xcum) [B,H,W] fp32
    from mask_pyramid skips the two cumsum calls (mask may then be None)."""
    if cums is None:
        _gpu(mask, "sine_pos_tokens_into")
        valid = ~mask
        ycum = valid.cumsum(1, dtype=torch.float32).contiguous()
        xcum = valid.cumsum(2, dtype=torch.float32).contiguous()
    else:
        ycum, xcum = cums
        _gpu(ycum, "sine_pos_tokens_into")
    if dest.dtype not in (torch.float16, torch.bfloat16) or not dest.is_contiguous() or dest.shape[2] != 2 * num_feats:
        raise AssertionError("destination must be a contiguous f16 / bf16 [B, S, 2*num_feats] tensor")
    if level_embed is not None and level_embed.dtype != dest.dtype:
        level_embed = level_embed.to(dest.dtype)
    with torch.cuda.device(ycum.device):
        _cabi.sine_pos_tokens(ycum, xcum, level_embed, dest[0, row_start:], dest.shape[1] * dest.shape[2], num_feats,
                              temperature, scale, eps, offset, normalize)
    return dest


def conv2d(x, weight, bias=None, stride=1, padding=0):
    _gpu(x, "conv2d")
    return F.conv2d(x, weight, bias, stride=stride, padding=padding)


def window_attention(qkv, rel_bias, mask, num_heads):
    """qkv [nWB, N, 3*C] (q|k|v, each head-major) ; rel_bias [nH, N, N]; mask [nW, N, N] or None.
    Returns [nWB, N, C].  softmax((q*scale) k^T + bias (+mask)) v per window and head
    (reference codetr/swin.py:92-112)."""
    _gpu(qkv, "window_attention")
    nWB, N, C3 = qkv.shape
    C = C3 // 3
    hd = C // num_heads
    qkv = qkv.view(nWB, N, 3, num_heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    attn = (q * hd ** -0.5) @ k.transpose(-2, -1)
    attn = attn + rel_bias.unsqueeze(0)
    if mask is not None:
        nW = mask.shape[0]
        attn = (attn.view(nWB // nW, nW, num_heads, N, N) + mask[None, :, None]).view(nWB, num_heads, N, N)
    attn = attn.softmax(-1)
    return (attn @ v).transpose(1, 2).reshape(nWB, N, C)


def swin_window_attention_supported(x, embed_dims, num_heads, window_size):
    """True when the fused kernel serves this block (f16, head_dim 32, window in {4,7,8,12})."""
    return x.is_cuda and _cabi.window_attention_supported(x.dtype, embed_dims, num_heads, window_size)


WINDOW_BIAS_LANE = True   # route switch (A/B): False = the kernel reads the [nH, N, N] table as the reference lays it out
_LANE_INDEX = {}


def _rel_bias_lane_order(rel_bias, window_size):
    """rel_bias [nH, N, N] in the lane order of the window-attention kernel's score tiles (bias_layout 1 of
    codetr_window_attention_ex; same values, rows permuted along the key axis), cached on the table tensor; None where the
    window size has no lane order (7 x 7)."""
    if window_size not in _LANE_INDEX:
        _LANE_INDEX[window_size] = _cabi.window_attention_bias_index(window_size)
    idx = _LANE_INDEX[window_size]
    if idx is None:
        return None
    return derived((rel_bias,), "_codetr_rel_bias_lane",
                   lambda: rel_bias[..., torch.tensor(idx, dtype=torch.long, device=rel_bias.device)].contiguous())


def swin_window_attention(qkv, qkv_bias, rel_bias, hw_shape, num_heads, window_size, shift, out_scale=None, out_mx=False):
    """Fused (shifted-)window attention on the UNPADDED spatial token map.
    qkv [B, H*W, 3C] (output of the qkv Linear on real tokens only), qkv_bias [3C] or None,
    rel_bias [nH, N, N] -> [B, H*W, C].  Pad / roll / partition / softmax / reverse are all inside
    the kernel (reference codetr/swin.py:191-252, 92-112).
    out_scale (fp16 qkv only): the result is e4m3 = sat(f16(o) / out_scale), the operand of the fp8 proj GEMM;
    out_mx (fp16 qkv only): the result is (e4m3, MX block scales) -- one block per (token, head)."""
    _gpu(qkv, "swin_window_attention")
    B, L, C3 = qkv.shape
    H, W = hw_shape
    if L != H * W:
        raise AssertionError("input feature has wrong size")
    if not qkv.is_contiguous():
        qkv = qkv.contiguous()
    if qkv_bias is None:
        qkv_bias = torch.zeros(C3, dtype=qkv.dtype, device=qkv.device)
    if (out_scale is not None or out_mx) and qkv.dtype != torch.float16:
        raise ValueError("the e4m3 output forms take fp16 qkv")
    out = torch.empty((B, L, C3 // 3), dtype=FP8 if (out_scale is not None or out_mx) else qkv.dtype, device=qkv.device)
    scales = torch.empty(_cabi.mx_scale_bytes(B * L, C3 // 3), dtype=torch.uint8, device=qkv.device) if out_mx else None
    rel_bias = rel_bias.contiguous()
    lane = _rel_bias_lane_order(rel_bias, window_size) if WINDOW_BIAS_LANE else None
    with torch.cuda.device(qkv.device):
        _timed("window_attention", {"rows": B * L, "C": C3 // 3, "heads": num_heads, "window": window_size,
                                    "out_bytes": out.element_size()},
               lambda: _cabi.window_attention(qkv, qkv_bias, rel_bias if lane is None else lane, out, B, H, W, num_heads,
                                              window_size, shift, out_scale, scales, 0 if lane is None else 1), qkv.device)
    return (out, scales) if out_mx else out


def mha_self_attention(q, k, v, num_heads):
    """q,k,v [B, N, C] already projected (q / k may be column slices of one fused projection) -> [B, N, C]: dense
    softmax attention, 900 x 900 per head in the decoder (reference transformer_mmcv.py:394-428).  Native kernel for
    head_dim 32 and up to 1024 keys; the SDPA library call otherwise (fp32 parity runs, other head sizes)."""
    _gpu(q, "mha_self_attention")
    B, N, C = q.shape
    if _cabi.mha_attention_supported(q, k, v, num_heads):
        out = torch.empty((B, N, C), dtype=q.dtype, device=q.device)
        with torch.cuda.device(q.device):
            _cabi.mha_attention(q, k, v, num_heads, out)
        return out
    hd = C // num_heads
    sp = lambda t: t.reshape(B, -1, num_heads, hd).transpose(1, 2)  # noqa: E731
    o = F.scaled_dot_product_attention(sp(q), sp(k), sp(v))
    return o.transpose(1, 2).reshape(B, N, C)


def msda_fused_supported(value_dtype, head_dim, num_levels, num_points):
    return _cabi.msda_fused_supported(value_dtype, head_dim, num_levels, num_points)


def msda_head_major_supported(value_dtype, head_dim, num_levels, num_points):
    return _cabi.msda_head_major_supported(value_dtype, head_dim, num_levels, num_points)


def msda_fused(value, spatial_shapes, level_start_index, proj, off_col, logit_col, reference_points, num_levels,
               num_points, head_major=False):
    """MSDA with softmax + sampling-location arithmetic inside the kernel (reference
    multi_scale_deformable_attention.py:180-196 + the op).  value [B,S,M,D] (or [B,M,S,D] with head_major);
    proj [B,Nq,cols] = output of the fused (offsets | logits) projection; reference_points [B,Nq,L,2|4]
    -> [B,Nq,M*D]."""
    _gpu(value, "msda_fused")
    if head_major:
        B, M, S, D = value.shape
    else:
        B, S, M, D = value.shape
    out = torch.empty((B, proj.shape[1], M * D), dtype=value.dtype, device=value.device)
    ref32 = getattr(reference_points, "_codetr_ref32", None) if MSDA_FP32_REF else None
    ref = ref32 if ref32 is not None else reference_points.to(value.dtype).contiguous()
    if out.numel():
        with torch.cuda.device(value.device):
            _timed("msda_fused", {"B": B, "S": S, "Nq": proj.shape[1], "M": M, "D": D, "L": num_levels, "P": num_points},
                   lambda: _cabi.msda_fused(value.contiguous(), spatial_shapes, level_start_index, proj.contiguous(),
                                            off_col, logit_col, ref, num_levels, num_points, out,
                                            head_major=head_major), value.device)
    return out


def patch_embed_supported(x, weight, stride):
    kh, kw = weight.shape[-2:]
    return (x.is_cuda and x.dtype in (torch.float16, torch.bfloat16) and weight.dtype == x.dtype
            and kh == kw == 4 and tuple(stride) == (4, 4) and x.shape[1] * 16 <= 64 and not torch.is_grad_enabled())


def patch_embed(x, weight, bias):
    """PatchEmbed's Conv2d(C, E, 4, stride 4) with bottom / right zero padding as patch gather + GEMM: x [B,C,H,W] ->
    ([B, Hp*Wp, E] token-major, (Hp, Wp)).  The weight padded to 64 columns is cached on the parameter."""
    _gpu(x, "patch_embed")
    B, C, H, W = x.shape
    E = weight.shape[0]
    Hp, Wp = -(-H // 4), -(-W // 4)
    cache = getattr(weight, "_codetr_w64", None)
    if cache is None or cache[0] != weight._version or cache[1].device != weight.device:
        w64 = torch.zeros((E, 64), dtype=weight.dtype, device=weight.device)
        w64[:, :C * 16] = weight.detach().reshape(E, C * 16)
        cache = weight._codetr_w64 = (weight._version, w64)
    cols = torch.empty((B * Hp * Wp, 64), dtype=x.dtype, device=x.device)
    with torch.cuda.device(x.device):
        _cabi.patch_im2col(x.contiguous(), 4, 64, cols)
    return linear(cols, cache[1], bias).view(B, Hp * Wp, E), (Hp, Wp)


def im2col_tokens(x4d, k, stride, pad):
    """k x k / stride / zero-pad patches of a token-major map x4d [B,H,W,C] -> [B, Ho*Wo, k*k*C], K ordered
    (ky, kx, c).  Native 16-byte gather for 16-bit maps with C % 8 == 0; strided slices + cat otherwise."""
    _gpu(x4d, "im2col_tokens")
    B, H, W, C = x4d.shape
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    if x4d.dtype in (torch.float16, torch.bfloat16) and C % 8 == 0:
        out = torch.empty((B, Ho * Wo, k * k * C), dtype=x4d.dtype, device=x4d.device)
        with torch.cuda.device(x4d.device):
            _cabi.im2col_tokens(x4d.contiguous(), k, stride, pad, out)
        return out
    xp = F.pad(x4d, (0, 0, pad, pad, pad, pad))
    return torch.cat([xp[:, ky:ky + stride * (Ho - 1) + 1:stride, kx:kx + stride * (Wo - 1) + 1:stride, :]
                      for ky in range(k) for kx in range(k)], dim=-1).reshape(B, Ho * Wo, -1)


# (The library routes these three ops can also take -- SDPA, the MIOpen stem convolution, torch.topk -- were measured
# against the native kernels in rounds 1-2 and are no longer selectable here: tools/ab_library_routes.py times them by
# patching this module from the outside.  What remains below them is the route for shapes / dtypes the kernels do not
# take: fp32 parity runs, other head sizes.)


def topk(x, k, want_values=True):
    """torch.topk(x, k, dim=-1) for the head's two selections (reference transformer.py:560-561, co_dino_head.py:183):
    (values, indices), sorted descending; ties by ascending index and NaN first on the native path (f16 / bf16 rows,
    k <= 1024), torch.topk otherwise (fp32 parity runs)."""
    _gpu(x, "topk")
    x2 = x.reshape(-1, x.shape[-1])
    if not torch.is_grad_enabled():
        x2 = x2 if x2.is_contiguous() else x2.contiguous()
        if _cabi.topk_supported(x2, k):
            idx = torch.empty((x2.shape[0], k), dtype=torch.int64, device=x.device)
            val = torch.empty((x2.shape[0], k), dtype=x.dtype, device=x.device) if want_values else None
            with torch.cuda.device(x.device):
                _cabi.topk(x2, k, val, idx)
            shape = (*x.shape[:-1], k)
            return (val.view(shape) if val is not None else None), idx.view(shape)
    v, i = torch.topk(x, k, dim=-1)
    return v, i


# Route switches of the encoder MSDA (plain module attributes; tools/ab_host_routes.py patches them for A/B runs -- the
# package reads no CODETR_* environment variable except checkpoint.py's CODETR_ALLOW_PICKLE):
MSDA_ENCODER = True     # False = general fused kernel in the encoder
MSDA_FP32_REF = True    # False = reference points read in the model dtype


_SWITCH_DEFAULTS = {"ENC_POSGEN": True, "WINDOW_BIAS_LANE": True, "LINEAR_PP": True, "SWIN_MLP": True, "SWIN_MLP_MIN_ROWS": 32768, "LN_GEMM": True, "XADD": True, "XADD_MIN_ROWS": 0, "MERGE_LN": True, "MSDA_ENCODER": True,
                    "MSDA_FP32_REF": True, "FP8_MIN_TILES": 96}


def nondefault_switches():
    """Names of the host package's route switches that are not at their default (the measured-best path): the plan
    exporter refuses to record under any of them, tools/ab_host_routes.py sets them."""
    import sys

    from . import multi_scale_deformable_attention as msda_mod
    from . import transformer as tr_mod

    me = sys.modules[__name__]
    out = [k for k, v in _SWITCH_DEFAULTS.items() if getattr(me, k) != v]
    if msda_mod.HEAD_MAJOR_VALUE:
        out.append("HEAD_MAJOR_VALUE")
    out += [k for k in ("DEC_FUSED", "DEC_VPROJ") if not getattr(tr_mod, k)]
    return sorted(out)


# ---- round-5 encoder kernel (csrc/msda_encoder4.hip): lane-major packed projection, scalar geometry, zero border ----
MSDA_V4 = True                 # route switch: False = the general fused kernel on the unpacked projection
# Measured on MI355X at 4 x 1920x1280, offsets with 0 / 2 / 4 / 8 px of spread (profiles/r05_msda_encoder4_sweep.txt):
# 512 threads x 16x16 regions x 64 KiB = 918 / 973 / 1210 / 1637 us per launch; 256 x 16x8 x 40 KiB = 926 / 1019 / 1332 /
# 1832; 80 KiB windows trade 3 % at 2 px for 5 % at 8 px (round-4 kernel: 1220 / 1318 / 1810 / 2566).
MSDA_V4_THREADS = 512          # workgroup size (256: four workgroups per CU; 512: two)
MSDA_V4_REGION = (16, 16)      # region of a workgroup, pixels of the finest level
MSDA_V4_LDS_BUDGET = 64 * 1024   # bytes per workgroup
MSDA_V4_MARGIN_CAP = 40.0      # pixels: windows grow up to this margin around a head's bias points within the LDS budget
MSDA_V4_HEAD_MAJOR = True      # value projection writes [B, M, S, 32] for the packed encoder kernel
ENC_PROJ_FUSED = True          # value projection and packed (offsets | logits) projection as ONE launch (x, pos read once)
_SWITCH_DEFAULTS.update({"MSDA_V4": True, "MSDA_V4_THREADS": 512, "MSDA_V4_REGION": (16, 16),
                         "MSDA_V4_LDS_BUDGET": 64 * 1024, "MSDA_V4_MARGIN_CAP": 40.0,
                         "MSDA_V4_HEAD_MAJOR": True, "ENC_PROJ_FUSED": True, "FFN_OPROJ": True})


def msda_encoder_packed_supported(dtype, head_dim, num_levels, num_points):
    return (MSDA_V4 and MSDA_ENCODER and dtype in (torch.float16, torch.bfloat16) and head_dim == 32 and num_levels == 5
            and num_points == 4)


def value_projection_f16(x, weight, bias, row_mask, head_dim):
    """bf16 model: the encoder MSDA's value projection with an FP16, head-major result [B, M, S, head_dim] (the packed
    kernel's value map is fp16 whatever the model's type).  None when the library has no kernel for the shape."""
    _gpu(x, "value_projection_f16")
    if not (x.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16 and x.dim() == 3):
        return None
    B, S, K = x.shape
    N = weight.shape[0]
    x2 = x.reshape(-1, K)
    x2 = x2 if x2.is_contiguous() else x2.contiguous()
    w = weight if weight.is_contiguous() else weight.contiguous()
    mk = None
    if row_mask is not None:
        mk = row_mask.reshape(-1)
        if mk.dtype != torch.bool and mk.dtype != torch.uint8:
            mk = mk != 0
        mk = mk.contiguous()
    out = torch.empty((B * S, N), dtype=torch.float16, device=x.device)
    with torch.cuda.device(x.device):
        ok = _cabi.linear_bf16_f16out(x2, w, bias, out, mk, S, int(head_dim))
    return out.view(B, N // head_dim, S, head_dim) if ok else None


ENC_POSGEN = True   # route switch (A/B): False = the encoder's projections read the positional encoding tensor


def encoder_projections(x, pos, w_cat, b_cat, row_mask, n_value, head_dim):
    """The encoder self-attention's value projection and packed (offsets | logits) projection as ONE launch
    (include/codetr_hip.h codetr_encoder_projections_*; reference multi_scale_deformable_attention.py:161-182 with
    value = query): value = x @ w_cat[:n_value]^T + b (rows where row_mask is True zeroed), returned head-major
    [B, n_value / head_dim, S, head_dim] in FP16 (fp16 and bf16 models alike: the packed MSDA kernel's value map), and
    packed [B, S, Np] = (x + pos) @ w_cat[n_value:]^T + b in x's type.  None where the library declines the shape (the
    caller then runs the two GEMMs)."""
    _gpu(x, "encoder_projections")
    if not (ENC_PROJ_FUSED and x.dim() == 3 and x.dtype in (torch.float16, torch.bfloat16) and pos is not None
            and pos.shape == x.shape and pos.dtype == x.dtype and w_cat.dtype == x.dtype and not torch.is_grad_enabled()):
        return None
    B, S, K = x.shape
    n_packed = w_cat.shape[0] - n_value
    x2, p2 = x.reshape(-1, K), pos.reshape(-1, K)
    x2 = x2 if x2.is_contiguous() else x2.contiguous()
    p2 = p2 if p2.is_contiguous() else p2.contiguous()
    mk = None
    if row_mask is not None:
        mk = row_mask.reshape(-1)
        if mk.dtype != torch.bool and mk.dtype != torch.uint8:
            mk = mk != 0
        mk = mk.contiguous()
    value = torch.empty((B * S, n_value), dtype=torch.float16, device=x.device)
    packed = torch.empty((B * S, n_packed), dtype=x.dtype, device=x.device)
    ok = [False]
    gen = getattr(pos, "_codetr_posgen", None) if ENC_POSGEN else None   # (set by the producer of `pos`: co_dino_head.py)

    def launch():
        if gen is not None and gen["S"] == S and gen["B"] == B:
            ok[0] = _cabi.encoder_projections_posgen(x2, S, gen["cums"], gen["shapes"], gen["level_embed"], gen["temperature"],
                                                     gen["scale"], gen["eps"], gen["offset"], gen["normalize"], w_cat, b_cat,
                                                     mk, value, packed, S, int(head_dim))
            if ok[0]:
                return
        ok[0] = _cabi.encoder_projections(x2, p2, w_cat, b_cat, mk, value, packed, S, int(head_dim))

    with torch.cuda.device(x.device):
        if LINEAR_PROFILE is None:
            launch()
        else:
            st = torch.cuda.current_stream(x.device)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            launch()
            e1.record(st)
            if ok[0]:
                LINEAR_PROFILE.append((e0, e1, 2.0 * B * S * (n_value + n_packed) * K, B * S, n_value + n_packed, K))
    if not ok[0]:
        return None
    return value.view(B, n_value // head_dim, S, head_dim), packed.view(B, S, n_packed)


def msda_packed_projection(w_off, b_off, w_aw, b_aw, num_heads, num_levels, num_points):
    """(sampling_offsets | attention_weights) as ONE [64 M, C] weight + [64 M] bias whose output row is the lane-major
    packed layout codetr_msda_encoder_forward_packed_f16 reads (include/codetr_hip.h): per head 4 x 16 columns = the
    (x, y) offsets of point p on the five levels, its five logits, one zero pad column.  Pure row permutation of the
    reference's two Linears (multi_scale_deformable_attention.py:83-85): no arithmetic changes."""
    idx = _cabi.msda_pack_projection_index(num_heads, num_levels, num_points)
    if idx is None:
        return None
    ii = torch.tensor(idx, dtype=torch.long, device=w_off.device)
    wcat = torch.cat((w_off, w_aw), 0)
    bcat = torch.cat((b_off, b_aw), 0)
    pad = ii < 0
    wp = wcat[ii.clamp_min(0)].clone()
    bp = bcat[ii.clamp_min(0)].clone()
    wp[pad] = 0
    bp[pad] = 0
    return wp.contiguous(), bp.contiguous()


def msda_encoder_windows_packed(bias, level_shapes, num_heads, num_levels, num_points):
    """Staged window per (head, level) for the packed encoder kernel, as msda_encoder_windows: the bounding box of the
    head's bias points on that level grown by the largest margin (steps of 1/2 pixel, at most MSDA_V4_MARGIN_CAP) that
    keeps the workgroup inside MSDA_V4_LDS_BUDGET, per pass {0}, {1, 2}, {3, 4}.  Windows are clamped to the level (plus
    its zero border) by the kernel, so on the coarse levels a generous margin ends as whole-level residency."""
    import math

    b = bias.detach().float().cpu().view(num_heads, num_levels, num_points, 2)
    b = torch.round(b * 1024) / 1024
    lo, hi = b.amin(2).tolist(), b.amax(2).tolist()
    groups = [[0], [1, 2], [3, 4]]
    steps = int(2 * MSDA_V4_MARGIN_CAP)

    def window(m, l, mg):
        return (max(-127, math.floor(lo[m][l][0] - mg)), min(127, math.ceil(hi[m][l][0] + mg)),
                max(-127, math.floor(lo[m][l][1] - mg)), min(127, math.ceil(hi[m][l][1] + mg)))

    def need(trial):
        return _cabi.msda_encoder_packed_lds_bytes(level_shapes, num_heads, num_points, [trial] * num_heads,
                                                   MSDA_V4_REGION, MSDA_V4_THREADS)

    out = []
    for m in range(num_heads):
        win = [window(m, l, 0.0) for l in range(num_levels)]
        for grp in groups:
            a, z = 0, steps          # largest half-step count that fits (monotone): bisection
            while a < z:
                mid = (a + z + 1) // 2
                trial = list(win)
                for l in grp:
                    trial[l] = window(m, l, 0.5 * mid)
                n = need(trial)
                if 0 < n <= MSDA_V4_LDS_BUDGET:
                    a = mid
                else:
                    z = mid - 1
            for l in grp:
                win[l] = window(m, l, 0.5 * a)
        out.append(win)
    return out


def msda_encoder_packed(value, level_shapes, packed, num_points, windows, valid_counts, head_major=False):
    """Encoder self-attention MSDA on the lane-major packed projection (round-5 kernel).  value [B,S,M,32] fp16 (or
    [B,M,S,32] with head_major: what linear(..., head_major=32) returns), packed [B,S,64 M] fp16, valid_counts [B,L,2]
    fp32.  Returns None when the library does not take the shape."""
    _gpu(value, "msda_encoder_packed")
    if head_major:
        B, M, S, D = value.shape
    else:
        B, S, M, D = value.shape
    if not (valid_counts is not None and valid_counts.dtype == torch.float32 and valid_counts.is_contiguous()
            and valid_counts.shape == (B, len(level_shapes), 2) and packed.shape[:2] == (B, S) and packed.is_contiguous()):
        return None
    if value.dtype != torch.float16 or packed.dtype not in (torch.float16, torch.bfloat16):
        return None
    out = torch.empty((B, S, M * D), dtype=packed.dtype, device=value.device)
    ok = [True]

    def run():
        ok[0] = _cabi.msda_encoder_packed(value.contiguous(), level_shapes, packed, num_points, windows, valid_counts,
                                          MSDA_V4_REGION, MSDA_V4_THREADS, out, head_major)

    with torch.cuda.device(value.device):
        _timed("msda_fused", {"B": B, "S": S, "Nq": S, "M": M, "D": D, "L": len(level_shapes), "P": num_points},
               run, value.device)
    return out if ok[0] else None


def msda(value, spatial_shapes, level_start_index, sampling_locations, attention_weights, im2col_step):
    """The reference's native op, hand-written HIP behind the C ABI (always native)."""
    return torch.ops.codetr.multi_scale_deformable_attention(
        value, spatial_shapes, level_start_index, sampling_locations, attention_weights, im2col_step
    )


# ---------------------------------------------------------------------------------------------------------------
# fp8 (OCP e4m3) path -- BASELINE config 5.  Weights are quantised once per parameter (per-output-channel scales),
# activations with static per-tensor scales from a calibration forward (codetr/fp8.py); fp32 accumulation.
# ---------------------------------------------------------------------------------------------------------------
FP8 = _cabi.FP8
FP8_MAX = 448.0
FP8_MIN_TILES = 96   # 256x256 output tiles below which fp16 serves the layer


def fp8_weight(weight):
    """(w8 [N,K] e4m3, w_scale [N] fp32) of an nn.Linear weight: per-output-channel absmax / 448; cached on the parameter"""
    def build():
        w = weight.detach().float()
        scale = (w.abs().amax(1) / FP8_MAX).clamp_min(1e-12)
        return (w / scale[:, None]).to(FP8).contiguous(), scale.contiguous()

    return derived((weight,), "_codetr_fp8_w", build)


def ffn_fp8_supported(x, w1, w2, act):
    """the fp8 fused FFN serves the same layers as ffn_fused with hidden a multiple of 128 (<= 2048), fp16 storage"""
    return (ffn_fused_supported(x, w1, w2, act) and x.dtype == torch.float16 and w1.shape[0] % 128 == 0
            and w1.shape[0] <= 2048)


def ffn_fp8_weights(w1, w2):
    """(w1q [hidden,256] e4m3, s1 [hidden], w2q_packed [256,hidden] e4m3, s2 [256]) of an FFN's two Linear weights:
    per-row absmax / 448 scales; W2's columns permuted inside every 128-block into the order in which the first
    product's accumulators become the second product's operand (packed column 32 g + 4 t + r <- unit 16 t + 4 g + r).
    Cached on w1."""
    def build():
        a, b = w1.detach().float(), w2.detach().float()
        s1 = (a.abs().amax(1) / FP8_MAX).clamp_min(1e-12)
        s2 = (b.abs().amax(1) / FP8_MAX).clamp_min(1e-12)
        p = torch.arange(128, device=w2.device)
        g, t, r = p // 32, (p % 32) // 4, p % 4
        src = 16 * t + 4 * g + r
        bq = (b / s2[:, None]).to(FP8).view(torch.uint8).view(b.shape[0], -1, 128)[:, :, src].reshape(b.shape).contiguous()
        return (a / s1[:, None]).to(FP8).contiguous(), s1.contiguous(), bq.view(FP8), s2.contiguous()

    return derived((w1, w2), "_codetr_ffn_fp8_w", build)


def ffn_fp8(x, w1, b1, w2, b2, x_scale, h_scale, ln=None, pos=None, ln_in=None):
    """ffn_fused on the e4m3 matrix path: both products in fp8 with static activation scales (x_scale for the -- possibly
    LayerNorm'ed -- input, h_scale for the hidden activation), fp16 in and out; same epilogue options."""
    _gpu(x, "ffn_fp8")
    x2 = x.reshape(-1, x.shape[-1])
    if not x2.is_contiguous():
        x2 = x2.contiguous()
    out = torch.empty_like(x2)
    p2 = out2 = None
    if pos is not None:
        p2 = pos.reshape(-1, pos.shape[-1])
        if p2.shape != x2.shape or p2.dtype != x2.dtype:
            raise ValueError("pos must match x in shape and dtype")
        if not p2.is_contiguous():
            p2 = p2.contiguous()
        out2 = torch.empty_like(x2)
    w1q, s1, w2q, s2 = ffn_fp8_weights(w1, w2)
    if x2.shape[0] > 0:
        with torch.cuda.device(x.device):
            _timed("ffn_fp8", {"M": x2.shape[0], "C": x2.shape[1], "hidden": w1.shape[0]},
                   lambda: _cabi.ffn_fp8(x2, w1q, s1, b1, w2q, s2, b2, out, x_scale, h_scale, ln, p2, out2, ln_in), x.device)
    if pos is not None:
        return out.view(x.shape), out2.view(x.shape)
    return out.view(x.shape)


def linear_fp8_supported(rows, weight):
    """True when the fp8 GEMM serves this layer: K a multiple of 128 bytes, N of 8, and enough 256x256 output tiles to
    fill the chip (smaller problems stay on the fp16 kernels)"""
    N, K = weight.shape
    return (weight.is_cuda and weight.dtype == torch.float16 and K % 128 == 0 and N % 8 == 0
            and -(-rows // 256) * -(-N // 256) >= FP8_MIN_TILES
            and rows * N <= 0x7fffffff)   # (the kernel's 32-bit output offsets: larger batches stay on the fp16 kernels)


def linear_fp8(x8, x_scale, weight, bias=None, act=None, residual=None, out_scale=None):
    """act((x8 * x_scale) @ weight.T + bias) (+ residual) with x8 e4m3 [..., K] and the fp16 `weight` quantised per
    output channel (cached).  Returns fp16, or e4m3 = sat(y / out_scale) when out_scale is given."""
    _gpu(x8, "linear_fp8")
    w8, ws = fp8_weight(weight)
    K, N = x8.shape[-1], weight.shape[0]
    x2 = x8.reshape(-1, K)
    r2 = None if residual is None else residual.reshape(-1, N).contiguous()
    out = torch.empty((x2.shape[0], N), dtype=FP8 if out_scale is not None else torch.float16, device=x8.device)
    with torch.cuda.device(x8.device):
        launch = lambda: _cabi.linear_fp8(x2, w8, ws, x_scale, bias, r2, act, out, out_scale or 0.0)  # noqa: E731
        if LINEAR_PROFILE is None:
            launch()
        else:
            st = torch.cuda.current_stream(x8.device)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            launch()
            e1.record(st)
            LINEAR_PROFILE.append((e0, e1, 2.0 * x2.shape[0] * N * K, x2.shape[0], N, K, "fp8"))
    return out.view(*x8.shape[:-1], N)


def linear_fp8mx(x8, x_scales, weight, bias=None, act=None, residual=None, out_mx=False):
    """act((x8 (*) block scales) @ weight.T + bias) (+ residual): x8 e4m3 [..., K] with its MX scale tensor, the fp16
    `weight` quantised per output channel (cached).  Returns fp16, or (e4m3, scales) with block scales along N."""
    _gpu(x8, "linear_fp8mx")
    w8, ws = fp8_weight(weight)
    K, N = x8.shape[-1], weight.shape[0]
    x2 = x8.reshape(-1, K)
    r2 = None if residual is None else residual.reshape(-1, N).contiguous()
    out = torch.empty((x2.shape[0], N), dtype=FP8 if out_mx else torch.float16, device=x8.device)
    sy = torch.empty(_cabi.mx_scale_bytes(x2.shape[0], N), dtype=torch.uint8, device=x8.device) if out_mx else None
    with torch.cuda.device(x8.device):
        launch = lambda: _cabi.linear_fp8mx(x2, x_scales, w8, ws, bias, r2, act, out, sy)  # noqa: E731
        if LINEAR_PROFILE is None:
            launch()
        else:
            st = torch.cuda.current_stream(x8.device)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            launch()
            e1.record(st)
            LINEAR_PROFILE.append((e0, e1, 2.0 * x2.shape[0] * N * K, x2.shape[0], N, K, "fp8"))
    out = out.view(*x8.shape[:-1], N)
    return (out, sy) if out_mx else out


def cast_fp8mx(x):
    """fp16 [..., C] -> (e4m3, MX block scales), C % 128 == 0"""
    _gpu(x, "cast_fp8mx")
    x2 = x.reshape(-1, x.shape[-1])
    x2 = x2 if x2.is_contiguous() else x2.contiguous()
    out = torch.empty(x2.shape, dtype=FP8, device=x.device)
    sy = torch.empty(_cabi.mx_scale_bytes(*x2.shape), dtype=torch.uint8, device=x.device)
    with torch.cuda.device(x.device):
        _cabi.cast_fp8mx(x2, out, sy)
    return out.view(x.shape), sy


def layer_norm_fp8mx(x, weight, bias, eps):
    """LayerNorm(x) -> (e4m3, MX block scales) in one kernel (fp16 input; the norm's fp16 output is never written)"""
    _gpu(x, "layer_norm_fp8mx")
    C = x.shape[-1]
    x2 = x.reshape(-1, C)
    x2 = x2 if x2.is_contiguous() else x2.contiguous()
    out = torch.empty(x2.shape, dtype=FP8, device=x.device)
    sy = torch.empty(_cabi.mx_scale_bytes(x2.shape[0], C), dtype=torch.uint8, device=x.device)
    with torch.cuda.device(x.device):
        _cabi.layernorm_fp8mx(x2, weight, bias, eps, out, sy)
    return out.view(x.shape), sy


def mx_dequant(x8, scales):
    """(tests / diagnostics) the fp32 values an (e4m3, MX scales) pair stands for"""
    x2 = x8.reshape(-1, x8.shape[-1])
    M, K = x2.shape
    m = torch.arange(M, device=x8.device)[:, None]
    kb = torch.arange(K // 32, device=x8.device)[None, :]
    MB = -(-M // 128)
    idx = ((((kb >> 2) * MB + (m >> 7)) * 64 + (kb & 3) * 16 + (m & 15)) * 8 + ((m >> 4) & 7)).long()
    e = scales.long()[idx].float() - 127.0                       # [M, K / 32]
    return (x2.float().view(M, K // 32, 32) * torch.exp2(e)[:, :, None]).view(x8.shape)


def cast_fp8(x, scale):
    """sat(x / scale) -> e4m3, fp16 input"""
    _gpu(x, "cast_fp8")
    x = x if x.is_contiguous() else x.contiguous()
    out = torch.empty(x.shape, dtype=FP8, device=x.device)
    with torch.cuda.device(x.device):
        _cabi.cast_fp8(x, scale, out)
    return out


def layer_norm_fp8(x, weight, bias, eps, scale):
    """sat(LayerNorm(x) / scale) -> e4m3 in one kernel (fp16 input; the norm's fp16 output is never written)"""
    _gpu(x, "layer_norm_fp8")
    C = x.shape[-1]
    x2 = x.reshape(-1, C)
    x2 = x2 if x2.is_contiguous() else x2.contiguous()
    out = torch.empty(x2.shape, dtype=FP8, device=x.device)
    with torch.cuda.device(x.device):
        _cabi.layernorm_fp8(x2, weight, bias, eps, scale, out)
    return out.view(x.shape)


# ---------------------------------------------------------------------------------------------------------------
# small element-wise / gather kernels of the decoder and the head (csrc/small_ops.hip); the torch formulation serves
# fp32 / bf16 parity runs and autograd
# ---------------------------------------------------------------------------------------------------------------
def _native16(*ts):
    return ((not torch.is_grad_enabled()) and ts[0].dtype in (torch.float16, torch.bfloat16)
            and all(t.is_cuda and t.dtype == ts[0].dtype and t.data_ptr() % 16 == 0 for t in ts))


def add(a, b):
    """a + b for same-shape fp16 tensors, or `a` broadcast over b's leading dimension (a stride-0 batch view of a
    parameter, e.g. query_embed.weight[None].expand(B, ...)): `query + query_pos` of the attention layers"""
    _gpu(b, "add")
    if _native16(a, b) and a.shape == b.shape and b.is_contiguous() and b.numel() % 8 == 0:
        period = None
        if a.is_contiguous():
            period = b.numel()
        elif a.dim() >= 2 and a.stride(0) == 0 and a[0].is_contiguous() and a[0].numel() % 8 == 0:
            period = a[0].numel()
        if period is not None:
            out = torch.empty_like(b)
            with torch.cuda.device(b.device):
                _cabi.add_f16(a, b, out, period)
            return out
    return a + b


def sigmoid(x):
    _gpu(x, "sigmoid")
    if _native16(x) and x.is_contiguous():
        out = torch.empty_like(x)
        with torch.cuda.device(x.device):
            _cabi.sigmoid_f16(x, out)
        return out
    return x.sigmoid()


def gather_rows(src, idx):
    """src [B,S,C], idx [B,K] int64 -> [B,K,C] = torch.gather(src, 1, idx[..., None].expand(-1, -1, C))"""
    _gpu(src, "gather_rows")
    C = src.shape[-1]
    if (_native16(src) and src.is_contiguous() and idx.dtype == torch.int64 and idx.is_contiguous() and C % 4 == 0
            and src.data_ptr() % 16 == 0):
        out = torch.empty((idx.shape[0], idx.shape[1], C), dtype=src.dtype, device=src.device)
        with torch.cuda.device(src.device):
            _cabi.gather_rows(src, idx, out)
        return out
    return torch.gather(src, 1, idx.unsqueeze(-1).expand(-1, -1, C))


def decode_boxes_supported(coords_unact, idx):
    return _native16(coords_unact) and coords_unact.is_contiguous() and coords_unact.shape[-1] == 4 and idx.is_contiguous()


def decode_boxes(coords_unact, idx, num_classes, img_w, img_h):
    """head decode in one launch (reference co_dino_head.py:177-209): coords_unact [B,Nq,4] (box branch + reference,
    before the sigmoid), idx [B,K] into the flattened (query, class) scores -> (boxes [B,K,4] xyxy pixels, labels [B,K])"""
    _gpu(coords_unact, "decode_boxes")
    B, K = idx.shape
    boxes = torch.empty((B, K, 4), dtype=coords_unact.dtype, device=coords_unact.device)
    labels = torch.empty((B, K), dtype=torch.int64, device=coords_unact.device)
    with torch.cuda.device(coords_unact.device):
        _cabi.decode_boxes(coords_unact, idx, num_classes, img_w, img_h, boxes, labels)
    return boxes, labels


def valid_ratios(counts, level_wh):
    """counts [B,L,2] fp32 (mask_pyramid) , level_wh [L,2] (W_l, H_l) in the model dtype -> valid ratios [B,L,2]"""
    _gpu(counts, "valid_ratios")
    if _native16(level_wh) and counts.dtype == torch.float32 and counts.is_contiguous() and level_wh.is_contiguous():
        out = torch.empty(counts.shape, dtype=level_wh.dtype, device=counts.device)
        out32 = torch.empty(counts.shape, dtype=torch.float32, device=counts.device)
        with torch.cuda.device(counts.device):
            _cabi.valid_ratios(counts, level_wh, out, out32)
        if out32 is not None:
            out._codetr_f32 = out32   # counts / size unrounded: what the fp32 reference computes (query_sine_embed)
        return out
    return counts.to(level_wh.dtype) / level_wh
