"""codetr_decoder_layer_f16 (csrc/decoder_layer.hip): one launch per DINO decoder layer against the same decoder run as
separate launches (hip_ops linears / LayerNorms / MSDA / fused FFN / query_sine_embed) and against an fp32 ATen
formulation of reference codetr/transformer.py:193-230 + the layer (:233-277) on the same fp16 weights."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _decoder(layers, seed=0, levels=5, dtype=torch.float16):
    from codetr.transformer import DinoTransformerDecoder, build_MLP

    torch.manual_seed(seed)
    cfg = dict(type="DetrTransformerDecoderLayer",
               attn_cfgs=[dict(type="MultiheadAttention", embed_dims=256, num_heads=8, dropout=0.0),
                          dict(type="MultiScaleDeformableAttention", embed_dims=256, num_levels=levels, dropout=0.0)],
               feedforward_channels=2048, ffn_dropout=0.0,
               operation_order=("self_attn", "norm", "cross_attn", "norm", "ffn", "norm"))
    dec = DinoTransformerDecoder(return_intermediate=True, transformerlayers=cfg, num_layers=layers)
    reg = torch.nn.ModuleList(build_MLP(256, 256, 4, 3) for _ in range(layers))
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():   # trained-like scales: every branch contributes, nothing saturates
        for p in list(dec.parameters()) + list(reg.parameters()):
            if p.dim() == 2:
                p.copy_(torch.randn(p.shape, generator=g) * (1.0 / p.shape[1]) ** 0.5)
            else:
                p.copy_(torch.randn(p.shape, generator=g) * 0.1)
        for layer in dec.layers:
            for n in layer.norms:
                n.weight.add_(1.0)
            ca = layer.attentions[1]
            ca.sampling_offsets.weight.mul_(0.5)
            ca.sampling_offsets.bias.copy_(torch.randn(ca.sampling_offsets.bias.shape, generator=g) * 2.0)
        dec.norm.weight.add_(1.0)
        for r in reg:
            r[4].weight.mul_(0.2)
    return dec.to(DEV).to(dtype).eval(), reg.to(DEV).to(dtype).eval()


def _inputs(B, Nq, shapes, seed=5, masked=True, dtype=torch.float16):
    g = torch.Generator(device=DEV).manual_seed(seed)
    S = sum(h * w for h, w in shapes)
    query = torch.randn(B, Nq, 256, device=DEV, generator=g).to(dtype)
    memory = torch.randn(B, S, 256, device=DEV, generator=g).to(dtype)
    mask = (torch.rand(B, S, device=DEV, generator=g) < 0.1) if masked else None
    ref = (torch.randn(B, Nq, 4, device=DEV, generator=g) * 1.5).to(dtype)
    vr32 = (0.6 + 0.4 * torch.rand(B, len(shapes), 2, device=DEV, generator=g)).float()
    vr = vr32.to(dtype)
    vr._codetr_f32 = vr32
    ss = torch.tensor(shapes, dtype=torch.int64, device=DEV)
    ls = torch.cat((ss.new_zeros(1), ss.prod(1).cumsum(0)[:-1]))
    return query, memory, mask, ref, vr, ss, ls


def _run(dec, reg, inp, fused):
    from codetr import transformer as tr

    query, memory, mask, ref, vr, ss, ls = inp
    saved = tr.DEC_FUSED
    tr.DEC_FUSED = bool(fused)   # (route switch: a module attribute since round 5, no environment variable)
    try:
        with torch.no_grad():
            return dec.forward_bf(query, memory, mask, ref, vr, reg, spatial_shapes=ss, level_start_index=ls)
    finally:
        tr.DEC_FUSED = saved


PYR = [(40, 60), (20, 30), (10, 15), (5, 8), (3, 4)]


@pytest.mark.parametrize("B,Nq,layers,levels", [(1, 900, 6, 5), (2, 37, 2, 5), (3, 16, 1, 5), (1, 5, 1, 5), (2, 300, 2, 4)])
def test_one_launch_per_layer_matches_the_separate_launches(B, Nq, layers, levels):
    from codetr import _cabi

    dec, reg = _decoder(layers, levels=levels)
    inp = _inputs(B, Nq, PYR[:levels])
    before = dict(_cabi.CALLS)
    out_f, ref_f = _run(dec, reg, inp, True)
    assert _cabi.CALLS["decoder_layer"] == before["decoder_layer"] + layers + 1
    assert _cabi.CALLS["mha_attention"] == before["mha_attention"] + layers
    assert _cabi.CALLS["layernorm"] == before["layernorm"] and _cabi.CALLS["ffn_fused"] == before["ffn_fused"]
    mid = dict(_cabi.CALLS)
    out_u, ref_u = _run(dec, reg, inp, False)
    assert _cabi.CALLS["decoder_layer"] == mid["decoder_layer"] and _cabi.CALLS["layernorm"] > mid["layernorm"]
    assert out_f.shape == out_u.shape == (B, Nq, 256) and ref_f.shape == ref_u.shape == (B, Nq, 4)
    assert torch.isfinite(out_f.float()).all() and torch.isfinite(ref_f.float()).all()
    # same rounding points, different summation order: a few fp16 ulps per layer on LayerNorm-ed (unit-scale) rows
    d = (out_f.float() - out_u.float())
    rel = float(d.norm() / out_u.float().norm())
    assert rel < 1e-3 * (1 + layers), rel     # measured: 2.3e-3 after 6 layers, 1e-3 after 2
    assert float(d.abs().max()) < 0.06 * max(1, layers // 2)
    assert float((ref_f.float() - ref_u.float()).abs().max()) < 0.02 * layers


def test_bf16_one_launch_per_layer_matches_the_separate_launches():
    """codetr_decoder_layer_bf16 (the same source compiled with bf16 storage): same launch counts as fp16, and as close to
    the separate bf16 launches as 8 mantissa bits allow (different summation order; a bf16 ulp is 2^-8)"""
    from codetr import _cabi

    layers = 3
    dec, reg = _decoder(layers, seed=7, dtype=torch.bfloat16)
    inp = _inputs(2, 300, PYR, seed=8, dtype=torch.bfloat16)
    before = dict(_cabi.CALLS)
    out_f, ref_f = _run(dec, reg, inp, True)
    assert _cabi.CALLS["decoder_layer"] == before["decoder_layer"] + layers + 1
    assert _cabi.CALLS["layernorm"] == before["layernorm"]
    out_u, ref_u = _run(dec, reg, inp, False)
    assert out_f.dtype == torch.bfloat16 and torch.isfinite(out_f.float()).all() and torch.isfinite(ref_f.float()).all()
    rel = float((out_f.float() - out_u.float()).norm() / out_u.float().norm())
    assert rel < 8e-3 * (1 + layers), rel
    assert float((ref_f.float() - ref_u.float()).abs().max()) < 0.15 * layers


def test_against_fp32_formulation_of_the_reference():
    """fp32 ATen walk of the reference's decoder on the same (fp16-valued) weights: the fused launch is as close to it
    as the separate launches are"""
    from codetr.ops import multi_scale_deformable_attention_pytorch
    from codetr.transformer import DinoTransformerDecoder
    import torch.nn.functional as F

    layers = 2
    dec, reg = _decoder(layers, seed=3)
    inp = _inputs(2, 50, PYR, seed=9)
    query, memory, mask, ref, vr, ss, ls = inp
    out_f, ref_f = _run(dec, reg, inp, True)
    out_u, ref_u = _run(dec, reg, inp, False)

    with torch.no_grad():
        x, rp = query.float(), ref.float()
        vr32 = vr._codetr_f32
        for lid, layer in enumerate(dec.layers):
            sa, ca, ffn = layer.attentions[0], layer.attentions[1], layer.ffns[0]
            n1, n2, n3 = layer.norms
            ref_in = rp.sigmoid()[:, :, None] * torch.cat((vr32, vr32), -1)[:, None]
            sine = DinoTransformerDecoder.gen_sineembed_for_position(ref_in[:, :, 0, :], 128)
            qpos = F.linear(F.relu(F.linear(sine, dec.ref_point_head[0].weight.float(), dec.ref_point_head[0].bias.float())),
                            dec.ref_point_head[2].weight.float(), dec.ref_point_head[2].bias.float())
            W, b = sa.attn.in_proj_weight.float(), sa.attn.in_proj_bias.float()
            q = F.linear(x + qpos, W[:256], b[:256]).view(*x.shape[:2], 8, 32).transpose(1, 2)
            k = F.linear(x + qpos, W[256:512], b[256:512]).view(*x.shape[:2], 8, 32).transpose(1, 2)
            v = F.linear(x, W[512:], b[512:]).view(*x.shape[:2], 8, 32).transpose(1, 2)
            o = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(x.shape)
            x1 = F.layer_norm(x + F.linear(o, sa.attn.out_proj.weight.float(), sa.attn.out_proj.bias.float()), (256,),
                              n1.weight.float(), n1.bias.float(), n1.eps)
            val = F.linear(memory.float(), ca.value_proj.weight.float(), ca.value_proj.bias.float())
            val = val.masked_fill(mask[..., None], 0.0).view(*memory.shape[:2], 8, 32)
            q2 = x1 + qpos
            off = F.linear(q2, ca.sampling_offsets.weight.float(), ca.sampling_offsets.bias.float()).view(*x.shape[:2], 8, 5, 4, 2)
            aw = F.linear(q2, ca.attention_weights.weight.float(), ca.attention_weights.bias.float()).view(*x.shape[:2], 8, 20)
            aw = aw.softmax(-1).view(*x.shape[:2], 8, 5, 4)
            loc = ref_in[:, :, None, :, None, :2] + off / 4 * ref_in[:, :, None, :, None, 2:] * 0.5
            s = multi_scale_deformable_attention_pytorch(val, ss, loc, aw)
            x2 = F.layer_norm(x1 + F.linear(s, ca.output_proj.weight.float(), ca.output_proj.bias.float()), (256,),
                              n2.weight.float(), n2.bias.float(), n2.eps)
            h = F.relu(F.linear(x2, ffn.layers[0][0].weight.float(), ffn.layers[0][0].bias.float()))
            x = F.layer_norm(x2 + F.linear(h, ffn.layers[1].weight.float(), ffn.layers[1].bias.float()), (256,),
                             n3.weight.float(), n3.bias.float(), n3.eps)
            r = reg[lid]
            d = F.linear(F.relu(F.linear(F.relu(F.linear(x, r[0].weight.float(), r[0].bias.float())), r[2].weight.float(),
                                         r[2].bias.float())), r[4].weight.float(), r[4].bias.float())
            rp = rp + d
        out32 = F.layer_norm(x, (256,), dec.norm.weight.float(), dec.norm.bias.float(), dec.norm.eps)

    def err(a):
        return float((a.float() - out32).norm() / out32.norm())

    e_f, e_u = err(out_f), err(out_u)
    assert e_f < 1.5e-2 and e_f < 1.5 * e_u + 1e-3, (e_f, e_u)
    assert float((ref_f.float() - rp).abs().max()) < 0.03


def test_contract():
    from codetr import _cabi

    assert _cabi.decoder_layer_supported(256, 8, 5, 4, 2048, 4, 128)
    assert not _cabi.decoder_layer_supported(256, 8, 5, 4, 2048, 2, 128)      # 2-d reference points
    assert not _cabi.decoder_layer_supported(384, 12, 5, 4, 2048, 4, 192)
    assert not _cabi.decoder_layer_supported(256, 8, 5, 4, 1024, 4, 128)
    assert _cabi.decoder_layer_blob_halfs(1, 5, 4, 2048) == 3 * 256 * 256 + 3 * 256
    assert _cabi.decoder_layer_blob_halfs(3, 5, 4, 2048) == 512
    x = torch.zeros(16, 256, device=DEV, dtype=torch.float16)
    with pytest.raises(RuntimeError):    # neither a tail nor a head
        _cabi.decoder_layer(x, None, None, x, x.float(), None, None, None, None, None, None, None, None, None, None, None,
                            None, 1, 16, 10, 5, 4, 2048, 1e-5, 10000.0)
