"""ORACLE -- TEST INFRASTRUCTURE ONLY (not part of the product path).

A functional, state-dict driven restatement of the reference's ``CoDETR.forward`` in plain
PyTorch on the CPU (fp32, or fp64 for tight checks).  No nn.Module classes, no config system:
every function takes the flat ``state_dict`` (mmdet checkpoint key names) plus tensors, so it is
structurally independent both of the reference's class hierarchy and of the product package in
``co-detr-tensorrt_amd/codetr``.  The MSDA step calls the C oracle (oracle/msda_ref.c), not
``grid_sample``.

What each function follows (paths relative to the reference tree):

  swin_*                codetr/swin.py:80-116 (window attention), :175-252 (pad / shift / mask /
                        partition / reverse), :368-379 (block), :725-749 (stages + out norms);
                        PatchEmbed / PatchMerging: codetr/transformer_mmcv.py:191-210, 276-316
  channel_mapper        mmdet v3.3.0 ``ChannelMapper`` (third party, built at codetr/codetr.py:53-54
                        from configs lsj:40-47 + swin:29): 1x1 conv (no bias) + GN(32) per level,
                        extra 3x3 stride-2 conv + GN on the RAW last input
  sine_positional_encoding   codetr/positional_encoding.py:58-93
  msda_module           codetr/multi_scale_deformable_attention.py:117-218
  mha_module            codetr/transformer_mmcv.py:366-428 (nn.MultiheadAttention semantics)
  ffn                   codetr/transformer_mmcv.py:484-500
  encoder / decoder     codetr/transformer.py:81-92, 156-229 + transformer_mmcv.py:709-749
  transformer           codetr/transformer.py:280-400 (helpers), 480-582 (forward)
  head                  codetr/co_dino_head.py:120-210 (+ mmdet ``bbox_cxcywh_to_xyxy``)
  codetr_forward        codetr/codetr.py:87-90

Pinning: tests/golden/model_*.npz hold outputs of the reference's own modules (imported live by
tests/golden/make_golden.py with seeded weights) for the MSDA module, positional encoding, one
Swin stage incl. padding/shift/merging, encoder, decoder with forced top-k, and the transformer;
tests/test_oracle_model.py checks this file against them.  ChannelMapper / the head's
top-k+box decode / ResNet have no importable reference here (mmdet absent): "parity unpinned"
for those three, covered by hand-derived known-answer tests instead.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg may import this.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

import msda_oracle

# ------------------------------------------------------------------------------------------
# small helpers
# ------------------------------------------------------------------------------------------


def _lin(sd, p, x):
    return F.linear(x, sd[p + ".weight"], sd.get(p + ".bias"))


def _ln(sd, p, x, eps=1e-5):
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], eps)


def _count(sd, prefix):
    """number of consecutive integer children under `prefix` (e.g. 'backbone.stages.')"""
    n = 0
    while any(k.startswith(f"{prefix}{n}.") for k in sd):
        n += 1
    return n


# ------------------------------------------------------------------------------------------
# Swin backbone
# ------------------------------------------------------------------------------------------


def _rel_pos_index(ws):
    """[ws*ws, ws*ws] index into the (2ws-1)^2 bias table: entry (i, j) addresses the offset
    (dy, dx) = (yi - yj + ws - 1, xi - xj + ws - 1)   (swin.py:63-67 builds the same table)."""
    coords = torch.stack(torch.meshgrid(torch.arange(ws), torch.arange(ws), indexing="ij")).flatten(1)  # [2, N]
    rel = coords[:, :, None] - coords[:, None, :] + (ws - 1)
    return rel[0] * (2 * ws - 1) + rel[1]


def _window_partition(x, ws):
    B, H, W, C = x.shape
    x = x.view(B, H // ws, ws, W // ws, ws, C).permute(0, 1, 3, 2, 4, 5)
    return x.reshape(-1, ws * ws, C)


def _window_reverse(win, ws, B, H, W):
    C = win.shape[-1]
    x = win.view(B, H // ws, W // ws, ws, ws, C).permute(0, 1, 3, 2, 4, 5)
    return x.reshape(B, H, W, C)


def swin_window_attention(sd, p, x, hw, num_heads, ws, shift):
    """x [B, H*W, C] (already norm1'ed) -> attention branch output [B, H*W, C]."""
    B, _, C = x.shape
    H, W = hw
    x = x.view(B, H, W, C)
    pad_b, pad_r = (-H) % ws, (-W) % ws
    x = F.pad(x, (0, 0, 0, pad_r, 0, pad_b))  # zeros AFTER norm1: pad tokens become the qkv bias
    Hp, Wp = H + pad_b, W + pad_r
    mask = None
    if shift > 0:
        x = torch.roll(x, (-shift, -shift), (1, 2))
        region = torch.zeros(Hp, Wp, dtype=x.dtype)
        cnt = 0
        for hs in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
            for wsl in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
                region[hs, wsl] = cnt
                cnt += 1
        rw = _window_partition(region[None, :, :, None], ws).squeeze(-1)  # [nW, N]
        diff = rw[:, None, :] - rw[:, :, None]
        mask = torch.where(diff != 0, torch.full_like(diff, -100.0), torch.zeros_like(diff))  # [nW, N, N]
    win = _window_partition(x, ws)  # [B*nW, N, C]
    N = ws * ws
    hd = C // num_heads
    qkv = _lin(sd, p + ".w_msa.qkv", win).view(-1, N, 3, num_heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0] * hd ** -0.5, qkv[1], qkv[2]
    attn = q @ k.transpose(-2, -1)  # [B*nW, nH, N, N]
    table = sd[p + ".w_msa.relative_position_bias_table"]  # [(2ws-1)^2, nH]
    bias = table[_rel_pos_index(ws).reshape(-1)].view(N, N, num_heads).permute(2, 0, 1)
    attn = attn + bias[None]
    if mask is not None:
        nW = mask.shape[0]
        attn = (attn.view(B, nW, num_heads, N, N) + mask[None, :, None]).view(-1, num_heads, N, N)
    attn = attn.softmax(-1)
    out = (attn @ v).transpose(1, 2).reshape(-1, N, C)
    out = _lin(sd, p + ".w_msa.proj", out)
    x = _window_reverse(out, ws, B, Hp, Wp)
    if shift > 0:
        x = torch.roll(x, (shift, shift), (1, 2))
    return x[:, :H, :W, :].reshape(B, H * W, C)


def swin_block(sd, p, x, hw, num_heads, ws, shift):
    x = x + swin_window_attention(sd, p + ".attn", _ln(sd, p + ".norm1", x), hw, num_heads, ws, shift)
    h = _ln(sd, p + ".norm2", x)
    h = _lin(sd, p + ".ffn.layers.1", F.gelu(_lin(sd, p + ".ffn.layers.0.0", h)))
    return x + h


def patch_merging(sd, p, x, hw):
    """2x2 neighbourhood -> 4C (nn.Unfold channel order: c*4 + ky*2 + kx) -> LN -> Linear(no bias)."""
    B, _, C = x.shape
    H, W = hw
    x = x.view(B, H, W, C)
    x = F.pad(x, (0, 0, 0, W % 2, 0, H % 2))
    H2, W2 = (H + 1) // 2, (W + 1) // 2
    x = x.view(B, H2, 2, W2, 2, C).permute(0, 1, 3, 5, 2, 4).reshape(B, H2 * W2, 4 * C)
    x = _ln(sd, p + ".norm", x)
    return F.linear(x, sd[p + ".reduction.weight"]), (H2, W2)


def swin_forward(sd, img, num_heads, window_size, prefix="backbone", out_indices=(0, 1, 2, 3)):
    """img [B,3,H,W] -> list of [B, C_i, H_i, W_i]."""
    w = sd[prefix + ".patch_embed.projection.weight"]
    ps = w.shape[-1]
    H, W = img.shape[-2:]
    img = F.pad(img, (0, (-W) % ps, 0, (-H) % ps))
    x = F.conv2d(img, w, sd[prefix + ".patch_embed.projection.bias"], stride=ps)
    hw = tuple(x.shape[-2:])
    x = x.flatten(2).transpose(1, 2)
    if prefix + ".patch_embed.norm.weight" in sd:
        x = _ln(sd, prefix + ".patch_embed.norm", x)
    outs = []
    n_stages = _count(sd, prefix + ".stages.")
    for i in range(n_stages):
        sp = f"{prefix}.stages.{i}"
        for j in range(_count(sd, sp + ".blocks.")):
            x = swin_block(sd, f"{sp}.blocks.{j}", x, hw, num_heads[i], window_size,
                           window_size // 2 if j % 2 else 0)
        if i in out_indices:
            o = _ln(sd, f"{prefix}.norm{i}", x)
            outs.append(o.view(-1, hw[0], hw[1], o.shape[-1]).permute(0, 3, 1, 2).contiguous())
        if sp + ".downsample.reduction.weight" in sd:
            x, hw = patch_merging(sd, sp + ".downsample", x, hw)
    return outs


# ------------------------------------------------------------------------------------------
# ResNet-50 (config 1; mmdet ResNet depth 50, style='pytorch', frozen BN in eval)
# ------------------------------------------------------------------------------------------


def _bn(sd, p, x, eps=1e-5):
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"],
                        False, 0.0, eps)


def resnet50_forward(sd, img, prefix="backbone"):
    x = F.relu(_bn(sd, prefix + ".bn1", F.conv2d(img, sd[prefix + ".conv1.weight"], stride=2, padding=3)))
    x = F.max_pool2d(x, 3, 2, 1)
    outs = []
    for li, nblk in enumerate((3, 4, 6, 3)):
        for b in range(nblk):
            p = f"{prefix}.layer{li + 1}.{b}"
            stride = 2 if (b == 0 and li > 0) else 1
            idt = x
            y = F.relu(_bn(sd, p + ".bn1", F.conv2d(x, sd[p + ".conv1.weight"])))
            y = F.relu(_bn(sd, p + ".bn2", F.conv2d(y, sd[p + ".conv2.weight"], stride=stride, padding=1)))  # style='pytorch'
            y = _bn(sd, p + ".bn3", F.conv2d(y, sd[p + ".conv3.weight"]))
            if p + ".downsample.0.weight" in sd:
                idt = _bn(sd, p + ".downsample.1", F.conv2d(x, sd[p + ".downsample.0.weight"], stride=stride))
            x = F.relu(y + idt)
        outs.append(x)
    return outs


# ------------------------------------------------------------------------------------------
# neck
# ------------------------------------------------------------------------------------------


def channel_mapper(sd, feats, prefix="neck", groups=32):
    outs = []
    for i, f in enumerate(feats):
        y = F.conv2d(f, sd[f"{prefix}.convs.{i}.conv.weight"], sd.get(f"{prefix}.convs.{i}.conv.bias"),
                     padding=(sd[f"{prefix}.convs.{i}.conv.weight"].shape[-1] - 1) // 2)
        outs.append(F.group_norm(y, groups, sd[f"{prefix}.convs.{i}.gn.weight"], sd[f"{prefix}.convs.{i}.gn.bias"]))
    for i in range(_count(sd, prefix + ".extra_convs.")):
        src = feats[-1] if i == 0 else outs[-1]
        y = F.conv2d(src, sd[f"{prefix}.extra_convs.{i}.conv.weight"], sd.get(f"{prefix}.extra_convs.{i}.conv.bias"),
                     stride=2, padding=1)
        outs.append(F.group_norm(y, groups, sd[f"{prefix}.extra_convs.{i}.gn.weight"],
                                 sd[f"{prefix}.extra_convs.{i}.gn.bias"]))
    return outs


# ------------------------------------------------------------------------------------------
# positional encoding
# ------------------------------------------------------------------------------------------


def sine_positional_encoding(mask, dtype, num_feats=128, temperature=20, scale=2 * math.pi, eps=1e-6, offset=0.0):
    """mask [B,H,W] bool (True = padding) -> [B, 2*num_feats, H, W]; y-features first."""
    valid = (~mask).to(dtype)
    y = valid.cumsum(1)
    x = valid.cumsum(2)
    y = (y + offset) / (y[:, -1:, :] + eps) * scale
    x = (x + offset) / (x[:, :, -1:] + eps) * scale
    i = torch.arange(num_feats, dtype=dtype)
    dim_t = temperature ** (2 * torch.div(i, 2, rounding_mode="floor") / num_feats)
    px, py = x[..., None] / dim_t, y[..., None] / dim_t
    B, H, W = mask.shape
    px = torch.stack((px[..., 0::2].sin(), px[..., 1::2].cos()), -1).view(B, H, W, -1)
    py = torch.stack((py[..., 0::2].sin(), py[..., 1::2].cos()), -1).view(B, H, W, -1)
    return torch.cat((py, px), 3).permute(0, 3, 1, 2)


# ------------------------------------------------------------------------------------------
# attention modules (batch-first internally: [B, N, C])
# ------------------------------------------------------------------------------------------


def msda_op(value, spatial_shapes, level_start, loc, w):
    """tensor wrapper around the C oracle; arithmetic type = the tensors' dtype (f32/f64)."""
    npdt = np.float64 if value.dtype == torch.float64 else np.float32
    out = msda_oracle.msda_forward_c(value.numpy(), spatial_shapes.numpy(), level_start.numpy(), loc.numpy(),
                                     w.numpy(), im2col_step=value.shape[0], dtype=npdt)
    return torch.from_numpy(out).to(value.dtype)


def msda_module(sd, p, query, value, query_pos, key_padding_mask, reference_points, spatial_shapes, level_start,
                num_heads=8, num_points=4):
    """query [B,Nq,C], value [B,S,C] (None -> query), reference_points [B,Nq,L,2|4] -> query-shaped output
    INCLUDING the residual (identity = the un-positioned query)."""
    identity = query
    if value is None:
        value = query
    if query_pos is not None:
        query = query + query_pos
    B, Nq, C = query.shape
    S = value.shape[1]
    L = spatial_shapes.shape[0]
    v = _lin(sd, p + ".value_proj", value)
    if key_padding_mask is not None:
        v = v.masked_fill(key_padding_mask[..., None], 0.0)
    v = v.view(B, S, num_heads, -1)
    off = _lin(sd, p + ".sampling_offsets", query).view(B, Nq, num_heads, L, num_points, 2)
    aw = _lin(sd, p + ".attention_weights", query).view(B, Nq, num_heads, L * num_points).softmax(-1)
    aw = aw.view(B, Nq, num_heads, L, num_points)
    if reference_points.shape[-1] == 2:
        wh = torch.stack((spatial_shapes[:, 1], spatial_shapes[:, 0]), -1).to(query.dtype)
        loc = reference_points[:, :, None, :, None, :] + off / wh[None, None, None, :, None, :]
    else:
        loc = (reference_points[:, :, None, :, None, :2]
               + off / num_points * reference_points[:, :, None, :, None, 2:] * 0.5)
    out = msda_op(v.contiguous(), spatial_shapes, level_start, loc.contiguous(), aw.contiguous())
    return _lin(sd, p + ".output_proj", out) + identity


def mha_module(sd, p, query, query_pos, num_heads=8):
    """self-attention as nn.MultiheadAttention computes it; q = k = query + pos, v = query; + residual."""
    B, N, C = query.shape
    qk = query + query_pos if query_pos is not None else query
    Wi, bi = sd[p + ".attn.in_proj_weight"], sd[p + ".attn.in_proj_bias"]
    q = F.linear(qk, Wi[:C], bi[:C])
    k = F.linear(qk, Wi[C:2 * C], bi[C:2 * C])
    v = F.linear(query, Wi[2 * C:], bi[2 * C:])
    hd = C // num_heads
    sp = lambda t: t.view(B, N, num_heads, hd).transpose(1, 2)  # noqa: E731
    a = (sp(q) * hd ** -0.5) @ sp(k).transpose(-2, -1)
    o = (a.softmax(-1) @ sp(v)).transpose(1, 2).reshape(B, N, C)
    return query + _lin(sd, p + ".attn.out_proj", o)


def ffn(sd, p, x):
    return x + _lin(sd, p + ".layers.1", F.relu(_lin(sd, p + ".layers.0.0", x)))


# ------------------------------------------------------------------------------------------
# encoder / decoder / transformer
# ------------------------------------------------------------------------------------------


def encoder(sd, p, x, pos, pad_mask, ref_by_level, spatial_shapes, level_start):
    for i in range(_count(sd, p + ".layers.")):
        lp = f"{p}.layers.{i}"
        x = msda_module(sd, lp + ".attentions.0", x, None, pos, pad_mask, ref_by_level, spatial_shapes, level_start)
        x = _ln(sd, lp + ".norms.0", x)
        x = ffn(sd, lp + ".ffns.0", x)
        x = _ln(sd, lp + ".norms.1", x)
    return x


def _sine_embed(pos, feat=128, temperature=10000.0):
    """pos [B,N,4] (x,y,w,h in [0,1]) -> [B,N,4*feat] ordered (y, x, w, h)."""
    i = torch.arange(feat, dtype=pos.dtype)
    dim_t = temperature ** (2 * torch.div(i, 2, rounding_mode="floor") / feat)

    def emb(c):
        e = (c * (2 * math.pi))[..., None] / dim_t
        return torch.stack((e[..., 0::2].sin(), e[..., 1::2].cos()), -1).flatten(-2)

    parts = [emb(pos[..., 1]), emb(pos[..., 0])]
    if pos.shape[-1] == 4:
        parts += [emb(pos[..., 2]), emb(pos[..., 3])]
    return torch.cat(parts, -1)


def reg_branch(sd, p, x):
    x = F.relu(_lin(sd, p + ".0", x))
    x = F.relu(_lin(sd, p + ".2", x))
    return _lin(sd, p + ".4", x)


def decoder(sd, p, query, memory, pad_mask, ref_unact, valid_ratios, spatial_shapes, level_start, reg_prefix,
            layer_capture=None):
    """query [B,Nq,C]; ref_unact [B,Nq,4] (logits). Returns (normed final state [B,Nq,C], refs [B,Nq,4]).
    layer_capture: a list that receives, per layer, dict(x_in, ref_in_unact, qpos, x_out, ref_out_unact) -- the layer's
    own inputs and outputs (reference transformer.py:193-230), for single-layer parity tests."""
    x = query
    vr4 = torch.cat((valid_ratios, valid_ratios), -1)  # [B,L,4]
    n_layers = _count(sd, p + ".layers.")
    for i in range(n_layers):
        lp = f"{p}.layers.{i}"
        ref_in = ref_unact.sigmoid()[:, :, None, :] * vr4[:, None]  # [B,Nq,L,4]
        qpos = _lin(sd, p + ".ref_point_head.2", F.relu(_lin(sd, p + ".ref_point_head.0", _sine_embed(ref_in[:, :, 0, :]))))
        if layer_capture is not None:
            layer_capture.append(dict(x_in=x, ref_in_unact=ref_unact, qpos=qpos))
        x = mha_module(sd, lp + ".attentions.0", x, qpos)
        x = _ln(sd, lp + ".norms.0", x)
        x = msda_module(sd, lp + ".attentions.1", x, memory, qpos, pad_mask, ref_in, spatial_shapes, level_start)
        x = _ln(sd, lp + ".norms.1", x)
        x = ffn(sd, lp + ".ffns.0", x)
        x = _ln(sd, lp + ".norms.2", x)
        ref_unact = reg_branch(sd, f"{reg_prefix}.{i}", x) + ref_unact  # no detach, no sigmoid
        if layer_capture is not None:
            layer_capture[-1].update(x_out=x, ref_out_unact=ref_unact)
    return _ln(sd, p + ".norm", x), ref_unact


def _valid_ratio(mask, dtype):
    H, W = mask.shape[1:]
    vh = (~mask[:, :, 0]).sum(1).to(dtype) / H
    vw = (~mask[:, 0, :]).sum(1).to(dtype) / W
    return torch.stack((vw, vh), -1)


def transformer(sd, feats, masks, pos_embeds, p="query_head.transformer", head="query_head", num_query=900,
                forced_topk=None, capture=None):
    """feats: list [B,C,H,W]; masks: list [B,H,W] bool; pos_embeds: list [B,C,H,W].
    Returns (final_state [B,Nq,C], refs_unact [B,Nq,4]).  `forced_topk` [B,Nq] int64 (or a callable
    (enc_cls, enc_coord, k) -> indices) overrides the proposal selection (parity on random weights, SURVEY.md
    section 4); `capture` dict receives intermediates."""
    dtype = feats[0].dtype
    B = feats[0].shape[0]
    shapes = [tuple(f.shape[-2:]) for f in feats]
    x = torch.cat([f.flatten(2).transpose(1, 2) for f in feats], 1)
    pad = torch.cat([m.flatten(1) for m in masks], 1)
    lvl = sd[p + ".level_embeds"]
    pos = torch.cat([pe.flatten(2).transpose(1, 2) + lvl[i].view(1, 1, -1) for i, pe in enumerate(pos_embeds)], 1)
    ss = torch.tensor(shapes, dtype=torch.long)
    counts = ss.prod(1)
    start = torch.cat((ss.new_zeros(1), counts.cumsum(0)[:-1]))
    vr = torch.stack([_valid_ratio(m, dtype) for m in masks], 1)  # [B,L,2]
    refs = []
    for l, (H, W) in enumerate(shapes):
        ry, rx = torch.meshgrid(torch.linspace(0.5, H - 0.5, H, dtype=dtype), torch.linspace(0.5, W - 0.5, W, dtype=dtype),
                                indexing="ij")
        refs.append(torch.stack((rx.reshape(1, -1) / (vr[:, l, 0:1] * W), ry.reshape(1, -1) / (vr[:, l, 1:2] * H)), -1))
    ref = torch.cat(refs, 1)  # [B,S,2]
    ref_by_level = ref[:, :, None] * vr[:, None]
    memory = encoder(sd, p + ".encoder", x, pos, pad, ref_by_level, ss, start)
    # proposals
    lvl_of = torch.repeat_interleave(torch.arange(len(shapes), dtype=dtype), counts)
    wh = (0.05 * 2.0 ** lvl_of).view(1, -1, 1).expand(B, -1, 1)
    prop = torch.cat((ref, wh, wh), -1)
    prop = torch.log(prop / (1 - prop))
    ok = ((prop > -4.6) & (prop < 4.6)).to(dtype).prod(-1, keepdim=True) * (~pad).to(dtype)[..., None]
    prop = prop * ok + (1.0 - ok) * torch.finfo(dtype).max
    om = memory * ok
    om = _ln(sd, p + ".enc_output_norm", _lin(sd, p + ".enc_output", om))
    n_dec = _count(sd, p + ".decoder.layers.")
    enc_cls = _lin(sd, f"{head}.cls_branches.{n_dec}", om)
    enc_coord = reg_branch(sd, f"{head}.reg_branches.{n_dec}", om) + prop
    if callable(forced_topk):   # e.g. helpers_model.valid_topk: a selection rule evaluated on this pass's own scores
        topk = forced_topk(enc_cls, enc_coord, num_query)
    else:
        topk = forced_topk if forced_topk is not None else torch.topk(enc_cls.max(-1)[0], num_query, dim=1)[1]
    ref_unact = torch.gather(enc_coord, 1, topk[..., None].expand(-1, -1, 4))
    query = sd[p + ".query_embed.weight"][None].expand(B, -1, -1)
    if capture is not None:
        capture.update(memory=memory, enc_outputs_class=enc_cls, enc_outputs_coord_unact=enc_coord, topk_indices=topk,
                       spatial_shapes=ss, level_start_index=start, valid_ratios=vr, reference_points=ref)
    return decoder(sd, p + ".decoder", query, memory, pad, ref_unact, vr, ss, start, f"{head}.reg_branches")


def head(sd, feats, img_masks, p="query_head", num_query=900, max_per_img=300, num_classes=80, forced_topk=None,
         capture=None):
    """feats list [B,256,H_l,W_l], img_masks [B,H,W] (1 = padding) -> boxes [B,300,4] xyxy px, scores, labels."""
    dtype = feats[0].dtype
    Himg, Wimg = img_masks.shape[-2:]
    masks, pos = [], []
    for f in feats:
        m = F.interpolate(img_masks[:, None].to(dtype), size=f.shape[-2:]).to(torch.bool).squeeze(1)  # nearest
        masks.append(m)
        pos.append(sine_positional_encoding(m, dtype))
    state, refs = transformer(sd, feats, masks, pos, p + ".transformer", p, num_query, forced_topk, capture)
    last = _count(sd, p + ".transformer.decoder.layers.") - 1
    cls = _lin(sd, f"{p}.cls_branches.{last}", state)
    coords = (reg_branch(sd, f"{p}.reg_branches.{last}", state) + refs).sigmoid()
    if capture is not None:
        capture.update(final_state=state, final_refs_unact=refs, outputs_classes=cls, outputs_coords=coords)
    return decode_detections(cls, coords, Himg, Wimg, max_per_img, num_classes)


def decode_detections(cls, coords, Himg, Wimg, max_per_img=300, num_classes=80):
    """class logits [B,Nq,C] + normalised cxcywh [B,Nq,4] -> (xyxy pixels [B,K,4], scores [B,K], labels [B,K]):
    top-k over the flattened (query, class) sigmoid scores, label = idx % C, query = idx // C,
    cxcywh -> xyxy, scale by (W,H), clamp to the image (reference co_dino_head.py:181-209)."""
    B = cls.shape[0]
    scores, idx = torch.topk(cls.sigmoid().view(B, -1), max_per_img, dim=-1)
    labels = idx % num_classes
    box = torch.gather(coords, 1, (idx // num_classes)[..., None].expand(-1, -1, 4))
    cx, cy, w, h = box.unbind(-1)
    xyxy = torch.stack((cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h), -1)
    scale = xyxy.new_tensor([Wimg, Himg, Wimg, Himg])
    xyxy = torch.minimum(torch.clamp(xyxy * scale, min=0), scale)
    return xyxy, scores, labels


def codetr_forward(sd, batch_inputs, img_masks, backbone="swin", num_heads=(6, 12, 24, 48), window_size=12,
                   num_query=900, max_per_img=300, num_classes=80, forced_topk=None, capture=None):
    """The reference's CoDETR.forward (codetr.py:87-90) on CPU.  `sd` tensors define the dtype."""
    if backbone == "swin":
        feats = swin_forward(sd, batch_inputs, num_heads, window_size)
    else:
        feats = resnet50_forward(sd, batch_inputs)
    if capture is not None:
        capture["backbone_feats"] = feats
    feats = channel_mapper(sd, feats)
    if capture is not None:
        capture["neck_feats"] = feats
    return head(sd, feats, img_masks, "query_head", num_query, max_per_img, num_classes, forced_topk, capture)
