"""The single-layer decoder parity case: ONE DetrTransformerDecoderLayer step of the DINO decoder
(reference codetr/transformer.py:193-230 + the layer :233-277) on 900 queries, BASELINE's 1920x1280 pyramid, 4-d reference
points and a padded memory, with the layer's own inputs taken from the fp32 oracle -- so the product's per-layer kernel
(csrc/decoder_layer.hip) is compared phase by phase without the six-layer refinement chain amplifying anything.

Shared by tests/golden/make_decoder_layer_fixture.py (build container, CPU: runs oracle/codetr_fp32.decoder and writes the
fixture) and tests/test_decoder_layer_oracle_gpu.py (GPU: rebuilds the same seeded weights / memory and checks the product).
Fixtures store seeds and the oracle's rows, not weights."""
import os

import numpy as np
import torch

SHAPES = [(320, 480), (160, 240), (80, 120), (40, 60), (20, 30)]     # 1920x1280: strides 4 .. 64
NQ, C, LAYERS, LAYER = 900, 256, 3, 1                                 # the layer under test: index 1 of a 3-layer decoder
FIXTURE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "decoder_layer_1920x1280.npz")
SEED = 20260504


def build_decoder(layers=LAYERS, seed=SEED, levels=5):
    """(DinoTransformerDecoder, reg branches) on the CPU in fp32 whose values are fp16-representable (the product runs
    them as fp16, the oracle as the same numbers in fp32); trained-like scales: every branch contributes, nothing saturates"""
    from codetr.transformer import DinoTransformerDecoder, build_MLP

    torch.manual_seed(seed)
    cfg = dict(type="DetrTransformerDecoderLayer",
               attn_cfgs=[dict(type="MultiheadAttention", embed_dims=C, num_heads=8, dropout=0.0),
                          dict(type="MultiScaleDeformableAttention", embed_dims=C, num_levels=levels, dropout=0.0)],
               feedforward_channels=2048, ffn_dropout=0.0,
               operation_order=("self_attn", "norm", "cross_attn", "norm", "ffn", "norm"))
    dec = DinoTransformerDecoder(return_intermediate=True, transformerlayers=cfg, num_layers=layers)
    reg = torch.nn.ModuleList(build_MLP(C, C, 4, 3) for _ in range(layers))
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
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
        for p in list(dec.parameters()) + list(reg.parameters()):
            p.copy_(p.half().float())
    return dec.eval(), reg.eval()


def state_dict(dec, reg):
    """keys as oracle/codetr_fp32.decoder reads them: prefix `dec` for the decoder, `reg.<i>` for the reg branches"""
    sd = {"dec." + k: v.detach().float() for k, v in dec.state_dict().items()}
    sd.update({"reg." + k: v.detach().float() for k, v in reg.state_dict().items()})
    return sd


def memory_and_masks(seed=SEED):
    """memory [1, S, 256] (fp16-representable), padding mask [1, S] (the last 10 % of rows and columns of every level),
    valid ratios [1, L, 2] fp32, level shapes / start indices"""
    g = torch.Generator().manual_seed(seed + 2)
    # a SMOOTH random field per level (a coarse grid of 1/8 the resolution, bilinearly upsampled, + 5 % white noise), like a
    # feature map: with white-noise memory a sampling position that moves by the fp16 rounding of an offset (0.2-0.5 pixel
    # for boxes that span the image) changes the sampled value by O(1), and the comparison measures that instead of the kernel
    levels = []
    for h, w in SHAPES:
        coarse = torch.randn(1, C, h // 8 + 2, w // 8 + 2, generator=g)
        fine = torch.nn.functional.interpolate(coarse, size=(h, w), mode="bilinear", align_corners=True)
        fine = fine + 0.05 * torch.randn(1, C, h, w, generator=g)
        levels.append(fine.flatten(2).transpose(1, 2))
    memory = torch.cat(levels, 1).contiguous().half().float()
    masks, vr = [], []
    for h, w in SHAPES:
        vh, vw = int(round(0.9 * h)), int(round(0.9 * w))
        m = torch.ones(h, w, dtype=torch.bool)
        m[:vh, :vw] = False
        masks.append(m.flatten())
        vr.append([vw / w, vh / h])
    pad = torch.cat(masks)[None]
    ss = torch.tensor(SHAPES, dtype=torch.long)
    start = torch.cat((ss.new_zeros(1), ss.prod(1).cumsum(0)[:-1]))
    return memory, pad, torch.tensor([vr], dtype=torch.float32), ss, start


def first_inputs(seed=SEED):
    """what the decoder starts from (query embedding, un-activated 4-d reference boxes): seeded"""
    g = torch.Generator().manual_seed(seed + 3)
    query = torch.randn(1, NQ, C, generator=g).half().float()
    ref = (torch.randn(1, NQ, 4, generator=g) * 1.5).half().float()
    return query, ref


def load_fixture():
    return {k: v for k, v in np.load(FIXTURE).items()}
