"""FP8 (OCP e4m3) inference mode -- BASELINE config 5, "fp8 weights + activations (CDNA4 fp8 MFMA for Swin + transformer
GEMMs)".  No reference counterpart (the reference's dtypes stop at half: codetr/csrc/ms_deform_attn.cu:946,
export.py:39-44); parity target = the fp32 oracle at a wider, stated tolerance (tests/test_fp8_gpu.py).

Scheme (standard post-training static quantisation):
  * weights: per-output-channel scale = absmax / 448, quantised once (``hip_ops.fp8_weight``, cached on the parameter);
  * activations: one static scale per GEMM input tensor = (running absmax over a calibration forward) / 448 x margin;
    the producer of each GEMM input emits e4m3 directly -- LayerNorm (``layer_norm_fp8``), the GELU epilogue of fc1
    (``out_scale``), the window-attention kernel's output store (``swin_window_attention(..., out_scale=)``);
  * arithmetic: e4m3 x e4m3 on v_mfma_scale_f32_16x16x128_f8f6f4 (twice the fp16 MFMA rate), fp32 accumulation, scales
    applied once in the epilogue; residual stream, attention, norms' statistics and everything else stay fp16 / fp32.

What runs in fp8: the four Linears of every Swin block whose K is a multiple of 128 and whose GEMMs fill the chip
(stages 1-3 at the bench's batch: 88 of the backbone's 96 block GEMMs, ~85 % of the Swin flops), and BOTH products of
the deformable encoder's fused FFN (``codetr_ffn_fp8``: the LayerNorm'ed input is quantised in registers, the hidden
activation is quantised as it leaves the first product's accumulators and never exists in fp16).  Stage 0 (C = 192),
patch merging, the neck, the attention projections, the decoder and the heads stay on the fp16 kernels; ``report()``
says so."""
import torch

from . import hip_ops
from .swin import SwinBlock
from .transformer_layers import FFN

MARGIN = 2.0   # static mode: scale = absmax * MARGIN / 448.  e4m3 is a floating-point format: one binade of headroom costs no
               # relative precision (only the smallest binade of the activations moves into the subnormals) and keeps inputs
               # up to twice the calibration set's maximum from saturating at +-448
# "mx" (default): MX block scales on the Swin activations -- one e8m0 exponent per 32 channels, chosen by the producer
# kernel from the block's own maximum and applied by the scaled MFMA in hardware: no calibration, no saturation on unseen
# inputs.  "static": the round-2 scheme (one calibrated scale per tensor), kept for A/B.  The encoder's fused FFN uses
# static scales in both modes (its calibration is the only thing `calibrate` still has to do in "mx" mode).
MODE = "mx"


def _blocks(model):
    return [m for m in model.modules() if isinstance(m, SwinBlock)]


def _ffns(model):
    return [m for m in model.modules() if isinstance(m, FFN)]


@torch.no_grad()
def calibrate(model, batch_inputs, img_masks):
    """One fp16 forward with every Swin block recording the absolute maxima of its four GEMM inputs; sets the static
    activation scales and pre-quantises the weights.  Returns the number of blocks prepared."""
    blocks, ffns = _blocks(model), _ffns(model)
    for b in blocks + ffns:
        b.fp8_mode = "calibrate"
        b.__dict__.pop("_fp8_amax", None)
    try:
        model(batch_inputs, img_masks)
    finally:
        for b in blocks + ffns:
            b.fp8_mode = None
    n = 0
    for f in ffns:      # only the FFNs the fused kernel served observed anything (the encoder's, at the bench's sizes)
        amax = f.__dict__.get("_fp8_amax")
        if amax:
            f._fp8_scales = {k: max(float(v) * MARGIN / hip_ops.FP8_MAX, 1e-8) for k, v in amax.items()}
            hip_ops.ffn_fp8_weights(f.layers[0][0].weight, f.layers[1].weight)
    for b in blocks:
        amax = b.__dict__.get("_fp8_amax")
        if not amax:
            continue
        b._fp8_scales = {k: max(float(v) * MARGIN / hip_ops.FP8_MAX, 1e-8) for k, v in amax.items()}   # (one host sync each)
        for w in b._fp8_weights():
            if w.shape[1] % 128 == 0:
                hip_ops.fp8_weight(w)
        n += 1
    return n


@torch.no_grad()
def saturation(model, batch_inputs, img_masks):
    """How close these inputs come to the calibrated static ranges: one fp16 forward in recording mode (the fp8 path is
    switched back to what it was afterwards) and, per recorded tensor, its absolute maximum over what the static scale
    represents without clamping (scale * 448).  Returns {"tensors": n, "saturating": number of tensors with a ratio > 1,
    "worst_ratio": max}.  MX-scaled tensors cannot saturate and are not counted."""
    blocks, ffns = _blocks(model), _ffns(model)
    saved = [(b, getattr(b, "fp8_mode", None)) for b in blocks + ffns]
    for b in blocks + ffns:
        b.fp8_mode = "calibrate"
        b.__dict__.pop("_fp8_amax", None)
    try:
        model(batch_inputs, img_masks)
    finally:
        for b, m in saved:
            b.fp8_mode = m
    n = sat = 0
    worst = 0.0
    for b in blocks + ffns:
        amax, scales = b.__dict__.pop("_fp8_amax", None), getattr(b, "_fp8_scales", None)
        if not amax or not scales:
            continue
        for k, v in amax.items():
            if k in scales:
                r = float(v) / (scales[k] * hip_ops.FP8_MAX)
                n += 1
                sat += r > 1.0
                worst = max(worst, r)
    return {"tensors": n, "saturating": int(sat), "worst_ratio": round(worst, 4)}


# The e4m3 GEMMs that ship: the selection of the sensitivity map (tools/fp8_sensitivity.py, profiles/r04_fp8_sensitivity.json).
# {stage: ops}; stage = log2(C / 192) of the Swin block (stage 0 has K = 192, not a multiple of 128: never fp8);
# None = every GEMM the kernels take (the round-3 behaviour, kept as `select="all"`).
DEFAULT_SELECT = None
DEFAULT_FFN = True
# What the map says (8 proxy images, every GEMM group switched to e4m3 alone; memory error against the fp16 product):
# stage 3 (2 blocks) 0.8-1.3e-2 per GEMM, stage 1 (2 blocks) 1.0-1.7e-2, stage 2 (18 blocks) 1.6-2.3e-2, the encoder FFN
# 6.1e-2; the errors add in quadrature (all: 7.8e-2).  "accurate" is the largest selection under 2e-2 -- and it is 1.6 of
# the model's 16.8 TFLOP: +1 % images/s.  e4m3's 3-bit mantissa cannot carry this model's GEMMs at that bound, with or
# without block scales; the full selection stays available as a FAST mode and is reported as such (not recommended where
# the accuracy bar of the fp16 path applies).
PRESETS = {"all": (None, True), "accurate": ({3: ("qkv", "fc1", "fc2"), 1: ("fc2",)}, False)}


def block_stage(block):
    """Swin stage of a block from its width: C = 192 * 2^stage in Swin-L (embed_dims of the config otherwise)"""
    c = block.attn.w_msa.qkv.weight.shape[1]
    s, base = 0, getattr(block, "_stage_base", 192)
    while base * 2 <= c:
        base *= 2
        s += 1
    return s


def enable(model, on=True, mode=None, select="default", ffn="default"):
    """switch the blocks to the fp8 path: Swin blocks to MX block scales (mode "mx", no calibration needed) or to their
    calibrated static scales (mode "static"); calibrated FFNs to the fused e4m3 kernel.  Blocks whose shapes the fp8
    GEMM does not take at run time keep running fp16.
    select (mode "mx"): which GEMMs run in e4m3 -- "default" = DEFAULT_SELECT, "all" / None = all four of every block,
    or {stage: iterable of "qkv" | "proj" | "fc1" | "fc2"} (stages not named stay fp16); ffn: the encoder FFNs too."""
    mode = mode or MODE
    if isinstance(select, str) and select in PRESETS and select != "all":
        select, preset_ffn = PRESETS[select]
        ffn = preset_ffn if ffn == "default" else ffn
    sel = DEFAULT_SELECT if select == "default" else (None if select == "all" else select)
    use_ffn = DEFAULT_FFN if ffn == "default" else bool(ffn)
    for b in _blocks(model):
        b.fp8_ops = None
        if not on:
            b.fp8_mode = None
        elif mode == "mx":
            if sel is None:
                b.fp8_mode = "mx"
            else:
                ops = tuple(o for o in ("qkv", "proj", "fc1", "fc2") if o in tuple(sel.get(block_stage(b), ())))
                b.fp8_mode = "mx" if ops else None
                b.fp8_ops = ops or None
        else:
            b.fp8_mode = "run" if hasattr(b, "_fp8_scales") else None
    for f in _ffns(model):
        f.fp8_mode = "run" if (on and use_ffn and hasattr(f, "_fp8_scales")) else None


def report(model):
    blocks = _blocks(model)
    ready = [b for b in blocks if getattr(b, "fp8_mode", None) in ("run", "mx")]
    k_ok = [b for b in ready if all(w.shape[1] % 128 == 0 for w in b._fp8_weights())]
    gemms = sum(len(getattr(b, "fp8_ops", None) or ("qkv", "proj", "fc1", "fc2")) for b in k_ok)
    ffns = _ffns(model)
    return {"swin_blocks": len(blocks), "swin_blocks_fp8": len(k_ok), "swin_gemms_fp8": gemms,
            "activation_scales": "MX blocks (e8m0 per 32 channels, dynamic)" if any(getattr(b, "fp8_mode", None) == "mx" for b in blocks)
            else "static per tensor (calibrated)",
            "ffns": len(ffns), "ffns_fp8": sum(1 for f in ffns if f.fp8_mode == "run"),
            "fp8_layers": f"qkv / proj / fc1 / fc2 of Swin blocks with K a multiple of 128 (stages 1-3) when the GEMM has "
                          f">= {hip_ops.FP8_MIN_TILES} 256x256 tiles; both products of the encoder's fused FFN "
                          f"(rows >= {hip_ops.FFN_FUSED_MIN_ROWS})",
            "fp16_layers": "Swin stage 0, patch merging, neck, attention projections, decoder, heads, window attention, MSDA"}
