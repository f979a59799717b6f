#!/usr/bin/env python3
"""bench.py -- Co-DINO Swin-L 1920x1280 fp16 inference throughput on N MI355X (one process per GPU).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B_per_gpu] [--res WxH]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

``python bench.py --gpus N`` with N > 1 and no launcher environment starts the N ranks itself: before anything in this
process touches the GPU it runs ``python -m torch.distributed.run --nproc-per-node N bench.py ...`` as a CHILD process
(never an exec), relays rank 0's JSON line and exits with the child's code.

A "step" is one pass of the hot path (``CoDETR.forward``: Swin-L -> ChannelMapper -> CoDINOHead incl.
12 MSDA HIP launches per image) over one batch of synthetic images already resident in HBM, random-init
weights of the real architecture (no network: no checkpoint, no COCO).  Images shard across ranks
(independent replicas, weights replicated; default 8 images per GPU per step = BASELINE config 4, batch 64 over
8 GPUs); the only collective is one all_gather of the final detections [B,300,6] per step over RCCL.  Timing: W warm-up steps, then exactly K steps between
barrier + torch.cuda.synchronize() on both sides, MAX over ranks.  Rank 0 prints ONE JSON line.

Weights: seeded default init, except that every MSDA ``sampling_offsets.weight`` (zero in the default init, i.e. every
query would sample the fixed bias grid -- the best case for the LDS-staged encoder kernel) is drawn so that the sampling
offsets carry ``--offset-noise-px`` (default 2) pixels of query-dependent spread on top of the bias grid.  2 px is an
ASSUMPTION, not a measurement of a trained checkpoint (none is available offline; the offsets of reference
codetr/multi_scale_deformable_attention.py:186-191 are in level pixels and unbounded): the same kernel is therefore also
measured at 0, 4 and 8 px (``roofline_msda_zero_noise`` / ``_4px`` / ``_8px``, with the share of samples that leave the
staged windows), and the whole model at 8 px (``value_8px``).

Extra objects on that line: ``host_feed`` -- the same K steps with every step's images copied from pinned host memory
(PCIe) on the sub-batch streams, i.e. the rate including the input transfer (never ``value``); and at N=1 only:
``latency_batch1`` -- p50 / p90 of single-image forwards (the quantity the
reference publishes); ``roofline`` -- the dominant kernel group of the forward, the hand-written
MFMA linears, every launch of one forward timed live with HIP events on its launch stream and priced with
2*M*N*K flops against the 2.5 PF dense fp16 MFMA peak (``composite``: the same launches against
max(flops / 2.5 PF, bytes / 8 TB/s) per launch -- the short-K layers are HBM-bound); ``roofline_ffn`` -- the fused encoder FFN kernel, same
peak; ``roofline_msda`` -- the fused MSDA gather kernel as the model launches it at the encoder shape, priced with
the algorithmic bytes of BASELINE.md section 3 against 8 TB/s (``roofline_msda_op``: the same for the stand-alone
operator ``torch.ops.codetr.multi_scale_deformable_attention`` on synthetic sampling locations); ``traffic`` in each
= HBM-side bytes per launch from the committed PMC passes (profiles/*_pmc_traffic.json); ``cpu_baseline`` -- the
fp32 CPU oracle (oracle/codetr_fp32.py + the C MSDA restatement) timed on the host cores on a bounded sample.

``--dry-run`` (CPU, gloo): the launcher, sharding, gather, timing and JSON assembly with stand-in detections instead of
the model -- what tests/test_bench_launch_cpu.py runs with 2 processes; ``value`` is null there.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (os.path.join(ROOT, "co-detr-tensorrt_amd"), os.path.join(ROOT, "oracle")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

CFG = os.path.join(ROOT, "co-detr-tensorrt_amd", "configs", "co_dino_5scale_swin_l_16xb1_16e_o365tococo.py")
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured streaming copy)
MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense fp16/bf16 MFMA peak (~2.5 PF, not the 2:1-sparse 5 PF)


def pyramid(h, w):
    c = lambda a, b: -(-a // b)  # noqa: E731
    lv = [(c(h, 4), c(w, 4)), (c(h, 8), c(w, 8)), (c(h, 16), c(w, 16)), (c(h, 32), c(w, 32))]
    lv.append((c(lv[-1][0], 2), c(lv[-1][1], 2)))
    return lv


def msda_algorithmic_bytes(B, S, Nq, M=8, D=32, L=5, P=4, e=2):
    """BASELINE.md section 3 / SURVEY.md 8(d): value + loc(2) + weight(1) + output, each touched once."""
    return e * (B * S * M * D + 3 * B * Nq * M * L * P + B * Nq * M * D) + 24 * L


def build_model(device, dtype, seed=42, offset_noise_px=2.0):
    """Seeded random-init weights of the real architecture.  offset_noise_px > 0: every MSDA sampling_offsets.weight
    (zero in the default init, reference multi_scale_deformable_attention.py:90-115) is drawn N(0, s^2) with s chosen
    so that offsets = W q + b spread by about that many pixels around the bias grid for unit-variance-ish queries
    (the measured spread is reported by `msda_offset_stats`)."""
    import codetr

    torch.manual_seed(seed)
    model = codetr.build_CoDETR(CFG, None, "cpu")
    model.init_weights()  # seeded default init of every sub-module (random weights of the real architecture)
    set_offset_noise(model, offset_noise_px, seed)
    return model.to(device=device, dtype=dtype).eval()


def set_offset_noise(model, px, seed=42):
    """(re)draw the sampling_offsets weights of every MSDA module for `px` pixels of query-dependent spread (0: the
    reference's default init, all zeros).  The encoder's query + pos has mean square ~2.6 per channel on these weights
    (LayerNorm output + sine encoding + level embedding); the decoder's ~2."""
    from codetr.multi_scale_deformable_attention import MultiScaleDeformableAttention

    g = torch.Generator().manual_seed(seed + 7)
    for m in model.modules():
        if isinstance(m, MultiScaleDeformableAttention):
            w = m.sampling_offsets.weight
            with torch.no_grad():
                if px > 0:
                    std = px / (w.shape[1] * 2.6) ** 0.5
                    w.copy_((torch.randn(w.shape, generator=g) * std).to(device=w.device, dtype=w.dtype))
                else:
                    w.zero_()


def msda_offset_stats(model, images, masks, halo=4):
    """One eager forward with a hook on the encoder's (offsets | logits) projections: the spread of the sampling
    offsets around their bias grid (pixels of the sampled level, std over queries / heads / points) and the fraction
    of sample points farther than `halo` pixels from the query's own location on that level -- the ones the LDS-staged
    encoder kernel has to fetch from global memory instead of its staged neighbourhood."""
    from codetr import hip_ops
    from codetr.multi_scale_deformable_attention import MultiScaleDeformableAttention

    enc_atts = [layer.attentions[0] for layer in model.query_head.transformer.encoder.layers]
    stats = []
    orig_linear, orig_xadd = hip_ops.linear, hip_ops.linear_xadd
    pyramid_shapes = pyramid(int(images.shape[-2]), int(images.shape[-1])) if enc_atts and enc_atts[0].num_levels == 5 else None

    def record(att, proj, packed):
        n_off = att.num_heads * att.num_levels * att.num_points * 2
        if packed:   # lane-major packed projection (round-5 encoder kernel): back to the reference's column order
            from codetr import _cabi as cabi
            idx = torch.tensor(cabi.msda_pack_projection_index(att.num_heads, att.num_levels, att.num_points), device=proj.device)
            std = proj.new_zeros(*proj.shape[:-1], n_off + n_off // 2)
            std[..., idx[idx >= 0]] = proj[..., (idx >= 0).nonzero().flatten()]
            proj = std
        off = proj[..., :n_off].float()
        bias = att.sampling_offsets.bias.float()
        spread = (off - bias).std().item()
        miss = (off.abs().view(*off.shape[:-1], -1, 2).amax(-1) > halo).float().mean().item()
        # ... and the windows the encoder kernel actually stages (per head and level): a sample is in when both bilinear
        # corners of floor(offset) are (exact for level-0 queries, whose centre sits on the level's grid; an estimate for
        # the coarser levels' sub-pixel phases)
        wmiss = None
        if pyramid_shapes is not None:
            wl = att._encoder_windows_packed(pyramid_shapes)
            win = torch.tensor(wl, dtype=torch.float32, device=off.device)             # [M, L, 4] lox, hix, loy, hiy
            o = off.view(*off.shape[:-1], att.num_heads, att.num_levels, att.num_points, 2)
            x0, y0 = torch.floor(o[..., 0]), torch.floor(o[..., 1])
            w = win[:, :, None, :]
            inside = (x0 >= w[..., 0]) & (x0 + 1 <= w[..., 1]) & (y0 >= w[..., 2]) & (y0 + 1 <= w[..., 3])
            wmiss = 1.0 - inside.float().mean().item()
        stats.append((spread, miss, wmiss))

    def find(weight):
        for a in enc_atts:
            if a._fused_projection()[0] is weight:
                return a, False
            if hip_ops.msda_encoder_packed_supported(torch.float16, 32, a.num_levels, a.num_points) and \
                    a._packed_projection()[0] is weight:
                return a, True
        return None, False

    def linear(x, weight, *args, **kw):
        y = orig_linear(x, weight, *args, **kw)
        a, packed = find(weight)
        if a is not None and y.shape[-1] == weight.shape[0] and y.dim() == 3 and y.shape[1] > 10000:
            record(a, y, packed)
        return y

    def linear_xadd(x, x_add, weight, bias=None):
        y = orig_xadd(x, x_add, weight, bias)
        a, packed = find(weight)
        if a is not None:
            record(a, y, packed)
        return y

    hip_ops.linear, hip_ops.linear_xadd = linear, linear_xadd
    try:
        with torch.no_grad():
            model(images, masks)
        torch.cuda.synchronize()
    finally:
        hip_ops.linear, hip_ops.linear_xadd = orig_linear, orig_xadd
    if not stats:
        return None
    rec = {"offset_spread_px_per_encoder_layer": [round(a, 2) for a, _, _ in stats],
           "fraction_of_samples_outside_halo_per_layer": [round(b, 4) for _, b, _ in stats],
           "halo_px": halo}
    if all(c is not None for _, _, c in stats):
        rec["fraction_of_samples_outside_staged_windows_per_layer"] = [round(c, 4) for _, _, c in stats]
        rec["windows_note"] = ("the encoder kernel stages per-(head, level) windows derived from the offset bias (three "
                               "passes, clamped to the level + a zero border), not the symmetric halo: this is the share "
                               "that takes its fix-up queue")
    return rec


def _msda_op_pmc():
    """HBM-side bytes per launch of the public op's kernels from the committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes
    (profiles/rNN_msda_op_pmc.json, newest round first; made by tools/pmc_msda_op.sh + tools/fold_msda_op_pmc.py on
    tools/bench_msda_op.py --pmc: the same shapes and inputs as below; reads corrected x2 as MI355X_MICROARCH.md prescribes for
    gfx950).  Counters cannot be collected from inside this process."""
    import glob

    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_msda_op_pmc.json")), reverse=True):
        try:
            with open(path) as f:
                d = json.load(f)
            d["_file"] = os.path.relpath(path, ROOT)
            return d
        except (OSError, ValueError):
            continue
    return {}


def msda_roofline(B, H, W, dtype, device, iters=30):
    """Time the PUBLIC op torch.ops.codetr.multi_scale_deformable_attention alone with HIP events on the launch stream:
    at the encoder shape (Nq = S: the windowed kernel of csrc/msda_op4.hip + the general kernel's skipped launch behind it)
    and at the decoder shape (Nq = 900: the general kernel).  Returns (encoder record, decoder record)."""
    from codetr import _cabi

    shapes = pyramid(H, W)
    ss = torch.tensor(shapes, dtype=torch.int64, device=device)
    ls = torch.cat((ss.new_zeros(1), ss.prod(1).cumsum(0)[:-1]))
    S = int(ss.prod(1).sum())
    M, D, L, P = 8, 32, 5, 4
    g = torch.Generator(device=device).manual_seed(0)
    value = torch.randn(B, S, M, D, device=device, generator=g).to(dtype)
    refs = []
    for (h, w) in shapes:  # encoder-like sampling: own pixel centre + offsets of a few pixels
        ys, xs = torch.meshgrid(torch.linspace(0.5, h - 0.5, h, device=device) / h,
                                torch.linspace(0.5, w - 0.5, w, device=device) / w, indexing="ij")
        refs.append(torch.stack((xs.reshape(-1), ys.reshape(-1)), -1))
    ref = torch.cat(refs)[None, :, None, None, None, :]
    norm = torch.stack((ss[:, 1], ss[:, 0]), -1).float()[None, None, None, :, None, :]
    op = torch.ops.codetr.multi_scale_deformable_attention
    st = torch.cuda.current_stream(device)
    pmc = _msda_op_pmc()

    def timed(loc, w):
        # warm-up by TIME, not by count: the op is timed behind tensor set-up during which the GPU sits nearly idle, and a
        # latency-bound launch reads 7-8 % longer in that clock state than right behind sustained work
        # (tools/micro/op_after_burn.py: 359 us cold, 335 behind a 20 s GEMM burn, 363 again after 5 s of idling)
        t0 = time.time()
        while time.time() - t0 < 0.25:
            for _ in range(50):
                op(value, ss, ls, loc, w, 64)
            torch.cuda.synchronize(device)
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
        for a, b in evs:
            a.record(st)
            op(value, ss, ls, loc, w, 64)
            b.record(st)
        torch.cuda.synchronize(device)
        ts = sorted(a.elapsed_time(b) for a, b in evs)
        ts = ts[len(ts) // 10: len(ts) - len(ts) // 10]   # (a host-side hiccup -- an allocator call between two launches -- is not the kernel)
        return sum(ts) / len(ts) * 1e-3

    tname = "BF16" if dtype == torch.bfloat16 else "F16"
    # ---- encoder shape ----
    loc = (ref + torch.randn(B, S, M, L, P, 2, device=device, generator=g) * 3.0 / norm).to(dtype).contiguous()
    w = torch.rand(B, S, M, L, P, device=device, generator=g)
    w = (w / w.sum((-1, -2), keepdim=True)).to(dtype).contiguous()
    windowed = bool(_cabi.load().codetr_msda_op4_supported(value.element_size(), B, S, M, D, L, S, P)) and dtype == torch.float16
    avg_s = timed(loc, w)
    nbytes = msda_algorithmic_bytes(B, S, S, e=value.element_size())
    achieved = nbytes / avg_s / 1e9
    enc = {
        "kernel": ("msda_op4_kernel (windowed: per-(region, head) LDS windows + zero border, locations / weights through LDS "
                   "records, fp32 blend, fix-up path) + msda_tiled_kernel<%s,4>'s skipped launch behind it" % tname if windowed
                   else "msda_tiled_kernel<%s,4>" % tname) + " (encoder call, Nq=S=%d, batch %d, 3 px of spread)" % (S, B),
        "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 4),
        "traffic": pmc.get("encoder", {}).get("hbm_bytes_per_launch") if windowed else None,
        "traffic_source": pmc.get("_file") if windowed else None,
        "algorithmic_bytes_per_launch": nbytes, "avg_launch_us": round(avg_s * 1e6, 1),
    }
    del loc, w
    # ---- decoder shape: 900 queries anywhere in the image, 10 px of spread around each query's point ----
    Nq = 900
    refq = torch.rand(B, Nq, 1, 1, 1, 2, device=device, generator=g) * 0.8 + 0.1
    loc = (refq + torch.randn(B, Nq, M, L, P, 2, device=device, generator=g) * 10.0 / norm).to(dtype).contiguous()
    w = torch.softmax(torch.randn(B, Nq, M, L * P, device=device, generator=g), -1).view(B, Nq, M, L, P).to(dtype).contiguous()
    avg_s = timed(loc, w)
    e = value.element_size()
    baseline_bytes = msda_algorithmic_bytes(B, S, Nq, e=e)                      # the whole value map read once
    touched = e * B * (Nq * M * L * P * 4 * D + 3 * Nq * M * L * P + Nq * M * D)   # at most 4 corner rows per sample are ever read
    nbytes = min(baseline_bytes, touched)
    achieved = nbytes / avg_s / 1e9
    dec = {
        "kernel": "msda_tiled_kernel<%s,4> (decoder call, Nq=%d, S=%d, batch %d)" % (tname, Nq, S, B),
        "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 4),
        "traffic": pmc.get("decoder", {}).get("hbm_bytes_per_launch"), "traffic_source": pmc.get("_file"),
        "algorithmic_bytes_per_launch": nbytes, "avg_launch_us": round(avg_s * 1e6, 1),
        "note": "BASELINE.md section 3 prices this call at %d bytes (the whole value map read once); 900 queries x 8 heads x 20 "
                "samples x 4 corner rows can touch at most %d bytes of it, which is what `achieved` uses -- a launch of 113 "
                "workgroups on 256 CUs is bound by latency, not by bandwidth" % (baseline_bytes, touched),
    }
    return enc, dec


def _pmc_traffic():
    """HBM-side bytes per launch from the committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this same
    workload (profiles/rNN_pmc_traffic.json, newest round first; made by tools/pmc_traffic.py; reads corrected x2 as
    MI355X_MICROARCH.md prescribes for gfx950).  Counters cannot be collected from inside this process."""
    import glob

    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")), reverse=True):
        try:
            with open(path) as f:
                d = json.load(f)["kernels"]
            d["_file"] = os.path.relpath(path, ROOT)
            return d
        except (OSError, KeyError, ValueError):
            continue
    return {}


def kernel_rooflines(model, images, masks, device):
    """One eager forward with every launch of the three heavy hand-written kernels bracketed by HIP events on its
    launch stream (hip_ops.LINEAR_PROFILE / KERNEL_PROFILE):
      roofline       linear_kernel, the dominant kernel (~40 % of the GPU time of a forward): sum of 2*M*N*K over its
                     launches / sum of their durations, against the dense fp16 MFMA peak
      roofline_ffn   ffn_fused_kernel (encoder FFN, 4*M*256*2048 flops per launch -- + 2*M*256*256 where the launch also does
                     the attention output projection, round 5), same peak
      roofline_msda  the fused MSDA gather kernel at the encoder shape (Nq = S), algorithmic bytes of BASELINE.md
                     section 3 (value + offsets + logits + output, each once) against the 8 TB/s HBM peak."""
    from codetr import hip_ops

    hip_ops.LINEAR_PROFILE, hip_ops.KERNEL_PROFILE = [], {}
    try:
        with torch.no_grad():
            model(images, masks)
        torch.cuda.synchronize(device)
        prof, kprof = hip_ops.LINEAR_PROFILE, hip_ops.KERNEL_PROFILE
    finally:
        hip_ops.LINEAR_PROFILE = hip_ops.KERNEL_PROFILE = None
    pmc = _pmc_traffic()
    out = {}
    prof8 = [p for p in prof if len(p) > 6]
    prof = [p for p in prof if len(p) == 6]
    if prof8:
        f8 = sum(p[2] for p in prof8)
        t8 = sum(p[0].elapsed_time(p[1]) for p in prof8) * 1e-3
        out["roofline_fp8"] = {
            "kernel": "linear_256_fp8_kernel (the %d e4m3 GEMM launches of one forward)" % len(prof8),
            "bound": "mfma", "achieved": round(f8 / t8 / 1e12, 1), "peak": 2 * MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(f8 / t8 / 1e12 / (2 * MFMA_PEAK_TFLOPS), 4), "traffic": None,
            "sum_launch_ms": round(t8 * 1e3, 3), "algorithmic_flops_per_forward": f8,
            "peak_note": "dense fp8 MFMA peak ~5 PF (block-scaled K = 128 instruction)"}
    flops = sum(p[2] for p in prof)
    secs = sum(p[0].elapsed_time(p[1]) for p in prof) * 1e-3
    achieved = flops / secs / 1e12
    big = max(prof, key=lambda p: p[0].elapsed_time(p[1]))
    # per-launch composite bound: a launch cannot finish sooner than its flops at the MFMA peak NOR than its
    # algorithmic bytes (X + W + Y once, 2 B each) at the HBM peak -- the K <= 384 layers sit under the HBM roof
    bound_s = sum(max(p[2] / (MFMA_PEAK_TFLOPS * 1e12), 2.0 * (p[3] * p[5] + p[4] * p[5] + p[3] * p[4]) / (HBM_PEAK_GBS * 1e9))
                  for p in prof)
    hbm_bound_launches = sum(1 for p in prof if 2.0 * (p[3] * p[5] + p[4] * p[5] + p[3] * p[4]) / (HBM_PEAK_GBS * 1e9)
                             > p[2] / (MFMA_PEAK_TFLOPS * 1e12))
    alg_bytes = sum(2.0 * (p[3] * p[5] + p[4] * p[5] + p[3] * p[4]) for p in prof)   # X + W + Y once per launch, 2 B each
    # per-launch figure from the PMC pass's PER-FORWARD bytes over THIS run's launch count: the committed pass may have
    # been taken with a different launch list (round 4 moved the decoder's 72 small Linears into decoder_layer_kernel)
    pl = pmc.get("linear_kernel", {})
    traffic = pl["hbm_bytes_per_forward"] / len(prof) if "hbm_bytes_per_forward" in pl else pl.get("hbm_bytes_per_launch")
    out["roofline"] = {
        "kernel": "linear_pp_kernel / linear_sk_kernel / linear_kernel / linear_256_kernel / linear_xs_kernel (16-bit operands of the run's "
                  "dtype, fp32 accumulation; all %d launches of one forward)" % len(prof),
        "bound": "mfma", "achieved": round(achieved, 1), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
        "frac": round(achieved / MFMA_PEAK_TFLOPS, 4),
        "traffic": traffic,
        "traffic_note": "HBM-side bytes of the group per forward / this run's launches (committed PMC pass %s)" % pmc.get("_file"),
        "algorithmic_bytes_per_forward": alg_bytes, "algorithmic_bytes_per_launch": round(alg_bytes / len(prof)),
        "traffic_over_algorithmic": round(traffic * len(prof) / alg_bytes, 3) if traffic else None,
        "composite": {"frac": round(bound_s / secs, 4), "bound_ms": round(bound_s * 1e3, 3),
                      "hbm_bound_launches": hbm_bound_launches,
                      "note": "sum over launches of max(flops / 2.5 PF, (M*K + N*K + M*N) * 2 B / 8 TB/s) / sum of "
                              "measured durations"},
        "algorithmic_flops_per_forward": flops, "sum_launch_ms": round(secs * 1e3, 3),
        "avg_launch_us": round(secs / len(prof) * 1e6, 1),
        "longest_launch": {"M": big[3], "N": big[4], "K": big[5], "us": round(big[0].elapsed_time(big[1]) * 1e3, 1),
                           "TFLOP/s": round(big[2] / (big[0].elapsed_time(big[1]) * 1e-3) / 1e12, 1)},
    }
    ffn = kprof.get("ffn_fused", [])
    if ffn:
        fl = sum(4.0 * m["M"] * m["C"] * m["hidden"] + (2.0 * m["M"] * m["C"] * m["C"] if m.get("oproj") else 0.0)
                 for _, _, m in ffn)
        t = sum(a.elapsed_time(b) for a, b, _ in ffn) * 1e-3
        out["roofline_ffn"] = {
            "kernel": "ffn_fused_kernel<2> (%d encoder launches, M = %d%s)" % (
                len(ffn), ffn[0][2]["M"],
                "; the attention output projection + identity + norm1 inside: 2*M*256*256 flops more per launch"
                if ffn[0][2].get("oproj") else ""),
            "bound": "mfma", "achieved": round(fl / t / 1e12, 1), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(fl / t / 1e12 / MFMA_PEAK_TFLOPS, 4),
            "traffic": pmc.get("ffn_fused", {}).get("hbm_bytes_per_launch"),
            "avg_launch_us": round(t / len(ffn) * 1e6, 1),
        }
    smlp = kprof.get("swin_mlp", [])
    if smlp:
        # round 6: the one-launch MLPs of Swin stages 0 / 1 (norm2, fc1, GELU, fc2, identity): 2 * M * C * 4C * 2 flops each
        fl = sum(16.0 * m["M"] * m["C"] * m["C"] for _, _, m in smlp)
        t = sum(a.elapsed_time(b) for a, b, _ in smlp) * 1e-3
        out["roofline_swin_mlp"] = {
            "kernel": "swin_mlp_kernel (%d launches: C = %s)" % (len(smlp), ", ".join(str(m["C"]) for _, _, m in smlp)),
            "bound": "mfma", "achieved": round(fl / t / 1e12, 1), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(fl / t / 1e12 / MFMA_PEAK_TFLOPS, 4), "traffic": None,
            "avg_launch_us": round(t / len(smlp) * 1e6, 1),
            "note": "the three launches it replaces (norm2 | fc1 + GELU | fc2 + identity) ran at 395 / 580 TF/s, bound by the "
                    "bytes of the hidden activation (profiles/r06_swin_mlp.txt); this kernel is bound by the issue slots of its one "
                    "wave per SIMD (GELU: as many vector-issue cycles as the chunk's MFMAs at C = 192)",
        }
    ffn8 = kprof.get("ffn_fp8", [])
    if ffn8:
        fl = sum(4.0 * m["M"] * m["C"] * m["hidden"] for _, _, m in ffn8)
        t = sum(a.elapsed_time(b) for a, b, _ in ffn8) * 1e-3
        # algorithmic HBM bytes per launch: the rows in, out, + pos in and rows + pos out (2 B each); W is L2-resident
        nbytes = sum(4.0 * m["M"] * m["C"] * 2 for _, _, m in ffn8)
        out["roofline_ffn_fp8"] = {
            "kernel": "ffn_fp8_kernel (%d encoder launches, M = %d: LayerNorm, both products in e4m3, LayerNorm, + pos)"
                      % (len(ffn8), ffn8[0][2]["M"]),
            "bound": "mfma", "achieved": round(fl / t / 1e12, 1), "peak": 2 * MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(fl / t / 1e12 / (2 * MFMA_PEAK_TFLOPS), 4), "traffic": None,
            "avg_launch_us": round(t / len(ffn8) * 1e6, 1),
            "hbm_floor_us": round(nbytes / len(ffn8) / (HBM_PEAK_GBS * 1e9) * 1e6, 1),
            "note": "at the e4m3 rate the kernel's row traffic (x, y, pos, y + pos) is within 2x of its MFMA time: "
                    "the composite floor is max(flops / 5 PF, bytes / 8 TB/s) per launch",
        }
    wa = kprof.get("window_attention", [])
    if wa:
        # HBM-bound: the qkv rows in (3 C halves per token) and the attention output out, each once; the MFMA work
        # (QK^T and PV over the window's 144 keys, head_dim 32) alongside
        nbytes = sum(m["rows"] * m["C"] * (3 * 2 + m["out_bytes"]) for _, _, m in wa)
        fl = sum(4.0 * m["rows"] * m["window"] ** 2 * m["C"] for _, _, m in wa)
        t = sum(a.elapsed_time(b) for a, b, _ in wa) * 1e-3
        out["roofline_window_attention"] = {
            "kernel": "window_attention_kernel (the %d Swin blocks of one forward: shifted-window softmax attention, "
                      "pad / roll / partition inside)" % len(wa),
            "bound": "hbm", "achieved": round(nbytes / t / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(nbytes / t / 1e9 / HBM_PEAK_GBS, 4),
            "traffic": pmc.get("window_attention", {}).get("hbm_bytes_per_launch"),
            "algorithmic_bytes_per_forward": nbytes, "sum_launch_ms": round(t * 1e3, 3),
            "avg_launch_us": round(t / len(wa) * 1e6, 1),
            "mfma": {"achieved_TFLOP/s": round(fl / t / 1e12, 1), "frac": round(fl / t / 1e12 / MFMA_PEAK_TFLOPS, 4)},
        }
    enc = [(a, b, m) for a, b, m in kprof.get("msda_fused", []) if m["Nq"] == m["S"]]
    if enc:
        from codetr import _cabi as _cabi_mod

        enc_native = _cabi_mod.CALLS.get("msda_encoder_packed", 0) > 0   # the LDS-staged encoder kernel served the launches
        m = enc[0][2]
        e = 2
        # value + offsets (2 per point) + logits (1 per point) + output, each touched once
        nbytes = e * m["B"] * (m["S"] * m["M"] * m["D"] + 3 * m["Nq"] * m["M"] * m["L"] * m["P"] + m["Nq"] * m["M"] * m["D"])
        t = sum(a.elapsed_time(b) for a, b, _ in enc) * 1e-3 / len(enc)
        out["roofline_msda"] = {
            "kernel": "%s (the %d encoder launches of one forward, Nq = S = %d)" % (
                "msda_encoder_v4_kernel (lane-major packed projection, head-major value map, scalar geometry, zero-border "
                "windows; packed-half blend, three passes, fp32 reference points)" if enc_native
                else "msda_tiled_kernel<F16,4,fused>", len(enc), m["S"]),
            "bound": "hbm", "achieved": round(nbytes / t / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(nbytes / t / 1e9 / HBM_PEAK_GBS, 4),
            "traffic": (pmc.get("msda_encoder", {}) if enc_native else pmc.get("msda", {})).get("hbm_bytes_largest_launch"),
            "algorithmic_bytes_per_launch": nbytes, "avg_launch_us": round(t * 1e6, 1),
        }
        if enc_native:
            # the roof that binds this kernel next to its issue slots: every sample reads four 64-byte rows (head_dim 32 halves)
            # out of the LDS windows -- bytes per launch / measured time against the LDS arrays' aggregate rate (256 B per
            # clock and CU: MI355X_MICROARCH.md; the clock of the guide's 2.4 GHz peak).  Bank conflicts (43 % of the LDS-active
            # cycles at 2 px, profiles/r06_sq_counters.txt) are inside `achieved`, not inside the byte count.
            lds_bytes = m["B"] * m["Nq"] * m["M"] * m["L"] * m["P"] * 4 * m["D"] * e
            lds_peak = 256 * 256 * 2.4          # GB/s: CUs x bytes per clock x GHz
            out["roofline_msda"]["lds"] = {"gather_bytes_per_launch": lds_bytes, "achieved": round(lds_bytes / t / 1e9, 1),
                                           "peak": round(lds_peak, 1), "unit": "GB/s",
                                           "frac": round(lds_bytes / t / 1e9 / lds_peak, 4),
                                           "bank_conflict_share_of_lds_cycles": _sq_conflict_share("msda_encoder_v4_kernel"),
                                           "bank_conflict_source": "profiles/r06_sq_counters.txt (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE)"}
    return out


def _sq_conflict_share(kernel_substring):
    """confl% column of the committed per-kernel SQ counter table (tools/pmc_sq_model.sh), as a fraction; None if absent"""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r06_sq_counters.txt")
    try:
        for ln in open(path):
            if kernel_substring in ln and not ln.startswith("#"):
                return round(float(ln.split()[-2]) / 100.0, 3)
    except (OSError, ValueError, IndexError):
        pass
    return None


def batch1_latency(model, image, mask, device, steps=20):
    """p50 / p90 of one-image forwards replayed from their own hipGraph (the reference's published number, 79.5 ms on
    an RTX 4090, is this quantity: batch 1, README.md:33)"""
    def fwd():
        with torch.no_grad():
            return model(image, mask)

    for _ in range(2):
        fwd()
    torch.cuda.synchronize(device)
    run = fwd
    try:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            fwd()
        run = g.replay
    except Exception:  # noqa: BLE001
        torch.cuda.synchronize(device)
    st = torch.cuda.current_stream(device)
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    run()
    evs[0].record(st)
    for i in range(steps):
        run()
        evs[i + 1].record(st)
    torch.cuda.synchronize(device)
    ts = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(steps))
    return {"p50_ms": round(ts[len(ts) // 2], 3), "p90_ms": round(ts[int(len(ts) * 0.9)], 3), "steps": steps,
            "hipgraph": run is not fwd}


def cpu_baseline(model, full_hw=(1280, 1920), budget_s=45.0):
    """fp32 CPU oracle on the host cores.  A 608x608 image is timed first (a few seconds); if that predicts the
    full-size image fits the budget, ONE 1920x1280 image -- the workload itself -- is run and reported directly;
    otherwise the 608x608 time is scaled by the pixel ratio (and says so)."""
    # thread count: the oracle (ATen CPU kernels + the OpenMP C MSDA) is fastest at 16-32 threads on the GPU box's 256
    # logical CPUs and 4-5x slower at torch's default of 128 (tools/probe_cpu_oracle_threads.py: 0.97 s at 16, 1.08 s
    # at 32, 4.40 s at 128 for a 384x384 image) -- use what serves it best, and say so in `cores`
    threads = max(1, min(32, (os.cpu_count() or 8) // 2, torch.get_num_threads()))
    os.environ["OMP_NUM_THREADS"] = str(threads)   # read by the C oracle's OpenMP runtime when it is loaded below
    torch.set_num_threads(threads)
    import codetr_fp32 as M

    sd = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}

    def run(H, W):
        g = torch.Generator().manual_seed(42)
        img = torch.randn(1, 3, H, W, generator=g)
        mask = torch.zeros(1, H, W)
        t0 = time.perf_counter()
        with torch.no_grad():
            M.codetr_forward(sd, img, mask, backbone="swin", num_heads=(6, 12, 24, 48), window_size=12)
        return time.perf_counter() - t0

    t608 = run(608, 608)
    ratio = (608 * 608) / float(full_hw[0] * full_hw[1])
    who = (f"fp32 CPU oracle (oracle/codetr_fp32.py + C MSDA, {threads} threads, os.cpu_count()={os.cpu_count()})")
    if t608 / ratio <= budget_s:
        dt = run(*full_hw)
        return {"value": round(1.0 / dt, 5), "unit": "images/s", "cores": threads, "kind": "port",
                "sample": f"1 image {full_hw[1]}x{full_hw[0]} (the workload's own size) through the {who}: {dt:.1f} s "
                          f"(a 608x608 image took {t608:.1f} s)",
                "seconds_for_sample": round(dt + t608, 2)}
    return {"value": round(ratio / t608, 5), "unit": "images/s", "cores": threads, "kind": "port",
            "sample": f"1 image 608x608 through the {who} took {t608:.1f} s; value = 1/t scaled by the pixel ratio "
                      f"{ratio:.3f} to {full_hw[1]}x{full_hw[0]}-equivalent images/s (a full-size image would exceed the "
                      f"{budget_s:.0f} s budget)",
            "seconds_for_sample": round(t608, 2)}


def _free_port():
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launch_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: start the N ranks as a child torch.distributed.run (this process
    has not touched the GPU and never execs), relay the child's stdout (rank 0's JSON line), return its exit code."""
    import subprocess

    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL needs it on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n)))
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    sys.stdout.write(proc.stdout)
    sys.stdout.flush()
    return proc.returncode


def bind_rank(local_rank, local_world):
    """rank -> device is explicit (cuda:local_rank); the rank's CPU affinity is narrowed to the cores of the GPU's
    NUMA node when sysfs tells (pinned host buffers and the launch thread then sit next to the GPU's PCIe root),
    else to an even contiguous share of the cores.  Returns a short description for the report."""
    try:
        cpus = sorted(os.sched_getaffinity(0))
    except AttributeError:
        return "no affinity API"
    node = None
    try:
        prop = torch.cuda.get_device_properties(local_rank)
        bdf = "%04x:%02x:%02x.0" % (prop.pci_domain_id, prop.pci_bus_id, prop.pci_device_id)
        with open(f"/sys/bus/pci/devices/{bdf}/numa_node") as f:
            node = int(f.read().strip())
        if node >= 0:
            with open(f"/sys/devices/system/node/node{node}/cpulist") as f:
                want = set()
                for part in f.read().strip().split(","):
                    lo, _, hi = part.partition("-")
                    want.update(range(int(lo), int(hi or lo) + 1))
            share = [c for c in cpus if c in want]
            if share:
                os.sched_setaffinity(0, share)
                return f"cuda:{local_rank} numa node {node}, {len(share)} cpus"
    except (OSError, AttributeError, ValueError, RuntimeError):
        pass
    if local_world > 1 and len(cpus) >= local_world:
        per = len(cpus) // local_world
        share = cpus[local_rank * per:(local_rank + 1) * per]
        os.sched_setaffinity(0, share)
        return f"cuda:{local_rank} cpus {share[0]}-{share[-1]} (even split; numa node unknown)"
    return f"cuda:{local_rank} (affinity unchanged)"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=8,
                    help="images per GPU per step (default 8 = BASELINE config 4: batch 64 sharded over 8 GPUs)")
    ap.add_argument("--streams", type=int, default=2,
                    help="sub-batches replayed concurrently on separate HIP streams (graph mode; 1 = single stream)")
    ap.add_argument("--res", default="1920x1280", help="WxH")
    ap.add_argument("--dtype", default="fp16", choices=["fp16", "bf16", "fp32", "fp8"],
                    help="fp8 = BASELINE config 5: e4m3 weights + activations on the Swin stage 1-3 linears (MX block scales) "
                         "and the encoder FFN (static scales calibrated on other images), fp16 elsewhere; a separate "
                         "line, never the fp16 headline (the default fp16 run carries it as the `fp8` sub-record)")
    ap.add_argument("--offset-noise-px", type=float, default=2.0,
                    help="query-dependent spread of the MSDA sampling offsets in pixels (0 = the default init: fixed grid)")
    ap.add_argument("--feed", default="hbm", choices=["hbm", "host"],
                    help="hbm: inputs resident in HBM (the metric); host: every step copies its images from pinned host "
                         "memory inside the timed region (reported, never `value`)")
    ap.add_argument("--no-graph", action="store_true", help="do not replay the forward from a captured hipGraph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-host-feed", action="store_true", help="skip the extra host-feed pass")
    ap.add_argument("--no-fp8-line", action="store_true", help="skip the fp8 (config 5) sub-record of the fp16 run")
    ap.add_argument("--pad", type=float, default=0.9,
                    help="valid fraction per side of the `padded` sub-record's images (1.0 = skip it)")
    ap.add_argument("--quick", action="store_true", help="headline line only: no sub-records, rooflines or CPU baseline")
    ap.add_argument("--dry-run", action="store_true",
                    help="CPU / gloo: launcher, sharding, gather, timing and report with stand-in detections, no model")
    a = ap.parse_args()
    if a.quick:
        a.no_cpu_baseline = a.no_roofline = a.no_host_feed = a.no_fp8_line = True
        a.pad = 1.0

    launched = "WORLD_SIZE" in os.environ and "RANK" in os.environ
    if not launched and a.gpus > 1:
        # no launcher around us: start the ranks as a child process BEFORE any GPU call in this one
        sys.exit(launch_ranks(a.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    if world != a.gpus:
        a.gpus = world
    if a.dry_run:
        return dry_run(a, world, rank)
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X (torch.cuda.is_available() is False)")
    # CODETR_BENCH_SHARE_GPU=1 (control-flow test on a 1-GPU box): ranks share the visible GPUs round-robin and gather
    # over gloo -- RCCL refuses two ranks on one device.  Never a measurement configuration.
    share = os.environ.get("CODETR_BENCH_SHARE_GPU", "0") == "1"
    if local_rank >= torch.cuda.device_count() and not share:
        sys.exit(f"rank {rank}: local rank {local_rank} has no GPU ({torch.cuda.device_count()} visible)")
    device = torch.device("cuda", local_rank % torch.cuda.device_count())
    torch.cuda.set_device(device)
    binding = bind_rank(device.index, local_world)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)  # "nccl" is RCCL on ROCm

    dtype = {"fp16": torch.float16, "bf16": torch.bfloat16, "fp32": torch.float32, "fp8": torch.float16}[a.dtype]
    W, H = (int(x) for x in a.res.split("x"))
    model = build_model(device, dtype, offset_noise_px=a.offset_noise_px)
    g = torch.Generator(device=device).manual_seed(42 + rank)
    images = torch.randn(a.batch, 3, H, W, device=device, generator=g).to(dtype)  # ~ mean/std-normalised image
    masks = torch.zeros(a.batch, H, W, device=device, dtype=dtype)
    from codetr.sharding import gather_detections, pack_detections

    fp8_report = None
    if a.dtype == "fp8":
        from codetr import fp8

        nb0 = max(1, a.batch // max(1, min(a.streams, a.batch)))
        gc = torch.Generator(device=device).manual_seed(4242)      # calibration sees OTHER images than the timed ones
        calib = torch.randn(nb0, 3, H, W, device=device, generator=gc).to(dtype)
        fp8.calibrate(model, calib, masks[:nb0].contiguous())
        del calib
        fp8.enable(model)
        fp8_report = fp8.report(model)

    gathered = torch.empty(world * a.batch, 300, 6, device=device, dtype=torch.float32) if world > 1 else None

    # The per-GPU batch is cut into `streams` sub-batches, each captured into its own hipGraph on its own HIP stream and
    # replayed concurrently: kernels of different phases (MFMA-bound GEMMs, the L2-bound MSDA gather, HBM-bound norms,
    # store-bound epilogues) of the two sub-batches overlap on the chip (measured -2.7 % ms/image at 2 x 4 images).
    nstreams = max(1, min(a.streams, a.batch)) if not a.no_graph else 1
    bounds = [(i * a.batch) // nstreams for i in range(nstreams + 1)]
    subs = [(images[bounds[i]:bounds[i + 1]].contiguous(), masks[bounds[i]:bounds[i + 1]].contiguous())
            for i in range(nstreams)]
    static_out = torch.empty(a.batch, 300, 6, device=device, dtype=torch.float32)
    # host side of the PCIe feed: the same images in pinned memory, one buffer per sub-batch
    host_subs = [subs[i][0].cpu().pin_memory() for i in range(nstreams)]

    def forward(i):
        with torch.no_grad():
            boxes, scores, labels = model(*subs[i])
        static_out[bounds[i]:bounds[i + 1]].copy_(pack_detections(boxes, scores, labels))  # [b,300,6] fp32

    side = [torch.cuda.Stream(device) for _ in range(nstreams)]

    def capture():
        """eager warm-up (builds the shape-keyed caches, lets the allocator settle), then one hipGraph per sub-batch"""
        for _ in range(max(2, min(a.warmup, 3))):
            for i in range(nstreams):
                forward(i)
        torch.cuda.synchronize(device)
        if a.no_graph:
            return None
        try:
            gs = []
            for i in range(nstreams):
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.stream(side[i]):
                    with torch.cuda.graph(gr, stream=side[i]):
                        forward(i)
                gs.append(gr)
            torch.cuda.synchronize(device)
            return gs
        except Exception as e:  # noqa: BLE001 -- report and run eagerly; the number is still valid, just launch-bound
            if rank == 0:
                print(f"[bench] hipGraph capture unavailable ({type(e).__name__}: {e}); running eagerly", file=sys.stderr)
            torch.cuda.synchronize(device)
            return None

    graphs = capture()
    graph = graphs  # (name kept for the report below)

    def step(feed_host=False):
        main = torch.cuda.current_stream(device)
        if graphs is not None:
            for i in range(nstreams):
                side[i].wait_stream(main)       # the previous step's consumers are done with static_out
                with torch.cuda.stream(side[i]):
                    if feed_host:   # this sub-batch's images over PCIe, on its own stream: overlaps the other's compute
                        subs[i][0].copy_(host_subs[i], non_blocking=True)
                    graphs[i].replay()
            for i in range(nstreams):
                main.wait_stream(side[i])
        else:
            for i in range(nstreams):
                if feed_host:
                    subs[i][0].copy_(host_subs[i], non_blocking=True)
                forward(i)
        if world > 1:
            gather_detections(static_out, world * a.batch, out=gathered)  # the only collective: 7.2 KB per image

    def fence():
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    rank_stats = {}

    def timed(feed_host):
        for _ in range(a.warmup):
            step(feed_host)
        st = torch.cuda.current_stream(device)
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]
        fence()
        t0 = time.perf_counter()
        evs[0].record(st)
        for i in range(a.steps):
            step(feed_host)
            evs[i + 1].record(st)
        fence()
        elapsed = time.perf_counter() - t0
        per_step = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(a.steps))
        if world > 1:
            # MAX over ranks is the job's time; MIN and the count of ranks that answered make a straggler or a missing
            # rank visible in the line (ranks_seen: an all_reduce of ones over the same backend -- RCCL on a real run)
            dv = "cpu" if share else device
            t = torch.tensor([elapsed], device=dv, dtype=torch.float64)
            tmin, ones = t.clone(), torch.ones(1, device=dv, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dist.all_reduce(tmin, op=dist.ReduceOp.MIN)
            dist.all_reduce(ones, op=dist.ReduceOp.SUM)
            rank_stats.update(ranks_seen=int(ones.item()),
                              per_rank_images_per_s_min=round(a.steps * a.batch / float(t.item()), 3),
                              per_rank_images_per_s_max=round(a.steps * a.batch / float(tmin.item()), 3))
            elapsed = float(t.item())
        else:
            rank_stats.update(ranks_seen=1, per_rank_images_per_s_min=round(a.steps * a.batch / elapsed, 3),
                              per_rank_images_per_s_max=round(a.steps * a.batch / elapsed, 3))
        return elapsed, per_step

    elapsed, per_step_ms = timed(a.feed == "host")
    headline_rank_stats = dict(rank_stats)
    host_feed = None
    if a.feed == "hbm" and not a.no_host_feed:
        e2, _ = timed(True)
        host_feed = {"images_per_s": round(a.steps * a.batch * world / e2, 3), "ms_per_step": round(e2 / a.steps * 1e3, 3),
                     "bytes_per_rank_per_step": int(sum(h.numel() * h.element_size() for h in host_subs)),
                     "note": "same K steps with every step's images copied from pinned host memory over PCIe on the "
                             "sub-batch streams inside the timed region; reported beside `value`, never as it"}

    fp8_line = None
    if world == 1 and a.dtype == "fp16" and not a.no_fp8_line:
        # BASELINE config 5 as a sub-record of the same run (never `value`): the Swin stage 1-3 linears on MX block-scaled
        # e4m3 (no calibration) + the encoder FFN on e4m3 with static scales calibrated on OTHER images; same K steps
        from codetr import fp8

        fp16_graphs = graphs
        try:
            gc = torch.Generator(device=device).manual_seed(4242)
            nb0 = max(1, a.batch // nstreams)
            calib = torch.randn(nb0, 3, H, W, device=device, generator=gc).to(dtype)
            fp8.calibrate(model, calib, masks[:nb0].contiguous())
            sat8 = fp8.saturation(model, images[:nb0].contiguous(), masks[:nb0].contiguous())   # the TIMED images vs those scales
            fp8.enable(model)
            del calib
            graphs = capture()
            e8, per8 = timed(False)
            fp8_line = {"images_per_s": round(a.steps * a.batch / e8, 3), "ms_per_step": round(e8 / a.steps * 1e3, 3),
                        "p50_ms_per_image": round(per8[len(per8) // 2] / a.batch, 3), "config": fp8.report(model),
                        "static_scale_headroom": sat8,
                        "dtype": "fp8 e4m3: Swin stage 1-3 linears with MX block scales (e8m0 per 32 channels, hardware-applied), "
                                 "encoder FFN with static scales calibrated on other images; f16 elsewhere, f32 accumulation"}
            if not a.no_roofline:
                nb = max(1, a.batch // max(1, nstreams))
                r8 = kernel_rooflines(model, images[:nb].contiguous(), masks[:nb].contiguous(), device)
                for k in ("roofline_fp8", "roofline_ffn_fp8"):
                    if k in r8:
                        fp8_line[k] = r8[k]
            # what the sensitivity map leaves when the fp16 path's accuracy bar applies (profiles/r04_fp8_sensitivity.json:
            # encoder-memory error <= 2e-2 against the fp16 product): 8 of the 88 GEMMs -- timed so that the line says what
            # that selection is worth, and that the full selection above is a FAST mode
            fp8_line["recommended"] = False

            def _memory(nimg=nb0):
                # one eager forward on the TIMED images: the deformable encoder's output (captured intermediates)
                cap = {}
                with torch.no_grad():
                    model(images[:nimg].contiguous(), masks[:nimg].contiguous(), capture=cap)
                torch.cuda.synchronize(device)
                return cap["memory"].float()

            def _rel(m8, m16):
                per = [float(torch.linalg.vector_norm(x - y) / torch.linalg.vector_norm(y).clamp_min(1e-30))
                       for x, y in zip(m8, m16)]
                return round(max(per), 5)

            # accuracy of THIS run's kernels (the line carried figures of a round-4 file): encoder memory of the
            # e4m3 selection against the fp16 product's on the same images, worst image; the proxy AP needs the oracle's
            # detections and stays with tools/fp8_sensitivity.py
            mem8 = _memory()
            fp8.enable(model, False)
            mem16 = _memory()
            fp8_line["accuracy"] = {"encoder_memory_rel_l2_vs_fp16": _rel(mem8, mem16),
                                    "source": "measured in this run: eager forwards of the timed images (first %d), "
                                              "worst image" % nb0}
            del mem8
            fp8.enable(model, True, "mx", select="accurate")
            graphs = capture()
            ea, pera = timed(False)
            fp8_line["accurate_preset"] = {"images_per_s": round(a.steps * a.batch / ea, 3),
                                           "p50_ms_per_image": round(pera[len(pera) // 2] / a.batch, 3),
                                           "config": fp8.report(model),
                                           "encoder_memory_rel_l2_vs_fp16": _rel(_memory(), mem16),
                                           "note": "stage-3 qkv / fc1 / fc2 + stage-1 fc2 in e4m3, encoder FFN in fp16; the "
                                                   "memory error is measured in this run like `accuracy` above"}
            del mem16
        except Exception as e:  # noqa: BLE001 -- the optional sub-record must never cost the measured fp16 line
            torch.cuda.synchronize(device)
            fp8_line = {"error": repr(e)}
        finally:
            fp8.enable(model, False)
            graphs = fp16_graphs

    # ---- the same K steps on PADDED images (reference inferencer.py:354-358: mask = 1 on the last 10 %
    # of the rows and columns; the graphs read the masks from device memory, so the captured launch lists serve) ----
    padded_line = None
    if world == 1 and a.pad < 1.0:
        try:
            hv, wv = int(H * a.pad), int(W * a.pad)
            for _, m_ in subs:
                m_.zero_()
                m_[:, hv:, :] = 1
                m_[:, :, wv:] = 1
            ep, perp = timed(False)
            padded_line = {"valid_fraction_per_side": a.pad, "images_per_s": round(a.steps * a.batch / ep, 3),
                           "ms_per_step": round(ep / a.steps * 1e3, 3),
                           "p50_ms_per_image": round(perp[len(perp) // 2] / a.batch, 3),
                           "vs_unpadded": round((a.steps * a.batch / ep) / (a.steps * a.batch * world / elapsed), 4),
                           "note": "same graphs, same K steps, padding masks = 1 on the last %d %% of rows and columns "
                                   "(valid ratios skew the encoder's reference points)" % round((1 - a.pad) * 100)}
        except Exception as e:  # noqa: BLE001
            torch.cuda.synchronize(device)
            padded_line = {"error": repr(e)}
        finally:
            for _, m_ in subs:
                m_.zero_()

    # ---- the same K steps with 8 px of query-dependent MSDA offset spread (the 2 px of the headline is an
    # assumption; wider offsets push samples out of the encoder kernel's staged windows) ----
    wide_line = None
    if world == 1 and a.dtype == "fp16" and a.offset_noise_px > 0 and not a.no_roofline:
        base_graphs = graphs
        try:
            set_offset_noise(model, 8.0)
            graphs = capture()     # (derived weights -- the packed projections -- are rebuilt: new launch lists)
            e8px, per8px = timed(False)
            wide_line = {"images_per_s": round(a.steps * a.batch / e8px, 3), "ms_per_step": round(e8px / a.steps * 1e3, 3),
                         "p50_ms_per_image": round(per8px[len(per8px) // 2] / a.batch, 3),
                         "vs_headline": round((a.steps * a.batch / e8px) / (a.steps * a.batch * world / elapsed), 4),
                         "msda_offset_noise_px": 8.0}
        except Exception as e:  # noqa: BLE001
            torch.cuda.synchronize(device)
            wide_line = {"error": repr(e)}
        finally:
            set_offset_noise(model, a.offset_noise_px)
            graphs = base_graphs

    if rank == 0:
        total_images = a.steps * a.batch * world
        out = {
            "metric": "images/sec, Co-DINO Swin-L %s %s (p50 ms/image alongside)" % (a.res, a.dtype),
            "value": round(total_images / elapsed, 3),
            "unit": "images/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": round(elapsed / a.steps * 1e3, 3),
            "p50_ms_per_image": round(per_step_ms[len(per_step_ms) // 2] / a.batch, 3),
            "p90_ms_per_image": round(per_step_ms[int(len(per_step_ms) * 0.9)] / a.batch, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": {"fp16": "f16", "bf16": "bf16", "fp32": "f32",
                      "fp8": "fp8 e4m3 (Swin stage 1-3 linears and both products of the encoder FFN: weights + activations; f16 elsewhere, f32 accumulation)"}[a.dtype],
            "data": "synthetic (randn images %s, zero padding masks; seeded random-init weights of the "
                    "real architecture with %g px of query-dependent MSDA offset spread -- no checkpoint/COCO available "
                    "offline)" % ("already in HBM" if a.feed == "hbm" else "fed from pinned host memory every step",
                                  a.offset_noise_px),
            "config": {
                "workload": "CoDETR.forward, Co-DINO 5-scale Swin-L, %s, batch %d per GPU, 900 queries, 300 detections"
                            % (a.res, a.batch),
                "global_batch": a.batch * world, "parallelism": "image-sharded replicas x%d" % world,
                "hipgraph": graph is not None, "streams": nstreams, "feed": a.feed, "rank0_binding": binding,
                "shared_gpu_test_mode": share,
                "msda_offset_noise_px": a.offset_noise_px,
                "native_kernels": sorted(__import__("codetr.hip_ops", fromlist=["NATIVE"]).NATIVE),
            },
            "reference_note": "reference publishes 79.5 ms/image (TensorRT fp16, RTX 4090, batch 1, README.md:33); "
                              "not the same hardware, so vs_baseline stays null",
        }
        out.update(headline_rank_stats)
        if host_feed is not None:
            out["host_feed"] = host_feed
        if fp8_line is not None:
            out["fp8"] = fp8_line
        if padded_line is not None:
            out["padded"] = padded_line
        if wide_line is not None:
            out["value_8px"] = wide_line
        if fp8_report is not None:
            out["config"]["fp8"] = fp8_report
        if world == 1 and not a.no_roofline:
            # rooflines: one eager forward over the images of ONE replayed graph (batch / streams: the launches of the
            # timed region, same shapes and kernels; the committed PMC passes ran this shape).  The stand-alone MSDA
            # operator and the latency are single-image quantities.
            op, op_dec = msda_roofline(1, H, W, dtype, device)
            out["latency_batch1"] = batch1_latency(model, images[:1].contiguous(), masks[:1].contiguous(), device)
            # ... and at the reference's other two published sizes (README.md:33-35: 1920x1280, 1152x768, 608x608)
            by_size = {a.res: out["latency_batch1"]}
            for rw, rh in ((1152, 768), (608, 608)):
                if "%dx%d" % (rw, rh) == a.res or dtype == torch.float32:
                    continue
                try:
                    gi = torch.Generator(device=device).manual_seed(7)
                    xi1 = torch.randn(1, 3, rh, rw, device=device, generator=gi).to(dtype)
                    by_size["%dx%d" % (rw, rh)] = batch1_latency(model, xi1, torch.zeros(1, rh, rw, device=device, dtype=dtype), device)
                except Exception as e:  # noqa: BLE001
                    torch.cuda.synchronize(device)
                    by_size["%dx%d" % (rw, rh)] = {"error": repr(e)}
            out["latency_batch1_by_size"] = by_size
            if dtype in (torch.float16, torch.bfloat16):   # (fp16, fp8 and bf16 runs: every kernel has both 16-bit forms)
                nb = max(1, a.batch // max(1, nstreams))
                xi, xm = images[:nb].contiguous(), masks[:nb].contiguous()
                out.update(kernel_rooflines(model, xi, xm, device))
                out["roofline_images_per_launch"] = nb
                out["roofline_msda_op"] = op
                out["roofline_msda_op_dec"] = op_dec
                if "roofline_msda" in out:
                    out["roofline_msda"]["offsets"] = msda_offset_stats(model, xi, xm)
                    if a.offset_noise_px > 0:
                        # the same kernel with the default init's zero offset weights (round 1's number): best case
                        set_offset_noise(model, 0.0)
                        out["roofline_msda_zero_noise"] = kernel_rooflines(model, xi, xm, device).get("roofline_msda")
                        # ... and with a wider trained-like spread (the share of samples that leave the
                        # staged windows, and what they cost, beyond 2 px)
                        for px in (4.0, 8.0):
                            set_offset_noise(model, px)
                            r = kernel_rooflines(model, xi, xm, device).get("roofline_msda")
                            if r is not None:
                                r["offsets"] = msda_offset_stats(model, xi, xm)
                                out["roofline_msda_%dpx" % int(px)] = r
                        set_offset_noise(model, a.offset_noise_px)
                    if padded_line is not None and "error" not in padded_line:
                        # the encoder MSDA kernel on the padded images (eager forward, same launch shapes)
                        pm = xm.clone()
                        pm[:, int(H * a.pad):, :] = 1
                        pm[:, :, int(W * a.pad):] = 1
                        r = kernel_rooflines(model, xi, pm, device).get("roofline_msda")
                        if r is not None:
                            r["offsets"] = msda_offset_stats(model, xi, pm)
                            out["padded"]["roofline_msda"] = r
            else:
                out["roofline"] = op
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(model)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def dry_run(a, world, rank):
    """The control flow of a multi-rank run without GPUs or the model: gloo process group, shard assignment, the
    detections all_gather, barrier-fenced timing with MAX over ranks, rank 0's JSON line (value null)."""
    from codetr.sharding import gather_detections, shard_range

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
    s, e = shard_range(a.batch * world, rank, world)
    local = torch.stack([torch.full((300, 6), float(i)) for i in range(s, e)])
    ok = True
    t0 = time.perf_counter()
    for _ in range(a.warmup + a.steps):
        full = gather_detections(local, a.batch * world)
        ok = ok and bool((full[:, 0, 0] == torch.arange(a.batch * world, dtype=torch.float32)).all())
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    seen, tmin_s = 1, elapsed
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64)
        tmin, ones = t.clone(), torch.ones(1, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(tmin, op=dist.ReduceOp.MIN)
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)
        elapsed, tmin_s, seen = float(t.item()), float(tmin.item()), int(ones.item())
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"metric": "dry run (no model, no GPU)", "value": None, "unit": "images/s", "n_gpus": world,
                          "ranks_seen": seen, "slowest_over_fastest_rank": round(elapsed / max(tmin_s, 1e-9), 3),
                          "steps": a.steps, "warmup": a.warmup, "dry_run": True, "gather_ok": ok,
                          "global_batch": a.batch * world, "ms_per_step": round(elapsed / (a.warmup + a.steps) * 1e3, 3),
                          "scaling": "weak", "higher_is_better": True}))
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main() or 0)
