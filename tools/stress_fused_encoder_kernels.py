"""Determinism stress of the fused encoder launches (counted-wait / ring-reuse logic in the FFN + output-projection kernel
and the one-launch projections; the DPP hazard fix of the packed MSDA kernel, csrc/msda_encoder4.hip): the same inputs N
times at several row counts -- every run must be bit-identical to the first (a missed wait or a re-introduced hazard shows
up as a sporadic difference), while another stream keeps the memory system busy.
    python tools/stress_fused_encoder_kernels.py [iters]
The same loop runs in the -m gpu suite (tests/test_stress_fused_encoder_gpu.py, fewer iterations)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if os.path.join(ROOT, "co-detr-tensorrt_amd") not in sys.path:
    sys.path.insert(0, os.path.join(ROOT, "co-detr-tensorrt_amd"))

SIZES = (818400, 204600 + 77, 40000, 128 * 256 + 1)
PYRAMID = [(80, 120), (40, 60), (20, 30), (10, 15), (5, 8)]


def run_gemm_side(iters, sizes=SIZES, log=print):
    """ffn_oproj_fused + encoder_projections; returns the number of runs that differed from the first"""
    from codetr import hip_ops

    dev = "cuda"
    bad = 0
    side = torch.cuda.Stream()
    noise_a = torch.randn(64 << 20, device=dev).half()
    noise_b = torch.empty_like(noise_a)
    with torch.no_grad():
        for M in sizes:
            r = lambda *s, k=1.0: (torch.randn(*s, device=dev) * k).half()   # noqa: E731
            attn, ident, pos = r(M, 256), r(M, 256, k=2.0), r(M, 256)
            wo, bo = r(256, 256, k=1 / 16), r(256, k=0.5)
            w1, b1, w2, b2 = r(2048, 256, k=1 / 16), r(2048), r(256, 2048, k=1 / 45), r(256)
            gam, bet = torch.ones(256, device=dev).half(), torch.zeros(256, device=dev).half()
            ln = (gam, bet, 1e-5)
            wc, bc = r(768, 256, k=1 / 16), r(768)
            x3, p3 = attn.view(1, M, 256), pos.view(1, M, 256)
            mask = (torch.rand(1, M, device=dev) < 0.1)
            ref_f = ref_p = None
            for it in range(iters):
                with torch.cuda.stream(side):
                    noise_b.copy_(noise_a)                      # traffic from a second stream
                y, y2 = hip_ops.ffn_oproj_fused(attn, wo, bo, ident, w1, b1, w2, b2, ln, pos=pos, ln_in=ln)
                both = hip_ops.encoder_projections(x3, p3, wc, bc, mask, 256, 32)
                torch.cuda.synchronize()
                if ref_f is None:
                    ref_f, ref_p = (y.clone(), y2.clone()), (both[0].clone(), both[1].clone())
                    assert torch.isfinite(y.float()).all() and torch.isfinite(both[1].float()).all()
                else:
                    ok = (torch.equal(y, ref_f[0]) and torch.equal(y2, ref_f[1]) and torch.equal(both[0], ref_p[0])
                          and torch.equal(both[1], ref_p[1]))
                    if not ok:
                        bad += 1
                        log(f"M={M} iteration {it}: DIFFERENT from the first run")
            log(f"M={M}: {iters} runs, identical so far: {bad == 0}")
    return bad


def run_msda_packed(iters, batch=2, log=print):
    """the packed encoder MSDA kernel at hip_ops' shipped launch shape, offsets wide enough to use the fix-up queue"""
    from codetr import _cabi, hip_ops

    dev = "cuda"
    M, L, P, D = 8, 5, 4, 32
    S = sum(h * w for h, w in PYRAMID)
    g = torch.Generator(device="cpu").manual_seed(5)
    value = torch.randn(batch, M, S, D, generator=g).half().to(dev)             # head-major
    cat = torch.cat(((torch.randn(batch, S, M * L * P * 2, generator=g) * 3).half(),
                     (torch.randn(batch, S, M * L * P, generator=g) * 2).half()), -1)
    idx = torch.tensor(_cabi.msda_pack_projection_index(M, L, P))
    packed = cat[..., idx.clamp_min(0)].contiguous().to(dev)
    counts = torch.tensor([[[w, h] for h, w in PYRAMID]] * batch, dtype=torch.float32, device=dev)
    win = [[(-4, 4, -4, 4)] * L] * M
    side = torch.cuda.Stream()
    noise_a = torch.randn(32 << 20, device=dev).half()
    noise_b = torch.empty_like(noise_a)
    ref, bad = None, 0
    for it in range(iters):
        with torch.cuda.stream(side):
            noise_b.copy_(noise_a)
        out = torch.full((batch, S, M * D), float("nan"), dtype=torch.float16, device=dev)
        ok = _cabi.msda_encoder_packed(value, PYRAMID, packed, P, win, counts, tuple(hip_ops.MSDA_V4_REGION),
                                       hip_ops.MSDA_V4_THREADS, out, True)
        torch.cuda.synchronize()
        assert ok, "the packed encoder kernel did not take the shape"
        if ref is None:
            ref = out.clone()
            assert torch.isfinite(ref.float()).all()
        elif not torch.equal(out, ref):
            bad += 1
            log(f"packed MSDA iteration {it}: DIFFERENT from the first run")
    log(f"packed MSDA: {iters} runs, identical so far: {bad == 0}")
    return bad


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    failures = run_gemm_side(n) + run_msda_packed(n)
    print("FAILED" if failures else "all runs bit-identical")
    sys.exit(1 if failures else 0)
