"""Determinism stress of the two round-5 fused encoder launches (counted-wait / ring-reuse logic): the same inputs N times at
several row counts -- every run must be bit-identical to the first (a missed wait shows up as a sporadic difference), while
another stream keeps the memory system busy.     python tools/stress_fused_encoder_kernels.py [iters]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "co-detr-tensorrt_amd"))
from codetr import hip_ops  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = "cuda"
bad = 0
side = torch.cuda.Stream()
noise_a = torch.randn(64 << 20, device=dev).half()
noise_b = torch.empty_like(noise_a)
with torch.no_grad():
    for M in (818400, 204600 + 77, 40000, 128 * 256 + 1):
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
                    print(f"M={M} iteration {it}: DIFFERENT from the first run")
        print(f"M={M}: {iters} runs, identical so far: {bad == 0}")
print("FAILED" if bad else "all runs bit-identical")
sys.exit(1 if bad else 0)
