"""Probe: does replaying two half-batch forwards on two streams beat one full-batch forward? (GPU box only)"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "co-detr-tensorrt_amd"))
import bench  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
model = bench.build_model(dev, torch.float16)
H, W = 1280, 1920


def make_graph(B, stream):
    g = torch.Generator(device=dev).manual_seed(B)
    img = torch.randn(B, 3, H, W, device=dev, generator=g).half()
    msk = torch.zeros(B, H, W, device=dev, dtype=torch.float16)

    def fwd():
        with torch.no_grad():
            return model(img, msk)

    with torch.cuda.stream(stream):
        for _ in range(2):
            fwd()
        stream.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=stream):
            out = fwd()
    return gr, out


def timed(fn, steps=10):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


s0 = torch.cuda.Stream()
g8, _ = make_graph(8, s0)
t8 = timed(lambda: g8.replay())
print(f"one graph of 8 images: {t8 * 1e3:.2f} ms/step = {t8 * 1e3 / 8:.3f} ms/image")
for nstreams, per in ((2, 4), (4, 2), (2, 8)):
    streams = [torch.cuda.Stream() for _ in range(nstreams)]
    graphs = [make_graph(per, st)[0] for st in streams]

    def step():
        for st, gr in zip(streams, graphs):
            with torch.cuda.stream(st):
                gr.replay()

    t = timed(step)
    print(f"{nstreams} streams x {per} images: {t * 1e3:.2f} ms/step = {t * 1e3 / (nstreams * per):.3f} ms/image")

# free-running streams with an initial skew of half a forward (no per-step join)
streams = [torch.cuda.Stream() for _ in range(2)]
graphs = [make_graph(4, st)[0] for st in streams]
for skew_ms in (0, 25, 50):
    def run(steps):
        for j, (st, gr) in enumerate(zip(streams, graphs)):
            with torch.cuda.stream(st):
                if j == 1 and skew_ms:
                    torch.cuda._sleep(int(skew_ms * 2.0e6))   # ~2 GHz cycles
                for _ in range(steps):
                    gr.replay()
    run(2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(10)
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / 10
    print(f"free-running 2 x 4 images, initial skew {skew_ms} ms: {t * 1e3:.2f} ms/step = {t * 1e3 / 8:.3f} ms/image (skew included)")
