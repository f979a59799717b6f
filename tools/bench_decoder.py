"""Time the DINO decoder (6 layers, 900 queries) alone, as a replayed HIP graph, at the headline pyramid.
    python tools/bench_decoder.py [--batch 1] [--res 1920x1280]
Inputs are captured from one full forward of the seeded random-init model (bench.build_model)."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "co-detr-tensorrt_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--res", default="1920x1280")
    ap.add_argument("--iters", type=int, default=30)
    a = ap.parse_args()
    import bench
    from codetr import _cabi

    if os.environ.get("CODETR_LIB"):   # timing experiments: a diagnostic build of the library (tools/micro/build_dec_variants.sh)
        _cabi.LIB_PATH, _cabi._lib, _cabi._rec_lib = os.environ["CODETR_LIB"], None, None
    dev = torch.device("cuda:0")
    W, H = (int(v) for v in a.res.split("x"))
    model = bench.build_model(dev, torch.float16)
    dec = model.query_head.transformer.decoder
    g = torch.Generator().manual_seed(1)
    img = torch.randn(a.batch, 3, H, W, generator=g).to(dev, torch.float16)
    mask = torch.zeros(a.batch, H, W, device=dev, dtype=torch.float16)
    grabbed = {}
    orig = dec.forward_bf

    def spy(*args, **kw):
        grabbed["args"], grabbed["kw"] = args, kw
        return orig(*args, **kw)

    dec.forward_bf = spy
    with torch.no_grad():
        model(img, mask)
    dec.forward_bf = orig
    args, kw = grabbed["args"], grabbed["kw"]

    def run():
        with torch.no_grad():
            return orig(*args, **kw)

    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            run()
        before = dict(_cabi.CALLS)
        run()
        launches = sum(_cabi.CALLS.values()) - sum(before.values())
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=s):
            out = run()
        for _ in range(3):
            gr.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            gr.replay()
        e1.record()
        torch.cuda.synchronize()
    lib = _cabi.load()
    if hasattr(lib, "codetr_decoder_layer_debug_stamps"):   # -DCODETR_DEC_STAMPS build: phase boundaries of the last launch
        import ctypes
        buf = (ctypes.c_ulonglong * 32)()
        lib.codetr_decoder_layer_debug_stamps(buf)
        t = list(buf)
        print("stamps of a tail + head launch (clock ticks since the kernel's first stamp):", [int(v - t[0]) for v in t[:14]])
        print("stamps of the last (tail-only) launch:", [int(v - t[16]) for v in t[16:26]])
    print(f"decoder batch {a.batch} {a.res}: {e0.elapsed_time(e1) / a.iters * 1e3:.1f} us per replay, {launches} C-ABI launches")


if __name__ == "__main__":
    main()
