#!/usr/bin/env python3
"""Ablation timing of the windowed op kernel (csrc/msda_op4.hip): diagnostic builds with parts compiled out
(-DMSDA_OP4_ABL=mask: 2 no staging, 4 no gather, 8 no preparation, 32 no output stores; WRONG results by construction),
each a shared object holding only that file, called through its own codetr_msda_op4_forward_f16.
    for m in 0 2 4 6 8 32; do hipcc $(make -s -C co-detr-tensorrt_amd/csrc print-flags) -shared -DMSDA_OP4_ABL=$m \
        co-detr-tensorrt_amd/csrc/msda_op4.hip -o tools/micro/_bin/libop4_abl$m.so; done
    python tools/bench_msda_op4_abl.py tools/micro/_bin/libop4_abl*.so"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from bench_msda_op import D, L, M, P, inputs, pyramid  # noqa: E402

spread = float(os.environ.get("SPREAD", "2"))
shapes = pyramid(1280, 1920)
value, ss, ls, loc, w, S = inputs(1, shapes, sum(h * w_ for h, w_ in shapes), spread, "cuda:0")
out = torch.empty(1, S, M * D, dtype=torch.float16, device="cuda:0")
st = torch.cuda.current_stream()
for path in sys.argv[1:]:
    lib = ctypes.CDLL(os.path.abspath(path))
    fn = lib.codetr_msda_op4_forward_f16
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p] * 6 + [ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int64,
                                            ctypes.c_int, ctypes.c_void_p]

    def run():
        return fn(st.cuda_stream, value.data_ptr(), ss.data_ptr(), ls.data_ptr(), loc.data_ptr(), w.data_ptr(), 1, S, M, D, L, S, P,
                  out.data_ptr())

    for _ in range(3):
        assert run() == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(20):
        run()
    e1.record(st)
    torch.cuda.synchronize()
    print(f"{os.path.basename(path):24s} {e0.elapsed_time(e1) / 20 * 1e3:8.1f} us  (spread {spread} px)", flush=True)
