import ctypes, os, sys, torch
sys.path.insert(0, "tools")
from bench_msda_op import D, L, M, P, inputs, pyramid
shapes = pyramid(1280, 1920)
st = torch.cuda.current_stream()
for Nq in (900, None):
    S0 = sum(h * w for h, w in shapes)
    value, ss, ls, loc, w, S = inputs(1, shapes, Nq or S0, 10.0 if Nq else 3.0, "cuda:0", seed=1)
    nq = Nq or S0
    out = torch.empty(1, nq, M * D, dtype=torch.float16, device="cuda:0")
    for rep in range(2):
        for path in sys.argv[1:]:
            lib = ctypes.CDLL(os.path.abspath(path))
            fn = lib.codetr_msda_forward_f16
            fn.restype = ctypes.c_int
            fn.argtypes = [ctypes.c_void_p] * 6 + [ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_int, ctypes.c_int64, ctypes.c_void_p]
            run = lambda: fn(st.cuda_stream, value.data_ptr(), ss.data_ptr(), ls.data_ptr(), loc.data_ptr(), w.data_ptr(), 1, S, M, D, L, nq, P, 64, out.data_ptr())
            for _ in range(3):
                assert run() == 0
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for _ in range(50):
                run()
            e1.record(st)
            torch.cuda.synchronize()
            print(f"Nq {nq:7d} {os.path.basename(path):18s} {e0.elapsed_time(e1) / 50 * 1e3:8.1f} us", flush=True)
