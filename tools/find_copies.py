"""Which host lines of the package allocate-and-copy (contiguous / clone / cat / to / masked_fill / index ...) inside one
steady-state CoDETR.forward: a TorchFunctionMode over one eager forward, reporting every call whose result does not share
storage with an input.  Round 5 at 608x608: none (57 `contiguous()` calls, all on contiguous tensors; 271 reshapes, all
views) -- the `__amd_rocclr_copyBuffer` rows of the kernel traces are the parameter uploads of `model.to(device)` and the
bench loop's own copies, not the forward.     python tools/find_copies.py [WxH]"""
import os
import sys
from collections import Counter

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "co-detr-tensorrt_amd"))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

W, H = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "608x608").split("x"))
model = bench.build_model("cuda", torch.float16)
x = torch.randn(1, 3, H, W, device="cuda").half()
m = torch.zeros(1, H, W, device="cuda").half()
import traceback
from torch.overrides import TorchFunctionMode

sites = Counter()
ALL = Counter()
WATCH = ("copy_", "clone", "contiguous", "cat", "stack", "to", "float", "half", "repeat", "expand_as", "masked_fill", "index_select",
         "gather", "__getitem__", "__setitem__", "flatten", "reshape", "permute", "transpose")


class Spy(TorchFunctionMode):
    def __torch_function__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = getattr(func, "__name__", str(func))
        ALL[name] += 1
        if name in ("copy_", "clone", "contiguous", "cat", "stack", "to", "float", "half", "repeat", "masked_fill", "index_select",
                    "gather", "__setitem__", "__getitem__", "reshape"):
            # does it allocate / copy?  (a view shares storage with an input)
            ins = [a for a in args if isinstance(a, torch.Tensor)]
            fresh = isinstance(out, torch.Tensor) and out.is_cuda and all(out.untyped_storage().data_ptr() != a.untyped_storage().data_ptr() for a in ins)
            if fresh or name in ("copy_", "__setitem__"):
                fr = [f for f in traceback.extract_stack() if "/codetr/" in f.filename]
                site = f"{os.path.basename(fr[-1].filename)}:{fr[-1].lineno}" if fr else "?"
                nbytes = out.numel() * out.element_size() if isinstance(out, torch.Tensor) else 0
                sites[(name, site, nbytes)] += 1
        return out


with torch.no_grad():
    for _ in range(2):
        model(x, m)
    torch.cuda.synchronize()
    with Spy():
        model(x, m)
    torch.cuda.synchronize()
print("intercepted:", sum(ALL.values()), ALL.most_common(25))
for (name, site, nb), n in sorted(sites.items(), key=lambda kv: -kv[1] * 1)[:70]:
    print(f"{n:4d} {name:14s} {nb:10d} B  {site}")
