"""Generate the golden fixtures in tests/golden/ by running the REFERENCE's own Python.

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden.py            # all groups
    python tests/golden/make_golden.py msda       # one group

Every fixture is data: seeded inputs + the outputs the reference produced for them.  The
reference never travels to the GPU box; these .npz files do.  Recipes follow the reference's
tests:

  msda_g1  tests/test_multi_scale_deformable_attention.py:246-364  (seed 3, N=1,M=2,D=2,Lq=2,L=2,P=2)
  msda_g2  tests/test_multi_scale_deformable_attention.py:14-62    (B=2,M=4,Nq=8,D=16,L=3,P=4, 32/16/8 squares)
  msda_g3  model-shaped slice: M=8,D=32,L=5,P=4 on the pyramid of a 64x96 image, Nq=64 and
           Nq=S, locations in [-0.1,1.1) incl. exact borders
  msda_g4  perf-test shape tests/...:417-501 (N=1,M=8,D=64,Lq=100,L=4,P=4, seed 42; squares 16..2 instead of 64..8)
  model_*  module-level captures (see make_model_goldens)
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_import as R  # noqa: E402


def _lsi(shapes):
    return torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))


def _save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = v
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.1f} KiB)")


def make_msda_goldens():
    ops = R.ref("ops")
    f = ops.multi_scale_deformable_attention_pytorch

    # ---- G1: reference tests :246-364 ----
    N, M, D, Lq, L, P = 1, 2, 2, 2, 2, 2
    shapes = torch.as_tensor([(6, 4), (3, 2)], dtype=torch.long)
    S = int(shapes.prod(1).sum())
    torch.manual_seed(3)
    value = torch.rand(N, S, M, D) * 0.01
    loc = torch.rand(N, Lq, M, L, P, 2)
    w = torch.rand(N, Lq, M, L, P) + 1e-5
    w /= w.sum(-1, keepdim=True).sum(-2, keepdim=True)
    _save(
        "msda_g1",
        value=value, spatial_shapes=shapes, level_start_index=_lsi(shapes), sampling_loc=loc, attn_weight=w,
        out_f64=f(value.double(), shapes, loc.double(), w.double()),
        out_f32=f(value, shapes, loc, w),
        out_f16=f(value.half(), shapes, loc.half(), w.half()).float(),
    )

    # ---- G2: reference tests :14-62 (fixed seed added) ----
    torch.manual_seed(1234)
    B, M, Nq, D, L, P = 2, 4, 8, 16, 3, 4
    shapes = torch.tensor([[32, 32], [16, 16], [8, 8]], dtype=torch.long)
    S = int(shapes.prod(1).sum())
    value = torch.rand(B, S, M, D)
    loc = torch.rand(B, Nq, M, L, P, 2)
    w = torch.rand(B, Nq, M, L, P)
    _save(
        "msda_g2",
        value=value.half(), spatial_shapes=shapes, level_start_index=_lsi(shapes), sampling_loc=loc.half(),
        attn_weight=w.half(),
        # fp16-representable inputs; outputs from the reference in fp64 / fp32 / fp16
        out_f64=f(value.half().double(), shapes, loc.half().double(), w.half().double()),
        out_f32=f(value.half().float(), shapes, loc.half().float(), w.half().float()),
        out_f16=f(value.half(), shapes, loc.half(), w.half()).float(),
    )

    # ---- G3: model-shaped slice with border stress ----
    torch.manual_seed(42)
    M, D, L, P = 8, 32, 5, 4
    shapes = torch.tensor([[16, 24], [8, 12], [4, 6], [2, 3], [1, 2]], dtype=torch.long)  # 64x96 image pyramid
    S = int(shapes.prod(1).sum())
    for tag, B, Nq in (("dec", 2, 64), ("enc", 1, S)):
        value = torch.rand(B, S, M, D).half()
        loc = (torch.rand(B, Nq, M, L, P, 2) * 1.2 - 0.1)
        # exact-border and exact-pixel-centre locations
        loc[:, 0] = 0.0
        loc[:, 1] = 1.0
        loc[:, 2] = 0.5
        loc[:, 3, :, :, :, 0] = 0.0
        loc[:, 4, :, :, :, 1] = 1.0
        loc[:, 5] = -0.1
        loc[:, 6] = 1.1
        for l in range(L):  # pixel centres of level l: (i + 0.5) / W
            loc[:, 7, :, l, :, 0] = (torch.arange(P) % shapes[l, 1] + 0.5) / shapes[l, 1]
            loc[:, 7, :, l, :, 1] = (torch.arange(P) % shapes[l, 0] + 0.5) / shapes[l, 0]
        loc = loc.half()
        w = torch.rand(B, Nq, M, L, P)
        w = (w / w.sum((-1, -2), keepdim=True)).half()
        _save(
            f"msda_g3_{tag}",
            value=value, spatial_shapes=shapes, level_start_index=_lsi(shapes), sampling_loc=loc, attn_weight=w,
            out_f64=f(value.double(), shapes, loc.double(), w.double()).float(),  # stored as f32 to stay small
            out_f32=f(value.float(), shapes, loc.float(), w.float()),
        )

    # ---- G4: reference perf-test shape :417-501 ----
    torch.manual_seed(42)
    N, M, D, Lq, L, P = 1, 8, 64, 100, 4, 4
    shapes = torch.tensor([[16, 16], [8, 8], [4, 4], [2, 2]], dtype=torch.long)  # reference uses 64..8; scaled /4 to keep the fixture small
    S = int(shapes.prod(1).sum())
    value = torch.rand(N, S, M, D).half()
    loc = torch.rand(N, Lq, M, L, P, 2).half()
    w = torch.rand(N, Lq, M, L, P)
    w = (w / w.sum((-1, -2), keepdim=True)).half()
    _save(
        "msda_g4",
        value=value, spatial_shapes=shapes, level_start_index=_lsi(shapes), sampling_loc=loc, attn_weight=w,
        out_f32=f(value.float(), shapes, loc.float(), w.float()),
    )


GROUPS = {"msda": make_msda_goldens}


if __name__ == "__main__":
    if not R.reference_available():
        sys.exit("reference tree not found at " + R.REFERENCE_ROOT)
    torch.set_grad_enabled(False)
    torch.set_num_threads(8)
    which = sys.argv[1:] or list(GROUPS)
    for g in which:
        GROUPS[g]()
