"""Deterministic, reference-free parameter generation shared by the golden generator and the tests.

Golden fixtures for whole modules would be dominated by their weights (tens of MB), so fixtures
store only (ordered parameter names, shapes, seed); both sides rebuild the identical tensors here.
"""
import math

import numpy as np
import torch


def seeded_params(named_shapes, seed, scale=1.0, dtype=torch.float32):
    """named_shapes: ordered iterable of (name, shape).  One torch.Generator stream, drawn in order:
    matrices/kernels ~ N(0, scale^2 / fan_in); 1-d '...weight' (norm scales) ~ 1 + 0.1 N; other 1-d ~ 0.1 N."""
    g = torch.Generator().manual_seed(int(seed))
    out = {}
    for name, shape in named_shapes:
        shape = tuple(int(s) for s in shape)
        if len(shape) >= 2:
            fan_in = shape[1] if len(shape) == 2 else int(np.prod(shape[1:]))
            t = torch.randn(shape, generator=g) * (scale / math.sqrt(max(fan_in, 1)))
        elif name.endswith("weight"):
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)
        else:
            t = 0.1 * torch.randn(shape, generator=g)
        out[name] = t.to(dtype)
    return out


def pack_param_spec(named_shapes):
    """-> dict of small numpy arrays describing the ordered (name, shape) list (for np.savez)."""
    names = [n for n, _ in named_shapes]
    shapes = [tuple(s) for _, s in named_shapes]
    flat = np.array([d for s in shapes for d in s], dtype=np.int64)
    ranks = np.array([len(s) for s in shapes], dtype=np.int64)
    return {"spec.names": np.array(names), "spec.ranks": ranks, "spec.dims": flat}


def unpack_param_spec(npz):
    names = [str(n) for n in npz["spec.names"]]
    ranks = npz["spec.ranks"]
    dims = npz["spec.dims"]
    out, o = [], 0
    for n, r in zip(names, ranks):
        out.append((n, tuple(int(d) for d in dims[o:o + r])))
        o += r
    return out
