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


def assert_close_lowp(actual, ref, rel_l2=1e-2, max_abs=None, what=""):
    """Model-level closeness against an fp32 oracle: relative L2 error of the whole tensor (robust to
    the few elements that land next to a rounding boundary after many layers) plus an optional cap on
    the worst element, expressed as a fraction of max|ref| when `max_abs` is given.

    NaN-aware: the reference's proposal formula yields NaN at padded positions (log of a negative
    number times a zero mask, reference transformer.py:338, 379); those must be NaN on both sides
    and are excluded from the norms."""
    a = np.asarray(actual, dtype=np.float64)
    r = np.asarray(ref, dtype=np.float64)
    assert a.shape == r.shape, (a.shape, r.shape)
    bad_a, bad_r = ~np.isfinite(a), ~np.isfinite(r)
    assert np.array_equal(bad_a, bad_r), f"{what}: non-finite pattern differs ({bad_a.sum()} vs {bad_r.sum()})"
    a, r = a[~bad_r], r[~bad_r]
    err = np.linalg.norm(a - r) / max(np.linalg.norm(r), 1e-30)
    assert err <= rel_l2, f"{what}: relative L2 error {err:.3e} > {rel_l2:.1e}"
    if max_abs is not None:
        worst = np.abs(a - r).max() / max(np.abs(r).max(), 1e-30)
        assert worst <= max_abs, f"{what}: max abs error {worst:.3e} x max|ref| > {max_abs:.1e}"
    return err


def valid_topk(enc_cls, enc_coord, k):
    """Proposal selection for parity runs on random weights: the reference's rule (top-k of the max class
    logit, reference transformer.py:560) restricted to positions whose proposal is finite.  With
    trained weights padded positions never win; with random weights they can, and their NaN box
    (log of a negative number, reference transformer.py:338) then poisons the whole image through
    the decoder's self-attention -- a property of random weights, not of either implementation."""
    import torch

    score = enc_cls.max(-1)[0].clone()
    score[~torch.isfinite(enc_coord).all(-1)] = -float("inf")
    return torch.topk(score, k, dim=1)[1]
