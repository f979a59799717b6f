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


def _row_error_terms(actual, ref, rel_l2):
    a = np.asarray(actual, dtype=np.float64).reshape(-1, np.asarray(actual).shape[-1])
    r = np.asarray(ref, dtype=np.float64).reshape(a.shape)
    fin = np.isfinite(r) & np.isfinite(a)
    d = np.where(fin, a - r, 0.0)
    rz = np.where(fin, r, 0.0)
    row_norm = np.sqrt((rz * rz).sum(-1))
    rms_row = np.sqrt((row_norm ** 2).mean())
    rms_el = np.sqrt((rz * rz).sum() / max(fin.sum(), 1))
    ratio = np.sqrt((d * d).sum(-1)) / (rel_l2 * np.maximum(row_norm, max(rms_row, 1e-30)))
    out = np.abs(d) > 4.0 * rel_l2 * (np.abs(rz) + rms_el)
    return ratio, out, fin


def row_error_stats(actual, ref, rel_l2):
    """Per-row companions of the whole-tensor relative L2 error, for [rows, C] samples of a stage (a tile-edge bug that
    corrupts a handful of rows moves the whole-tensor number by nothing):
      row_ratio  = max over rows of ||a_row - r_row|| / (rel_l2 * max(||r_row||, rms row norm of the tensor)),
      elem_frac  = max over rows of the fraction of the row's elements with |a - r| > 4 rel_l2 (|r| + rms element)
                   (rows narrower than 32 elements: that fraction over the whole tensor; assert_rows_close does NOT
                   compare it with a share there, see narrow_row_outliers).
    Non-finite elements must agree in position (checked by assert_close_lowp) and are left out."""
    ratio, out, fin = _row_error_terms(actual, ref, rel_l2)
    if out.shape[-1] >= 32:
        frac = out.sum(-1) / np.maximum(fin.sum(-1), 1)
    else:
        frac = np.asarray(out.sum() / max(fin.sum(), 1))
    return float(ratio.max()), float(frac.max()), int(ratio.argmax())


def narrow_row_outliers(actual, ref, rel_l2):
    """Rows of a few elements (box coordinates [rows, 4]): one element is already 25 % of its row, so a per-row SHARE says
    nothing and a share over the whole tensor would let dozens of coordinates through.  Counted instead:
    (outlier elements in the whole tensor, the largest number of outliers any one row holds)."""
    _, out, _ = _row_error_terms(actual, ref, rel_l2)
    return int(out.sum()), int(out.sum(-1).max()) if out.size else 0


NARROW_MAX_OUTLIERS = 3      # elements of the whole tensor (3 600 box coordinates in the headline test), never a share


def assert_rows_close(actual, ref, rel_l2, what="", row_factor=5.0, elem_frac=0.01):
    """assert_close_lowp plus the per-row bound: no row's error above row_factor x the tensor's bound (relative to
    the larger of its own norm and the tensor's rms row norm), no row with more than elem_frac of its elements beyond
    4 rel_l2 (|ref| + rms).  Rows narrower than 32 elements (boxes): at most NARROW_MAX_OUTLIERS such elements in the
    WHOLE tensor and never two in one row -- an absolute count (round 5 compared a 1 % / 5 % share here, which let
    36 - 180 box coordinates through).  The lone coordinates this admits are order-of-summation
    differences between a batch-of-4 and a single-image launch landing on a box whose fp16 offset rounds the other
    way; a row with two wrong coordinates is a wrong box and fails.
    Returns (tensor rel-L2, worst row ratio / row_factor, worst element fraction)."""
    err = assert_close_lowp(actual, ref, rel_l2, None, what)
    ratio, frac, worst = row_error_stats(actual, ref, rel_l2)
    assert ratio <= row_factor, f"{what}: row {worst} is {ratio:.2f} x the tensor bound {rel_l2:.1e} (limit {row_factor})"
    if np.asarray(actual).shape[-1] >= 32:
        assert frac <= elem_frac, f"{what}: a row has {100 * frac:.1f} % of its elements beyond 4 x {rel_l2:.1e} (|ref| + rms)"
    else:
        total, per_row = narrow_row_outliers(actual, ref, rel_l2)
        assert total <= NARROW_MAX_OUTLIERS and per_row <= 1, (
            f"{what}: {total} elements beyond 4 x {rel_l2:.1e} (|ref| + rms), up to {per_row} in one row "
            f"(limits {NARROW_MAX_OUTLIERS} / 1)")
    return err, ratio / row_factor, frac


def valid_topk(enc_cls, enc_coord, k, bound=None):
    """Proposal selection for parity runs on random weights: the reference's rule (top-k of the max class
    logit, reference transformer.py:560) restricted to positions whose proposal is finite.  With
    trained weights padded positions never win; with random weights they can, and their NaN box
    (log of a negative number, reference transformer.py:338) then poisons the whole image through
    the decoder's self-attention -- a property of random weights, not of either implementation."""
    import torch

    score = enc_cls.max(-1)[0].clone()
    score[~torch.isfinite(enc_coord).all(-1)] = -float("inf")
    if bound is not None:
        # also drop positions whose proposal was replaced by finfo.max (outside (-4.6, 4.6) or on padding, reference
        # transformer.py:365-380): finite in fp32, +inf once an fp16 model adds the box branch to 65504
        score[(enc_coord.abs() > bound).any(-1)] = -float("inf")
    return torch.topk(score, k, dim=1)[1]


def unmatched_detections(own, expected, score_tol=2e-6, box_tol=1e-3, tie_gap=1e-5):
    """Detections of `expected` = (boxes [K,4], scores [K], labels [K]) that `own` (same layout) does not contain.

    Both sides are top-k selections over the same kind of score table, possibly evaluated by different devices
    (sigmoid differs by an ulp between CPU and GPU), so membership is by tolerance -- same label, |score diff| <=
    score_tol, |box diff| <= box_tol pixels -- never by rounding to a grid (a value next to a rounding boundary
    would land in different cells on the two sides).  Expected detections whose score lies within `tie_gap` of
    another expected score or of the selection threshold are skipped: WHICH member of a (near-)tie wins the top-k is
    implementation-defined.  Non-finite rows (padded proposals, see valid_topk) are skipped on both sides."""
    import torch

    ob, os_, ol = (torch.as_tensor(t).detach().cpu() for t in own)
    eb, es, el = (torch.as_tensor(t).detach().cpu() for t in expected)
    ob, os_, eb, es = ob.double(), os_.double(), eb.double(), es.double()
    ok_o = torch.isfinite(ob).all(-1) & torch.isfinite(os_)
    ok_e = torch.isfinite(eb).all(-1) & torch.isfinite(es)
    fin = es[ok_e]
    thresh = fin.min() if fin.numel() else torch.tensor(0.0, dtype=torch.float64)
    missing = []
    for i in range(es.shape[0]):
        if not ok_e[i]:
            continue
        gap = (es - es[i]).abs()
        gap[i] = float("inf")
        if gap[ok_e].min() <= tie_gap or es[i] - thresh <= tie_gap:
            continue
        hit = ok_o & (ol == el[i]) & ((os_ - es[i]).abs() <= score_tol) & ((ob - eb[i]).abs().max(-1)[0] <= box_tol)
        if not bool(hit.any()):
            missing.append((float(es[i]), int(el[i]), [float(v) for v in eb[i]]))
    return missing


def poison_allocator(device, total_mb=512):
    """Fill the caching allocator's free lists with NaN bit patterns, so that the next torch.empty of (almost) any
    size hands out NaNs: a kernel that leaves part of its output or workspace unwritten then shows up as NaN / as a
    run-to-run difference instead of silently reading a previous run's (correct) values."""
    import torch

    held, used = [], 0
    for s in [1 << k for k in range(9, 27)] * 3:   # 512 B .. 64 MiB blocks
        if used + s > total_mb << 20:
            continue
        t = torch.empty(s // 4, dtype=torch.float32, device=device)
        t.fill_(float("nan"))
        held.append(t)
        used += s
    torch.cuda.synchronize(device)
    del held


def detection_agreement(cap, cap_o, Himg, Wimg):
    """Detection-level distance between a product run and the oracle run with the SAME proposal selection: for every
    query the decoded box (pixels, xyxy) and for every (query, class) the sigmoid score -- the quantities the final
    top-k / NMS / AP consume.  Returns mean and 95th-percentile absolute box error in pixels and score error, plus the
    fraction of the oracle's 300 highest (query, class) scores whose pair is also in the product's 300 highest."""
    import torch

    def boxes(c):
        cx, cy, w, h = c.float().cpu().unbind(-1)
        s = torch.tensor([Wimg, Himg, Wimg, Himg], dtype=torch.float32)
        return torch.stack((cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h), -1) * s

    db = (boxes(cap["outputs_coords"]) - boxes(cap_o["outputs_coords"])).abs()
    sa, so = cap["outputs_classes"].float().cpu().sigmoid(), cap_o["outputs_classes"].float().cpu().sigmoid()
    ds = (sa - so).abs()
    B = sa.shape[0]
    ta = torch.topk(sa.reshape(B, -1), 300, dim=1)[1]
    to = torch.topk(so.reshape(B, -1), 300, dim=1)[1]
    common = sum(len(set(a.tolist()) & set(o.tolist())) for a, o in zip(ta, to)) / (300.0 * B)
    fin = torch.isfinite(db)
    return {"box_err_px_mean": float(db[fin].mean()), "box_err_px_p95": float(db[fin].quantile(0.95)),
            "score_err_mean": float(ds.mean()), "score_err_p95": float(ds.flatten()[::7].quantile(0.95)),
            "top300_pairs_in_common": common}
