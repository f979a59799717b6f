"""ORACLE -- TEST INFRASTRUCTURE ONLY (not part of the product path).

Python face of the CPU restatement of the reference's multi-scale deformable attention
forward.  Two independent restatements live here:

* ``msda_forward_c``      -- ctypes call into ``oracle/_build/libmsda_ref.so`` (msda_ref.c: a
  per-output-scalar loop following reference codetr/csrc/ms_deform_attn.cu:31-77, 211-261,
  899-956).  Fast enough to serve as ``bench.py``'s ``cpu_baseline`` ("port", OpenMP threads).
* ``msda_forward_numpy``  -- a vectorised numpy gather formulation of the same arithmetic
  (same pixel-coordinate transform cu:246-247, range gate cu:249, per-corner zero padding
  cu:52-71), written without looking at the loop structure so the two check each other.

Pinning (see msda_ref.c header): both are checked against outputs of the reference's own
Python op (codetr/ops.py:129-186) captured into tests/golden/*.npz by make_golden.py, and
live against the imported reference when /root/reference is present.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg may import
this module.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libmsda_ref.so")
_lib = None


def build(force: bool = False) -> str:
    """Compile msda_ref.c with gcc (idempotent). Returns the .so path."""
    src = os.path.join(_HERE, "msda_ref.c")
    if force or not os.path.isfile(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


def _load():
    global _lib
    if _lib is None:
        build()
        lib = ctypes.CDLL(_SO)
        i64, i32, vp = ctypes.c_int64, ctypes.c_int, ctypes.c_void_p
        for name in ("msda_ref_forward_f32", "msda_ref_forward_f64"):
            fn = getattr(lib, name)
            fn.restype = ctypes.c_int
            fn.argtypes = [vp, vp, vp, vp, vp, i64, i64, i32, i32, i32, i64, i32, i64, vp]
        _lib = lib
    return _lib


def _as(a, dt):
    return np.ascontiguousarray(np.asarray(a), dtype=dt)


def msda_forward_c(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step=64,
                   dtype=np.float32):
    """value [B,S,M,D]; spatial_shapes [L,2] (h,w); level_start_index [L];
    sampling_loc [B,Nq,M,L,P,2] (x,y in [0,1]); attn_weight [B,Nq,M,L,P] -> [B,Nq,M*D].

    ``dtype`` float32 or float64 selects the arithmetic type (inputs are converted to it)."""
    lib = _load()
    dt = np.dtype(dtype)
    assert dt in (np.dtype(np.float32), np.dtype(np.float64))
    v = _as(value, dt)
    loc = _as(sampling_loc, dt)
    w = _as(attn_weight, dt)
    ss = _as(spatial_shapes, np.int64)
    ls = _as(level_start_index, np.int64)
    B, S, M, D = v.shape
    _, Nq, _, L, P, _ = loc.shape
    out = np.empty((B, Nq, M * D), dtype=dt)
    fn = lib.msda_ref_forward_f32 if dt == np.float32 else lib.msda_ref_forward_f64
    rc = fn(v.ctypes.data, ss.ctypes.data, ls.ctypes.data, loc.ctypes.data, w.ctypes.data,
            B, S, M, D, L, Nq, P, int(im2col_step), out.ctypes.data)
    if rc != 0:
        # reference: AT_ASSERTM(batch % im2col_step_ == 0, ...) (cu:924-926)
        raise ValueError(f"batch({B}) must divide im2col_step({min(B, im2col_step)})")
    return out


def msda_forward_numpy(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, dtype=np.float64):
    """Vectorised gather restatement (second, independent formulation)."""
    dt = np.dtype(dtype)
    v = np.asarray(value, dtype=dt)
    loc = np.asarray(sampling_loc, dtype=dt)
    w = np.asarray(attn_weight, dtype=dt)
    ss = np.asarray(spatial_shapes, dtype=np.int64)
    ls = np.asarray(level_start_index, dtype=np.int64)
    B, S, M, D = v.shape
    _, Nq, _, L, P, _ = loc.shape
    out = np.zeros((B, Nq, M, D), dtype=dt)
    b_idx = np.arange(B)[:, None, None, None]
    m_idx = np.arange(M)[None, None, :, None]
    half = dt.type(0.5)
    for l in range(L):
        H, W = int(ss[l, 0]), int(ss[l, 1])
        x = loc[:, :, :, l, :, 0] * dt.type(W) - half  # [B,Nq,M,P]   (cu:247)
        y = loc[:, :, :, l, :, 1] * dt.type(H) - half  # (cu:246)
        gate = (y > -1) & (x > -1) & (y < H) & (x < W)  # (cu:249)
        x0 = np.floor(x).astype(np.int64)
        y0 = np.floor(y).astype(np.int64)
        lx = x - x0.astype(dt)
        ly = y - y0.astype(dt)
        acc = np.zeros((B, Nq, M, P, D), dtype=dt)
        for dy, dx, cw in ((0, 0, (1 - ly) * (1 - lx)), (0, 1, (1 - ly) * lx), (1, 0, ly * (1 - lx)), (1, 1, ly * lx)):
            yy, xx = y0 + dy, x0 + dx
            ok = gate & (yy >= 0) & (yy <= H - 1) & (xx >= 0) & (xx <= W - 1)  # (cu:52-71)
            flat = ls[l] + np.clip(yy, 0, H - 1) * W + np.clip(xx, 0, W - 1)
            samp = v[b_idx, flat, m_idx, :]  # [B,Nq,M,P,D]
            acc += np.where(ok, cw, 0)[..., None] * samp
        out += (acc * w[:, :, :, l, :, None]).sum(axis=3)
    return out.reshape(B, Nq, M * D)


def level_start_index_from_shapes(spatial_shapes):
    ss = np.asarray(spatial_shapes, dtype=np.int64)
    counts = ss[:, 0] * ss[:, 1]
    return np.concatenate([[0], np.cumsum(counts)[:-1]]).astype(np.int64)
