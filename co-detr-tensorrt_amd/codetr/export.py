"""Launch-plan export: the MI355X counterpart of the reference's ``export.py`` -> TensorRT engine -> C++ runner
(``codetr_inference.cpp:322-438``) flow, without a tracing compiler.

The fp16 forward of ``CoDETR`` is, in the steady state, nothing but a fixed sequence of ``libcodetr_hip.so`` launches on
one stream (tests/test_small_ops_gpu.py asserts that no library kernel is left on it).  ``export_plan`` records that
sequence once -- entry-point names and raw arguments -- together with the device memory it runs in, and writes a
self-contained *plan* file:

  * the caching allocator's segments the launches touch (sizes only) and every device pointer rebased to
    (segment, offset), so a runner can re-create the same layout with a handful of ``hipMalloc`` calls;
  * the contents of the blocks that were live before the forward started: weights, derived / packed weights,
    shape-dependent constants and the inputs of the recorded run;
  * the launch list: name + arguments (integers, floats, device pointers, host byte arrays such as level shapes; the
    stream argument is marked so the runner substitutes its own);
  * where the inputs (``batch_inputs``, ``img_masks``) and outputs (``boxes``, ``scores``, ``labels``) live.

``runner/codetr_runner`` (C++, no Python, no PyTorch) loads the plan and ``libcodetr_hip.so``, replays the launches --
eagerly or captured once into a hipGraph -- and returns the detections.  A plan is specific to the model weights, the
input shape / batch and the library build (ABI version is recorded and checked)."""
import ctypes
import struct

import torch

from . import _cabi

MAGIC = b"CODETRPLAN\x00\x02"
import re

from ._kernel_names import KERNEL_NAMES

# WHITELIST: a device activity of the recorded forward is replayable iff it is one of libcodetr_hip.so's own kernels
# (their __global__ names, tools/gen_kernel_names.py; all live in an anonymous namespace).  Anything else -- ATen,
# rocBLAS / hipBLASLt, MIOpen, rocPRIM, a Triton kernel, a device copy, whatever a future PyTorch names its kernels --
# is foreign: the plan would skip it and replay on stale memory.
_OWN_DEMANGLED = re.compile(r"^(?:void )?\(anonymous namespace\)::(" + "|".join(KERNEL_NAMES) + r")\s*[<(]")
_OWN_MANGLED = re.compile(r"^_ZN12_GLOBAL__N_1(\d+)([A-Za-z_][A-Za-z0-9_]*)")


def is_own_kernel(name: str) -> bool:
    if _OWN_DEMANGLED.match(name):
        return True
    m = _OWN_MANGLED.match(name)
    return bool(m) and m.group(2)[:int(m.group(1))] in KERNEL_NAMES
KIND_INT, KIND_FLOAT, KIND_DEV, KIND_NULL, KIND_HOST, KIND_STREAM = 0, 1, 2, 3, 4, 5


def _hip():
    for name in ("libamdhip64.so", "libamdhip64.so.7", "libamdhip64.so.6"):
        try:
            return ctypes.CDLL(name)
        except OSError:
            continue
    raise RuntimeError("libamdhip64.so not found")


def _read_device(hip, addr, nbytes):
    buf = ctypes.create_string_buffer(nbytes)
    rc = hip.hipMemcpy(buf, ctypes.c_void_p(addr), ctypes.c_size_t(nbytes), 2)  # hipMemcpyDeviceToHost
    if rc != 0:
        raise RuntimeError(f"hipMemcpy D2H of {nbytes} bytes at {addr:#x} failed with {rc}")
    return buf.raw


def _segments(device):
    segs = []
    for s in torch.cuda.memory_snapshot():
        if s.get("device", 0) != device.index:
            continue
        blocks, addr = [], s["address"]
        for b in s["blocks"]:
            a = b.get("address", addr)
            blocks.append((a, b["size"], b["state"]))
            addr = a + b["size"]
        segs.append({"address": s["address"], "size": s["total_size"], "blocks": blocks})
    return sorted(segs, key=lambda s: s["address"])


@torch.no_grad()
def export_plan(model, batch_inputs, img_masks, path, warmup=2):
    """Record ``model(batch_inputs, img_masks)`` (fp16, on a HIP device) and write the plan to `path`.
    Returns a summary dict (launch count, bytes)."""
    if not batch_inputs.is_cuda or batch_inputs.dtype != torch.float16:
        raise ValueError("export_plan records the fp16 GPU path")
    # a plan must describe the default kernels: the host package's route switches (module attributes, see
    # hip_ops.nondefault_switches) select other launch lists, and a plan recorded under one would pin it for every replay
    from . import hip_ops

    ab = hip_ops.nondefault_switches()
    if ab:
        raise RuntimeError(f"export_plan refuses to record under A/B switches: reset {ab}")
    dev = batch_inputs.device
    batch_inputs, img_masks = batch_inputs.contiguous(), img_masks.contiguous()
    hip = _hip()
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    with torch.cuda.device(dev):
        for _ in range(warmup):          # shape-keyed caches, derived weights, allocator segments
            model(batch_inputs, img_masks)
        torch.cuda.synchronize(dev)
        live_before = [(a, n) for s in _segments(dev) for a, n, st in s["blocks"] if st == "active_allocated"]
        initial = [(a, n, _read_device(hip, a, n)) for a, n in live_before]
        _cabi.RECORDER = []
        try:
            # the plan can only hold libcodetr_hip.so launches: profile the recorded forward and refuse to export if
            # anything else ran on the device (an ATen kernel or a device copy -- e.g. PatchMerging's F.pad on an
            # odd-sized Swin map, or a dtype the small kernels do not take)
            with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA]) as prof:
                boxes, scores, labels = model(batch_inputs, img_masks)
                torch.cuda.synchronize(dev)
            calls = _cabi.RECORDER
        finally:
            _cabi.RECORDER = None
        foreign = sorted({e.name[:100] for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA
                          and not is_own_kernel(e.name)})
        if foreign:
            raise RuntimeError("this model / input shape is not exportable: the forward ran device work outside "
                               f"libcodetr_hip.so that a plan cannot replay: {foreign}")
        # ... and replay the recorded launches once in-process, on the same memory, with the outputs zeroed first: a launch
        # list that misses device work the recorder did not see cannot reproduce the recorded detections
        want = [t.cpu() for t in (boxes, scores, labels)]
        for t in (boxes, scores, labels):
            t.zero_()
        lib = _cabi.load()
        for name, args in calls:
            rc = getattr(lib, name)(*args)
            if rc:
                raise RuntimeError(f"replay of the recorded launch list failed in {name}: error {rc}")
        torch.cuda.synchronize(dev)
        for what, t, w in zip(("boxes", "scores", "labels"), (boxes, scores, labels), want):
            if not torch.equal(t.cpu(), w):
                raise RuntimeError(f"the recorded launch list does not reproduce the recorded {what}: not exportable")
        segs = _segments(dev)

    def locate(addr, what):
        for i, s in enumerate(segs):
            if s["address"] <= addr < s["address"] + s["size"]:
                return i, addr - s["address"]
        raise RuntimeError(f"{what}: device pointer {addr:#x} is outside every allocator segment")

    used = set()
    enc_calls = []
    for name, args in calls:
        argtypes = _cabi.SIGNATURES[name][1]
        enc = []
        for j, (a, t) in enumerate(zip(args, argtypes)):
            if j == 0:
                enc.append((KIND_STREAM, None))
            elif isinstance(a, (ctypes.Array, ctypes.Structure, ctypes._SimpleCData)) and t is _cabi._vp:
                enc.append((KIND_HOST, bytes(a)))
            elif t is _cabi._vp:
                if a is None or a == 0:
                    enc.append((KIND_NULL, None))
                else:
                    seg, off = locate(int(a), name)
                    used.add(seg)
                    enc.append((KIND_DEV, (seg, off)))
            elif t is ctypes.c_float:
                enc.append((KIND_FLOAT, float(a)))
            else:
                enc.append((KIND_INT, int(a)))
        enc_calls.append((name, enc))

    def io(name, t):
        seg, off = locate(t.data_ptr(), name)
        used.add(seg)
        return (name, seg, off, t.numel() * t.element_size(), str(t.dtype).replace("torch.", ""), tuple(t.shape))

    ios = [io("batch_inputs", batch_inputs), io("img_masks", img_masks), io("boxes", boxes.contiguous()),
           io("scores", scores.contiguous()), io("labels", labels.contiguous())]
    if not (boxes.is_contiguous() and scores.is_contiguous() and labels.is_contiguous()):
        raise RuntimeError("outputs must be contiguous to be addressable by the plan")
    blobs = []
    for a, n, data in initial:
        try:
            seg, off = locate(a, "initial block")
        except RuntimeError:
            continue
        if seg in used:
            blobs.append((seg, off, data))

    with open(path, "wb") as f:
        f.write(MAGIC)
        f.write(struct.pack("<iI", _cabi.ABI_VERSION, len(segs)))
        for i, s in enumerate(segs):
            f.write(struct.pack("<Q", s["size"] if i in used else 0))   # 0: not touched by the plan, not allocated
        f.write(struct.pack("<I", len(blobs)))
        for seg, off, data in blobs:
            f.write(struct.pack("<IQQ", seg, off, len(data)))
            f.write(data)
        f.write(struct.pack("<I", len(enc_calls)))
        for name, enc in enc_calls:
            nb = name.encode()
            f.write(struct.pack("<H", len(nb)) + nb + struct.pack("<B", len(enc)))
            for kind, v in enc:
                f.write(struct.pack("<B", kind))
                if kind == KIND_INT:
                    f.write(struct.pack("<q", v))
                elif kind == KIND_FLOAT:
                    f.write(struct.pack("<f", v))
                elif kind == KIND_DEV:
                    f.write(struct.pack("<IQ", *v))
                elif kind == KIND_HOST:
                    f.write(struct.pack("<I", len(v)) + v)
        f.write(struct.pack("<I", len(ios)))
        for name, seg, off, nbytes, dtype, shape in ios:
            nb, db = name.encode(), dtype.encode()
            f.write(struct.pack("<H", len(nb)) + nb + struct.pack("<H", len(db)) + db)
            f.write(struct.pack("<IQQB", seg, off, nbytes, len(shape)) + struct.pack(f"<{len(shape)}q", *shape))
    return {"launches": len(enc_calls), "segments": len(segs), "segments_used": len(used),
            "initial_bytes": sum(len(b[2]) for b in blobs), "outputs": {n: s for n, _, _, _, _, s in ios[2:]}}
