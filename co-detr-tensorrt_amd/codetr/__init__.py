"""codetr -- MI355X-native host package for the Co-DETR inference hot path.

Drop-in for the reference's ``codetr`` package on that path: same module / class names,
``torch.ops.codetr.multi_scale_deformable_attention`` and
``build_CoDETR`` / ``CoDETR.forward(batch_inputs, img_masks) -> (boxes, scores, labels)``.

Like the reference (codetr/__init__.py:8-12) the native library is loaded at import time and
its absence is an ImportError; unlike the reference there is one library
(``libcodetr_hip.so``, C ABI in include/codetr_hip.h) and no TensorRT plugin.
"""
from . import _cabi

_cabi.load()  # raises ImportError if the HIP extension has not been built

from . import ops  # noqa: E402,F401  (registers torch.ops.codetr.*)

try:  # model assembly (pure-Python host code on top of the ops)
    from .codetr import CoDETR, build_CoDETR  # noqa: E402,F401
except ModuleNotFoundError as _e:  # pragma: no cover - only while the tree is being bootstrapped
    if _e.name not in ("codetr.codetr",):
        raise
