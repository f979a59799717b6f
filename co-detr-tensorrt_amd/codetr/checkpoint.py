"""mmdet-checkpoint loading: the counterpart of mmengine's ``_load_checkpoint`` +
``_load_checkpoint_to_model`` as used at reference codetr/codetr.py:165-166 -- non-strict, so the
published Co-DINO checkpoint's training-only keys (``rpn_head.*``, ``roi_head.*``, ``bbox_head.*``,
``query_head.label_embedding.weight``, ...) are reported and ignored."""
import warnings

import torch


def load_checkpoint(filename, map_location="cpu", allow_pickle=None):
    """Tensors-and-containers only by default (``weights_only=True``: an mmdet checkpoint needs nothing else --
    ``state_dict`` + a ``meta`` dict of strings / tuples).  A checkpoint that carries other pickled objects is refused
    unless the caller opts in with ``allow_pickle=True`` (or CODETR_ALLOW_PICKLE=1): unpickling arbitrary objects runs
    code from the file."""
    import os
    import pickle

    if allow_pickle is None:
        allow_pickle = os.environ.get("CODETR_ALLOW_PICKLE", "0") == "1"
    try:
        ckpt = torch.load(filename, map_location=map_location, weights_only=True)
    except (pickle.UnpicklingError, RuntimeError) as e:
        if not allow_pickle:
            raise RuntimeError(
                f"{filename} holds pickled objects beyond tensors and plain containers ({str(e).splitlines()[0]}); "
                "pass allow_pickle=True (or set CODETR_ALLOW_PICKLE=1) only for files you trust") from e
        ckpt = torch.load(filename, map_location=map_location, weights_only=False)
    if not isinstance(ckpt, dict):
        raise RuntimeError(f"No state_dict found in checkpoint file {filename}")
    return ckpt


def load_checkpoint_to_model(model, checkpoint, strict=False):
    state = checkpoint.get("state_dict", checkpoint.get("model", checkpoint))
    state = {(k[7:] if k.startswith("module.") else k): v for k, v in state.items()}
    own = model.state_dict()
    mismatched = [k for k, v in state.items() if k in own and tuple(own[k].shape) != tuple(v.shape)]
    for k in mismatched:
        state.pop(k)
    res = model.load_state_dict(state, strict=False)
    problems = []
    if res.missing_keys:
        problems.append(f"missing keys: {', '.join(res.missing_keys)}")
    if res.unexpected_keys:
        problems.append(f"unexpected keys: {', '.join(res.unexpected_keys)}")
    if mismatched:
        problems.append(f"size mismatch (skipped): {', '.join(mismatched)}")
    if problems:
        if strict:
            raise RuntimeError("; ".join(problems))
        warnings.warn("The model and loaded state dict do not match exactly: " + "; ".join(problems))
    return res
