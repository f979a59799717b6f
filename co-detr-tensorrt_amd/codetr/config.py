"""Minimal loader for mmengine-style Python config files (no mmengine / mmdet dependency).

Covers exactly what the reference's configs use (reference configs/*.py, loaded by
``Config.fromfile`` at codetr/codetr.py:153):

* ``_base_ = 'file.py'`` or a list of files, resolved relative to the including file;
* ``mmdet::...`` bases (configs lsj:1 inherits ``mmdet::common/ssj_scp_270k_coco-instance.py``,
  which is not in the reference tree): resolved against ``configs/_mmdet_base_/`` next to this
  package, a tiny vendored stand-in that provides the few names the children read
  (``backend_args``, ``dataset_type``, ``data_root``, the dataloader / schedule dicts);
* ``_base_.name`` references inside a child file (configs r50:9, 49-55);
* recursive dict merge with ``_delete_=True`` (configs swin:9);
* attribute access on the result (``cfg.model.backbone.type``), ``.get`` / ``.pop`` / ``del``.
"""
import ast
import copy
import os

__all__ = ["Config", "ConfigDict"]


class ConfigDict(dict):
    """dict with attribute access.  Missing attributes raise AttributeError (not KeyError) so
    ``copy.deepcopy`` / ``hasattr`` behave."""

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)

    def __setattr__(self, name, value):
        self[name] = value

    def __delattr__(self, name):
        try:
            del self[name]
        except KeyError:
            raise AttributeError(name)

    def __deepcopy__(self, memo):
        return ConfigDict({k: copy.deepcopy(v, memo) for k, v in self.items()})


def _wrap(obj):
    if isinstance(obj, dict):
        return ConfigDict({k: _wrap(v) for k, v in obj.items()})
    if isinstance(obj, list):
        return [_wrap(v) for v in obj]
    if isinstance(obj, tuple):
        return tuple(_wrap(v) for v in obj)
    return obj


def _merge(base, child):
    """child overrides base, dicts merge recursively, `_delete_=True` replaces instead of merging."""
    out = dict(base)
    for k, v in child.items():
        if isinstance(v, dict):
            v = dict(v)
            delete = v.pop("_delete_", False)
            if not delete and isinstance(out.get(k), dict):
                out[k] = _merge(out[k], v)
            else:
                out[k] = _merge({}, v)
        else:
            out[k] = v
    return out


_MMDET_BASE_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "configs", "_mmdet_base_")


def _resolve(base, here):
    if base.startswith("mmdet::"):
        return os.path.normpath(os.path.join(_MMDET_BASE_DIR, base[len("mmdet::"):]))
    return os.path.normpath(os.path.join(here, base))


def _load_file(path):
    path = os.path.abspath(path)
    if not os.path.isfile(path):
        raise FileNotFoundError(path)
    here = os.path.dirname(path)
    src = open(path).read()
    tree = ast.parse(src, filename=path)
    # lift the `_base_ = ...` statement out of the module (a str or list literal); what remains may
    # refer to `_base_.name`, which must see the MERGED base config, not the path string
    base_names = None
    body = []
    for node in tree.body:
        if (isinstance(node, ast.Assign) and len(node.targets) == 1 and isinstance(node.targets[0], ast.Name)
                and node.targets[0].id == "_base_"):
            base_names = ast.literal_eval(node.value)
        else:
            body.append(node)
    tree.body = body
    merged_base = {}
    if base_names is not None:
        for b in [base_names] if isinstance(base_names, str) else list(base_names):
            merged_base = _merge(merged_base, _load_file(_resolve(b, here)))
    ns = {"_base_": _wrap(copy.deepcopy(merged_base))}
    exec(compile(tree, path, "exec"), ns)
    own = {k: v for k, v in ns.items() if not k.startswith("__") and k != "_base_" and not callable(v)
           and not type(v).__name__ == "module"}
    return _merge(merged_base, _unwrap(own))


def _unwrap(obj):
    if isinstance(obj, dict):
        return {k: _unwrap(v) for k, v in obj.items()}
    if isinstance(obj, list):
        return [_unwrap(v) for v in obj]
    if isinstance(obj, tuple):
        return tuple(_unwrap(v) for v in obj)
    return obj


class Config(ConfigDict):
    """``Config.fromfile(path)`` -> attribute-accessible nested dict of the merged config."""

    @classmethod
    def fromfile(cls, filename):
        cfg = cls(_wrap(_load_file(filename)))
        dict.__setattr__(cfg, "_filename", os.path.abspath(filename))
        return cfg

    @property
    def filename(self):
        return dict.__getattribute__(self, "__dict__").get("_filename")
