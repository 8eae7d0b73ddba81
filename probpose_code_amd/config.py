"""Minimal loader for MMPose-style python config files (reference: mmengine ``Config`` [3P], used at
``mmpose/apis/inference.py:88-93`` and ``tools/test.py:115``). Only what the inference path needs:
``Config.fromfile`` (python files, ``_base_`` inheritance with dict merge and ``_delete_``),
attribute/item access, ``merge_from_dict`` with dotted keys (``--cfg-options``), ``custom_imports``.
When mmengine is importable its own ``Config`` is used instead."""
import importlib
import os
from typing import Any, Dict

try:  # pragma: no cover
    from mmengine.config import Config, ConfigDict  # type: ignore  # noqa: F401

    USING_MMENGINE = True
except Exception:  # noqa: BLE001
    USING_MMENGINE = False

    class ConfigDict(dict):
        def __getattr__(self, name):
            try:
                return self[name]
            except KeyError as e:
                raise AttributeError(f"'ConfigDict' object has no attribute '{name}'") from e

        def __setattr__(self, name, value):
            self[name] = value

    def _wrap(x):
        if isinstance(x, dict):
            return ConfigDict({k: _wrap(v) for k, v in x.items()})
        if isinstance(x, list):
            return [_wrap(v) for v in x]
        if isinstance(x, tuple):
            return tuple(_wrap(v) for v in x)
        return x

    def _merge(base: dict, over: dict) -> dict:
        out = dict(base)
        for k, v in over.items():
            if isinstance(v, dict) and isinstance(out.get(k), dict) and not v.get("_delete_", False):
                out[k] = _merge(out[k], v)
            else:
                out[k] = {kk: vv for kk, vv in v.items() if kk != "_delete_"} if isinstance(v, dict) else v
        return out

    def _exec_file(path: str) -> Dict[str, Any]:
        ns: Dict[str, Any] = {"__file__": path}
        with open(path) as f:
            exec(compile(f.read(), path, "exec"), ns)  # noqa: S102 - config files are python by design
        cfg = {k: v for k, v in ns.items() if not k.startswith("__") and not callable(v) and not isinstance(v, type(os))}
        bases = cfg.pop("_base_", [])
        bases = [bases] if isinstance(bases, str) else list(bases)
        merged: Dict[str, Any] = {}
        for b in bases:
            merged = _merge(merged, _exec_file(os.path.normpath(os.path.join(os.path.dirname(path), b))))
        return _merge(merged, cfg)

    class Config:
        def __init__(self, cfg_dict: dict = None, filename: str = None):
            object.__setattr__(self, "_cfg_dict", _wrap(cfg_dict or {}))
            object.__setattr__(self, "filename", filename)

        @staticmethod
        def fromfile(filename: str, import_custom_modules: bool = True) -> "Config":
            filename = str(filename)
            if not os.path.isfile(filename):
                raise FileNotFoundError(f'file "{filename}" does not exist')
            if not filename.endswith(".py"):
                raise OSError("Only py type are supported now!")
            cfg = Config(_exec_file(os.path.abspath(filename)), filename)
            ci = cfg.get("custom_imports")
            if import_custom_modules and ci:
                for mod in ci.get("imports", []):
                    try:
                        importlib.import_module(mod)
                    except ImportError:
                        if not ci.get("allow_failed_imports", False):
                            raise
            return cfg

        def merge_from_dict(self, options: dict) -> None:
            for full_key, v in options.items():
                d = self._cfg_dict
                keys = full_key.split(".")
                for k in keys[:-1]:
                    d = d.setdefault(k, ConfigDict())
                d[keys[-1]] = _wrap(v)

        def get(self, key, default=None):
            return self._cfg_dict.get(key, default)

        def __getattr__(self, name):
            return getattr(self._cfg_dict, name)

        def __setattr__(self, name, value):
            self._cfg_dict[name] = _wrap(value)

        def __getitem__(self, name):
            return self._cfg_dict[name]

        def __contains__(self, name):
            return name in self._cfg_dict

        def to_dict(self):
            return dict(self._cfg_dict)
