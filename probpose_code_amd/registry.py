"""Registry surface of the drop-in (reference: ``mmpose/registry.py:50,92``).

The reference looks components up by their string ``type`` in mmengine registries
(``MODELS``, ``KEYPOINT_CODECS``). Two situations:

* **mmpose + mmengine importable** (a real MMPose install): the MI355X classes register
  into the real ``mmpose.registry`` registries, so a config whose ``custom_imports``
  names ``probpose_code_amd`` drives them from ``demo/image_demo.py`` / ``tools/test.py``
  unchanged. With ``PROBPOSE_MI355X_OVERRIDE=1`` they replace the reference classes
  under the reference's own names (``ProbMap``, ``ProbMapHead``, ``TopdownPoseEstimator``,
  ``mmpretrain.VisionTransformer``, ``PoseDataPreprocessor``) so the reference config
  file itself runs on the HIP path.
* **otherwise** (this container, the GPU box): a minimal ``Registry`` with the same
  ``register_module`` / ``build`` / ``get`` behaviour backs the same names.
"""
import inspect
import os
from typing import Any, Callable, Dict, Optional


class Registry:
    """The slice of ``mmengine.registry.Registry`` the hot path relies on."""

    def __init__(self, name: str, scope: str = "mmpose"):
        self.name = name
        self.scope = scope
        self._module_dict: Dict[str, type] = {}

    @property
    def module_dict(self):
        return self._module_dict

    def __contains__(self, key):
        return self.get(key) is not None

    def __len__(self):
        return len(self._module_dict)

    def get(self, key: str) -> Optional[type]:
        if not isinstance(key, str):
            raise TypeError(f"The key argument of `Registry.get` must be a str, got {type(key)}")
        if key in self._module_dict:
            return self._module_dict[key]
        # cross-scope names such as "mmpretrain.VisionTransformer" (config :57) or "mmpose.ProbMap"
        if "." in key:
            _, real = key.split(".", 1)
            return self._module_dict.get(key) or self._module_dict.get(real)
        return None

    def _register(self, module: type, name=None, force: bool = False) -> None:
        if not callable(module):
            raise TypeError(f"module must be Callable, but got {type(module)}")
        names = [module.__name__] if name is None else ([name] if isinstance(name, str) else list(name))
        for n in names:
            if not force and n in self._module_dict:
                raise KeyError(f"{n} is already registered in {self.name} at {self._module_dict[n].__module__}")
            self._module_dict[n] = module

    def register_module(self, name=None, force: bool = False, module: Optional[type] = None):
        if not isinstance(force, bool):
            raise TypeError(f"force must be a boolean, but got {type(force)}")
        if module is not None:
            self._register(module, name, force)
            return module

        def _deco(cls: Callable) -> Callable:
            self._register(cls, name, force)
            return cls

        return _deco

    def build(self, cfg: Dict[str, Any], *args, **default_args):
        if cfg is None:
            return None
        if not isinstance(cfg, dict):
            raise TypeError(f"cfg should be a dict, but got {type(cfg)}")
        if "type" not in cfg:
            raise KeyError(f'`cfg` must contain the key "type", but got {cfg}')
        kwargs = dict(cfg)
        for k, v in default_args.items():
            kwargs.setdefault(k, v)
        obj_type = kwargs.pop("type")
        if isinstance(obj_type, str):
            cls = self.get(obj_type)
            if cls is None:
                raise KeyError(
                    f"{obj_type} is not in the {self.scope}::{self.name} registry. "
                    "Please check whether the value of `type` is correct or it was registered as expected."
                )
        elif inspect.isclass(obj_type) or inspect.isfunction(obj_type):
            cls = obj_type
        else:
            raise TypeError(f"type must be a str or valid type, but got {type(obj_type)}")
        return cls(*args, **kwargs)


def _real_registries():
    try:
        from mmpose.registry import KEYPOINT_CODECS as KC  # type: ignore
        from mmpose.registry import MODELS as M  # type: ignore

        try:
            from mmpose.registry import TRANSFORMS as T  # type: ignore
        except ImportError:  # an MMPose without the transform registry: the val pipeline uses this package's own
            T = None
        return M, KC, T
    except Exception:  # noqa: BLE001 -- mmpose / mmengine absent or broken: use the shim
        return None


_real = _real_registries()
USING_MMENGINE = _real is not None
if USING_MMENGINE:
    MODELS, KEYPOINT_CODECS, TRANSFORMS = _real
    if TRANSFORMS is None:
        TRANSFORMS = Registry("transform")
else:
    MODELS = Registry("model")
    KEYPOINT_CODECS = Registry("KEYPOINT_CODECS")
    TRANSFORMS = Registry("transform")  # mmpose/registry.py:36: the val pipeline's LoadImage / GetBBoxCenterScale / TopdownAffine / PackPoseInputs

# Under a real MMPose, registering under the reference's own names needs force=True.
OVERRIDE_REFERENCE_NAMES = (not USING_MMENGINE) or os.environ.get("PROBPOSE_MI355X_OVERRIDE", "0") == "1"


def register(registry, reference_name: str, mi355x_name: str):
    """Class decorator: register under ``mi355x_name`` always, and under the reference's
    ``reference_name`` when we own the registry or were asked to override."""

    def _deco(cls):
        registry.register_module(name=mi355x_name, force=True, module=cls)
        if OVERRIDE_REFERENCE_NAMES:
            registry.register_module(name=reference_name, force=True, module=cls)
        return cls

    return _deco
