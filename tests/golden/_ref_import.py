"""Isolated import of the reference's hot-path modules (THIS container only).

`import mmpose` fails here (mmengine/mmcv/cv2/... are absent), so the
reference files on the decode path are loaded one by one behind minimal stubs:

* ``cv2``                       -> empty module (only used by gaussian_blur*, never by decode)
* ``mmpose``, ``mmpose.codecs`` -> bare package shells (their ``__init__`` must not run)
* ``mmengine.utils.is_method_overridden`` -> 3-line functional stub
* ``mmpose.registry.KEYPOINT_CODECS``     -> dict-backed register/build

Used ONLY by ``tests/golden/make_golden.py`` to produce committed fixtures. Nothing in
``tests/``'s collected tests, ``bench.py`` or ``__graft_entry__`` imports this file:
``/root/reference`` does not exist on the GPU box.
"""
import importlib
import importlib.util
import os
import sys
import types

REF = os.environ.get("PROBPOSE_REFERENCE", "/root/reference")


def _shell(name, path=None):
    m = types.ModuleType(name)
    if path is not None:
        m.__path__ = [path]
    sys.modules[name] = m
    return m


def _load(name, relpath):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, relpath))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


class _DictRegistry:
    def __init__(self):
        self.module_dict = {}

    def register_module(self, name=None, force=False, module=None):
        def deco(cls):
            self.module_dict[name or cls.__name__] = cls
            return cls

        return deco if module is None else deco(module)

    def build(self, cfg):
        cfg = dict(cfg)
        return self.module_dict[cfg.pop("type")](**cfg)


def load_reference():
    """Returns a namespace with the reference's own functions/classes."""
    if not os.path.isdir(REF):
        raise RuntimeError(f"reference tree not found at {REF}")
    if "cv2" not in sys.modules:
        _shell("cv2")
    _shell("mmpose", os.path.join(REF, "mmpose"))
    _shell("mmpose.codecs", os.path.join(REF, "mmpose/codecs"))
    mmengine = _shell("mmengine")
    mmengine_utils = _shell("mmengine.utils")

    def is_method_overridden(method, base_class, derived_class):
        if not isinstance(derived_class, type):
            derived_class = derived_class.__class__
        return getattr(derived_class, method) != getattr(base_class, method)

    mmengine_utils.is_method_overridden = is_method_overridden
    mmengine.utils = mmengine_utils
    reg = _shell("mmpose.registry")
    reg.KEYPOINT_CODECS = _DictRegistry()

    ns = types.SimpleNamespace()
    # real package: mmpose/codecs/utils/__init__.py imports only numpy/scipy/torch/cv2 users
    ns.utils = importlib.import_module("mmpose.codecs.utils")
    ns.post = sys.modules["mmpose.codecs.utils.post_processing"]
    ns.base = _load("mmpose.codecs.base", "mmpose/codecs/base.py")
    ns.probmap = _load("mmpose.codecs.probmap", "mmpose/codecs/probmap.py")
    ns.tta = _load("_ref_tta", "mmpose/models/utils/tta.py")
    ns.KEYPOINT_CODECS = reg.KEYPOINT_CODECS
    return ns


def load_reference_eval():
    """The reference's Ex-OKS evaluator (mmpose/evaluation/metrics/_cocoeval.py) and its bbox helper, behind stubs for
    ``xtcocotools._mask`` (mask IoU only) and the ``mmpose.structures.keypoint`` package (re-exports the real
    ``fix_bbox_aspect_ratio``). Returns a namespace with ``COCOeval`` and ``fix_bbox_aspect_ratio``."""
    if not os.path.isdir(REF):
        raise RuntimeError(f"reference tree not found at {REF}")
    if "mmpose" not in sys.modules:
        _shell("mmpose", os.path.join(REF, "mmpose"))
    _shell("mmpose.structures")
    kp = _shell("mmpose.structures.keypoint")
    minpad = _load("_ref_keypoints_min_padding", "mmpose/structures/keypoint/keypoints_min_padding.py")
    kp.fix_bbox_aspect_ratio = minpad.fix_bbox_aspect_ratio
    _shell("mmpose.evaluation")
    _shell("mmpose.evaluation.metrics", os.path.join(REF, "mmpose/evaluation/metrics"))
    _shell("mmpose.evaluation.metrics._mask")
    ce = _load("mmpose.evaluation.metrics._cocoeval", "mmpose/evaluation/metrics/_cocoeval.py")
    ns = types.SimpleNamespace()
    ns.COCOeval = ce.COCOeval
    ns.fix_bbox_aspect_ratio = minpad.fix_bbox_aspect_ratio
    return ns


def load_reference_bbox():
    """mmpose/structures/bbox/transforms.py behind the empty ``cv2`` stub (cv2 is only used by get_warp_matrix's
    callers, not by the functions the fixtures need)."""
    if "cv2" not in sys.modules:
        _shell("cv2")
    return _load("_ref_bbox_transforms", "mmpose/structures/bbox/transforms.py")


def load_reference_nms():
    """mmpose/evaluation/functional/nms.py (oks_iou / oks_nms) behind a stub for ``mmpose.structures.bbox.bbox_overlaps``
    (used only by the box NMS variants)."""
    if "mmpose" not in sys.modules:
        _shell("mmpose", os.path.join(REF, "mmpose"))
    if "mmpose.structures" not in sys.modules:
        _shell("mmpose.structures")
    b = _shell("mmpose.structures.bbox")
    b.bbox_overlaps = None
    return _load("_ref_nms", "mmpose/evaluation/functional/nms.py")
