"""Output containers of the hot path (reference: ``mmpose/structures/pose_data_sample.py:9-104``
on top of mmengine's ``BaseDataElement`` / ``InstanceData`` / ``PixelData``).

With mmengine importable the real classes are re-exported so results are the very types
``demo/image_demo.py`` and ``CocoMetric.process`` expect. Without it (this container, the
GPU box) the look-alikes below provide the slice of behaviour the path touches:
attribute + item access, ``set_field``, ``metainfo`` / ``set_metainfo``, ``keys`` /
``all_items``, ``in``.
"""
from typing import Any, Dict, Iterator, Tuple

try:  # pragma: no cover - exercised only in a real MMPose install
    from mmengine.structures import BaseDataElement, InstanceData, PixelData  # type: ignore
    from mmpose.structures import PoseDataSample  # type: ignore

    USING_MMENGINE = True
except Exception:  # noqa: BLE001
    USING_MMENGINE = False

    class BaseDataElement:
        def __init__(self, *, metainfo: Dict[str, Any] = None, **kwargs):
            object.__setattr__(self, "_metainfo_fields", set())
            object.__setattr__(self, "_data_fields", set())
            if metainfo is not None:
                self.set_metainfo(metainfo)
            for k, v in kwargs.items():
                setattr(self, k, v)

        # -- metainfo
        def set_metainfo(self, metainfo: Dict[str, Any]) -> None:
            assert isinstance(metainfo, dict), f"metainfo should be a ``dict`` but got {type(metainfo)}"
            for k, v in metainfo.items():
                self.set_field(v, k, field_type="metainfo")

        @property
        def metainfo(self) -> Dict[str, Any]:
            return {k: getattr(self, k) for k in self._metainfo_fields}

        def metainfo_keys(self):
            return list(self._metainfo_fields)

        # -- data
        def set_field(self, value: Any, name: str, dtype=None, field_type: str = "data") -> None:
            assert field_type in ("metainfo", "data")
            if dtype is not None:
                assert isinstance(value, dtype), f"{value} should be a {dtype} but got {type(value)}"
            if field_type == "metainfo":
                if name in self._data_fields:
                    raise AttributeError(f"Cannot set {name} to be a field of metainfo because it is a data field")
                self._metainfo_fields.add(name)
            else:
                if name in self._metainfo_fields:
                    raise AttributeError(f"Cannot set {name} to be a field of data because it is a metainfo field")
                self._data_fields.add(name)
            object.__setattr__(self, name, value)

        def __setattr__(self, name: str, value: Any) -> None:
            if name in ("_metainfo_fields", "_data_fields"):
                raise AttributeError(f"{name} has been used as a private attribute, which is immutable.")
            self.set_field(value, name)

        def _set_data_fields(self, fields: Dict[str, Any]) -> None:
            """Several data fields at once without the per-field checks of ``set_field`` (the caller guarantees that none of the
            names is a metainfo field): the packaging loop of a batch sets ~10 fields on each of its samples."""
            self._data_fields.update(fields)
            d = self.__dict__
            for k, v in fields.items():
                d[k] = v

        def __delattr__(self, name: str) -> None:
            object.__delattr__(self, name)
            self._data_fields.discard(name)
            self._metainfo_fields.discard(name)

        __setitem__ = lambda self, k, v: setattr(self, k, v)  # noqa: E731

        def __getitem__(self, name: str) -> Any:
            return getattr(self, name)

        def __contains__(self, name: str) -> bool:
            return name in self._data_fields or name in self._metainfo_fields

        def get(self, name: str, default=None):
            return getattr(self, name, default)

        def keys(self):
            private = {"_" + k for k in self._data_fields}  # property-backed fields
            return [k for k in self._data_fields if k not in private]

        def values(self):
            return [getattr(self, k) for k in self.keys()]

        def items(self) -> Iterator[Tuple[str, Any]]:
            for k in self.keys():
                yield k, getattr(self, k)

        def all_keys(self):
            return self.metainfo_keys() + self.keys()

        def all_items(self) -> Iterator[Tuple[str, Any]]:
            for k in self.all_keys():
                yield k, getattr(self, k)

        def __repr__(self):
            body = ", ".join(f"{k}={type(v).__name__}" for k, v in self.all_items())
            return f"<{self.__class__.__name__}({body})>"

    class InstanceData(BaseDataElement):
        """Instance-level fields; all values share their first dimension."""

        def __len__(self) -> int:
            vals = self.values()
            return len(vals[0]) if vals else 0

        @staticmethod
        def cat(instances_list):
            """mmengine InstanceData.cat: concatenate every data field along the first dimension."""
            import numpy as np
            import torch

            assert len(instances_list) > 0 and all(isinstance(i, InstanceData) for i in instances_list)
            if len(instances_list) == 1:
                return instances_list[0]
            out = InstanceData(metainfo=instances_list[0].metainfo)
            for k in instances_list[0].keys():
                vals = [getattr(i, k) for i in instances_list]
                v0 = vals[0]
                if isinstance(v0, torch.Tensor):
                    merged = torch.cat(vals, dim=0)
                elif isinstance(v0, np.ndarray):
                    merged = np.concatenate(vals, axis=0)
                elif isinstance(v0, (str, list, tuple)):
                    merged = [x for v in vals for x in (v if isinstance(v, (list, tuple)) else [v])]
                else:
                    raise ValueError(f"The type of `{k}` is `{type(v0)}` which has no concatenation rule")
                setattr(out, k, merged)
            return out

    class PixelData(BaseDataElement):
        """Pixel-level fields of shape (C, H, W)."""

        @property
        def shape(self):
            vals = self.values()
            return tuple(vals[0].shape[-2:]) if vals else None

    def _prop(private, dtype):
        def getter(self):
            return getattr(self, private)

        def setter(self, value):
            self.set_field(value, private, dtype=dtype)

        def deleter(self):
            delattr(self, private)

        return property(getter, setter, deleter)

    class PoseDataSample(BaseDataElement):
        """pose_data_sample.py:9-104: gt_instances / pred_instances / gt_fields / pred_fields."""

        gt_instances = _prop("_gt_instances", InstanceData)
        gt_instance_labels = _prop("_gt_instance_labels", InstanceData)
        pred_instances = _prop("_pred_instances", InstanceData)
        gt_fields = _prop("_gt_fields", PixelData)
        pred_fields = _prop("_pred_heatmaps", PixelData)

        def __setattr__(self, name, value):
            p = getattr(type(self), name, None)
            if isinstance(p, property):
                p.fset(self, value)
            else:
                super().__setattr__(name, value)

        def __contains__(self, name):
            priv = {
                "gt_instances": "_gt_instances",
                "pred_instances": "_pred_instances",
                "gt_fields": "_gt_fields",
                "pred_fields": "_pred_heatmaps",
                "gt_instance_labels": "_gt_instance_labels",
            }.get(name, name)
            return super().__contains__(priv)




def _image_padding(centers, scales, ori_shape):
    """[left, top, right, bottom] padding that keeps every activation window (+10 px) inside (structures/utils.py:70-87)."""
    import numpy as np

    pad = np.zeros(4, np.int64)
    for c, s in zip(centers, scales):
        pad = np.maximum(pad, [int(max(s[0] / 2 - c[0] + 10, 0)), int(max(s[1] / 2 - c[1] + 10, 0)),
                               int(max(c[0] + s[0] / 2 - ori_shape[1] + 10, 0)), int(max(c[1] + s[1] / 2 - ori_shape[0] + 10, 0))])
    return pad


def revert_heatmaps_max(heatmaps, input_centers, input_scales, img_shape, device="cuda"):
    """The (K, h, w) maps of n persons back on an image of ``img_shape`` = (H, W), merged by maximum - what
    ``np.max([revert_heatmap(...) for ...], axis=0)`` gives in the reference (structures/utils.py:105-123), in one launch
    (pp_revert_heatmaps_max). ``heatmaps``: (n, K, h, w) array / tensor or a list of (K, h, w). Returns a float32 device
    tensor (K, H, W)."""
    import numpy as np
    import torch

    from . import _lib
    from .transforms import get_warp_matrix, invert_affine

    device = torch.device(device)
    if device.type != "cuda":
        raise RuntimeError("revert_heatmaps_max runs on the GPU only (no CPU fallback)")
    if isinstance(heatmaps, (list, tuple)):
        heatmaps = torch.stack([torch.as_tensor(h) for h in heatmaps])
    hm = torch.as_tensor(heatmaps).to(device=device, dtype=torch.float32).contiguous()
    if hm.dim() == 3:
        hm = hm[None]
    n, K, h, w = hm.shape
    centers = np.asarray(input_centers, np.float64).reshape(n, 2)
    scales = np.asarray(input_scales, np.float64).reshape(n, 2)
    inv = np.stack([invert_affine(get_warp_matrix(centers[i], scales[i], 0, (w, h), inv=True)) for i in range(n)])
    inv_d = torch.from_numpy(inv).to(device)
    out = torch.empty((K, int(img_shape[0]), int(img_shape[1])), dtype=torch.float32, device=device)
    _lib.call("pp_revert_heatmaps_max", hm.data_ptr(), inv_d.data_ptr(), out.data_ptr(), n, K, h, w, int(img_shape[0]),
              int(img_shape[1]), torch.cuda.current_stream(device).cuda_stream)
    return out


def revert_heatmap(heatmap, input_center, input_scale, img_shape):
    """structures/utils.py:146-175: one (K, h, w) or (h, w) map back on the image; returns a numpy array like the
    reference."""
    import torch

    hm = torch.as_tensor(heatmap)
    out = revert_heatmaps_max(hm[None] if hm.dim() == 3 else hm[None, None], [input_center], [input_scale], img_shape)
    out = out.cpu().numpy()
    return out if hm.dim() == 3 else out[0]


def merge_data_samples(data_samples):
    """mmpose/structures/utils.py:16-143: the top-down predictions of one image - one PoseDataSample per box - merged
    into a single sample with all instances; metainfo of the first sample, ``input_center`` / ``input_scale`` stacked.
    Predicted heatmaps (``pred_fields.heatmaps``, present with ``test_cfg.output_heatmaps``) are put back on the image
    padded so that every activation window fits, and merged by maximum (:48-128); ground-truth heatmaps on the
    un-padded image (:130-141)."""
    import warnings

    import numpy as np

    if not isinstance(data_samples, (list, tuple)) or not all(isinstance(d, PoseDataSample) for d in data_samples):
        raise ValueError("Invalid input type, should be a list of :obj:`PoseDataSample`")
    if len(data_samples) == 0:
        warnings.warn("Try to merge an empty list of data samples.")
        return PoseDataSample()
    metadata = dict(data_samples[0].metainfo)
    metadata["input_center"] = np.array([ds.input_center for ds in data_samples])
    metadata["input_scale"] = np.array([ds.input_scale for ds in data_samples])
    merged = PoseDataSample(metainfo=metadata)
    if "gt_instances" in data_samples[0]:
        merged.gt_instances = InstanceData.cat([d.gt_instances for d in data_samples])
    if "pred_instances" in data_samples[0]:
        merged.pred_instances = InstanceData.cat([d.pred_instances for d in data_samples])
    centers = [np.asarray(ds.input_center, np.float64).reshape(2) for ds in data_samples]
    scales = [np.asarray(ds.input_scale, np.float64).reshape(2) for ds in data_samples]
    ori_shape = data_samples[0].metainfo.get("ori_shape")
    if "pred_fields" in data_samples[0] and "heatmaps" in data_samples[0].pred_fields:
        pad = _image_padding(centers, scales, ori_shape)
        padded_shape = (ori_shape[0] + pad[1] + pad[3], ori_shape[1] + pad[0] + pad[2])
        maps = revert_heatmaps_max([ds.pred_fields.heatmaps for ds in data_samples], [c + pad[:2] for c in centers], scales,
                                   padded_shape)
        merged.pred_fields = PixelData(heatmaps=maps.cpu().numpy())
        merged.set_metainfo(dict(image_pad=pad))
    if "gt_fields" in data_samples[0] and "heatmaps" in data_samples[0].gt_fields:
        maps = revert_heatmaps_max([ds.gt_fields.heatmaps for ds in data_samples], centers, scales, ori_shape)
        merged.gt_fields = PixelData(heatmaps=maps.cpu().numpy())
    return merged


def posterior_heatmaps(heatmaps, keypoints_probs, device="cuda"):
    """What ``--draw-heatmap`` shows (mmpose/visualization/local_visualizer.py:827-837): every map normalised to sum 1,
    times the presence probability of its keypoint averaged over the instances. ``heatmaps`` (K, H, W), ``keypoints_probs``
    (n, K). Returns a float32 device tensor (pp_heatmap_posterior)."""
    import torch

    from . import _lib

    device = torch.device(device)
    if device.type != "cuda":
        raise RuntimeError("posterior_heatmaps runs on the GPU only (no CPU fallback)")
    hm = torch.as_tensor(heatmaps).to(device=device, dtype=torch.float32).contiguous().clone()
    K, H, W = hm.shape
    pr = torch.as_tensor(keypoints_probs).to(device=device, dtype=torch.float32).reshape(-1, K).mean(dim=0).contiguous()
    scratch = torch.empty(K * 64, dtype=torch.float64, device=device)
    _lib.call("pp_heatmap_posterior", hm.data_ptr(), pr.data_ptr(), scratch.data_ptr(), K, H, W,
              torch.cuda.current_stream(device).cuda_stream)
    return hm


__all__ = ["BaseDataElement", "InstanceData", "PixelData", "PoseDataSample", "USING_MMENGINE", "merge_data_samples",
           "revert_heatmap", "revert_heatmaps_max", "posterior_heatmaps"]
