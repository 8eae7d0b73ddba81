"""Host-side mirror of the reference's model classes for the ProbPose inference path.

Same registry names, constructor arguments, method names and outputs as

* ``PoseDataPreprocessor``  -- mmpose/models/data_preprocessors/data_preprocessor.py:13-133
* ``VisionTransformer``     -- mmpretrain 1.2.0 [3P], ctor args at
  configs/body_2d_keypoint/topdown_probmap/coco/td-pm_ProbPose-small_8xb64-210e_coco-256x192.py:56-67
* ``ProbMapHead``           -- mmpose/models/heads/hybrid_heads/probmap_head.py:25-804 (inference methods)
* ``TopdownPoseEstimator``  -- mmpose/models/pose_estimators/topdown.py:12-194, base.py:17-243

but every ``forward`` / ``predict`` lands in ``ProbPoseEngine`` (hand-written HIP through the C ABI).
The modules are ``torch.nn.Module`` parameter containers whose ``state_dict`` keys are exactly the
reference's, so ``load_checkpoint`` / ``load_state_dict`` of ``ProbPose-s.pth`` work unchanged;
torch never computes anything here. Training-side members (``loss``, the loss modules, ``encode``)
are outside the hot path and raise ``NotImplementedError``.
"""
from itertools import zip_longest
from typing import Dict, List, Optional, Sequence, Tuple, Union

import numpy as np
import torch
from torch import Tensor, nn

from .engine import ProbPoseEngine
from .registry import KEYPOINT_CODECS, MODELS, register
from .structures import InstanceData, PixelData, PoseDataSample

TOWERS = ("probability", "visibility", "oks", "error")


# =================================================================================================
@register(MODELS, reference_name="PoseDataPreprocessor", mi355x_name="PoseDataPreprocessorMI355X")
class PoseDataPreprocessor(nn.Module):
    """Stacks the uint8 CHW crops of a batch on the device. Channel swap and ``(x - mean) / std``
    (mmengine ``ImgDataPreprocessor`` [3P]) are NOT done here: they are fused into the patch-embed
    im2col kernel, which reads the raw bytes once. ``mean`` / ``std`` / ``bgr_to_rgb`` are kept and
    handed to the engine."""

    def __init__(self, mean: Sequence[float] = None, std: Sequence[float] = None, pad_size_divisor: int = 1,
                 pad_value: Union[float, int] = 0, bgr_to_rgb: bool = False, rgb_to_bgr: bool = False,
                 non_blocking: Optional[bool] = False, batch_augments: Optional[List[dict]] = None):
        super().__init__()
        assert not (bgr_to_rgb and rgb_to_bgr), "`bgr2rgb` and `rgb2bgr` cannot be set to True at the same time"
        assert (mean is None) == (std is None), "mean and std should be both None or tuple"
        if pad_size_divisor != 1:
            raise NotImplementedError("pad_size_divisor != 1 is not used by top-down crops")
        self.mean_values = tuple(mean) if mean is not None else (0.0, 0.0, 0.0)
        self.std_values = tuple(std) if std is not None else (1.0, 1.0, 1.0)
        self.channel_conversion = bool(bgr_to_rgb or rgb_to_bgr)
        self.pad_size_divisor = pad_size_divisor
        self.non_blocking = bool(non_blocking)
        self.batch_augments = None  # training-time only
        self.register_buffer("_device_probe", torch.zeros(()), persistent=False)

    @property
    def device(self):
        return self._device_probe.device

    def forward(self, data: dict, training: bool = False) -> dict:
        inputs, data_samples = data["inputs"], data.get("data_samples")
        if isinstance(inputs, (list, tuple)):
            for t in inputs:
                assert t.dim() == 3 and t.shape[0] == 3, f"expects (3, H, W) crops, got {tuple(t.shape)}"
            inputs = torch.stack([t.to(self.device, non_blocking=self.non_blocking) for t in inputs])
        elif isinstance(inputs, torch.Tensor):
            assert inputs.dim() == 4, (
                "The input of `ImgDataPreprocessor` should be a NCHW tensor or a list of tensor, "
                f"but got a tensor with shape: {inputs.shape}"
            )
            inputs = inputs.to(self.device, non_blocking=self.non_blocking)
        else:
            raise TypeError(f"Output of `cast_data` should be a dict of list/tuple with inputs and data_samples, but got {type(data)}")
        if data_samples is not None:
            shape = tuple(inputs.shape[-2:])
            for ds in data_samples:
                ds.set_metainfo({"batch_input_shape": shape, "pad_shape": shape})
        return {"inputs": inputs, "data_samples": data_samples}


# =================================================================================================
_VIT_ARCHS = {
    "small": dict(embed_dims=768, num_layers=8, num_heads=8, feedforward_channels=768 * 3),
    "base": dict(embed_dims=768, num_layers=12, num_heads=12, feedforward_channels=3072),
    "large": dict(embed_dims=1024, num_layers=24, num_heads=16, feedforward_channels=4096),
    "deit-small": dict(embed_dims=384, num_layers=12, num_heads=6, feedforward_channels=384 * 4),
}
_VIT_ARCHS.update({"s": _VIT_ARCHS["small"], "b": _VIT_ARCHS["base"], "l": _VIT_ARCHS["large"]})


class _Holder(nn.Module):
    """Bare parameter namespace (never called)."""


@register(MODELS, reference_name="VisionTransformer", mi355x_name="VisionTransformerMI355X")
class VisionTransformer(nn.Module):
    """Parameter container with mmpretrain's ``VisionTransformer`` key names; forward on the MI355X.

    Supported configuration == what ProbPose / ViTPose use: ``with_cls_token=False``,
    ``out_type='featmap'``, ``final_norm=True``, ``pre_norm=False``, no layer scale, patch 16.
    """

    def __init__(self, arch="base", img_size=224, patch_size=16, in_channels=3, out_indices=-1, drop_rate=0.0,
                 drop_path_rate=0.0, qkv_bias=True, norm_cfg=dict(type="LN", eps=1e-6), final_norm=True,
                 out_type="cls_token", with_cls_token=True, frozen_stages=-1, interpolate_mode="bicubic",
                 layer_scale_init_value=0.0, patch_cfg=dict(), layer_cfgs=dict(), pre_norm=False, init_cfg=None):
        super().__init__()
        if isinstance(arch, str):
            arch = arch.lower()
            assert arch in _VIT_ARCHS, f"Arch {arch} is not in default archs {set(_VIT_ARCHS)}"
            arch = _VIT_ARCHS[arch]
        else:
            essential = {"embed_dims", "num_layers", "num_heads", "feedforward_channels"}
            assert isinstance(arch, dict) and essential <= set(arch), f"Custom arch needs a dict with keys {essential}"
        unsupported = []
        if with_cls_token: unsupported.append("with_cls_token=True")  # noqa: E701
        if out_type != "featmap": unsupported.append(f"out_type={out_type!r}")  # noqa: E701
        if not final_norm: unsupported.append("final_norm=False")  # noqa: E701
        if pre_norm: unsupported.append("pre_norm=True")  # noqa: E701
        if layer_scale_init_value: unsupported.append("layer_scale_init_value != 0")  # noqa: E701
        if patch_size != 16 or in_channels != 3: unsupported.append("patch_size != 16 or in_channels != 3")  # noqa: E701
        if out_indices not in (-1, [-1], (-1,)): unsupported.append(f"out_indices={out_indices}")  # noqa: E701
        if unsupported:
            raise NotImplementedError("VisionTransformer on MI355X covers the ProbPose/ViTPose setting only; got " + ", ".join(unsupported))
        self.arch_settings = dict(arch)
        self.embed_dims, self.num_layers = arch["embed_dims"], arch["num_layers"]
        self.num_heads, self.ffn_dims = arch["num_heads"], arch["feedforward_channels"]
        self.img_size = (img_size, img_size) if isinstance(img_size, int) else tuple(img_size)
        self.patch_size = patch_size
        self.patch_padding = int(dict(patch_cfg).get("padding", 0))
        self.ln_eps = float(dict(norm_cfg).get("eps", 1e-5))
        self.qkv_bias = qkv_bias
        E, Fd, P = self.embed_dims, self.ffn_dims, patch_size
        Hp = (self.img_size[0] + 2 * self.patch_padding - P) // P + 1
        Wp = (self.img_size[1] + 2 * self.patch_padding - P) // P + 1
        self.patch_resolution = (Hp, Wp)
        self.patch_embed = _Holder()
        self.patch_embed.projection = nn.Conv2d(3, E, P, stride=P, padding=self.patch_padding)
        self.pos_embed = nn.Parameter(torch.zeros(1, Hp * Wp, E))
        self.layers = nn.ModuleList()
        for _ in range(self.num_layers):
            lyr = _Holder()
            lyr.ln1 = nn.LayerNorm(E, eps=self.ln_eps)
            lyr.attn = _Holder()
            lyr.attn.qkv = nn.Linear(E, 3 * E, bias=qkv_bias)
            lyr.attn.proj = nn.Linear(E, E)
            lyr.ln2 = nn.LayerNorm(E, eps=self.ln_eps)
            lyr.ffn = _Holder()
            lyr.ffn.layers = nn.Sequential(nn.Sequential(nn.Linear(E, Fd), nn.GELU()), nn.Linear(Fd, E))
            self.layers.append(lyr)
        self.ln1 = nn.LayerNorm(E, eps=self.ln_eps)
        self._owner = None  # set by the estimator: the engine lives there (it needs the head's weights too)

    def init_weights(self):
        nn.init.trunc_normal_(self.pos_embed, std=0.02)

    def forward(self, x: Tensor) -> Tuple[Tensor]:
        """(B, 3, H, W) float32 preprocessed (reference contract) or uint8 raw crops -> ``(feat,)`` with
        feat of logical shape (B, E, Hp, Wp), channels-last in memory. The result is the caller's own copy (the
        engine's workspace is reused by the next call with the same batch size: the reference's flip-test call pattern,
        ``feats = extract_feat(x); feats_flip = extract_feat(x.flip(-1))``, topdown.py:109-112, must see two tensors)."""
        if self._owner is None:
            raise RuntimeError("VisionTransformer (MI355X) must be built inside a TopdownPoseEstimator: "
                               "the HIP engine is owned by the estimator")
        eng = self._owner().engine
        feat = eng.export_features(eng.run_backbone(x, flip_test=False))
        return (feat.permute(0, 3, 1, 2),)


# =================================================================================================
@register(MODELS, reference_name="ProbMapHead", mi355x_name="ProbMapHeadMI355X")
class ProbMapHead(nn.Module):
    """probmap_head.py:25-804, inference side. Constructor arguments are the reference's; loss configs are
    accepted and ignored (training is out of scope)."""

    _version = 2

    def __init__(self, in_channels: Union[int, Sequence[int]], out_channels: int,
                 deconv_out_channels: Optional[Sequence[int]] = (256, 256, 256),
                 deconv_kernel_sizes: Optional[Sequence[int]] = (4, 4, 4),
                 conv_out_channels: Optional[Sequence[int]] = None, conv_kernel_sizes: Optional[Sequence[int]] = None,
                 final_layer_dict: dict = dict(kernel_size=1), keypoint_loss=None, probability_loss=None,
                 visibility_loss=None, oks_loss=None, error_loss=None, normalize: float = None,
                 detach_probability: bool = True, detach_visibility: bool = True,
                 learn_heatmaps_from_zeros: bool = False, freeze_heatmaps: bool = False,
                 freeze_probability: bool = False, freeze_visibility: bool = False, freeze_oks: bool = False,
                 freeze_error: bool = False,
                 decoder=dict(type="UDPHeatmap", input_size=(192, 256), heatmap_size=(48, 64), sigma=2),
                 init_cfg=None):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.temperature = 0.5  # probmap_head.py:135
        self.normalize = normalize
        self.freeze_oks, self.freeze_error = freeze_oks, freeze_error
        self.decoder = KEYPOINT_CODECS.build(decoder) if decoder is not None else None
        if deconv_out_channels:
            if deconv_kernel_sizes is None or len(deconv_out_channels) != len(deconv_kernel_sizes):
                raise ValueError(
                    '"deconv_out_channels" and "deconv_kernel_sizes" should be integer sequences with the same '
                    f"length. Got mismatched lengths {deconv_out_channels} and {deconv_kernel_sizes}"
                )
            layers, cin = [], in_channels
            for cout, ks in zip(deconv_out_channels, deconv_kernel_sizes):
                if ks not in (4, 3, 2):
                    raise ValueError(f"Unsupported kernel size {ks} fordeconvlutional layers in {self.__class__.__name__}")
                if ks != 4:
                    raise NotImplementedError("the MI355X deconv kernel implements kernel 4 / stride 2 / pad 1 (ProbPose)")
                layers += [nn.ConvTranspose2d(cin, cout, 4, stride=2, padding=1, output_padding=0, bias=False),
                           nn.BatchNorm2d(cout), nn.ReLU(inplace=True)]
                cin = cout
            self.deconv_layers = nn.Sequential(*layers)
        else:
            raise NotImplementedError("ProbMapHead on MI355X needs at least one deconv layer (ProbPose uses two)")
        if conv_out_channels:
            raise NotImplementedError("intermediate conv layers are not part of the ProbPose head")
        ks = dict(final_layer_dict or {}).get("kernel_size", 1)
        if final_layer_dict is None or ks != 1:
            raise NotImplementedError("final layer must be the 1x1 conv of the ProbPose config")
        self.final_layer = nn.Conv2d(cin, out_channels, 1)
        for t in TOWERS:
            mods = []
            for pool in ((4, 3), (2, 2), (2, 2)):  # probmap_head.py:264
                mods += [nn.Conv2d(in_channels, in_channels, 3, 1, 1), nn.BatchNorm2d(in_channels),
                         nn.MaxPool2d(pool, pool, 0), nn.ReLU(inplace=True)]
            mods += [nn.Conv2d(in_channels, out_channels, 1), nn.ReLU(inplace=True) if t == "error" else nn.Sigmoid()]
            setattr(self, f"{t}_layers", nn.Sequential(*mods))
        self._register_load_state_dict_pre_hook(self._load_state_dict_pre_hook)
        self._owner = None
        self.init_weights()

    def init_weights(self):
        """default_init_cfg (probmap_head.py:592-598): Normal(std=0.001) convs, BN weight 1."""
        for m in self.modules():
            if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
                nn.init.normal_(m.weight, std=0.001)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.ones_(m.weight)

    def _load_state_dict_pre_hook(self, state_dict, prefix, local_meta, *args, **kwargs):
        """probmap_head.py:1014-1061: pre-v2 checkpoints name the convs behind the deconvolutions ``final_layer.n.*``
        (intermediate conv layers first, the last one the final layer). This head has no intermediate conv layers
        (ProbPose: ``conv_layers`` is ``nn.Identity``), and for that case the reference's hook asserts
        (``assert isinstance(self.conv_layers, nn.Sequential)``, :1047; pinned in tests/golden/head_estimator.npz) - so
        does this one; ``final_layer.weight / .bias`` pass through unchanged."""
        version = local_meta.get("version", None)
        if version and version >= self._version:
            return
        for _k in list(state_dict.keys()):
            if not _k.startswith(prefix):
                continue
            k_parts = _k[len(prefix):].split(".")
            if k_parts[0] == "final_layer" and len(k_parts) == 3:
                assert False, ("old-style key '" + _k + "' (final_layer.n.*) belongs to a head with intermediate conv "
                               "layers; the ProbPose head has none")

    # -- engine access
    @property
    def _engine(self) -> ProbPoseEngine:
        if self._owner is None:
            raise RuntimeError("ProbMapHead (MI355X) must be built inside a TopdownPoseEstimator: "
                               "the HIP engine is owned by the estimator")
        return self._owner().engine

    def _to_nhwc(self, feats) -> Tensor:
        x = feats[-1] if isinstance(feats, (tuple, list)) else feats
        eng = self._engine
        return eng.import_features(x.permute(0, 2, 3, 1))  # layout/format plumbing for feature maps handed in from outside

    def forward(self, feats: Tuple[Tensor]) -> Tuple[Tensor, Tensor, Tensor, Tensor, Tensor]:
        """probmap_head.py:600-625 -> heatmaps (B,K,H,W), probability, visibility, oks, error (B,K,1,1)."""
        out = self._engine.run_head(self._to_nhwc(feats), flip_test=False, return_heatmaps=True)
        s = out["scalars"]
        B, K = s.shape[1], s.shape[2]
        return (out["heatmaps"].clone(),) + tuple(s[i].reshape(B, K, 1, 1).clone() for i in range(4))

    def decode(self, batch_outputs: Tensor):
        """base_head.py:33-86 with the codec's batched branch (:57-62)."""
        if self.decoder is None:
            raise RuntimeError(
                f"The decoder has not been set in {self.__class__.__name__}. "
                "Please set the decoder configs in the init parameters to "
                "enable head methods `head.predict()` and `head.decode()`"
            )
        kpts, scores = self.decoder.batch_decode(batch_outputs)
        return [InstanceData(keypoints=k, keypoint_scores=s) for k, s in zip(kpts, scores)]

    def predict(self, feats, batch_data_samples, test_cfg: dict = {}):
        """probmap_head.py:715-804. ``feats`` is ``[feats, feats_flip]`` under flip_test."""
        flip = bool(test_cfg.get("flip_test", False))
        if flip:
            assert isinstance(feats, list) and len(feats) == 2
            if test_cfg.get("flip_mode", "heatmap") != "heatmap":
                raise NotImplementedError("MI355X head implements flip_mode='heatmap' (models/utils/tta.py:35-39; the ProbPose config): "
                                          "'udp_combined' / 'offset' belong to other heads' outputs")
            flip_indices = batch_data_samples[0].metainfo["flip_indices"]
            x = torch.cat([self._to_nhwc(feats[0]), self._to_nhwc(feats[1])])
        else:
            flip_indices = None
            x = self._to_nhwc(feats)
        out = self._engine.run_head(x, flip, flip_indices, return_heatmaps=bool(test_cfg.get("output_heatmaps", False)),
                                    shift_heatmap=flip and bool(test_cfg.get("shift_heatmap", False)))
        return self.pack_predictions(out, test_cfg)

    def pack_predictions(self, out: Dict[str, Tensor], test_cfg: dict = {}):
        """probmap_head.py:779-804: device results -> list of InstanceData (+ PixelData). One ``pp_pack_records`` launch and ONE
        device-to-host copy bring the whole batch back (the estimator's ``predict`` uses its own pinned buffer for that and
        calls ``pack_records`` directly)."""
        from .dist import pack_records as _pack

        rec = _pack(out).cpu().numpy()  # (B, K, 7) float64: x, y, conf, prob, vis, oks, err (raw)
        return self.pack_records(rec, test_cfg, out.get("heatmaps"))

    def pack_records(self, rec: np.ndarray, test_cfg: dict = {}, heatmaps: Optional[Tensor] = None):
        """probmap_head.py:779-804 from the batch's host record ``rec`` (B, K, 7) float64 [x, y, conf, prob, vis, oks, raw error]
        (float32 device results widened exactly, so narrowing them back is exact too). Every field of the batch is cut out of
        the record ONCE; the per-crop ``InstanceData`` hold (1, K[, 2]) views of those batch arrays - no per-crop copies."""
        eng = self._engine
        B, C = rec.shape[:2]
        if not np.isfinite(rec).all():
            bad = np.argwhere(~np.isfinite(rec).all(axis=(1, 2))).ravel().tolist()
            raise FloatingPointError(
                f"non-finite keypoints / scores for crop(s) {bad[:8]} of this batch: a value left the numeric domain of precision="
                f"{eng.precision!r} (f16x3: operands are fp16 pairs, |x| <= 65504 - include/probpose_mi355x.h, 'numeric domain'; "
                "ProbPoseEngine.domain_report(crops) shows the layer). Run the model with precision='f32' or rescale the checkpoint")
        kpts = np.ascontiguousarray(rec[..., :2])  # float64, like the reference's decode (codecs/probmap.py:218)
        sc = np.ascontiguousarray(np.moveaxis(rec[..., 2:7], -1, 0).astype(np.float32))  # (5, B, K): conf, prob, vis, oks, err
        conf, probabilities, visibilities, oks, errors = (sc[i].reshape(B, 1, C) for i in range(5))
        errors = errors / np.sqrt(eng.Hh**2 + eng.Wh**2)  # :786-787, the same expression (its result dtype is numpy's promotion rule's)
        preds = _BatchPreds()
        preds.keypoints_batch = kpts  # (add_pred_to_datasample maps the whole batch to image space through this array)
        fast = hasattr(InstanceData, "_set_data_fields")  # (the in-repo containers; mmengine's own InstanceData: field by field)
        for pi in range(B):
            if fast:
                p = InstanceData()
                p._set_data_fields(dict(keypoints=kpts[pi:pi + 1], keypoint_scores=conf[pi] if self.freeze_oks else oks[pi],
                                        keypoints_conf=conf[pi], keypoints_probs=probabilities[pi], keypoints_visible=visibilities[pi],
                                        keypoints_oks=oks[pi], keypoints_error=errors[pi]))
                preds.append(p)
                continue
            p = InstanceData(keypoints=kpts[pi:pi + 1], keypoint_scores=conf[pi])
            p.set_field(conf[pi], "keypoints_conf")
            p.set_field(probabilities[pi], "keypoints_probs")
            p.set_field(visibilities[pi], "keypoints_visible")
            p.set_field(oks[pi], "keypoints_oks")
            p.set_field(errors[pi], "keypoints_error")
            if not self.freeze_oks:
                p.set_field(oks[pi], "keypoint_scores")
            preds.append(p)
        if test_cfg.get("output_heatmaps", False):
            assert heatmaps is not None, "output_heatmaps needs the engine's heatmaps"
            return preds, [PixelData(heatmaps=hm) for hm in heatmaps.detach().clone()]
        return preds

    def loss(self, *a, **k):
        raise NotImplementedError("training (probmap_head.py:806-) is outside the MI355X inference hot path")


# =================================================================================================
@register(MODELS, reference_name="TopdownPoseEstimator", mi355x_name="TopdownPoseEstimatorMI355X")
class TopdownPoseEstimator(nn.Module):
    """topdown.py:12-194 / base.py:17-243 for inference. Extra keyword: ``precision`` in {"bf16", "f16x3", "f32"}
    (operand precision of the MFMA kernels: "f16x3" (the default) = split-fp16 operands, three fp16 MFMAs per product, meets
    the reference's 1e-3 tolerance; "bf16" = throughput mode, 0.3 - 0.45 px and a few % argmax flips away from the fp32
    reference - opt-in only; "f32" = exact fp32 products, bit-for-bit an fmaf chain, slowest)."""

    _version = 2

    def __init__(self, backbone: dict, neck: Optional[dict] = None, head: Optional[dict] = None,
                 train_cfg: Optional[dict] = None, test_cfg: Optional[dict] = None,
                 data_preprocessor: Optional[dict] = None, init_cfg=None, metainfo: Optional[dict] = None,
                 precision: str = "f16x3", graph_replay: bool = True, graph_capture_after: int = 3, max_graphs: int = 8):
        super().__init__()
        if neck is not None:
            raise NotImplementedError("the ProbPose config has no neck")
        self.metainfo = metainfo
        self.train_cfg = train_cfg if train_cfg else {}
        self.test_cfg = test_cfg if test_cfg else {}
        self.precision = precision
        # ``predict`` replays the engine's captured hipGraph for a batch size it has met before (the first batch of a size runs
        # the same launches one by one - a one-off size never pays for a capture); False: always kernel by kernel
        self.graph_replay = bool(graph_replay)
        # A capture costs two warm-up forwards, the capture itself and a device-wide synchronisation (torch.cuda.graph does one of its
        # own), i.e. a stall of a few steps that also drains any StepPipeline slot in flight: it is paid at the
        # ``graph_capture_after``-th batch of a size (3: the size has repeated twice - the fixed batches of a test dataloader get there in
        # their third step, the changing person counts of a video mostly never), and the engine keeps at most ``max_graphs`` captured
        # graphs (least recently used out, with their static inputs and workspaces).
        self.graph_capture_after = max(2, int(graph_capture_after))
        self.max_graphs = max(1, int(max_graphs))
        self._sizes_seen: Dict[tuple, int] = {}
        self._gather = None  # ResultGather of ``predict``: pinned host record buffer, grown to the largest batch met
        self.backbone = MODELS.build(backbone)
        self.head = MODELS.build(head) if head is not None else None
        self.data_preprocessor = MODELS.build(data_preprocessor) if data_preprocessor is not None else PoseDataPreprocessor()
        import weakref

        self.backbone._owner = weakref.ref(self)
        if self.head is not None:
            self.head._owner = weakref.ref(self)
        self._engine: Optional[ProbPoseEngine] = None
        self._register_load_state_dict_pre_hook(self._load_state_dict_pre_hook)
        self.register_load_state_dict_post_hook(lambda *_: self.reset_engine())

    # -- properties of the reference
    @property
    def with_neck(self) -> bool:
        return False

    @property
    def with_head(self) -> bool:
        return self.head is not None

    # -- engine lifecycle: (re)built lazily from the module's own state_dict
    def reset_engine(self):
        self._engine = None
        self._sizes_seen = {}
        self._gather = None

    def _apply(self, fn, *args, **kwargs):  # .to(device) / .cuda() move the parameters -> rebuild
        self.reset_engine()
        return super()._apply(fn, *args, **kwargs)

    @property
    def engine(self) -> ProbPoseEngine:
        if self._engine is None:
            dev = self.backbone.pos_embed.device
            if dev.type != "cuda":
                raise RuntimeError("the MI355X pose estimator has no CPU path: move the model to the GPU "
                                   "(`model.to('cuda')`, as init_model does, apis/inference.py:128)")
            dp, bb, hd = self.data_preprocessor, self.backbone, self.head
            codec = hd.decoder
            self._engine = ProbPoseEngine(
                self.state_dict(), num_heads=bb.num_heads, img_size=bb.img_size, patch_size=bb.patch_size,
                patch_padding=bb.patch_padding, mean=dp.mean_values, std=dp.std_values,
                bgr_to_rgb=dp.channel_conversion, temperature=hd.temperature, normalize=hd.normalize,
                input_size=tuple(codec.input_size), ln_eps=bb.ln_eps, precision=self.precision, device=dev)
            assert (self._engine.Wh, self._engine.Hh) == tuple(codec.heatmap_size), (
                f"decoder heatmap_size {tuple(codec.heatmap_size)} does not match the head's output "
                f"{(self._engine.Wh, self._engine.Hh)}")
        return self._engine

    def _load_state_dict_pre_hook(self, state_dict, prefix, local_meta, *args, **kwargs):
        """base.py:212-243: drop data_preprocessor.mean/std; keypoint_head -> head for pre-1.0 checkpoints."""
        keys = list(state_dict.keys())
        for k in keys:
            if k in ("data_preprocessor.mean", "data_preprocessor.std"):
                del state_dict[k]
        version = local_meta.get("version", None)
        if version and version >= self._version:
            return
        for k in keys:
            if "keypoint_head" in k:
                state_dict[k.replace("keypoint_head", "head")] = state_dict.pop(k)

    # -- mmengine BaseModel surface
    def test_step(self, data: dict) -> list:
        data = self.data_preprocessor(data, False)
        return self.forward(data["inputs"], data["data_samples"], mode="predict")

    val_step = test_step

    def forward(self, inputs: Tensor, data_samples=None, mode: str = "tensor"):
        if isinstance(inputs, list):
            inputs = torch.stack(inputs)
        if mode == "loss":
            return self.loss(inputs, data_samples)
        elif mode == "predict":
            if self.metainfo is not None:
                for data_sample in data_samples:
                    data_sample.set_metainfo(self.metainfo)
            return self.predict(inputs, data_samples)
        elif mode == "tensor":
            return self._forward(inputs)
        else:
            raise RuntimeError(f'Invalid mode "{mode}". Only supports loss, predict and tensor mode.')

    def loss(self, inputs, data_samples):
        raise NotImplementedError("training is outside the MI355X inference hot path")

    def extract_feat(self, inputs: Tensor) -> Tuple[Tensor]:
        return self.backbone(inputs)

    def _forward(self, inputs: Tensor, data_samples=None):
        x = self.extract_feat(inputs)
        return self.head.forward(x) if self.with_head else x

    def _check_flip_cfg(self) -> bool:
        flip = bool(self.test_cfg.get("flip_test", False))
        if flip and self.test_cfg.get("flip_mode", "heatmap") != "heatmap":
            raise NotImplementedError(
                f"flip_mode={self.test_cfg.get('flip_mode')!r}: the MI355X path merges the flipped pass as flip_mode='heatmap' "
                "(models/utils/tta.py:35-39; the ProbPose config); 'udp_combined' / 'offset' belong to other heads' outputs")
        return flip

    @property
    def _shift_heatmap(self) -> bool:
        """``test_cfg.shift_heatmap`` (flip_heatmaps(..., shift_heatmap=True), tta.py:64-66): the flipped-back map moves one pixel to
        the right before the average - done inside the fused flip-merge + decode kernel (PP_DECODE_SHIFT_HEATMAP)."""
        return bool(self.test_cfg.get("flip_test", False)) and bool(self.test_cfg.get("shift_heatmap", False))

    def predict(self, inputs: Tensor, data_samples: list) -> list:
        """topdown.py:86-126, as ONE launch sequence: both flip-test passes are batched through the
        backbone and the head consumes the features in place. A batch size met ``graph_capture_after`` times replays the engine's captured
        hipGraph (bit-identical to the launches one by one, tests/test_estimator_gpu.py); the results come back as ONE
        fixed-layout record (``pp_pack_records``) through ONE copy into pinned host memory."""
        from .dist import ResultGather

        assert self.with_head, "The model must have head to perform prediction."
        flip = self._check_flip_cfg()
        flip_indices = data_samples[0].metainfo["flip_indices"] if flip else None
        want_hm = bool(self.test_cfg.get("output_heatmaps", False))
        eng = self.engine
        B = int(inputs.shape[0])
        key = (B, flip, tuple(flip_indices) if flip_indices is not None else None, want_hm)
        seen = self._sizes_seen.get(key, 0)
        self._sizes_seen[key] = seen + 1
        shift = self._shift_heatmap
        eng.max_graphs = self.max_graphs
        # replay a captured graph; capture a new one only when that does not evict a graph still in use (more recurring batch sizes than
        # graphs kept - the person counts of a video - would otherwise capture on every call: engine.capture_would_thrash)
        use_graph = self.graph_replay and inputs.dtype == torch.uint8 and (
            eng.has_graph(B, flip, flip_indices, want_hm, 0, shift)
            or (seen + 1 >= self.graph_capture_after and not eng.capture_would_thrash()))
        if use_graph:
            out = eng.forward_graph(inputs, flip, flip_indices, return_heatmaps=want_hm, shift_heatmap=shift)
        else:
            out = eng.forward(inputs, flip, flip_indices, return_heatmaps=want_hm, shift_heatmap=shift)
        if self._gather is None or self._gather.batch < B:
            self._gather = ResultGather(max(B, 64), eng.K, eng.device, 1)
        self._gather(out)
        rec = self._gather.wait()[0, :B].numpy().copy()  # the pinned buffer is reused by the next batch
        preds = self.head.pack_records(rec, self.test_cfg, out.get("heatmaps"))
        if isinstance(preds, tuple):
            batch_pred_instances, batch_pred_fields = preds
        else:
            batch_pred_instances, batch_pred_fields = preds, None
        return self.add_pred_to_datasample(batch_pred_instances, batch_pred_fields, data_samples)

    def test_step_stream(self, batches, depth: int = 2, max_batch: int = 64):
        """Generator over an iterable of ``test_step`` batch dicts (``inputs``, ``data_samples``): yields what ``test_step``
        returns for each, in order, while up to ``depth`` batches are in flight on the device (pipeline.StepPipeline: own
        stream and workspace per slot; the loop of tools/test.py / the video loop of demo/topdown_demo_with_mmdet.py finishes
        one batch before the next starts). Batches may differ in size (persons per frame) up to ``max_batch``: a batch of
        exactly ``max_batch`` crops replays the slot's captured hipGraph (captured when the first such batch arrives - the
        fixed-size batches of a test dataloader), smaller ones are launched kernel by kernel. Results are bit-identical to
        ``test_step``: the records that come back are float64 images of the same float32 / float64 device results
        (``output_heatmaps`` is not carried by the record: use ``test_step`` for that)."""
        from .pipeline import StepPipeline

        assert self.with_head, "The model must have head to perform prediction."
        if bool(self.test_cfg.get("output_heatmaps", False)):
            raise NotImplementedError("test_step_stream returns the per-keypoint records only; output_heatmaps needs test_step")
        flip = self._check_flip_cfg()
        pipe, pending = None, []  # pending: (ticket, n, data_samples)

        def collect(entry):
            ticket, n, samples = entry
            rec = pipe.result(ticket)[0, :n].numpy().copy()  # (n, K, 7) float64: x, y, conf, prob, vis, oks, err (raw)
            return self.add_pred_to_datasample(self.head.pack_records(rec, self.test_cfg), None, samples)

        for data in batches:
            data = self.data_preprocessor(data, False)
            inputs, samples = data["inputs"], data["data_samples"]
            if isinstance(inputs, list):
                inputs = torch.stack(inputs)
            if self.metainfo is not None:
                for ds in samples:
                    ds.set_metainfo(self.metainfo)
            if pipe is None:
                fi = samples[0].metainfo["flip_indices"] if flip else None
                pipe = StepPipeline(self.engine, max_batch, fi, flip_test=flip, depth=depth,
                                    use_graph="full" if self.graph_replay else False, shift_heatmap=self._shift_heatmap)
            if inputs.shape[0] > max_batch:
                raise ValueError(f"batch of {inputs.shape[0]} crops exceeds max_batch={max_batch}")
            while len(pending) >= depth:  # the slot about to be reused must have been read
                yield collect(pending.pop(0))
            pending.append((pipe.submit(inputs), inputs.shape[0], samples))
        while pending:
            yield collect(pending.pop(0))

    def add_pred_to_datasample(self, batch_pred_instances, batch_pred_fields, batch_data_samples):
        """topdown.py:128-194. The input -> image space map of the keypoints (:165-167) is evaluated for the whole batch in one
        numpy expression when the per-crop keypoint arrays are views of one batch array (``ProbMapHead.pack_records``) and the
        samples' ``input_center`` / ``input_scale`` / ``input_size`` agree in dtype - the same elementwise operations in the same
        dtypes as the per-sample expression, hence the same bits - and sample by sample otherwise."""
        assert len(batch_pred_instances) == len(batch_data_samples)
        if batch_pred_fields is None:
            batch_pred_fields = []
        output_keypoint_indices = self.test_cfg.get("output_keypoint_indices", None)
        mapped = self._map_batch_to_image_space(batch_pred_instances, batch_data_samples)
        for pred_instances, pred_fields, data_sample in zip_longest(batch_pred_instances, batch_pred_fields,
                                                                    batch_data_samples):
            if pred_instances is None:
                continue
            gt_instances = data_sample.gt_instances
            if not mapped:
                input_center = data_sample.metainfo["input_center"]
                input_scale = data_sample.metainfo["input_scale"]
                input_size = data_sample.metainfo["input_size"]
                pred_instances.keypoints[..., :2] = (
                    pred_instances.keypoints[..., :2] / input_size * input_scale + input_center - 0.5 * input_scale
                )
            if "keypoints_visible" not in pred_instances:
                pred_instances.keypoints_visible = pred_instances.keypoint_scores
            if output_keypoint_indices is not None:
                num_keypoints = pred_instances.keypoints.shape[1]
                for key, value in pred_instances.all_items():
                    if key.startswith("keypoint"):
                        pred_instances.set_field(value[:, output_keypoint_indices], key)
            pred_instances.bboxes = gt_instances.bboxes
            pred_instances.bbox_scores = gt_instances.bbox_scores
            data_sample.pred_instances = pred_instances
            if pred_fields is not None:
                if output_keypoint_indices is not None:
                    for key, value in pred_fields.all_items():
                        if value.shape[0] != num_keypoints:
                            continue
                        pred_fields.set_field(value[output_keypoint_indices], key)
                data_sample.pred_fields = pred_fields
        return batch_data_samples

    @staticmethod
    def _map_batch_to_image_space(batch_pred_instances, batch_data_samples) -> bool:
        """One evaluation of topdown.py:165-167 for the batch; False = not applicable (the caller maps sample by sample)."""
        base = getattr(batch_pred_instances, "keypoints_batch", None)
        n = len(batch_pred_instances)
        if base is None or n < 2 or base.ndim != 3 or base.shape[0] != n or base.dtype != np.float64:
            return False
        for p in batch_pred_instances:  # every sample still holds ITS row of the batch array (nobody swapped a field in between)
            if p is None or p.keypoints.base is not base or p.keypoints.shape[0] != 1:
                return False
        try:
            cen = [np.asarray(ds.input_center) for ds in batch_data_samples]
            sca = [np.asarray(ds.input_scale) for ds in batch_data_samples]
            siz = [np.asarray(ds.input_size) for ds in batch_data_samples]
        except AttributeError:
            return False
        for arrs in (cen, sca, siz):
            d0 = arrs[0].dtype
            if any(a.shape != (2,) or a.dtype != d0 for a in arrs):
                return False
        cen, sca, siz = (np.stack(a)[:, None, :] for a in (cen, sca, siz))  # (B, 1, 2): broadcast over the K keypoints
        base[..., :2] = base[..., :2] / siz * sca + cen - 0.5 * sca
        # the batch array is now in image space: a second add_pred_to_datasample on the same preds takes the per-sample path (which, like the
        # reference's in-place `pred_instances.keypoints[..., :2] = ...`, maps whatever it is given once more - per sample, visibly)
        batch_pred_instances.keypoints_batch = None
        return True


class _BatchPreds(list):
    """``list[InstanceData]`` of one batch whose ``keypoints`` are rows of ONE (B, K, 2) array, kept in ``keypoints_batch``."""

    keypoints_batch = None


def build_pose_estimator(cfg: dict):
    """mmpose/models/builder.py:33-35."""
    return MODELS.build(cfg)
