"""Oracle (test infrastructure, NOT product): torch-CPU float32 restatement of the ProbPose
network and of ``TopdownPoseEstimator.predict`` around it.

PARITY UNPINNED for the network: none of these pieces can be imported from the reference
in this container and the reference ships no test or golden vector for them.
Each function below follows the cited reference lines / the published third-party
definition with stock ``torch.nn.functional`` ops; ``tests/test_model_oracle.py`` holds
known-answer tests that pin the restatement itself.

* data preprocessor ... ``mmpose/models/data_preprocessors/data_preprocessor.py:79-104`` + mmengine
  ``ImgDataPreprocessor`` [3P]: BGR->RGB channel swap, float(), (x - mean) / std.
* backbone ............ ``mmpretrain==1.2.0`` ``VisionTransformer`` [3P, un-vendored]; ctor args at
  ``configs/body_2d_keypoint/topdown_probmap/coco/td-pm_ProbPose-small_8xb64-210e_coco-256x192.py:56-67``:
  patch-embed Conv2d(3->E, k16, s16, zero pad 2) -> + pos_embed -> L x [x += Attn(LN(x)); x += FFN(LN(x))]
  (qkv Linear with bias, softmax(q k^T / sqrt(hd)) v, proj; FFN Linear-GELU(erf)-Linear; LN eps 1e-6)
  -> final LN -> (B, E, Hp, Wp).
* Sparsemax ........... PyPI ``sparsemax`` [3P, unpinned]; Martins & Astudillo 2016, the sort +
  cumulative-sum form that package implements, in float32.
* ProbMapHead ......... ``mmpose/models/heads/hybrid_heads/probmap_head.py``: ``:435-472`` deconv spec,
  ``:244-251`` final conv + Sparsemax, ``:627-648`` order (/T -> sparsemax -> *normalize -> clamp),
  ``:261-410`` the four scalar towers, ``:746-798`` flip-test + packaging.
* estimator ........... ``mmpose/models/pose_estimators/topdown.py:86-194``.
"""
import math
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F

from . import decode_ref as D

TOWERS = ("probability", "visibility", "oks", "error")
POOLS = ((4, 3), (2, 2), (2, 2))  # probmap_head.py:264


# --------------------------------------------------------------------------- preprocessing
def preprocess(imgs_u8_bgr: torch.Tensor, mean: Sequence[float], std: Sequence[float], bgr_to_rgb=True):
    """(B,3,H,W) uint8 (BGR, as cv2/`LoadImage` produce) -> float32 normalised RGB."""
    x = imgs_u8_bgr
    if bgr_to_rgb:
        x = x[:, [2, 1, 0]]
    x = x.float()
    m = torch.tensor(mean, dtype=torch.float32).view(1, 3, 1, 1)
    s = torch.tensor(std, dtype=torch.float32).view(1, 3, 1, 1)
    return (x - m) / s


# --------------------------------------------------------------------------- backbone
def vit_forward(sd: Dict[str, torch.Tensor], x: torch.Tensor, num_heads: int, patch: int = 16, pad: int = 2,
                ln_eps: float = 1e-6, prefix: str = "backbone.") -> torch.Tensor:
    """mmpretrain VisionTransformer (with_cls_token=False, out_type='featmap', final_norm=True)."""
    p = lambda k: sd[prefix + k]  # noqa: E731
    x = F.conv2d(x, p("patch_embed.projection.weight"), p("patch_embed.projection.bias"), stride=patch, padding=pad)
    B, E, Hp, Wp = x.shape
    x = x.flatten(2).transpose(1, 2)  # (B, N, E)
    x = x + p("pos_embed")
    n_layers = 1 + max(int(k.split(".")[2]) for k in sd if k.startswith(prefix + "layers."))
    hd = E // num_heads
    for i in range(n_layers):
        q = lambda k: p(f"layers.{i}.{k}")  # noqa: E731
        h = F.layer_norm(x, (E,), q("ln1.weight"), q("ln1.bias"), ln_eps)
        qkv = F.linear(h, q("attn.qkv.weight"), q("attn.qkv.bias"))
        qkv = qkv.reshape(B, -1, 3, num_heads, hd).permute(2, 0, 3, 1, 4)
        att = (qkv[0] @ qkv[1].transpose(-2, -1)) * (hd**-0.5)
        att = att.softmax(dim=-1)
        h = (att @ qkv[2]).transpose(1, 2).reshape(B, -1, E)
        x = x + F.linear(h, q("attn.proj.weight"), q("attn.proj.bias"))
        h = F.layer_norm(x, (E,), q("ln2.weight"), q("ln2.bias"), ln_eps)
        h = F.gelu(F.linear(h, q("ffn.layers.0.0.weight"), q("ffn.layers.0.0.bias")))  # exact erf GELU
        x = x + F.linear(h, q("ffn.layers.1.weight"), q("ffn.layers.1.bias"))
    x = F.layer_norm(x, (E,), p("ln1.weight"), p("ln1.bias"), ln_eps)
    return x.reshape(B, Hp, Wp, E).permute(0, 3, 1, 2).contiguous()


# --------------------------------------------------------------------------- sparsemax
def sparsemax(z: torch.Tensor) -> torch.Tensor:
    """Euclidean projection of each row (last dim) onto the simplex, float32, sort-based.

    z <- z - max z; sort descending; k = max{j : 1 + j z_(j) > sum_{i<=j} z_(i)};
    tau = (sum_{i<=k} z_(i) - 1) / k; out = max(z - tau, 0).
    """
    z = z - z.max(dim=-1, keepdim=True).values
    zs = z.sort(dim=-1, descending=True).values
    rng = torch.arange(1, z.shape[-1] + 1, dtype=z.dtype).expand_as(zs)
    is_gt = (1 + rng * zs > zs.cumsum(-1)).to(z.dtype)
    k = (is_gt * rng).max(dim=-1, keepdim=True).values
    tau = ((is_gt * zs).sum(-1, keepdim=True) - 1) / k
    return torch.clamp(z - tau, min=0)


# --------------------------------------------------------------------------- head
def _bn(x, sd, name, eps=1e-5):
    return F.batch_norm(x, sd[name + ".running_mean"], sd[name + ".running_var"], sd[name + ".weight"],
                        sd[name + ".bias"], training=False, eps=eps)


def head_heatmap(sd, feat, temperature=0.5, normalize: Optional[float] = 1.0, prefix="head.", return_logits=False):
    """probmap_head.py:627-648 with deconv_layers = [deconv,BN,ReLU]*n (:435-472) and final 1x1 conv (:244-249)."""
    x = feat
    i = 0
    while prefix + f"deconv_layers.{i}.weight" in sd:
        x = F.conv_transpose2d(x, sd[prefix + f"deconv_layers.{i}.weight"], None, stride=2, padding=1, output_padding=0)
        x = F.relu(_bn(x, sd, prefix + f"deconv_layers.{i + 1}"))
        i += 3
    logits = F.conv2d(x, sd[prefix + "final_layer.weight"], sd[prefix + "final_layer.bias"])
    B, C, H, W = logits.shape
    x = logits.reshape(B, C, H * W)
    if normalize is not None:
        x = sparsemax(x / temperature)
        x = x * normalize
    else:
        x = x / temperature
    x = torch.clamp(x, 0, 1).reshape(B, C, H, W)
    return (x, logits) if return_logits else x


def head_tower(sd, feat, name, prefix="head."):
    """probmap_head.py:261-294 (identical for visibility :296-339, oks :341-375, error :377-410):
    3 x [Conv3x3(p1) -> BN -> MaxPool(k=s) -> ReLU] -> Conv1x1 -> Sigmoid (error: ReLU)."""
    x = feat
    base = f"{prefix}{name}_layers."
    for j, pool in enumerate(POOLS):
        x = F.conv2d(x, sd[base + f"{4 * j}.weight"], sd[base + f"{4 * j}.bias"], padding=1)
        x = _bn(x, sd, base + f"{4 * j + 1}")
        x = F.relu(F.max_pool2d(x, pool, pool))
    x = F.conv2d(x, sd[base + "12.weight"], sd[base + "12.bias"])
    return F.relu(x) if name == "error" else torch.sigmoid(x)


def head_forward(sd, feat, **kw):
    """probmap_head.py:600-625 -> heatmaps, probabilities, visibilities, oks, errors."""
    return (head_heatmap(sd, feat, **kw),) + tuple(head_tower(sd, feat, t) for t in TOWERS)


# --------------------------------------------------------------------------- estimator
def predict(sd, imgs_u8_bgr: torch.Tensor, num_heads: int, mean, std, input_size=(192, 256), flip_test=True,
            flip_indices=D.COCO_FLIP_INDICES, input_center=None, input_scale=None, decode_backend="scipy",
            normalize=1.0, freeze_oks=False, shift_heatmap=False, dtype=torch.float32) -> Dict[str, np.ndarray]:
    """TopdownPoseEstimator.predict (topdown.py:86-126) + ProbMapHead.predict (probmap_head.py:715-804)
    + add_pred_to_datasample (topdown.py:128-194), batched; returns the pred_instances fields stacked
    over the batch plus the intermediate tensors parity tests compare. ``dtype=torch.float64`` runs the SAME network in double precision
    (not what the reference does - it is the yardstick that says how far the fp32 reference itself sits from the exact result, i.e. what
    two correct fp32 implementations may differ by); the maps go to the decode as float32 like the reference's."""
    if dtype != torch.float32:
        sd = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in sd.items()}
    with torch.no_grad():
        x = preprocess(imgs_u8_bgr, mean, std).to(dtype)
        feat = vit_forward(sd, x, num_heads)
        htm, prob, vis, oks, err = head_forward(sd, feat, normalize=normalize)
        if flip_test:
            feat_f = vit_forward(sd, x.flip(-1), num_heads)
            htm_f, prob_f, vis_f, oks_f, err_f = head_forward(sd, feat_f, normalize=normalize)
            fi = list(flip_indices)
            back = htm_f.flip(-1)[:, fi]
            if shift_heatmap:  # flip_heatmaps(..., shift_heatmap=True), tta.py:64-66
                back = torch.cat([back[..., :1], back[..., :-1]], dim=-1)
            heat = (htm + back) * 0.5
            prob = (prob + prob_f[:, fi]) * 0.5
            vis = (vis + vis_f[:, fi]) * 0.5
            oks = (oks + oks_f[:, fi]) * 0.5
            err = (err + err_f[:, fi]) * 0.5
        else:
            heat = htm
    B, K, H, W = heat.shape
    heat, prob, vis, oks, err, feat = (t.float() for t in (heat, prob, vis, oks, err, feat))
    heat_np = heat.numpy()
    kpts, conf = [], []
    for b in range(B):  # base_head.py:69-77: per-sample loop
        k_, s_ = D.probmap_decode(heat_np[b], input_size, (W, H), backend=decode_backend)
        kpts.append(k_)
        conf.append(s_)
    kpts = np.stack(kpts)  # (B,1,K,2) f64
    conf = np.stack(conf)  # (B,1,K) f32
    out = dict(
        heatmaps=heat_np,
        keypoints_input_space=kpts.copy(),
        keypoints_conf=conf,
        keypoints_probs=prob.numpy().reshape(B, 1, K),
        keypoints_visible=vis.numpy().reshape(B, 1, K),
        keypoints_oks=oks.numpy().reshape(B, 1, K),
        keypoints_error=err.numpy().reshape(B, 1, K) / np.sqrt(H**2 + W**2),  # probmap_head.py:786-787
        features=feat.numpy(),
    )
    out["keypoint_scores"] = out["keypoints_conf"] if freeze_oks else out["keypoints_oks"]  # :797-798
    if input_center is not None:
        out["keypoints"] = np.stack([
            D.to_image_space(kpts[b], np.array(input_size), np.asarray(input_center[b]), np.asarray(input_scale[b]))
            for b in range(B)
        ])
    else:
        out["keypoints"] = kpts
    return out
