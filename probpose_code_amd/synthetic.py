"""Seeded synthetic weights and inputs for tests and ``bench.py`` (there is no network for
the published ``ProbPose-s.pth``, reference ``README.md:119-120``).

The state dict uses the reference's own key names (``backbone.*`` as
mmpretrain's ``VisionTransformer`` names them, ``head.*`` from ``probmap_head.py``), so the
same loader path that would ingest the real checkpoint is exercised.

Why not the reference's default init (``probmap_head.py:592-598``: Normal std 0.001)? It gives
logits ~ 0, Sparsemax ~ uniform 1/3072 and an argmax that is pure rounding noise,
which makes keypoint parity meaningless. The init below keeps activations O(1) through the
network and scales the last 1x1 conv so that logits have a spread of a few units: after
``/temperature`` + Sparsemax the maps are sparse and peaked like a trained model's. BatchNorm
running statistics and affine parameters are randomised so that BN folding is really tested.
"""
import math
from typing import Dict, Sequence

import torch

COCO_FLIP_INDICES = (0, 2, 1, 4, 3, 6, 5, 8, 7, 10, 9, 12, 11, 14, 13, 16, 15)
IMG_MEAN = (123.675, 116.28, 103.53)
IMG_STD = (58.395, 57.12, 57.375)

ARCHS = {
    # config :58 (explicit dict): embed 384, 12 layers, 12 heads (head_dim 32), FFN 1536
    "small": dict(embed_dims=384, num_layers=12, num_heads=12, feedforward_channels=1536),
    # mmpretrain preset "base" (BASELINE config 4): 768 / 12 / 12 heads (head_dim 64) / 3072
    "base": dict(embed_dims=768, num_layers=12, num_heads=12, feedforward_channels=3072),
}


TRAINED_MASSIVE = (250.0, -120.0, 60.0)  # residual-stream values of the massive-activation channels (stats="trained")


def _trained_like(sd: Dict[str, torch.Tensor], E: int, L: int, heads: int, g: torch.Generator,
                  massive=TRAINED_MASSIVE, row_offset: float = 8.0, gamma_decades: float = 2.0,
                  small_rows: float = 1e-2) -> Dict[str, torch.Tensor]:
    """Re-parametrise the O(1) network above into the statistics a TRAINED ViT shows (the published ``ProbPose-s.pth``,
    ``README.md:119-120``, cannot be fetched; this is what its numerics must survive) - in place, same key names:

      * massive activations (Sun et al. 2024): ``len(massive)`` channels of the residual stream sit at 60 .. 250x the
        typical magnitude from the second layer on - injected through that layer's ``fc2`` bias, its ``fc2`` rows for
        those channels 20x wider so that the values also vary per token; every LayerNorm behind it carries a small
        gamma there and a gamma ~ row std on the ordinary channels (what training arrives at: the normalised ordinary
        channels would otherwise shrink to 1 / std);
      * LayerNorm gamma log-uniform over ``gamma_decades`` decades, the Linear layer behind it with the inverse column
        scale (function preserved, the stored weights are what changes);
      * a per-token offset of the whole row of up to ``row_offset`` (token mean / std up to ~ ``row_offset`` in the first
        layers) through ``pos_embed``;
      * weight rows down to ``small_rows`` of their width: q rows scaled by s with the k rows of the same head
        dimension by 1 / s, v rows by s with the ``proj`` columns by 1 / s, ``fc1`` rows by s with the ``fc2`` columns
        by 1 / sqrt(s) (fp16 subnormal low halves in the split format).
    """
    u = lambda *s: torch.rand(*s, generator=g)  # noqa: E731
    hd = E // heads
    n_mass = len(massive)
    chans = torch.randperm(E, generator=g)[:n_mass]
    inject = min(1, L - 1)
    mass_std = math.sqrt(sum(m * m for m in massive) / E)  # what the massive channels add to a row's std
    sd["backbone.pos_embed"] = sd["backbone.pos_embed"] + (u(1, sd["backbone.pos_embed"].shape[1], 1) * 2 - 1) * row_offset
    p = f"backbone.layers.{inject}."
    sd[p + "ffn.layers.1.bias"][chans] += torch.tensor(massive)
    sd[p + "ffn.layers.1.weight"][chans] *= 20.0

    def widen(depth):
        """What a LayerNorm behind the injection must multiply the ordinary channels with to hand on what the O(1) network's
        LayerNorm would: row std with the massive channels / row std without (the latter grows ~ 0.8 + 0.38 per layer)."""
        return math.sqrt(1.0 + (mass_std / (0.8 + 0.38 * depth)) ** 2)

    def regamma(ln_key, lin_keys, depth, after_massive):
        gam = 10.0 ** ((u(E) - 0.5) * gamma_decades)
        comp = 1.0 / gam  # the Linear behind sees gamma * norm(x): keep the product
        sd[ln_key + ".bias"] = sd[ln_key + ".bias"] * gam
        if after_massive:
            gam = gam * widen(depth)
            gam[chans] = 0.02 + 0.08 * u(n_mass)
            comp[chans] = 1.0
        sd[ln_key + ".weight"] = sd[ln_key + ".weight"] * gam
        for k in lin_keys:
            sd[k] = sd[k] * comp[None, :]

    for i in range(L):
        p = f"backbone.layers.{i}."
        regamma(p + "ln1", [p + "attn.qkv.weight"], i, i > inject)
        regamma(p + "ln2", [p + "ffn.layers.0.0.weight"], i + 0.5, i > inject)
        # small rows, function preserved
        s = small_rows ** u(heads, hd)  # log-uniform in [small_rows, 1]
        wq = sd[p + "attn.qkv.weight"].reshape(3, heads, hd, E)
        bq = sd[p + "attn.qkv.bias"].reshape(3, heads, hd)
        sv = small_rows ** u(heads, hd)
        for part, sc in ((0, s), (1, 1.0 / s), (2, sv)):
            wq[part] *= sc[..., None]
            bq[part] *= sc
        sd[p + "attn.proj.weight"] = sd[p + "attn.proj.weight"] / sv.reshape(1, E)
        sf = small_rows ** u(sd[p + "ffn.layers.0.0.weight"].shape[0])
        sd[p + "ffn.layers.0.0.weight"] = sd[p + "ffn.layers.0.0.weight"] * sf[:, None]
        sd[p + "ffn.layers.0.0.bias"] = sd[p + "ffn.layers.0.0.bias"] * sf
        sd[p + "ffn.layers.1.weight"] = sd[p + "ffn.layers.1.weight"] / sf.sqrt()[None, :]
    # the final LayerNorm feeds the head: small gamma on the massive channels, ordinary channels back at O(1)
    if L > inject + 1:
        gf = sd["backbone.ln1.weight"] * widen(L)
        gf[chans] = 0.02 + 0.08 * u(n_mass)
        sd["backbone.ln1.weight"] = gf
    return sd


def synthetic_state_dict(arch: str = "small", img_size=(256, 192), num_keypoints: int = 17,
                         deconv_out_channels: Sequence[int] = (256, 256), seed: int = 0,
                         logit_scale: float = 3.0, stats: str = "unit", **trained_kw) -> Dict[str, torch.Tensor]:
    """``stats="unit"``: activations O(1) everywhere (the weights every round-1..5 parity figure was taken on);
    ``stats="trained"``: the same network re-parametrised to a trained ViT's statistics (``_trained_like``)."""
    if stats not in ("unit", "trained"):
        raise ValueError(f"stats must be 'unit' or 'trained', got {stats!r}")
    a = ARCHS[arch] if isinstance(arch, str) else arch
    E, L, Fd = a["embed_dims"], a["num_layers"], a["feedforward_channels"]
    g = torch.Generator().manual_seed(seed)
    n = lambda *s, std=1.0: torch.randn(*s, generator=g) * std  # noqa: E731
    u = lambda *s, lo=0.5, hi=1.5: torch.rand(*s, generator=g) * (hi - lo) + lo  # noqa: E731
    xav = lambda o, i: n(o, i, std=math.sqrt(2.0 / (o + i)))  # noqa: E731
    sd: Dict[str, torch.Tensor] = {}
    P = 16
    Hp, Wp = (img_size[0] + 4 - P) // P + 1, (img_size[1] + 4 - P) // P + 1
    sd["backbone.patch_embed.projection.weight"] = n(E, 3, P, P, std=math.sqrt(1.0 / (3 * P * P)))
    sd["backbone.patch_embed.projection.bias"] = n(E, std=0.1)
    sd["backbone.pos_embed"] = n(1, Hp * Wp, E, std=0.3)
    for i in range(L):
        p = f"backbone.layers.{i}."
        for ln in ("ln1", "ln2"):
            sd[p + ln + ".weight"] = 1.0 + n(E, std=0.1)
            sd[p + ln + ".bias"] = n(E, std=0.1)
        sd[p + "attn.qkv.weight"] = n(3 * E, E, std=1.5 / math.sqrt(E))
        sd[p + "attn.qkv.bias"] = n(3 * E, std=0.1)
        sd[p + "attn.proj.weight"] = xav(E, E)
        sd[p + "attn.proj.bias"] = n(E, std=0.05)
        sd[p + "ffn.layers.0.0.weight"] = xav(Fd, E)
        sd[p + "ffn.layers.0.0.bias"] = n(Fd, std=0.1)
        sd[p + "ffn.layers.1.weight"] = xav(E, Fd)
        sd[p + "ffn.layers.1.bias"] = n(E, std=0.05)
    sd["backbone.ln1.weight"] = 1.0 + n(E, std=0.1)
    sd["backbone.ln1.bias"] = n(E, std=0.1)

    def bn(prefix, c):
        sd[prefix + ".weight"] = u(c)
        sd[prefix + ".bias"] = n(c, std=0.2)
        sd[prefix + ".running_mean"] = n(c, std=0.2)
        sd[prefix + ".running_var"] = u(c)
        sd[prefix + ".num_batches_tracked"] = torch.tensor(1000)

    cin = E
    for j, cout in enumerate(deconv_out_channels):
        # ConvTranspose2d weight is (Cin, Cout, 4, 4); each output pixel sees 2x2 taps x Cin inputs
        sd[f"head.deconv_layers.{3 * j}.weight"] = n(cin, cout, 4, 4, std=math.sqrt(2.0 / (4 * cin)))
        bn(f"head.deconv_layers.{3 * j + 1}", cout)
        cin = cout
    sd["head.final_layer.weight"] = n(num_keypoints, cin, 1, 1, std=logit_scale / math.sqrt(cin))
    sd["head.final_layer.bias"] = n(num_keypoints, std=0.1)
    for t in ("probability", "visibility", "oks", "error"):
        for j in range(3):
            sd[f"head.{t}_layers.{4 * j}.weight"] = n(E, E, 3, 3, std=math.sqrt(2.0 / (9 * E)))
            sd[f"head.{t}_layers.{4 * j}.bias"] = n(E, std=0.1)
            bn(f"head.{t}_layers.{4 * j + 1}", E)
        sd[f"head.{t}_layers.12.weight"] = n(num_keypoints, E, 1, 1, std=1.0 / math.sqrt(E))
        sd[f"head.{t}_layers.12.bias"] = n(num_keypoints, std=0.3)
    if stats == "trained":
        _trained_like(sd, E, L, a["num_heads"], g, **trained_kw)
    return sd


def synthetic_crops(batch: int, img_size=(256, 192), seed: int = 0) -> torch.Tensor:
    """uint8 BGR crops (B,3,H,W), uniform [0,255] as ``mmpose/testing/_utils.py:117`` does, but
    low-pass filtered a little so that neighbouring patches correlate like an image's."""
    g = torch.Generator().manual_seed(seed)
    H, W = img_size
    coarse = torch.rand((batch, 3, H // 8, W // 8), generator=g)
    fine = torch.rand((batch, 3, H, W), generator=g)
    img = 0.6 * torch.nn.functional.interpolate(coarse, size=(H, W), mode="bilinear", align_corners=False) + 0.4 * fine
    return (img * 255).round().clamp(0, 255).to(torch.uint8)


def whole_image_bbox_meta(batch: int, img_size=(256, 192), padding: float = 1.25):
    """input_center / input_scale of a whole-image bbox (apis/inference.py:168 + TopdownAffine
    padding, topdown_transforms.py:96-101): centre (W/2, H/2), scale (W, H) * 1.25."""
    import numpy as np

    H, W = img_size
    center = np.tile(np.array([W / 2, H / 2], np.float32), (batch, 1))
    scale = np.tile(np.array([W * padding, H * padding], np.float32), (batch, 1))
    return center, scale
