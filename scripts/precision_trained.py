#!/usr/bin/env python
"""f16x3 off the O(1)-weights manifold: the operand-model emulation of scripts/precision_ablation.py on
`synthetic_state_dict(stats="trained")` - massive-activation channels, LayerNorm gamma over two decades, token mean / std up to ~10,
weight rows down to 1e-2 of their width - with the LayerNorm FOLD modelled as the kernels compute it:

    plain     h = LN(x) in fp32, split(h) x split(W)                          (pp_qkv_attention_split, the unfolded plan)
    fold      split(x) x split(W gamma);  rstd (acc - mean colsum) + bias'     (pp_qkv_attention_split_folded / pp_linear_ln_folded, round 5)
    center    split(x - mean) x split(W gamma);  rstd acc + bias'              (the producer subtracts the row mean it already has)

against the SAME network in fp64 (the fp32 oracle's own distance to fp64 is printed beside it: that is the floor any fp32-accumulating
implementation shares). Columns: keypoint L_inf px on agreeing argmaxes / argmax flips / probability L_inf.

    python scripts/precision_trained.py [--crops 8] [--stats trained] [--arch small]

Test/measurement infrastructure: imports oracle/, never imported by the product.
"""
import argparse
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))

from oracle import decode_ref as D  # noqa: E402
from oracle import model_ref as M  # noqa: E402
from probpose_code_amd import synthetic as S  # noqa: E402
import precision_ablation as PA  # noqa: E402


def split_sum(x):
    hi, lo = PA.split(x, torch.float16)
    return hi + lo


class FoldNet(PA.Net):
    """PA.Net with the encoder's LayerNorm -> Linear pairs computed in the folded form where `fold[stage]` says so."""

    def __init__(self, sd, modes, fold, heads):
        super().__init__(sd, modes)
        self.fold, self.heads = fold, heads

    def ln_lin(self, stage, x, gamma, beta, w, b, eps, first):
        how = "plain" if first else self.fold.get(stage, "plain")
        if how == "plain" or self.m[stage] != "f16x3":
            return self.lin(stage, F.layer_norm(x, (x.shape[-1],), gamma, beta, eps), w, b)
        mean = x.mean(-1, keepdim=True)
        rstd = torch.rsqrt(x.var(-1, unbiased=False, keepdim=True) + eps)
        wf = (w.double() * gamma.double()[None, :]).float()
        bias = (b.double() + w.double() @ beta.double()).float()
        if how == "fold":
            colsum = split_sum(wf).double().sum(1).float()
            acc = PA.contract(lambda a, ww: a @ ww.t(), x, wf, "f16x3")
            return rstd * (acc - mean * colsum) + bias
        acc = PA.contract(lambda a, ww: a @ ww.t(), x - mean, wf, "f16x3")
        return rstd * acc + bias

    def vit(self, x, eps=1e-6):
        sd, heads = self.sd, self.heads
        p = lambda k: sd["backbone." + k]  # noqa: E731
        w = p("patch_embed.projection.weight")
        x = PA.contract(lambda a, ww: F.conv2d(a, ww, None, stride=16, padding=2), x, w, self.m["patch"])
        x = x + p("patch_embed.projection.bias").view(1, -1, 1, 1)
        B, E, Hp, Wp = x.shape
        x = x.flatten(2).transpose(1, 2) + p("pos_embed")
        hd = E // heads
        L = 1 + max(int(k.split(".")[2]) for k in sd if k.startswith("backbone.layers."))
        for i in range(L):
            q = lambda k: p(f"layers.{i}.{k}")  # noqa: E731
            qkv = self.ln_lin("qkv", x, q("ln1.weight"), q("ln1.bias"), q("attn.qkv.weight"), q("attn.qkv.bias"), eps, i == 0)
            qkv = qkv.reshape(B, -1, 3, heads, hd).permute(2, 0, 3, 1, 4)
            att = PA.contract(lambda a, b: a @ b.transpose(-2, -1), qkv[0], qkv[1], self.m["attn"]) * hd**-0.5
            att = att.softmax(-1)
            h = PA.contract(lambda a, b: a @ b, att, qkv[2], self.m["attn"]).transpose(1, 2).reshape(B, -1, E)
            x = x + self.lin("proj", h, q("attn.proj.weight"), q("attn.proj.bias"))
            if self.fold.get("residual") == "split" and self.m["qkv"] == "f16x3":
                x = split_sum(x)  # the residual stream between the Linear layers in the operand format
            h = F.gelu(self.ln_lin("fc1", x, q("ln2.weight"), q("ln2.bias"), q("ffn.layers.0.0.weight"), q("ffn.layers.0.0.bias"), eps, False))
            x = x + self.lin("fc2", h, q("ffn.layers.1.weight"), q("ffn.layers.1.bias"))
            if self.fold.get("residual") in ("split", "split_layer") and self.m["qkv"] == "f16x3":
                x = split_sum(x)
        x = F.layer_norm(x, (E,), p("ln1.weight"), p("ln1.bias"), eps)
        return x.reshape(B, Hp, Wp, E).permute(0, 3, 1, 2).contiguous()


def run64(sd, crops, heads):
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    with torch.no_grad():
        x = M.preprocess(crops, S.IMG_MEAN, S.IMG_STD).double()
        fi = list(S.COCO_FLIP_INDICES)
        f, ff = M.vit_forward(sd64, x, heads), M.vit_forward(sd64, x.flip(-1), heads)
        heat = (M.head_heatmap(sd64, f) + M.head_heatmap(sd64, ff).flip(-1)[:, fi]) * 0.5
        prob = (M.head_tower(sd64, f, "probability") + M.head_tower(sd64, ff, "probability")[:, fi]) * 0.5
    H, W = heat.shape[-2:]
    kp = np.stack([D.probmap_decode(h, (W * 4, H * 4), (W, H))[0] for h in heat.float().numpy()])
    return kp, prob.float().numpy().reshape(len(crops), -1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--crops", type=int, default=8)
    ap.add_argument("--stats", default="trained")
    ap.add_argument("--arch", default="small")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--row-offset", type=float, default=None)
    ap.add_argument("--massive", default=None, help="comma list, e.g. 2500,-1200,600")
    args = ap.parse_args()
    torch.set_num_threads(os.cpu_count() or 1)
    kw = {}
    if args.row_offset is not None:
        kw["row_offset"] = args.row_offset
    if args.massive is not None:
        kw["massive"] = tuple(float(v) for v in args.massive.split(","))
    heads = S.ARCHS[args.arch]["num_heads"]
    sd = S.synthetic_state_dict(args.arch, seed=args.seed, logit_scale=2.0, stats=args.stats, **kw)
    crops = S.synthetic_crops(args.crops, seed=100)
    ref64 = run64(sd, crops, heads)
    base = {s: "f32" for s in PA.STAGES}
    allx3 = {s: "f16x3" for s in PA.STAGES}
    rows = [
        ("fp32 oracle (torch CPU)", base, {}),
        ("f16x3 plain LayerNorm", allx3, {}),
        ("f16x3 ViT-S chain: ln1 fold, rows split per layer", allx3, {"qkv": "fold", "residual": "split_layer"}),
        ("f16x3 ViT-S chain: ln1 CENTERED fold", allx3, {"qkv": "center", "residual": "split_layer"}),
        ("f16x3 ViT-B plan: ln1 + ln2 fold, split residual", allx3, {"qkv": "fold", "fc1": "fold", "residual": "split"}),
        ("f16x3 ViT-B plan, CENTERED", allx3, {"qkv": "center", "fc1": "center", "residual": "split"}),
    ]
    print(f"{args.arch} stats={args.stats} {kw} {args.crops} crops x 17 keypoints vs the fp64 network: keypoint L_inf px / flips / probs L_inf")
    ref32 = None
    for name, modes, fold in rows:
        net = FoldNet(sd, modes, fold, heads)
        kp, prob = net.run(crops)
        if ref32 is None:
            ref32 = (kp, prob)
        a = PA.compare(kp, prob, ref64)
        b = PA.compare(kp, prob, ref32)
        print(f"  {name:52s} vs fp64 {a[0]:9.2e} px {a[1]:2d} flips probs {a[2]:.1e} | vs fp32 oracle {b[0]:9.2e} px {b[1]:2d} flips probs {b[2]:.1e}")


if __name__ == "__main__":
    main()
