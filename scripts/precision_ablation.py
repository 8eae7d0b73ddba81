#!/usr/bin/env python
"""Where does reduced-precision operand error enter the path?

CPU experiment on the oracle's network (oracle/model_ref.py): every contraction of the path takes its two operands
through an operand model

    f32      exact fp32 products                       (what the reference computes, probmap_head.py:627-648)
    bf16     both operands rounded to bf16              (PP_PREC_BF16: v_mfma_f32_16x16x32_bf16)
    f16      both operands rounded to fp16
    bf16x3   x = hi + lo in bf16; hi*hi + hi*lo + lo*hi  (3 MFMAs)
    f16x3    x = hi + lo in fp16; hi*hi + hi*lo + lo*hi  (3 MFMAs; PP_PREC_F16X3)

with fp32 accumulation, one stage at a time or all together, and reports the keypoint L_inf (input-space px, on
keypoints whose argmax agrees) and the number of argmax flips against the all-f32 run on the bench's synthetic batch.

    python scripts/precision_ablation.py [--crops 16]

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

from oracle import decode_ref as D  # noqa: E402
from oracle import model_ref as M  # noqa: E402
from probpose_code_amd import synthetic as S  # noqa: E402

STAGES = ("patch", "qkv", "attn", "proj", "fc1", "fc2", "deconv1", "deconv2", "final", "towers")


def split(x, dt):
    hi = x.to(dt)
    lo = (x - hi.float()).to(dt)
    return hi.float(), lo.float()


def contract(op, a, b, mode):
    """op(a, b) bilinear in (a, b); operands through the operand model `mode`."""
    if mode == "f32":
        return op(a, b)
    if mode in ("bf16", "f16"):
        dt = torch.bfloat16 if mode == "bf16" else torch.float16
        return op(a.to(dt).float(), b.to(dt).float())
    dt = torch.bfloat16 if mode == "bf16x3" else torch.float16
    ah, al = split(a, dt)
    bh, bl = split(b, dt)
    return op(ah, bh) + (op(ah, bl) + op(al, bh))


class Net:
    def __init__(self, sd, modes):
        self.sd, self.m = sd, modes

    def lin(self, stage, x, w, b):
        return contract(lambda a, ww: a @ ww.t(), x, w, self.m[stage]) + b

    def vit(self, x, heads=12, eps=1e-6):
        sd = self.sd
        p = lambda k: sd["backbone." + k]  # noqa: E731
        w = p("patch_embed.projection.weight")
        x = contract(lambda a, ww: F.conv2d(a, ww, None, stride=16, padding=2), x, w, self.m["patch"])
        x = x + p("patch_embed.projection.bias").view(1, -1, 1, 1)
        B, E, Hp, Wp = x.shape
        x = x.flatten(2).transpose(1, 2) + p("pos_embed")
        hd = E // heads
        for i in range(12):
            q = lambda k: p(f"layers.{i}.{k}")  # noqa: E731
            h = F.layer_norm(x, (E,), q("ln1.weight"), q("ln1.bias"), eps)
            qkv = self.lin("qkv", h, q("attn.qkv.weight"), q("attn.qkv.bias"))
            qkv = qkv.reshape(B, -1, 3, heads, hd).permute(2, 0, 3, 1, 4)
            att = contract(lambda a, b: a @ b.transpose(-2, -1), qkv[0], qkv[1], self.m["attn"]) * hd**-0.5
            att = att.softmax(-1)
            h = contract(lambda a, b: a @ b, att, qkv[2], self.m["attn"]).transpose(1, 2).reshape(B, -1, E)
            x = x + self.lin("proj", h, q("attn.proj.weight"), q("attn.proj.bias"))
            h = F.layer_norm(x, (E,), q("ln2.weight"), q("ln2.bias"), eps)
            h = F.gelu(self.lin("fc1", h, q("ffn.layers.0.0.weight"), q("ffn.layers.0.0.bias")))
            x = x + self.lin("fc2", h, q("ffn.layers.1.weight"), q("ffn.layers.1.bias"))
        x = F.layer_norm(x, (E,), p("ln1.weight"), p("ln1.bias"), eps)
        return x.reshape(B, Hp, Wp, E).permute(0, 3, 1, 2).contiguous()

    def heat(self, feat):
        sd, x = self.sd, feat
        for j, st in enumerate(("deconv1", "deconv2")):
            w = sd[f"head.deconv_layers.{3 * j}.weight"]
            x = contract(lambda a, ww: F.conv_transpose2d(a, ww, None, stride=2, padding=1), x, w, self.m[st])
            x = F.relu(M._bn(x, sd, f"head.deconv_layers.{3 * j + 1}"))
        lg = contract(lambda a, ww: F.conv2d(a, ww), x, sd["head.final_layer.weight"], self.m["final"])
        lg = lg + sd["head.final_layer.bias"].view(1, -1, 1, 1)
        B, C, H, W = lg.shape
        return torch.clamp(M.sparsemax(lg.reshape(B, C, -1) / 0.5), 0, 1).reshape(B, C, H, W)

    def tower(self, feat, name):
        sd, x = self.sd, feat
        base = f"head.{name}_layers."
        for j, pool in enumerate(M.POOLS):
            x = contract(lambda a, ww: F.conv2d(a, ww, None, padding=1), x, sd[base + f"{4 * j}.weight"], self.m["towers"])
            x = M._bn(x + sd[base + f"{4 * j}.bias"].view(1, -1, 1, 1), sd, base + f"{4 * j + 1}")
            x = F.relu(F.max_pool2d(x, pool, pool))
        x = F.conv2d(x, sd[base + "12.weight"], sd[base + "12.bias"])
        return torch.sigmoid(x)

    def run(self, crops):
        with torch.no_grad():
            x = M.preprocess(crops, S.IMG_MEAN, S.IMG_STD)
            f, ff = self.vit(x), self.vit(x.flip(-1))
            fi = list(S.COCO_FLIP_INDICES)
            heat = (self.heat(f) + self.heat(ff).flip(-1)[:, fi]) * 0.5
            prob = (self.tower(f, "probability") + self.tower(ff, "probability")[:, fi]) * 0.5
        kp = np.stack([D.probmap_decode(h, (192, 256), (48, 64))[0] for h in heat.numpy()])
        return kp, prob.numpy().reshape(len(crops), -1)


def compare(kp, prob, ref):
    d = np.abs(kp - ref[0]).max(-1)
    same = d < 2.0
    return float(d[same].max()), int((~same).sum()), float(np.abs(prob - ref[1]).max())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--crops", type=int, default=16)
    ap.add_argument("--modes", default="bf16,f16,bf16x3,f16x3")
    args = ap.parse_args()
    torch.set_num_threads(os.cpu_count() or 1)
    sd = S.synthetic_state_dict("small", seed=0, logit_scale=2.0)
    crops = S.synthetic_crops(args.crops, seed=100)
    base = {s: "f32" for s in STAGES}
    ref = Net(sd, base).run(crops)
    # the noise floor of fp32 itself: same network in fp64
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    print(f"{args.crops} crops x 17 keypoints; columns: keypoint L_inf px (same argmax) / argmax flips / probs L_inf")
    for mode in args.modes.split(","):
        print(f"--- operand model {mode}")
        for st in STAGES + ("ALL",):
            m = dict(base)
            if st == "ALL":
                m = {s: mode for s in STAGES}
            else:
                m[st] = mode
            linf, flips, pl = compare(*Net(sd, m).run(crops), ref)
            print(f"  {st:8s} {linf:10.3e} px  {flips:3d} flips  probs {pl:.2e}")
    del sd64


if __name__ == "__main__":
    main()
