"""Checkpoint -> packed device weights for the HIP pipeline.

Accepts a state dict with the reference's key names: ``backbone.*`` as
mmpretrain's ``VisionTransformer`` names them and ``head.*`` from
``mmpose/models/heads/hybrid_heads/probmap_head.py:219,247,290,325,371,406``. Honours the two
state-dict pre-hooks of the reference: ``keypoint_head.* -> head.*`` and dropping
``data_preprocessor.mean/std`` (``mmpose/models/pose_estimators/base.py:212-243``).

Packing (all on the host, once, in fp32; then cast to the operand precision):
  * every nn.Linear / 1x1 conv -> [N, K] row-major, K contiguous (what pp_gemm consumes);
  * patch-embed Conv2d(3->E, k16) -> [E, 768], column order (c, i, j);
  * ConvTranspose2d(k4, s2, p1, bias=False) + BatchNorm (eval) -> BN folded, then split into the
    four output phases: Wph[py, px][n, (ty*2+tx)*Cin + c] = w[c, n, 3-2ty-py, 3-2tx-px] * scale[n];
  * tower Conv2d(k3, p1) + BatchNorm -> folded, [tower][n, (ky*3+kx)*Cin + c];
  * tower Conv1x1 -> [tower][K, C] fp32.
"""
import math
from dataclasses import dataclass, field
from typing import Dict, List

import torch

TOWERS = ("probability", "visibility", "oks", "error")
BN_EPS = 1e-5  # nn.BatchNorm2d default, probmap_head.py:276


def to_split(x: torch.Tensor) -> torch.Tensor:
    """fp32 (..., K) -> the split-fp16 operand format of PP_PREC_F16X3 (include/probpose_mi355x.h, csrc/pp_split.h):
    hi = fp16(x), lo = fp16(x - hi), stored in blocks of 32 elements along the last axis as 32 hi halves then 32 lo
    halves. Returned as a float32 CONTAINER of the same shape (4 bytes per element; the values are not numbers)."""
    K = x.shape[-1]
    assert K % 32 == 0, f"split-fp16 tensors need a last dimension that is a multiple of 32, got {K}"
    x = x.float()
    hi = x.half()
    lo = (x - hi.float()).half()
    lead = x.shape[:-1]
    blocks = torch.stack([hi.reshape(*lead, K // 32, 32), lo.reshape(*lead, K // 32, 32)], dim=-2)
    return blocks.reshape(*lead, 2 * K).contiguous().view(torch.float32)


def from_split(c: torch.Tensor) -> torch.Tensor:
    """Inverse of ``to_split`` (hi + lo in fp32)."""
    K = c.shape[-1]
    v = c.contiguous().view(torch.float16).reshape(*c.shape[:-1], K // 32, 2, 32).float()
    return (v[..., 0, :] + v[..., 1, :]).reshape(c.shape)


SPLIT_MAX = 65504.0  # largest fp16: the high half of a split-fp16 operand (csrc/pp_split.h; numeric domain in include/probpose_mi355x.h)


def check_split_range(name: str, x: torch.Tensor) -> None:
    """Refuse a weight tensor the split-fp16 container cannot hold: a non-finite value, or |w| > 65504 (its high half would be inf and every
    product with it NaN). Raised by name, at load time - the alternative is a NaN heatmap at the first batch."""
    x = x.detach()
    if not bool(torch.isfinite(x).all()):
        raise ValueError(f"weight tensor {name!r} holds non-finite values")
    m = float(x.abs().max()) if x.numel() else 0.0
    if m > SPLIT_MAX:
        raise ValueError(f"weight tensor {name!r}: max |w| = {m:.4g} exceeds the split-fp16 operand range (65504); the f16x3 mode cannot "
                         "represent it - use precision='f32', or rescale the checkpoint (include/probpose_mi355x.h, numeric domain)")


def weight_scale_exponent(w: torch.Tensor) -> int:
    """e such that the largest element of ``w * 2^e`` lies in [2^12, 2^13): the power-of-two scale a split-fp16 Linear weight tensor is stored
    with (include/probpose_mi355x.h, numeric domain). The low half of a split value is a normal fp16 number only for |x| >= 2^-3: trained ViT
    weights (~ 0.02, rows down to 1e-3) sit far below and would keep 13 - 17 of the format's 22 bits; scaled, every element down to 2^-16 of the
    tensor's largest keeps them all. The kernels multiply their accumulators by 2^-e (exact)."""
    m = float(w.detach().abs().max()) if w.numel() else 0.0
    if not m > 0.0 or not math.isfinite(m):
        return 0
    return max(-40, min(40, 12 - math.floor(math.log2(m))))


def balance_attention_dims(sd: Dict[str, torch.Tensor], prefix: str, heads: int) -> None:
    """Exact re-parametrisation of one attention block for the split-fp16 format, in place: per head dimension d a power of two moves magnitude
    between the q row and the k row (q_d k_d is what the logits see: q_d * 2^a and k_d * 2^-a leave every product unchanged) and between the v row
    and the projection's column d (att_d = sum_t P_t v_td is linear in v_d). Chosen so that the two partners' largest weights agree to a factor
    of two. Why: the format carries a VALUE to an absolute 2^-25 below 2^-3 (include/probpose_mi355x.h, numeric domain). A checkpoint whose q
    dimension is 1000x smaller than the k dimension it meets computes the same logits in fp32, but its q activations sit in the split format's
    subnormal band while they are multiplied with k values of the hundreds - measured (scripts/r06/trained_stats_ablation.py, weight rows down to
    1e-3): features 2.6e-4 off, keypoints 1.6e-3 px, one argmax flip; balanced: the O(1) network's figures. mmpretrain MultiheadAttention [3P]:
    qkv rows [q | k | v], each (heads, head_dim)."""
    wq = sd[prefix + "attn.qkv.weight"]
    E = wq.shape[1]
    hd = E // heads
    w = wq.detach().clone().double().reshape(3, heads, hd, E)
    bq = sd.get(prefix + "attn.qkv.bias")
    b = bq.detach().clone().double().reshape(3, heads, hd) if bq is not None else None
    wp = sd[prefix + "attn.proj.weight"].detach().clone().double()  # (E, E): column h * hd + d multiplies att dim d of head h

    def exps(n_small, n_big):
        ok = (n_small > 0) & (n_big > 0)
        a = torch.zeros_like(n_small)
        a[ok] = torch.round(0.5 * torch.log2(n_big[ok] / n_small[ok]))
        return a.clamp(-30, 30)

    a_qk = exps(w[0].abs().amax(dim=-1), w[1].abs().amax(dim=-1))  # (heads, hd): q * 2^a, k * 2^-a
    a_vp = exps(w[2].abs().amax(dim=-1), wp.abs().amax(dim=0).reshape(heads, hd))  # v * 2^a, proj column * 2^-a
    w[0] *= (2.0 ** a_qk)[..., None]
    w[1] *= (2.0 ** -a_qk)[..., None]
    w[2] *= (2.0 ** a_vp)[..., None]
    wp *= (2.0 ** -a_vp).reshape(1, E)
    if b is not None:
        b[0] *= 2.0 ** a_qk
        b[1] *= 2.0 ** -a_qk
        b[2] *= 2.0 ** a_vp
        sd[prefix + "attn.qkv.bias"] = b.reshape(-1).to(bq.dtype)
    sd[prefix + "attn.qkv.weight"] = w.reshape(3 * E, E).to(wq.dtype)
    sd[prefix + "attn.proj.weight"] = wp.to(wq.dtype)


def fold_layernorm(w: torch.Tensor, b: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, name: str = "folded weights", scale_exp: int = 0):
    """Linear(LayerNorm(x)) with the LayerNorm's affine part folded into the Linear layer, as pp_linear_ln_folded consumes it
    (include/probpose_mi355x.h): ``W'[n, k] = W[n, k] gamma[k] 2^scale_exp`` in the split-fp16 container, ``colsum[n] = sum_k W'[n, k]`` of the
    ROUNDED split values (what the MFMAs multiply the row mean with: it carries the scale too), ``bias'[n] = b[n] + sum_k W[n, k] beta[k]``; sums
    in fp64. mmpretrain TransformerEncoderLayer [3P]: ``attn(ln1(x))`` / ``ffn(ln2(x))``."""
    wd, g, be = w.double(), gamma.double(), beta.double()
    check_split_range(name, wd * g[None, :])
    wf = to_split((wd * g[None, :] * 2.0 ** scale_exp).float().contiguous())
    colsum = from_split(wf).double().sum(dim=1).float().contiguous()
    bias = (b.double() + wd @ be).float().contiguous()
    return wf, colsum, bias


def normalize_state_dict(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    if "state_dict" in sd and isinstance(sd["state_dict"], dict):
        sd = sd["state_dict"]
    out = {}
    for k, v in sd.items():
        if k in ("data_preprocessor.mean", "data_preprocessor.std"):
            continue
        if k.startswith("keypoint_head."):
            k = "head." + k[len("keypoint_head."):]
        out[k] = v
    return out


@dataclass
class PackedWeights:
    dtype: torch.dtype
    embed_dims: int
    num_layers: int
    ffn_dims: int
    num_keypoints: int
    deconv_channels: List[int]
    t: Dict[str, torch.Tensor] = field(default_factory=dict)
    # name -> 2^-e of the split-fp16 Linear weight tensors stored as w * 2^e (weight_scale_exponent); absent = 1.0
    inv_scale: Dict[str, float] = field(default_factory=dict)

    def __getitem__(self, k):
        return self.t[k]

    def inv(self, k) -> float:
        """What a kernel multiplies its accumulators with for weight tensor ``k`` (the ``*_ws`` entry points' ``w_inv_scale``)."""
        return float(self.inv_scale.get(k, 1.0))

    def has(self, k) -> bool:
        return k in self.t


def _bn_fold(sd, name):
    scale = sd[name + ".weight"].float() / torch.sqrt(sd[name + ".running_var"].float() + BN_EPS)
    shift = sd[name + ".bias"].float() - sd[name + ".running_mean"].float() * scale
    return scale, shift


def pack_head_split(w32: torch.Tensor) -> torch.Tensor:
    """(32, 256) fp32 1x1-conv weights (rows >= K zero) -> the register image the fused deconvolution + 1x1 kernel loads
    (csrc/pp_panel_split.hip, HEAD): [column group 4][K block 2][map fragment 2][hi | lo][lane 64][8 halves]. Lane (fr, fg) of
    map fragment nf holds map 16 nf + fr and the channels 64 cg + 32 kb + {4 fg + i, 16 + 4 fg + i : i < 4} - the order in
    which a lane's accumulators of two neighbouring 16-channel fragments become the other operand."""
    assert tuple(w32.shape) == (32, 256)
    w32 = w32.float()
    hi = w32.half()
    lo = (w32 - hi.float()).half()
    lane = torch.arange(64)
    fr, fg = lane % 16, lane // 16
    j = torch.arange(8)
    ch_in_blk = torch.where(j < 4, 4 * fg[:, None] + j[None, :], 16 + 4 * fg[:, None] + (j[None, :] - 4))  # (64, 8)
    out = torch.empty((4, 2, 2, 2, 64, 8), dtype=torch.float16)
    for cg in range(4):
        for kb in range(2):
            ch = 64 * cg + 32 * kb + ch_in_blk
            for nf in range(2):
                rows = (16 * nf + fr)[:, None].expand(64, 8)
                out[cg, kb, nf, 0] = hi[rows, ch]
                out[cg, kb, nf, 1] = lo[rows, ch]
    return out.reshape(-1).view(torch.float32).contiguous()


def winograd_weights(w: torch.Tensor) -> torch.Tensor:
    """(Cout, Cin, 3, 3) folded 3 x 3 weights -> (16, Cout, Cin) fp32: U_p = (G g G^T)[a][b], p = 4 a + b, of Winograd's
    F(2x2, 3x3) (csrc/pp_winograd.hip), computed in fp64."""
    G = torch.tensor([[1.0, 0.0, 0.0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0.0, 0.0, 1.0]], dtype=torch.float64)
    u = torch.einsum("ai,ocij,bj->aboc", G, w.double(), G)
    return u.reshape(16, w.shape[0], w.shape[1]).float()


def pack(sd: Dict[str, torch.Tensor], dtype: torch.dtype, device, split: bool = False, scale_linear: bool = True,
         num_heads: int = 0, fold_ln: bool = True) -> PackedWeights:
    """``dtype``: operand dtype of the MFMA kernels (bf16 / fp32); ``split=True``: split-fp16 operands in a float32
    container (``to_split``); with it ``scale_linear``: the backbone's Linear weights (qkv, proj, fc1, fc2 and their LayerNorm-folded forms) are
    stored times a power of two per tensor, ``PackedWeights.inv(name)`` is what the ``*_ws`` launches are handed (``weight_scale_exponent``), and
    - given ``num_heads`` - every attention block's q / k and v / proj dimensions are balanced by exact powers of two (``balance_attention_dims``).
    ``fold_ln=False``: no LayerNorm-folded copies (the engine passes its ``ln_fold`` plan switch: ~200 MB at ViT-B that the unfolded plan never reads)."""
    sd = {k: v.detach().cpu() for k, v in normalize_state_dict(sd).items()}
    if split and scale_linear and num_heads > 0:
        n_layers = 1 + max(int(k.split(".")[2]) for k in sd if k.startswith("backbone.layers."))
        for i in range(n_layers):
            if sd[f"backbone.layers.{i}.attn.qkv.weight"].shape[1] % num_heads == 0:
                balance_attention_dims(sd, f"backbone.layers.{i}.", num_heads)
    f32 = lambda x: x.float().contiguous().to(device)  # noqa: E731
    inv_scale: Dict[str, float] = {}
    if split:
        def op(x, name="weights", scaled=False):
            check_split_range(name, x)
            x = x.float()
            if scaled and scale_linear:
                e = weight_scale_exponent(x)
                x = x * 2.0 ** e  # exact
                inv_scale[name] = 2.0 ** -e
            return to_split(x.contiguous()).to(device)
    else:
        op = lambda x, name=None, scaled=False: x.float().contiguous().to(dtype).to(device)  # noqa: E731
    pw = sd["backbone.patch_embed.projection.weight"]
    E = pw.shape[0]
    L = 1 + max(int(k.split(".")[2]) for k in sd if k.startswith("backbone.layers."))
    Fd = sd["backbone.layers.0.ffn.layers.0.0.weight"].shape[0]
    K = sd["head.final_layer.weight"].shape[0]
    t: Dict[str, torch.Tensor] = {}
    t["patch_w"] = op(pw.reshape(E, -1), "patch_w")
    t["patch_b"] = f32(sd["backbone.patch_embed.projection.bias"])
    t["pos_embed"] = f32(sd["backbone.pos_embed"].reshape(-1, E))
    for i in range(L):
        p = f"backbone.layers.{i}."
        for ln in ("ln1", "ln2"):
            t[f"l{i}.{ln}.w"] = f32(sd[p + ln + ".weight"])
            t[f"l{i}.{ln}.b"] = f32(sd[p + ln + ".bias"])
        t[f"l{i}.qkv.w"] = op(sd[p + "attn.qkv.weight"], f"l{i}.qkv.w", scaled=True)
        qb = sd.get(p + "attn.qkv.bias")
        t[f"l{i}.qkv.b"] = f32(qb if qb is not None else torch.zeros(3 * E))
        t[f"l{i}.proj.w"] = op(sd[p + "attn.proj.weight"], f"l{i}.proj.w", scaled=True)
        t[f"l{i}.proj.b"] = f32(sd[p + "attn.proj.bias"])
        t[f"l{i}.fc1.w"] = op(sd[p + "ffn.layers.0.0.weight"], f"l{i}.fc1.w", scaled=True)
        t[f"l{i}.fc1.b"] = f32(sd[p + "ffn.layers.0.0.bias"])
        t[f"l{i}.fc2.w"] = op(sd[p + "ffn.layers.1.weight"], f"l{i}.fc2.w", scaled=True)
        t[f"l{i}.fc2.b"] = f32(sd[p + "ffn.layers.1.bias"])
        if split and fold_ln and E == 384 and i >= 1:
            # ViT-S chain of fused layer kernels: ln1 of layers 1 .. L - 1 folded into the qkv projection (pp_qkv_attention_split_folded; the layer in front
            # leaves raw rows + statistics, pp_proj_ffn_split_folded)
            bb = sd.get(p + "attn.qkv.bias")
            e = weight_scale_exponent(sd[p + "attn.qkv.weight"].double() * sd[p + "ln1.weight"].double()[None, :]) if scale_linear else 0
            wf, _, bf = fold_layernorm(sd[p + "attn.qkv.weight"].float(), bb.float() if bb is not None else torch.zeros(3 * E),
                                       sd[p + "ln1.weight"].float(), sd[p + "ln1.bias"].float(), name=f"l{i}.qkv.wf", scale_exp=e)
            # (no column sums: the rows this projection is handed are centered, pp_qkv_attention_split_folded)
            t[f"l{i}.qkv.wf"], t[f"l{i}.qkv.bf"] = wf.to(device), bf.to(device)
            inv_scale[f"l{i}.qkv.wf"] = 2.0 ** -e
        if split and fold_ln and E % 192 == 0 and E != 384:
            # widths without a fused layer kernel (ViT-B): the folded form of the two Linear layers that follow a LayerNorm (pp_linear_ln_folded;
            # engine.py takes that plan from the row count at which the twelve-wave Linear kernel engages - the plain copies serve below it)
            for name, wk, bk, ln in (("qkv", "attn.qkv.weight", "attn.qkv.bias", "ln1"), ("fc1", "ffn.layers.0.0.weight", "ffn.layers.0.0.bias", "ln2")):
                bb = sd.get(p + bk)
                e = weight_scale_exponent(sd[p + wk].double() * sd[p + ln + ".weight"].double()[None, :]) if scale_linear else 0
                wf, cs, bf = fold_layernorm(sd[p + wk].float(), bb.float() if bb is not None else torch.zeros(sd[p + wk].shape[0]),
                                            sd[p + ln + ".weight"].float(), sd[p + ln + ".bias"].float(), name=f"l{i}.{name}.wf", scale_exp=e)
                t[f"l{i}.{name}.wf"], t[f"l{i}.{name}.cf"], t[f"l{i}.{name}.bf"] = wf.to(device), cs.to(device), bf.to(device)
                inv_scale[f"l{i}.{name}.wf"] = 2.0 ** -e
    t["ln_f.w"] = f32(sd["backbone.ln1.weight"])
    t["ln_f.b"] = f32(sd["backbone.ln1.bias"])

    # ---- heatmap branch
    deconv_channels = []
    j = 0
    while f"head.deconv_layers.{3 * j}.weight" in sd:
        w = sd[f"head.deconv_layers.{3 * j}.weight"].float()  # (Cin, Cout, 4, 4)
        assert w.shape[2:] == (4, 4), "only deconv kernel 4 / stride 2 / pad 1 (the ProbPose config) is packed"
        scale, shift = _bn_fold(sd, f"head.deconv_layers.{3 * j + 1}")
        w = w * scale.view(1, -1, 1, 1)
        cin, cout = w.shape[:2]
        ph = torch.empty((2, 2, cout, 4 * cin))
        for py in range(2):
            for px in range(2):
                for ty in range(2):
                    for tx in range(2):
                        tap = ty * 2 + tx
                        ph[py, px, :, tap * cin : (tap + 1) * cin] = w[:, :, 3 - 2 * ty - py, 3 - 2 * tx - px].t()
        t[f"deconv{j}.w"] = op(ph, f"deconv{j}.w")
        t[f"deconv{j}.b"] = f32(shift)
        deconv_channels.append(cout)
        j += 1
    t["final.w"] = op(sd["head.final_layer.weight"].reshape(K, -1), "final.w")
    t["final.b"] = f32(sd["head.final_layer.bias"])
    if K <= 32:  # zero-padded to 32 rows: the MFMA operand of the 1x1 conv fused into the last deconvolution (pp_deconv_head)
        wpad = torch.zeros((32, t["final.w"].shape[1]), dtype=torch.float32)
        wpad[:K] = sd["head.final_layer.weight"].reshape(K, -1).float()
        t["final.w_pad"] = op(wpad, "final.w_pad")
        if split and wpad.shape[1] == 256:
            t["final.w_head"] = pack_head_split(wpad).to(device)

    # ---- scalar towers
    for c in range(3):
        ws, bs, wino = [], [], []
        for tw in TOWERS:
            base = f"head.{tw}_layers."
            w = sd[base + f"{4 * c}.weight"].float()  # (Cout, Cin, 3, 3)
            b = sd[base + f"{4 * c}.bias"].float()
            scale, shift = _bn_fold(sd, base + f"{4 * c + 1}")
            w = w * scale.view(-1, 1, 1, 1)
            ws.append(w.permute(0, 2, 3, 1).reshape(w.shape[0], -1))  # [n, (ky, kx, c)]
            bs.append(b * scale + shift)
            if c == 0 and split:
                wino.append(winograd_weights(w))
        if c == 0 and split and w.shape[1] % 128 == 0 and w.shape[0] % 96 == 0:
            t["tower0.wino"] = op(torch.stack(wino), "tower0.wino")  # (4, 16, Cout, Cin): the first stage's Winograd form (pp_conv3x3_winograd_maxpool_relu)
        t[f"tower{c}.w"] = op(torch.stack(ws), f"tower{c}.w")
        t[f"tower{c}.b"] = f32(torch.stack(bs))
    t["tower_out.w"] = f32(torch.stack([sd[f"head.{tw}_layers.12.weight"].float().reshape(K, -1) for tw in TOWERS]))
    t["tower_out.b"] = f32(torch.stack([sd[f"head.{tw}_layers.12.bias"].float() for tw in TOWERS]))
    return PackedWeights(dtype=dtype, embed_dims=E, num_layers=L, ffn_dims=Fd, num_keypoints=K,
                         deconv_channels=deconv_channels, t=t, inv_scale=inv_scale)
