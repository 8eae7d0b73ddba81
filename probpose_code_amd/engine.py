"""The launch plan of the hot path: uint8 crops in HBM -> keypoints + per-keypoint scalars.

``ProbPoseEngine`` owns the packed weights and the per-batch-size workspace (device memory
comes from torch, that is all torch does here) and enqueues the hand-written HIP kernels of
``libprobpose_mi355x.so`` on torch's current stream, in the order of
``TopdownPoseEstimator.predict`` (mmpose/models/pose_estimators/topdown.py:86-126):

  preprocess + flip copy + patch im2col -> patch-embed GEMM (+bias +pos_embed)
  -> L x [LN -> qkv GEMM -> attention -> proj GEMM (+residual) -> LN -> fc1 GEMM (GELU) -> fc2 GEMM (+residual)]
  -> final LN -> deconv x2 (4 phase GEMMs each, BN folded, ReLU) -> 1x1 conv (planar logits)
  -> fused Sparsemax + flip-average + OKS-conv + argmax + sub-pixel decode
  and, from the same features, the four scalar towers (grouped implicit-GEMM 3x3 convs,
  MaxPool+ReLU, final 1x1 + sigmoid/ReLU + flip-average).

Both flip-test passes run as one batch of 2B sequences. Activations are token-major / NHWC
throughout, so the ViT output *is* the NHWC feature map the head convolutions gather from.
"""
import math
import os
import warnings
from typing import Dict, Optional, Sequence

import numpy as np
import ctypes

import torch

from . import _lib
from .codecs import oks_kernel_taps
from .weights import PackedWeights, from_split, pack, to_split

PREC = {"bf16": 0, "f32": 1, "f16x3": 2}  # PP_PREC_*
# torch dtype of the operand buffers; "f16x3" = split fp16 (x = hi + lo, csrc/pp_split.h) in a 4-byte-per-element container
_DTYPE = {"bf16": torch.bfloat16, "f32": torch.float32, "f16x3": torch.float32}
_FMT = {"bf16": 1, "f32": 0, "f16x3": 2}  # PP_OUT_*: format code of an operand-precision output
ACT_NONE, ACT_GELU, ACT_RELU = 0, 1, 2
CONV3X3, DECONV = 1, 2


# launch-plan switches and their shipped values (fuse_resln: None = on for E = 384 only, True = also the E = 768 row-owner kernel)
PLAN_DEFAULTS = dict(fuse_mlp=True, fuse_proj=True, fuse_qkv=True, split_k=True, fuse_attn=True, fuse_head=True, fuse_resln=None,
                     fuse_pool=True, fuse_qkv_attn=True, winograd=True, ln_fold=True, small_plan=True, head_two_streams=True)
# f16x3: batches with fewer token rows than this (B * passes * tokens) take the column-parallel plan of small batches (pp_skinny_linear): below it the
# row-owner layer kernels leave most of the chip idle (96 rows per workgroup). Measured crossover (scripts/r06/small_batch_profile.py, ms per replayed
# step one in flight, small plan / row-owner plan): B = 1 0.88 / 2.06, 4: 1.19 / 2.13, 8: 1.77 / 2.24, 16: 2.33 / 2.44, 24: 3.96 / 2.85 - 6 912 rows
# = up to 17 crops with flip test
SMALL_PLAN_ROWS_BELOW = 6912


def plan_from_env() -> Dict[str, object]:
    """PP_FUSE_MLP=0 / 1 ... -> {fuse_mlp: False / True, ...}. Dev convenience only, read ONCE per engine, in the constructor;
    any other value leaves the switch at its default (a stray variable must not change the plan)."""
    out: Dict[str, object] = {}
    for k in PLAN_DEFAULTS:
        v = os.environ.get("PP_" + k.upper())
        if v in ("0", "1"):
            out[k] = v == "1"
    return out


class _CapturedGraph:
    """A captured step: the graph, its static input, its outputs, the fork / join events recorded inside the capture and the HIP stream its
    two-stream head forked to - a stream of this graph's own (``pp_stream_create``), destroyed right after the graph: on ROCm 7.0 forking to a
    stream that took part in the capture of a graph destroyed since makes the first launch of the new graph crash inside hipGraphLaunch
    (scripts/r06/graph_eager_repro.py; streams from torch's pool come round again after 32 captures)."""

    def __init__(self, graph, static_in, out, events, side_stream, raw_stream):
        self.graph, self.static_in, self.out, self.events, self.side_stream, self.raw_stream = graph, static_in, out, events, side_stream, raw_stream

    def release(self):
        """Device idle -> graph, then its events and its stream (idempotent)."""
        if self.graph is None and self.raw_stream is None:
            return
        try:
            if torch.cuda.is_available():
                torch.cuda.synchronize()
        except Exception:
            pass
        self.graph = None
        self.out = self.static_in = None
        self.events = []
        self.side_stream = None
        raw, self.raw_stream = self.raw_stream, None
        if raw:
            try:
                _lib.lib.pp_stream_destroy(ctypes.c_void_p(raw))
            except Exception:
                pass

    def __del__(self):
        self.release()


class ProbPoseEngine:
    def __init__(self, state_dict: Dict[str, torch.Tensor], num_heads: int, img_size=(256, 192), patch_size: int = 16,
                 patch_padding: int = 2, mean=(123.675, 116.28, 103.53), std=(58.395, 57.12, 57.375),
                 bgr_to_rgb: bool = True, temperature: float = 0.5, normalize: Optional[float] = 1.0,
                 input_size: Optional[Sequence[int]] = None, ln_eps: float = 1e-6, precision: str = "f16x3",
                 device="cuda", plan: Optional[Dict[str, object]] = None):
        if precision not in PREC:
            raise ValueError(f"precision must be one of {list(PREC)}, got {precision!r}")
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("ProbPoseEngine runs on the MI355X only; there is no CPU fallback")
        # Launch-plan switches: every fusion below is on in the shipped plan; `plan` (constructor argument) turns single ones off
        # for A/B timing and for the tests of the unfused kernels, PP_FUSE_* environment variables do the same from outside a
        # script (read HERE, in the host-side mirror - the C library reads no environment, include/probpose_mi355x.h).
        pl = dict(PLAN_DEFAULTS)
        pl.update(plan_from_env())
        pl.update(plan or {})
        unknown = set(pl) - set(PLAN_DEFAULTS)
        if unknown:
            raise ValueError(f"unknown launch-plan switch(es) {sorted(unknown)}; known: {sorted(PLAN_DEFAULTS)}")
        self.plan = pl
        self.precision = precision
        self.prec = PREC[precision]
        self.dtype = _DTYPE[precision]
        self.fmt = _FMT[precision]
        self.w: PackedWeights = pack(state_dict, self.dtype, self.device, split=precision == "f16x3", num_heads=int(num_heads), fold_ln=bool(pl["ln_fold"]))
        self.heads = num_heads
        self.H, self.W = img_size
        self.P, self.pad = patch_size, patch_padding
        self.Hp = (self.H + 2 * self.pad - self.P) // self.P + 1
        self.Wp = (self.W + 2 * self.pad - self.P) // self.P + 1
        self.Np = self.Hp * self.Wp
        assert self.w["pos_embed"].shape[0] == self.Np, "pos_embed does not match img_size / patch grid"
        self.E = self.w.embed_dims
        self.hd = self.E // num_heads
        self.K = self.w.num_keypoints
        self.up = 2 ** len(self.w.deconv_channels)
        self.Hh, self.Wh = self.Hp * self.up, self.Wp * self.up  # heatmap size
        self.input_size = tuple(input_size) if input_size is not None else (self.W, self.H)
        self.mean = (np.asarray(mean, np.float32)).copy()
        self.std = (np.asarray(std, np.float32)).copy()
        self.bgr_to_rgb = bool(bgr_to_rgb)
        self.temperature = float(temperature)
        self.normalize = normalize
        self.ln_eps = float(ln_eps)
        taps, radius = oks_kernel_taps(self.K, self.Hh, self.Wh)
        self.taps = torch.from_numpy(taps).to(self.device)
        self.radius = torch.from_numpy(radius).to(self.device)
        self._ws: Dict[tuple, Dict[str, torch.Tensor]] = {}
        self._flip: Dict[tuple, torch.Tensor] = {}
        self._graphs: Dict[tuple, tuple] = {}  # insertion order = least recently used first (see capture / forward_graph)
        self.max_graphs = 8  # captured graphs kept (each pins a static input and, per batch size and slot, a workspace)
        self._tick = 0  # forward / forward_graph calls so far; _graph_tick: the call at which a captured graph last replayed
        self._graph_tick: Dict[tuple, int] = {}
        self.graph_captures = 0  # captures so far (diagnostics; tests assert it stays bounded when more sizes recur than graphs are kept)
        self._opt_epoch = _lib.option_epoch()
        self.fuse_mlp, self.fuse_proj, self.fuse_qkv, self.split_k = pl["fuse_mlp"], pl["fuse_proj"], pl["fuse_qkv"], pl["split_k"]
        # f16x3: the column-parallel plan of small batches (pp_skinny_linear, _layers_small); small_rows_below: the row count it ends at
        self.small_plan = bool(precision == "f16x3" and pl["small_plan"])
        # the final 1x1 conv's weights padded to 32 rows (zeros) for pp_skinny_conv1x1_planar
        self._final_padded = None
        if self.small_plan and self.w.has("final.w") and self.w["final.w"].dim() == 2:
            fw, fb = self.w["final.w"], self.w["final.b"]
            rows = 32 * ((fw.shape[0] + 31) // 32)
            wp = torch.zeros((rows, fw.shape[1]), dtype=fw.dtype, device=fw.device)
            wp[: fw.shape[0]] = fw
            bp = torch.zeros(rows, dtype=torch.float32, device=fb.device)
            bp[: fb.shape[0]] = fb
            self._final_padded = (wp, bp)
        self.small_rows_below = SMALL_PLAN_ROWS_BELOW
        # attention inside the bf16 layer kernel (pp_vit_layer, one launch per layer; 192-token sequences, head dim 32): about 2 %
        # faster per step than pp_attention + the fused rest since the residual rows load under its attention phase
        self.fuse_attn, self.fuse_head, self.fuse_pool = pl["fuse_attn"], pl["fuse_head"], pl["fuse_pool"]
        # residual GEMM + LayerNorm in one launch (pp_gemm_ln.hip; False: GEMM, then LayerNorm). Default on for E = 384. At E = 768
        # (ViT-B) the row-owner kernel exists and is tested, but is OFF unless fuse_resln is True: measured at 384x288 bs 32 it saves
        # the LayerNorm launches one step at a time (10.72 -> 10.47 ms) and loses with two steps in flight (9.64 -> 10.01 ms: one
        # 124 KiB workgroup per CU cannot share a CU with the other step's kernels, the 128 x 128 GEMM tiles can)
        self.fuse_resln = pl["fuse_resln"] is not False
        self._resln_768 = pl["fuse_resln"] is True
        # f16x3: fc1 - GELU - fc2 + residual + LayerNorm of a layer in one launch (pp_ffn_split.hip; the hidden activation stays
        # on the CU). The kernel takes W1 / W2 as one buffer in its consumption order, packed here once per layer.
        # With fuse_proj the attention output projection + residual + ln2 run in front of it in the same launch
        # (pp_proj_ffn_split_residual_layernorm): the intermediate residual stream never leaves the CU either.
        self._ffn_packed: Dict[int, torch.Tensor] = {}
        self._proj_packed: Dict[int, torch.Tensor] = {}
        if precision == "f16x3" and self.fuse_mlp and _lib.lib.pp_ffn_split_packed_bytes(self.E, self.w.ffn_dims) > 0:
            nbytes = _lib.lib.pp_ffn_split_packed_bytes(self.E, self.w.ffn_dims)
            pbytes = _lib.lib.pp_proj_split_packed_bytes(self.E) if self.fuse_proj else -1
            with torch.cuda.device(self.device):
                for i in range(self.w.num_layers):
                    buf = torch.empty(nbytes // 4, dtype=torch.float32, device=self.device)
                    _lib.call("pp_ffn_split_pack_weights", self.w[f"l{i}.fc1.w"].data_ptr(), self.w[f"l{i}.fc2.w"].data_ptr(),
                              buf.data_ptr(), self.E, self.w.ffn_dims, _lib.stream_ptr(self.device))
                    self._ffn_packed[i] = buf
                    if pbytes > 0:
                        pbuf = torch.empty(pbytes // 4, dtype=torch.float32, device=self.device)
                        _lib.call("pp_proj_split_pack_weights", self.w[f"l{i}.proj.w"].data_ptr(), pbuf.data_ptr(), self.E,
                                  _lib.stream_ptr(self.device))
                        self._proj_packed[i] = pbuf
                torch.cuda.synchronize(self.device)
            # the fused launches read only the packed copies: release the plain ones (55 MiB at ViT-S) - unless the plan of small batches
            # (pp_skinny_linear: column-parallel Linear layers on the plain tensors) is on
            if not (pl["small_plan"]):
                for i in range(self.w.num_layers):
                    for name in (f"l{i}.fc1.w", f"l{i}.fc2.w") + ((f"l{i}.proj.w",) if i in self._proj_packed else ()):
                        self.w.t.pop(name, None)
        # f16x3, 192-token sequences of 32-dim heads: qkv Linear + attention of a layer in one launch, one workgroup per
        # (sequence, head); the qkv tensor never reaches HBM (pp_qkv_attn_split.hip). PP_FUSE_QKV_ATTN=0: pp_gemm + pp_attention
        self.fuse_qkv_attn = precision == "f16x3" and pl["fuse_qkv_attn"] and self.Np == 192 and self.hd == 32 and self.E == 384
        # f16x3 widths without a fused layer kernel (ViT-B): ln1 / ln2 folded into qkv / fc1 (weights.fold_layernorm), the statistics emitted
        # by proj / fc2, the residual stream in the operand format: no LayerNorm launch inside the layers (pp_linear_ln_folded)
        self.ln_fold = bool(precision == "f16x3" and pl["ln_fold"] and not self._proj_packed and not self._ffn_packed
                            and self.w.has("l0.fc1.wf") and self.E % 192 == 0 and self.w.ffn_dims % 192 == 0)
        # the ViT-S chain of fused layer kernels with ln1 of layers 1 .. L - 1 folded into the qkv projection: a layer's projection + FFN launch leaves
        # its rows once, in the operand format, with (mean, rstd) per row (pp_proj_ffn_split_folded), the next qkv + attention launch applies them
        # (pp_qkv_attention_split_folded): 288 KiB less to store per 96 rows and launch (the paired twelve-wave kernel only)
        self.ln_fold_fused = bool(precision == "f16x3" and pl["ln_fold"] and self.fuse_qkv_attn and self._proj_packed and self.w.has("l1.qkv.wf")
                                  and (self.w.ffn_dims // 128) % 2 == 0 and self.w.num_layers >= 2)
        # Which layer plan this geometry gets - said once, loudly, when a ViT-S-like model misses the two-launch layer only because of
        # its token count (pp_qkv_attn_split.hip is written for 192-token sequences of 32-dim heads; the projection + FFN launch takes
        # any row count): it then runs qkv Linear + attention (pp_attention) + the fused projection / FFN launch, three launches per layer
        # with the qkv tensor through HBM. ViT-B (E = 768) has no fused layer kernel at all.
        if precision == "f16x3":
            if self.fuse_qkv_attn and self._proj_packed:
                self.layer_plan = "two launches per layer (pp_qkv_attention_split + pp_proj_ffn_split_residual_layernorm)"
                if self.ln_fold_fused:
                    self.layer_plan += "; ln1 of layers 1 .. L - 1 folded into the qkv projection (pp_proj_ffn_split_folded -> pp_qkv_attention_split_folded)"
            elif self._proj_packed:
                self.layer_plan = "three launches per layer (pp_gemm qkv + pp_attention + pp_proj_ffn_split_residual_layernorm)"
            else:
                self.layer_plan = "generic (pp_gemm / pp_attention / pp_layernorm per layer)"
                if self.ln_fold:
                    self.layer_plan += ("; from the row count at which the twelve-wave Linear kernel engages: LayerNorm folded into the Linear "
                                        "layers (pp_linear_ln_folded x4 + pp_attention per layer, residual stream in the operand format)")
            if pl["fuse_qkv_attn"] and not self.fuse_qkv_attn and self.E == 384 and self.hd == 32:
                warnings.warn(
                    f"ProbPoseEngine: {self.Np}-token sequences ({img_size[0]}x{img_size[1]} input) miss the fused qkv + attention kernel, "
                    f"which is built for 192 tokens (256x192); this model runs {self.layer_plan}", RuntimeWarning, stacklevel=2)
        else:
            self.layer_plan = "bf16 / f32 plan"
        # f16x3, 16 x 12 feature maps: the first tower stage in its Winograd F(2x2, 3x3) form (pp_winograd.hip: 2.25x fewer MFMAs)
        self.winograd = (precision == "f16x3" and pl["winograd"] and self.w.has("tower0.wino")
                         and _lib.lib.pp_winograd_scratch_bytes(1, self.Hp, self.Wp, self.E) > 0)
        if self.w.has("tower0.wino") and not self.winograd:
            self.w.t.pop("tower0.wino")  # (37.7 MB at ViT-S that no launch of this plan reads)
        self._logits_phased = False
        self._head_stream: Optional[torch.cuda.Stream] = None  # second stream of the head at small batches (run_head)
        self._capture_events: list = []   # fork / join events of the graph being captured (_fork_join)
        self._eager_events: dict = {}
        self.profile: Optional[Dict[str, list]] = None
        self.stage_hook = None  # callable(name) invoked between stages of the launch plan ("embed", "layer<i>", "backbone"); dev / scheduling experiments
        # tower pooling schedule (probmap_head.py:264) and the spatial sizes it produces
        self.pools = ((4, 3), (2, 2), (2, 2))
        hs, ws_ = self.Hp, self.Wp
        self.tower_hw = []
        for ph, pw_ in self.pools:
            self.tower_hw.append((hs, ws_))
            hs, ws_ = hs // ph, ws_ // pw_
        if (hs, ws_) != (1, 1):
            raise ValueError(
                f"scalar towers reduce the {self.Hp}x{self.Wp} feature map to {hs}x{ws_}; the reference "
                "reshapes them to (B, 1, K) (probmap_head.py:780-783), which needs 1x1"
            )

    def ksplit(self, nb: int, th: int, tw: int) -> int:
        """K-slices of a small tower convolution over nb images of th x tw pixels: the library's own rule
        (pp_conv3x3_splitk_slices; PP_WS_TOWER_PARTIAL is sized by the same rule). 3 = one kernel row of taps per slice; 9 = one tap
        (stages with fewer than 1024 rows - the 2 x 2 maps at bs 64: split-K launch 48 -> 27 us, the nine-way sum costs 5 us more);
        4 = four channel ranges on the split-fp16 wide-tile kernel (the 4 x 4 maps at bs 64: one 256 x 192 tile per CU)."""
        return int(_lib.lib.pp_conv3x3_splitk_slices(self.prec, int(nb), int(th), int(tw), self.E, self.E, 4))

    # ------------------------------------------------------------------ workspace
    def _workspace(self, B: int, passes: int, slot: int = 0) -> Dict[str, torch.Tensor]:
        """Buffers of one step at batch size B. ``slot`` > 0: a second (third ...) independent set, so that consecutive
        steps can be in flight at once on different streams (pipeline.py)."""
        self._check_options()
        key = (B, passes, slot)
        if key in self._ws:
            return self._ws[key]
        dev, T, f32 = self.device, self.dtype, torch.float32
        nb = B * passes
        M = nb * self.Np
        E, Fd = self.E, self.w.ffn_dims
        dc = self.w.deconv_channels
        shape = _lib.PlanShape(prec=self.prec, n_img=nb, n_tokens=self.Np, embed=E, ffn=Fd, patch_k=3 * self.P * self.P, n_keypoints=self.K,
                               feat_h=self.Hp, feat_w=self.Wp, heat_h=self.Hh, heat_w=self.Wh, deconv_channels=dc[0] if dc else 0)

        def buf(which, dims, dt=T, index=0):
            """One buffer of the plan, sized by the library (pp_workspace_bytes): the C side owns the formats' byte sizes."""
            nbytes = _lib.workspace_bytes(which, shape, index)
            t = torch.empty(nbytes, dtype=torch.uint8, device=dev).view(dt)
            assert t.numel() == math.prod(dims), (which, index, nbytes, dims)
            return t.view(*dims)

        e = lambda *s, dt=T: torch.empty(s, dtype=dt, device=dev)  # noqa: E731  (results: sizes fixed by the output contract)
        small = self._small_at(M)
        fused_layer = self.precision == "f16x3" and self.fuse_qkv_attn and bool(self._proj_packed) and not small
        ws = dict(
            patches=buf("patches", (M, 3 * self.P * self.P)), x=buf("x", (M, E), f32), h=buf("h", (M, E)),
            feat=buf("feat", (M, E)), logits=buf("logits", (nb, self.K, self.Hh * self.Wh), f32),
            # qkv / hidden activation only where a launch plan without the fused layer kernels needs them
            qkv=None if (fused_layer or (small and self.fuse_qkv_attn)) else buf("qkv", (M, 3 * E)),
            qkv2=None if (fused_layer or small) else buf("qkv", (M, 3 * E)),
            f=None if ((fused_layer or self._ffn_packed) and not small) else buf("ffn", (M, Fd)),
            att=buf("att", (M, E)) if (self.fuse_qkv_attn or small) else None,  # attention output of the fused qkv + attention launch
            hs=buf("ln2", (M, E)) if (self._proj_packed or small) else None,  # ln2 rows of the fused projection + FFN launch (scratch, parked in L2 / MALL)
            # small-batch plan: one arrival counter per 32-row block for pp_skinny_linear's LayerNorm tail (zero between launches)
            ln_count=torch.zeros((M + 31) // 32, dtype=torch.int32, device=dev) if small else None,
            # folded-LayerNorm plan: the residual stream in the operand format and the row statistics between its Linear layers
            xs=buf("h", (M, E)) if self._ln_fold_at(M) else None,
            lnst=buf("ln_stats", (M, E // 96, 2), f32) if self._ln_fold_at(M) else None,
            rowst=e(M, 2, dt=f32) if self.ln_fold_fused else None,  # (mean, rstd) per row between pp_proj_ffn_split_folded and pp_qkv_attention_split_folded
            scalars=e(4, B, self.K, dt=f32), locs=e(B, self.K, 2, dt=f32),
            keypoints=e(B, self.K, 2, dt=torch.float64), scores=e(B, self.K, dt=f32),
            heatmaps=e(B, self.K, self.Hh, self.Wh, dt=f32),
        )
        hh, ww = self.Hp, self.Wp
        for j, c in enumerate(dc):
            hh, ww = hh * 2, ww * 2
            if c == dc[0]:
                ws[f"d{j}"] = buf("deconv", (nb, hh, ww, c), index=j)
            else:
                ws[f"d{j}"] = e(nb, hh, ww, c)
        if self.winograd:
            nbytes = _lib.workspace_bytes("winograd", shape)
            # (the kernel addresses its planes with 32-bit offsets: batches beyond ~900 crops with flip test at ViT-S take the direct kernel)
            ws["wino"] = torch.empty(nbytes, dtype=torch.uint8, device=dev) if 0 < nbytes < 0x7FFFFFF0 else None
        for j, (th, tw) in enumerate(self.tower_hw):
            ph, pw_ = self.pools[j]
            ws[f"t{j}"] = buf("tower", (4, nb, th, tw, E), index=j)
            if nb * th * tw * 4 < 128 * 128:
                ks = self.ksplit(nb, th, tw)
                ws[f"tp{j}"] = buf("tower_partial", (ks, 4, nb, th, tw, E), f32, index=j)  # split-K partial sums of the small tower stages
            ws[f"p{j}"] = buf("tower_pooled", (4, nb, th // ph, tw // pw_, E), index=j)
        self._ws[key] = ws
        return ws

    def _small_at(self, M: int) -> bool:
        """The column-parallel plan of small batches: f16x3, fewer than small_rows_below token rows, shapes pp_skinny_linear serves."""
        return bool(self.small_plan and M < self.small_rows_below and self.E % 64 == 0 and self.w.ffn_dims % 64 == 0 and self.E <= 1024
                    and self.w.has("l0.fc1.w") and self.w.has("l0.proj.w") and (3 * self.P * self.P) % 64 == 0)

    def _ln_fold_at(self, M: int) -> bool:
        """The folded-LayerNorm layer plan runs from the row count at which pp_gemm itself would pick the twelve-wave Linear kernel for the
        narrowest layer (proj: N = E); below it the generic plan (128 x 128 / wide tiles + pp_layernorm)."""
        return (self.ln_fold and _lib.get_option("linear_dma") != 0
                and _lib.lib.pp_linear_ln_folded_supported(int(M), self.E, self.E, 1) == 2
                and _lib.lib.pp_linear_ln_folded_supported(int(M), self.E, self.w.ffn_dims, 0) == 2)

    def _check_options(self) -> None:
        """Library options (pp_set_option) decide kernel selection AND buffer sizes - the split-K slice count of the small tower
        convolutions, hence PP_WS_TOWER_PARTIAL, follows "panel" / "ksplit_channels" / "psplit_tap_inner" / "ksplit9_below": a workspace
        cached, or a graph captured, before an option changed would run another plan than a fresh engine (or fail with
        PP_ERR_UNSUPPORTED for a slice count its buffer was not sized for). Options are a development hook; when one changes, the
        caches go."""
        if _lib.option_epoch() != self._opt_epoch:
            self._opt_epoch = _lib.option_epoch()
            self._ws.clear()
            self._graphs.clear()
            self._graph_tick.clear()

    def _flip_indices(self, flip_indices) -> torch.Tensor:
        key = tuple(int(i) for i in flip_indices)
        if key not in self._flip:
            assert len(key) == self.K
            self._flip[key] = torch.tensor(key, dtype=torch.int32, device=self.device)
        return self._flip[key]

    # ------------------------------------------------------------------ launches
    def _call(self, tag: str, fn: str, *args):
        """One C-ABI launch. With ``self.profile`` set (a dict), HIP events are recorded on the launch
        stream around every launch and collected per kernel tag (bench.py reads them for the roofline)."""
        if self.profile is None:
            _lib.call(fn, *args)
            return
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        _lib.call(fn, *args)
        b.record()
        self.profile.setdefault(tag, []).append((a, b))

    def _gemm(self, st, a, w, bias, out, M, N, K, act=ACT_NONE, residual=None, res_mod=0, out_bf16=None, planar=0,
              ldc=None, winv: float = 1.0):
        """``out_bf16``: PP_OUT_* format of ``out``; default = the operand format of the precision mode. ``winv``: inverse power-of-two scale of
        a split-fp16 Linear weight tensor (``PackedWeights.inv``)."""
        ob = self.fmt if out_bf16 is None else out_bf16
        self._call("gemm_bf16out" if ob else "gemm_f32out", "pp_gemm_ws", self.prec, a.data_ptr(), w.data_ptr(), _lib.ptr(bias), _lib.ptr(residual),
                   res_mod, out.data_ptr(), M, N, K, K, K, N if ldc is None else ldc, act, ob, planar, float(winv), st)

    def backbone(self, imgs_u8: torch.Tensor, passes: int, ws, st) -> torch.Tensor:
        """uint8 (B,3,H,W) -> final-LN features, token-major (passes*B*Np, E) == NHWC (passes*B, Hp, Wp, E)."""
        B = imgs_u8.shape[0]
        M = B * passes * self.Np
        E, Fd, w = self.E, self.w.ffn_dims, self.w
        ob = self.fmt
        self._call("im2col", "pp_preproc_im2col", self.prec, imgs_u8.data_ptr(), int(imgs_u8.dtype == torch.float32),
                  ws["patches"].data_ptr(), B, passes, self.H, self.W, self.P, self.pad, self.mean.ctypes.data,
                  self.std.ctypes.data, int(self.bgr_to_rgb), st)
        Kp = 3 * self.P * self.P
        scale = self.hd ** -0.5
        # residual GEMM + LayerNorm in one kernel (pp_gemm_ln.hip: E = 384, and 768 = ViT-B); other widths: GEMM then LN
        fused = self.fuse_resln and (E == 384 or (E == 768 and self._resln_768))
        L = w.num_layers

        def res_ln(a, wk, bk, K, gamma, beta, h_out, residual=None, res_mod=0, winv=1.0):
            """x <- residual + a @ wk^T + bk ; h_out <- LN(x). ``winv``: the weight tensor's inverse power-of-two scale."""
            residual = ws["x"] if residual is None else residual
            if fused:
                self._call("gemm_res_ln", "pp_gemm_residual_layernorm_ws", self.prec, a.data_ptr(), wk.data_ptr(),
                           bk.data_ptr(), residual.data_ptr(), res_mod, ws["x"].data_ptr(), gamma.data_ptr(),
                           beta.data_ptr(), self.ln_eps, h_out.data_ptr(), ob, M, E, K, K, K, float(winv), st)
            else:
                self._gemm(st, a, wk, bk, ws["x"], M, E, K, residual=residual, res_mod=res_mod, out_bf16=0, winv=winv)
                self._call("layernorm", "pp_layernorm", ws["x"].data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                           h_out.data_ptr(), M, E, self.ln_eps, ob, st)

        if self._small_at(M):
            return self._layers_small(ws, st, B * passes, M)
        res_ln(ws["patches"], w["patch_w"], w["patch_b"], Kp, w["l0.ln1.w"], w["l0.ln1.b"], ws["h"],
               residual=w["pos_embed"], res_mod=self.Np)
        if self._ln_fold_at(M):
            return self._layers_ln_folded(ws, st, B * passes, M)
        qkv_done = False  # the fused layer kernel has already produced this layer's qkv
        one_launch = (fused and E == 384 and self.precision == "bf16" and Fd % 128 == 0 and self.fuse_mlp and self.fuse_proj and self.fuse_attn
                      and self.Np == 192 and self.hd == 32)
        qcur, qnext = ws["qkv"], ws["qkv2"]
        att = ws["att"] if self.fuse_qkv_attn else ws["h"]  # where a layer's attention output goes
        for i in range(L):
            if not qkv_done and not self.fuse_qkv_attn:
                self._gemm(st, ws["h"], w[f"l{i}.qkv.w"], w[f"l{i}.qkv.b"], qcur, M, 3 * E, E, winv=w.inv(f"l{i}.qkv.w"))
            if self.stage_hook is not None:
                self.stage_hook("embed" if i == 0 else f"layer{i - 1}")
            if one_launch:
                # a ViT layer in ONE launch: attention, projection, ln2, FFN, next LayerNorm and the next layer's qkv
                last = i + 1 == L
                gn, bn = (w["ln_f.w"], w["ln_f.b"]) if last else (w[f"l{i + 1}.ln1.w"], w[f"l{i + 1}.ln1.b"])
                nq = None if (last or not self.fuse_qkv) else (w[f"l{i + 1}.qkv.w"], w[f"l{i + 1}.qkv.b"])
                h_next = ws["feat"] if last else ws["h"]
                self._call("vit_layer", "pp_vit_layer", qcur.data_ptr(), self.Np, self.heads, scale, w[f"l{i}.proj.w"].data_ptr(),
                           w[f"l{i}.proj.b"].data_ptr(), ws["x"].data_ptr(), w[f"l{i}.ln2.w"].data_ptr(),
                           w[f"l{i}.ln2.b"].data_ptr(), w[f"l{i}.fc1.w"].data_ptr(), w[f"l{i}.fc1.b"].data_ptr(),
                           w[f"l{i}.fc2.w"].data_ptr(), w[f"l{i}.fc2.b"].data_ptr(), ws["x"].data_ptr(), gn.data_ptr(),
                           bn.data_ptr(), self.ln_eps, None if nq else h_next.data_ptr(), nq[0].data_ptr() if nq else None,
                           nq[1].data_ptr() if nq else None, qnext.data_ptr() if nq else None, M, E, Fd, st)
                qkv_done = nq is not None
                qcur, qnext = qnext, qcur
                continue
            qkv_done = False
            fold = self.ln_fold_fused
            if self.fuse_qkv_attn and fold and i >= 1:
                # ws["h"] holds the CENTERED rows the previous layer left (operand format), ws["rowst"] their (mean, rstd)
                self._call("qkv_attention", "pp_qkv_attention_split_folded", ws["h"].data_ptr(), w[f"l{i}.qkv.wf"].data_ptr(), w[f"l{i}.qkv.bf"].data_ptr(),
                           ws["rowst"].data_ptr(), att.data_ptr(), B * passes, self.Np, self.heads, self.hd, scale, w.inv(f"l{i}.qkv.wf"), st)
            elif self.fuse_qkv_attn:
                self._call("qkv_attention", "pp_qkv_attention_split_ws", ws["h"].data_ptr(), w[f"l{i}.qkv.w"].data_ptr(),
                           w[f"l{i}.qkv.b"].data_ptr(), att.data_ptr(), B * passes, self.Np, self.heads, self.hd, scale, w.inv(f"l{i}.qkv.w"), st)
            else:
                self._call("attention", "pp_attention", self.prec, qcur.data_ptr(), ws["h"].data_ptr(), B * passes,
                           self.Np, self.heads, self.hd, scale, st)
            last = i + 1 == L
            gn, bn = (w["ln_f.w"], w["ln_f.b"]) if last else (w[f"l{i + 1}.ln1.w"], w[f"l{i + 1}.ln1.b"])
            h_next = ws["feat"] if last else ws["h"]
            fuse_ffn = fused and E == 384 and self.precision == "bf16" and Fd % 128 == 0 and self.fuse_mlp
            if fuse_ffn and self.fuse_proj:
                # second half of the layer in one kernel: projection + residual, ln2, FFN + residual, next LayerNorm;
                # the intermediate residual stream and ln2 output stay on the CU
                # ... and, except after the last layer, the next layer's qkv Linear on that LayerNorm output
                nq = None if (last or not self.fuse_qkv) else (w[f"l{i + 1}.qkv.w"], w[f"l{i + 1}.qkv.b"])
                self._call("proj_mlp_res_ln", "pp_proj_mlp_residual_layernorm", ws["h"].data_ptr(),
                           w[f"l{i}.proj.w"].data_ptr(), w[f"l{i}.proj.b"].data_ptr(), ws["x"].data_ptr(),
                           w[f"l{i}.ln2.w"].data_ptr(), w[f"l{i}.ln2.b"].data_ptr(), w[f"l{i}.fc1.w"].data_ptr(),
                           w[f"l{i}.fc1.b"].data_ptr(), w[f"l{i}.fc2.w"].data_ptr(), w[f"l{i}.fc2.b"].data_ptr(),
                           ws["x"].data_ptr(), gn.data_ptr(), bn.data_ptr(), self.ln_eps,
                           None if nq else h_next.data_ptr(), nq[0].data_ptr() if nq else None,
                           nq[1].data_ptr() if nq else None, qcur.data_ptr() if nq else None, M, E, Fd, st)
                qkv_done = nq is not None
                continue
            if i in self._proj_packed and self.fuse_qkv_attn and fold:
                # the same launch in the folded chain: residual rows fp32 (layer 0: the patch embedding's) or operand format; every layer but the last
                # leaves its rows once, in the operand format, in ws["h"] with their statistics; the last applies ln_f as before
                res, res_fmt = (ws["x"], 0) if i == 0 else (ws["h"], 2)
                self._call("proj_ffn_split", "pp_proj_ffn_split_folded", att.data_ptr(), self._proj_packed[i].data_ptr(), w[f"l{i}.proj.b"].data_ptr(),
                           w[f"l{i}.ln2.w"].data_ptr(), w[f"l{i}.ln2.b"].data_ptr(), ws["hs"].data_ptr(), self._ffn_packed[i].data_ptr(),
                           w[f"l{i}.fc1.b"].data_ptr(), w[f"l{i}.fc2.b"].data_ptr(), res.data_ptr(), res_fmt, None if i == 0 else ws["rowst"].data_ptr(),
                           0 if last else 1, ws["x"].data_ptr() if last else None, gn.data_ptr() if last else None, bn.data_ptr() if last else None,
                           self.ln_eps, h_next.data_ptr(), None if last else ws["rowst"].data_ptr(), M, E, Fd, w.inv(f"l{i}.proj.w"),
                           w.inv(f"l{i}.fc1.w"), w.inv(f"l{i}.fc2.w"), st)
                continue
            if i in self._proj_packed:
                # f16x3: projection + residual, ln2, FFN + residual, next LayerNorm in one kernel
                self._call("proj_ffn_split", "pp_proj_ffn_split_residual_layernorm_ws", att.data_ptr(), self._proj_packed[i].data_ptr(),
                           w[f"l{i}.proj.b"].data_ptr(), w[f"l{i}.ln2.w"].data_ptr(), w[f"l{i}.ln2.b"].data_ptr(), ws["hs"].data_ptr(),
                           self._ffn_packed[i].data_ptr(), w[f"l{i}.fc1.b"].data_ptr(), w[f"l{i}.fc2.b"].data_ptr(),
                           ws["x"].data_ptr(), ws["x"].data_ptr(), gn.data_ptr(), bn.data_ptr(), self.ln_eps, h_next.data_ptr(),
                           M, E, Fd, w.inv(f"l{i}.proj.w"), w.inv(f"l{i}.fc1.w"), w.inv(f"l{i}.fc2.w"), st)
                continue
            res_ln(att, w[f"l{i}.proj.w"], w[f"l{i}.proj.b"], E, w[f"l{i}.ln2.w"], w[f"l{i}.ln2.b"], ws["h"], winv=w.inv(f"l{i}.proj.w"))
            if i in self._ffn_packed:
                # f16x3: whole FFN + residual + next LayerNorm in one kernel, hidden activation on the CU
                self._call("ffn_split", "pp_ffn_split_residual_layernorm_ws", ws["h"].data_ptr(), self._ffn_packed[i].data_ptr(),
                           w[f"l{i}.fc1.b"].data_ptr(), w[f"l{i}.fc2.b"].data_ptr(), ws["x"].data_ptr(), ws["x"].data_ptr(),
                           gn.data_ptr(), bn.data_ptr(), self.ln_eps, h_next.data_ptr(), M, E, Fd, w.inv(f"l{i}.fc1.w"), w.inv(f"l{i}.fc2.w"), st)
            elif fuse_ffn:
                # whole FFN + residual + next LayerNorm in one kernel: the 4x-wide hidden activation stays on the CU
                self._call("mlp_res_ln", "pp_mlp_residual_layernorm", ws["h"].data_ptr(), w[f"l{i}.fc1.w"].data_ptr(),
                           w[f"l{i}.fc1.b"].data_ptr(), w[f"l{i}.fc2.w"].data_ptr(), w[f"l{i}.fc2.b"].data_ptr(),
                           ws["x"].data_ptr(), ws["x"].data_ptr(), gn.data_ptr(), bn.data_ptr(), self.ln_eps,
                           h_next.data_ptr(), M, E, Fd, st)
            else:
                self._gemm(st, ws["h"], w[f"l{i}.fc1.w"], w[f"l{i}.fc1.b"], ws["f"], M, Fd, E, act=ACT_GELU, winv=w.inv(f"l{i}.fc1.w"))
                res_ln(ws["f"], w[f"l{i}.fc2.w"], w[f"l{i}.fc2.b"], Fd, gn, bn, h_next, winv=w.inv(f"l{i}.fc2.w"))
        return ws["feat"]

    def _layers_small(self, ws, st, nb: int, M: int) -> torch.Tensor:
        """Patch embedding and encoder layers of a SMALL batch (pp_skinny_linear: every Linear layer column-parallel over the whole chip, the
        LayerNorm behind a residual layer done by the workgroup that completes a row block - no LayerNorm launch). Per layer: qkv + attention
        (one workgroup per sequence and head where the fused kernel applies, else qkv Linear + pp_attention), proj (+ residual, ln2), fc1 (GELU),
        fc2 (+ residual, next ln1 / ln_f): four launches. ws["patches"] holds the im2col rows on entry."""
        E, Fd, w, L = self.E, self.w.ffn_dims, self.w, self.w.num_layers
        F32, SPLIT = 0, 2
        scale = self.hd ** -0.5
        cnt = ws["ln_count"]

        def lin(a, wname, bias, out, N, K, act=ACT_NONE, residual=None, res_mod=0, out_fmt=SPLIT, ln=None):
            g, b, h_out = ln if ln is not None else (None, None, None)
            self._call("skinny_linear", "pp_skinny_linear", a.data_ptr(), w[wname].data_ptr(), _lib.ptr(bias), _lib.ptr(residual), res_mod,
                       out.data_ptr(), out_fmt, M, N, K, act, w.inv(wname), _lib.ptr(g), _lib.ptr(b), self.ln_eps, _lib.ptr(h_out),
                       cnt.data_ptr() if ln is not None else None, st)

        # patch embedding (+ pos_embed) -> x (fp32), ln1 of layer 0 -> h
        lin(ws["patches"], "patch_w", w["patch_b"], ws["x"], E, 3 * self.P * self.P, residual=w["pos_embed"], res_mod=self.Np, out_fmt=F32,
            ln=(w["l0.ln1.w"], w["l0.ln1.b"], ws["h"]))
        for i in range(L):
            if self.fuse_qkv_attn:
                self._call("qkv_attention", "pp_qkv_attention_split_ws", ws["h"].data_ptr(), w[f"l{i}.qkv.w"].data_ptr(), w[f"l{i}.qkv.b"].data_ptr(),
                           ws["att"].data_ptr(), nb, self.Np, self.heads, self.hd, scale, w.inv(f"l{i}.qkv.w"), st)
            else:
                lin(ws["h"], f"l{i}.qkv.w", w[f"l{i}.qkv.b"], ws["qkv"], 3 * E, E)
                self._call("attention", "pp_attention", self.prec, ws["qkv"].data_ptr(), ws["att"].data_ptr(), nb, self.Np, self.heads, self.hd, scale, st)
            if self.stage_hook is not None:
                self.stage_hook("embed" if i == 0 else f"layer{i - 1}")
            lin(ws["att"], f"l{i}.proj.w", w[f"l{i}.proj.b"], ws["x"], E, E, residual=ws["x"], out_fmt=F32,
                ln=(w[f"l{i}.ln2.w"], w[f"l{i}.ln2.b"], ws["hs"]))
            lin(ws["hs"], f"l{i}.fc1.w", w[f"l{i}.fc1.b"], ws["f"], Fd, E, act=ACT_GELU)
            last = i + 1 == L
            gn, bn = (w["ln_f.w"], w["ln_f.b"]) if last else (w[f"l{i + 1}.ln1.w"], w[f"l{i + 1}.ln1.b"])
            lin(ws["f"], f"l{i}.fc2.w", w[f"l{i}.fc2.b"], ws["x"], E, Fd, residual=ws["x"], out_fmt=F32, ln=(gn, bn, ws["feat"] if last else ws["h"]))
        return ws["feat"]

    def _layers_ln_folded(self, ws, st, nb: int, M: int) -> torch.Tensor:
        """The encoder layers with every inner LayerNorm folded into the Linear layer behind it (pp_linear_ln_folded): per layer qkv,
        attention, proj, fc1, fc2 - five launches, none of them a LayerNorm. ws["x"] / ws["h"] hold the patch-embed output and ln1 of
        layer 0 on entry (res_ln above); the residual stream then lives in ws["xs"] (operand format) until the last fc2 writes fp32 rows for
        the final LayerNorm."""
        E, Fd, w, L = self.E, self.w.ffn_dims, self.w, self.w.num_layers
        F32, SPLIT = 0, 2
        xs, stt, scale = ws["xs"], ws["lnst"], self.hd ** -0.5

        def lin(a, wname, bias, out, N, K, act=ACT_NONE, residual=None, res_fmt=F32, out_fmt=SPLIT, ln=None, stats_out=None):
            self._call("linear_fold", "pp_linear_ln_folded_ws", a.data_ptr(), w[wname].data_ptr(), bias.data_ptr(), _lib.ptr(residual), res_fmt,
                       out.data_ptr(), out_fmt, M, N, K, act, stt.data_ptr() if ln is not None else None, _lib.ptr(ln), self.ln_eps,
                       stt.data_ptr() if stats_out else None, w.inv(wname), st)

        for i in range(L):
            if i == 0:  # ln1 of layer 0 came out of the patch-embed launch: the plain weights
                lin(ws["h"], "l0.qkv.w", w["l0.qkv.b"], ws["qkv"], 3 * E, E)
            else:
                lin(xs, f"l{i}.qkv.wf", w[f"l{i}.qkv.bf"], ws["qkv"], 3 * E, E, ln=w[f"l{i}.qkv.cf"])
            if self.stage_hook is not None:
                self.stage_hook("embed" if i == 0 else f"layer{i - 1}")
            self._call("attention", "pp_attention", self.prec, ws["qkv"].data_ptr(), ws["h"].data_ptr(), nb, self.Np, self.heads, self.hd, scale, st)
            # x <- x + att Wp^T + bp (layer 0: the fp32 rows of the patch embedding), statistics for ln2
            lin(ws["h"], f"l{i}.proj.w", w[f"l{i}.proj.b"], xs, E, E, residual=ws["x"] if i == 0 else xs, res_fmt=F32 if i == 0 else SPLIT,
                stats_out=True)
            lin(xs, f"l{i}.fc1.wf", w[f"l{i}.fc1.bf"], ws["f"], Fd, E, act=ACT_GELU, ln=w[f"l{i}.fc1.cf"])
            last = i + 1 == L
            # x <- x + f W2^T + b2: statistics for the next layer's ln1, or fp32 rows for the final LayerNorm
            lin(ws["f"], f"l{i}.fc2.w", w[f"l{i}.fc2.b"], ws["x"] if last else xs, E, Fd, residual=xs, res_fmt=SPLIT,
                out_fmt=F32 if last else SPLIT, stats_out=not last)
        self._call("layernorm", "pp_layernorm", ws["x"].data_ptr(), w["ln_f.w"].data_ptr(), w["ln_f.b"].data_ptr(), ws["feat"].data_ptr(), M, E,
                   self.ln_eps, self.fmt, st)
        return ws["feat"]

    def heatmap_logits(self, feat: torch.Tensor, nb: int, ws, st) -> torch.Tensor:
        """NHWC features -> planar logits (nb, K, Hh*Wh) fp32 (deconv x n + BN + ReLU, final 1x1 conv)."""
        w = self.w
        ob = self.fmt
        src, cin, hh, ww = feat, self.E, self.Hp, self.Wp
        self._logits_phased = False
        nd = len(w.deconv_channels)
        for j, cout in enumerate(w.deconv_channels):
            dst = ws[f"d{j}"]
            wj = w[f"deconv{j}.w"]
            if (j == nd - 1 and self.fuse_head and ob == 1 and cout == 256 and w.has("final.w_pad") and cin % 32 == 0
                    and ww % 4 == 0 and self.K <= 28):
                # last deconvolution + the 1x1 conv behind it in one kernel; the 256-channel map is never stored and the
                # logits come out phase-separated (the decode kernel reads that layout directly)
                self._call("deconv_head", "pp_deconv_head", src.data_ptr(), wj.data_ptr(), w[f"deconv{j}.b"].data_ptr(),
                           w["final.w_pad"].data_ptr(), w["final.b"].data_ptr(), ws["logits"].data_ptr(), nb, hh, ww, cin, cout,
                           self.K, st)
                self._logits_phased = True
                return ws["logits"]
            if (j == nd - 1 and self.fuse_head and ob == 2 and cout == 256 and w.has("final.w_head") and cin % 32 == 0
                    and ww % 4 == 0 and self.K <= 28 and nb * hh * ww * 4 >= 192 * 192):
                # f16x3: the same fusion, the 1x1 weights as a register image (weights.pack_head_split)
                self._call("deconv_head", "pp_deconv_head_split", src.data_ptr(), wj.data_ptr(), w[f"deconv{j}.b"].data_ptr(),
                           w["final.w_head"].data_ptr(), w["final.b"].data_ptr(), ws["logits"].data_ptr(), nb, hh, ww, cin, cout,
                           self.K, st)
                self._logits_phased = True
                return ws["logits"]
            if ob == 2 and self._small_at(nb * self.Np) and nb * hh * ww <= 1536 and cin % 32 == 0 and cout % 32 == 0:
                # a FEW input pixels (<= 1 536: both deconvolutions of one crop, the first one up to four crops with flip test): the four phases as
                # column-parallel GEMMs on pp_skinny_linear's tiles - 19.6 / 37.1 us against 61.2 / 44.8 for the 128 x 128 kernel's 24 / 96
                # workgroups at B = 1; from 3 072 pixels on the 128 x 128 kernel wins (scripts/r06/skinny_deconv_sweep.py)
                self._call("deconv", "pp_skinny_deconv", src.data_ptr(), wj.data_ptr(), w[f"deconv{j}.b"].data_ptr(), dst.data_ptr(), nb, hh, ww, cin,
                           cout, st)
                src, cin, hh, ww = dst, cout, hh * 2, ww * 2
                continue
            # all four output phases of the transposed conv in one persistent launch
            self._call("deconv", "pp_conv_gemm", self.prec, DECONV, src.data_ptr(), wj.data_ptr(),
                       w[f"deconv{j}.b"].data_ptr(), dst.data_ptr(), nb, hh, ww, cin, cout, -1, -1, 1, 0, 0, 0, 0, cout,
                       ACT_RELU, ob, st)
            src, cin, hh, ww = dst, cout, hh * 2, ww * 2
        P = hh * ww
        if ob == 2 and self._small_at(nb * self.Np) and cin % 64 == 0 and self._final_padded is not None:
            # small batches: 32 x 32 tiles over the pixels (192 workgroups for one crop + flip) instead of 128 x 128 ones (48)
            wp, bp = self._final_padded
            self._call("final_conv", "pp_skinny_conv1x1_planar", src.data_ptr(), wp.data_ptr(), bp.data_ptr(), ws["logits"].data_ptr(), nb, P, cin, self.K,
                       w.inv("final.w"), st)
            return ws["logits"]
        self._gemm(st, src, w["final.w"], w["final.b"], ws["logits"], nb * P, self.K, cin, planar=P, out_bf16=0)
        return ws["logits"]

    def towers(self, feat: torch.Tensor, B: int, passes: int, flip_indices, ws, st) -> torch.Tensor:
        """NHWC features -> (4, B, K) fp32: probability, visibility, oks, error (error NOT yet / diagonal)."""
        w, E = self.w, self.E
        nb = B * passes
        ob = self.fmt
        src, stride_src = feat, 0  # the four towers share the backbone features
        for j, (th, tw) in enumerate(self.tower_hw):
            ph, pw_ = self.pools[j]
            out = ws[f"t{j}"]
            rows = nb * th * tw
            if self.split_k and f"tp{j}" in ws:
                # few output rows: one workgroup per 128 x 128 tile would leave most CUs idle for a 54-step K loop;
                # cut K in three (one kernel row of taps each), reduce + bias in the pooling kernel
                part = ws[f"tp{j}"]
                ks = part.shape[0]
                self._call("conv3x3_splitk", "pp_conv3x3_splitk", self.prec, src.data_ptr(), w[f"tower{j}.w"].data_ptr(),
                           part.data_ptr(), nb, th, tw, E, E, 4, stride_src, E * 9 * E, ks, st)
                self._call("maxpool", "pp_sum_maxpool_relu_nhwc", part.data_ptr(), ks, 4 * rows * E, w[f"tower{j}.b"].data_ptr(),
                           nb, ws[f"p{j}"].data_ptr(), ob, 4 * nb, th, tw, E, ph, pw_, st)
                src = ws[f"p{j}"]
                stride_src = nb * (th // ph) * (tw // pw_) * E
                continue
            if j == 0 and self.winograd and ws.get("wino") is not None and stride_src == 0 and (ph, pw_) == (4, 3):
                # first stage, Winograd F(2x2, 3x3): input transform + 16 position GEMMs with output transform, pooling, bias, ReLU
                self._call("conv3x3", "pp_conv3x3_winograd_maxpool_relu", src.data_ptr(), w["tower0.wino"].data_ptr(), w["tower0.b"].data_ptr(),
                           ws["wino"].data_ptr(), ws["p0"].data_ptr(), nb, th, tw, E, E, ph, pw_, 4, st)
            elif self.fuse_pool:
                # conv + BN -> MaxPool -> ReLU in one launch where the halo-staged kernel holds whole images per tile (bf16,
                # 16 x 12 maps); the C side takes the two-launch route through `out` for every other shape
                self._call("conv3x3", "pp_conv3x3_maxpool_relu", self.prec, src.data_ptr(), w[f"tower{j}.w"].data_ptr(),
                           w[f"tower{j}.b"].data_ptr(), ws[f"p{j}"].data_ptr(), out.data_ptr(), nb, th, tw, E, E, ph, pw_, 4,
                           stride_src, E * 9 * E, E, ob, st)
            else:
                self._call("conv3x3", "pp_conv_gemm", self.prec, CONV3X3, src.data_ptr(), w[f"tower{j}.w"].data_ptr(),
                           w[f"tower{j}.b"].data_ptr(), out.data_ptr(), nb, th, tw, E, E, 0, 0, 4, stride_src,
                           E * 9 * E, nb * th * tw * E, E, E, ACT_NONE, ob, st)
                self._call("maxpool", "pp_maxpool_relu_nhwc", out.data_ptr(), ob, ws[f"p{j}"].data_ptr(), ob, 4 * nb, th, tw, E, ph,
                           pw_, st)
            src = ws[f"p{j}"]
            stride_src = nb * (th // ph) * (tw // pw_) * E
        fi = self._flip_indices(flip_indices) if passes == 2 else None
        self._call("tower_final", "pp_tower_final", src.data_ptr(), ob, w["tower_out.w"].data_ptr(), w["tower_out.b"].data_ptr(),
                  _lib.ptr(fi), ws["scalars"].data_ptr(), B, passes, E, self.K, 1.0, st)
        return ws["scalars"]

    # ------------------------------------------------------------------ public
    def _check_imgs(self, imgs: torch.Tensor):
        assert imgs.dim() == 4 and imgs.is_cuda, "expects a (B,3,H,W) tensor on the GPU"
        assert imgs.dtype in (torch.uint8, torch.float32), "crops are uint8 (raw) or float32 (already preprocessed)"
        assert tuple(imgs.shape[1:]) == (3, self.H, self.W), f"crop shape {tuple(imgs.shape)} != (B,3,{self.H},{self.W})"

    @torch.no_grad()
    def run_backbone(self, imgs: torch.Tensor, flip_test: bool, slot: int = 0) -> torch.Tensor:
        """(B,3,H,W) uint8 raw crops (or fp32 preprocessed) -> NHWC features (passes*B, Hp, Wp, E); with
        flip_test rows [B:] are the features of the horizontally flipped crops (topdown.py:109-112)."""
        self._check_imgs(imgs)
        imgs = imgs.contiguous()
        B, passes = imgs.shape[0], 2 if flip_test else 1
        ws = self._workspace(B, passes, slot)
        with torch.cuda.device(self.device):
            feat = self.backbone(imgs, passes, ws, _lib.stream_ptr(self.device))
        return feat.view(B * passes, self.Hp, self.Wp, self.E)

    # features cross the module-level interfaces (backbone(x) -> head.forward(feats)) as ordinary tensors: bf16 / fp32
    # as they are, split fp16 decoded to fp32 on the way out and re-encoded on the way in (host-side plumbing, not on the
    # fused predict path)
    def export_features(self, feat: torch.Tensor) -> torch.Tensor:
        """Engine feature buffer -> the caller's own tensor (a copy: the workspace is reused by the next call)."""
        return from_split(feat) if self.precision == "f16x3" else feat.clone()

    def import_features(self, x_nhwc: torch.Tensor) -> torch.Tensor:
        """NHWC feature tensor of any float dtype -> contiguous buffer in the engine's operand format."""
        if self.precision == "f16x3":
            return to_split(x_nhwc.float().contiguous())
        if x_nhwc.dtype != self.dtype or not x_nhwc.is_contiguous():
            x_nhwc = x_nhwc.to(self.dtype).contiguous()
        return x_nhwc

    @torch.no_grad()
    def run_head(self, feat_nhwc: torch.Tensor, flip_test: bool, flip_indices=None,
                 return_heatmaps: bool = False, slot: int = 0, shift_heatmap: bool = False) -> Dict[str, torch.Tensor]:
        """NHWC features (passes*B, Hp, Wp, E) in the engine's operand dtype -> decoded results
        (ProbMapHead.forward + the flip-test merge + BaseHead.decode, probmap_head.py:746-779)."""
        passes = 2 if flip_test else 1
        nb = feat_nhwc.shape[0]
        assert nb % passes == 0 and tuple(feat_nhwc.shape[1:]) == (self.Hp, self.Wp, self.E)
        assert feat_nhwc.dtype == self.dtype and feat_nhwc.is_contiguous() and feat_nhwc.is_cuda
        if flip_test and flip_indices is None:
            raise ValueError("flip_test needs flip_indices (dataset meta)")
        B = nb // passes
        ws = self._workspace(B, passes, slot)
        # Small batches: the heatmap branch (two deconvolutions, 1x1 conv, decode) and the four scalar towers read the same features and nothing of
        # each other - none of their launches fills the chip at these sizes, so the towers run on a second stream beside the heatmap branch
        # (fork / join by events: inside a capture they become two branches of the graph). B = 1: ~0.12 ms of the step.
        two = self.plan["head_two_streams"] and self._small_at(nb * self.Np)
        with torch.cuda.device(self.device):
            cur = torch.cuda.current_stream(self.device)
            scalars = None
            if two:
                if self._head_stream is None:
                    self._head_stream = torch.cuda.Stream(device=self.device)
                self._fork_join(cur, self._head_stream)
                with torch.cuda.stream(self._head_stream):
                    scalars = self.towers(feat_nhwc, B, passes, flip_indices, ws, _lib.stream_ptr(self.device))
            st = _lib.stream_ptr(self.device)
            logits = self.heatmap_logits(feat_nhwc, nb, ws, st)
            fi = self._flip_indices(flip_indices) if flip_test else None
            lf = logits[B:] if flip_test else None
            flags = 1 | (2 if self._logits_phased else 0) | (4 if (shift_heatmap and flip_test) else 0)  # PP_DECODE_LOGITS | _PHASED | _SHIFT_HEATMAP
            self._call("head_decode", "pp_probmap_decode_flags", logits.data_ptr(), _lib.ptr(lf), _lib.ptr(fi), self.taps.data_ptr(),
                      self.radius.data_ptr(), B, self.K, self.Hh, self.Wh, float(self.input_size[0]),
                      float(self.input_size[1]), self.temperature, -1.0 if self.normalize is None else float(self.normalize),  # (< 0: no Sparsemax)
                      ws["heatmaps"].data_ptr() if return_heatmaps else None, None, ws["locs"].data_ptr(),
                      ws["keypoints"].data_ptr(), ws["scores"].data_ptr(), flags, st)
            if two:
                self._fork_join(self._head_stream, cur)
            else:
                scalars = self.towers(feat_nhwc, B, passes, flip_indices, ws, st)
        out = dict(keypoints=ws["keypoints"], scores=ws["scores"], locs=ws["locs"], scalars=scalars)
        if return_heatmaps:
            out["heatmaps"] = ws["heatmaps"]
        return out

    @torch.no_grad()
    def forward(self, imgs: torch.Tensor, flip_test: bool = True, flip_indices=None, return_heatmaps: bool = False,
                return_features: bool = False, slot: int = 0, shift_heatmap: bool = False) -> Dict[str, torch.Tensor]:
        """imgs: (B, 3, H, W) uint8 on the device (BGR CHW as PackPoseInputs emits). Returns device
        tensors (views into the cached workspace, valid until the next call with the same batch size):
        ``keypoints`` (B,K,2) f64 input-pixel space, ``scores`` (B,K) f32 (keypoints_conf), ``locs``,
        ``scalars`` (4,B,K) f32 [probability, visibility, oks, raw error], optionally ``heatmaps``."""
        self._tick += 1
        feat = self.run_backbone(imgs, flip_test, slot)
        out = self.run_head(feat, flip_test, flip_indices, return_heatmaps, slot, shift_heatmap)
        if return_features:
            out["features"] = feat
        return out

    def _fork_join(self, producer: torch.cuda.Stream, consumer: torch.cuda.Stream) -> None:
        """``consumer.wait_stream(producer)`` with an event that OUTLIVES the call. `Stream.wait_stream` records a temporary event and destroys it on
        return; inside a stream capture the HIP runtime (ROCm 7.0) keeps referring to that event from the captured graph - after the event's memory
        had been reused by later host allocations, replaying a graph with the two-stream head crashed inside hipGraphLaunch
        (scripts/r06/graph_eager_repro2.py: `test_step` graphs + short-lived StepPipeline objects). Events recorded during a capture are kept with
        the graph being captured (`capture`), the others are two per engine, re-recorded."""
        if torch.cuda.is_current_stream_capturing():
            ev = torch.cuda.Event()
            self._capture_events.append(ev)
        else:
            key = (producer.cuda_stream, consumer.cuda_stream)
            ev = self._eager_events.get(key)
            if ev is None:
                ev = self._eager_events[key] = torch.cuda.Event()
        ev.record(producer)
        consumer.wait_event(ev)

    # ------------------------------------------------------------------ hipGraph replay
    def capture(self, B: int, flip_test: bool = True, flip_indices=None, return_heatmaps: bool = False, slot: int = 0,
                shift_heatmap: bool = False):
        """Capture the whole launch sequence for batch size B into a HIP graph (the ~110 launches of one
        forward are launch-latency-bound from Python). Returns the static input buffer to fill. Every ``slot`` has its
        own workspace, input buffer and graph."""
        self._check_options()
        key = (B, flip_test, tuple(flip_indices) if flip_indices is not None else None, return_heatmaps, bool(shift_heatmap), slot)
        if key in self._graphs:
            self._graphs[key] = self._graphs.pop(key)  # most recently used last
            return self._graphs[key].static_in
        while len(self._graphs) >= max(1, int(self.max_graphs)):  # least recently used out, with the workspace nothing else replays from
            old_key = next(iter(self._graphs))
            # the victim may still be replaying on another slot's stream (StepPipeline depth >= 2): its exec graph and the buffers it reads go
            # back to the allocator below, so the device must be through with it first (an eviction is rare; the capture syncs anyway)
            torch.cuda.synchronize(self.device)
            self._graphs.pop(old_key).release()
            self._graph_tick.pop(old_key, None)
            passes_old = 2 if old_key[1] else 1
            if not any(k[0] == old_key[0] and k[-1] == old_key[-1] and (2 if k[1] else 1) == passes_old for k in self._graphs):
                self._ws.pop((old_key[0], passes_old, old_key[-1]), None)
        static_in = torch.zeros((B, 3, self.H, self.W), dtype=torch.uint8, device=self.device)
        # The two-stream head forks to a side stream INSIDE the capture: a HIP stream of this graph's own (_CapturedGraph), never one an earlier
        # capture used; the kernel-by-kernel launches keep theirs.
        raw = ctypes.c_void_p()
        _lib.check("pp_stream_create", _lib.lib.pp_stream_create(ctypes.byref(raw)))
        eager_head_stream, self._head_stream = self._head_stream, torch.cuda.ExternalStream(raw.value, device=self.device)
        side = torch.cuda.Stream(device=self.device)
        try:
            side.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(side):  # warm-up: allocates the workspace, sets kernel attributes
                for _ in range(2):
                    self.forward(static_in, flip_test, flip_indices, return_heatmaps, slot=slot, shift_heatmap=shift_heatmap)
            torch.cuda.current_stream(self.device).wait_stream(side)
            side.synchronize()  # (this stream's warm-ups only; torch.cuda.graph below still synchronises the device when the capture begins)
            graph = torch.cuda.CUDAGraph()
            self._capture_events = []
            with torch.cuda.graph(graph):
                out = self.forward(static_in, flip_test, flip_indices, return_heatmaps, slot=slot, shift_heatmap=shift_heatmap)
            self._graphs[key] = _CapturedGraph(graph, static_in, out, self._capture_events, self._head_stream, raw.value)
            raw = None
        finally:
            self._capture_events = []
            self._head_stream = eager_head_stream
            if raw is not None and raw.value:  # the capture failed: nothing refers to the stream
                torch.cuda.synchronize(self.device)
                _lib.lib.pp_stream_destroy(raw)
        self._graph_tick[key] = self._tick
        self.graph_captures += 1
        return static_in

    @staticmethod
    def _graph_key(B, flip_test, flip_indices, return_heatmaps, shift_heatmap, slot):
        return (B, flip_test, tuple(flip_indices) if flip_indices is not None else None, return_heatmaps, bool(shift_heatmap), slot)

    def has_graph(self, B: int, flip_test: bool = True, flip_indices=None, return_heatmaps: bool = False, slot: int = 0,
                  shift_heatmap: bool = False) -> bool:
        self._check_options()
        return self._graph_key(B, flip_test, flip_indices, return_heatmaps, shift_heatmap, slot) in self._graphs

    def capture_would_thrash(self) -> bool:
        """True when a new capture would evict a graph that replayed within the last ``64 * max_graphs`` calls: with more recurring batch sizes
        than graphs kept (the person counts of a video: 0 .. 20 per frame) every call would otherwise evict, re-allocate a workspace, warm up,
        capture, synchronise the device and replay - several times the cost of launching kernel by kernel, for ever (at 4 * max_graphs a random
        mix of 15 sizes still captured on 8 % of its calls: scripts/r06/chaos_soak.py). The caller then runs this size eagerly; a graph nobody
        replays any more ages out and makes room."""
        if len(self._graphs) < max(1, int(self.max_graphs)):
            return False
        victim = next(iter(self._graphs))
        return self._tick - self._graph_tick.get(victim, 0) <= 64 * max(1, int(self.max_graphs))

    def forward_graph(self, imgs: torch.Tensor, flip_test: bool = True, flip_indices=None,
                      return_heatmaps: bool = False, slot: int = 0, shift_heatmap: bool = False) -> Dict[str, torch.Tensor]:
        """Same contract as ``forward`` for uint8 crops, replaying the captured graph (on torch's current stream)."""
        B = imgs.shape[0]
        static_in = self.capture(B, flip_test, flip_indices, return_heatmaps, slot, shift_heatmap)
        key = self._graph_key(B, flip_test, flip_indices, return_heatmaps, shift_heatmap, slot)
        graph, out = self._graphs[key].graph, self._graphs[key].out
        self._tick += 1
        self._graph_tick[key] = self._tick
        if imgs.data_ptr() != static_in.data_ptr():
            static_in.copy_(imgs, non_blocking=True)
        graph.replay()
        return out


# numeric domain of the split-fp16 operand format (include/probpose_mi355x.h, "numeric domain"; csrc/pp_split.h)
F16X3_MAX_OPERAND = 65504.0     # |x| beyond it: the high half is inf, products NaN -> NaN keypoints (pp_probmap_decode_flags), FloatingPointError in predict
F16X3_FULL_PRECISION_MIN = 0.125  # below it the low half is an fp16 subnormal: absolute 2^-25 instead of relative 2^-22
F16X3_FOLD_MEAN_OVER_STD = 30.0  # LayerNorm fold: rstd (acc - mean colsum) cancels ~ log2(|mean| / std) of the 22 operand bits (1e-4 relative at 30)


@torch.no_grad()
def domain_report(state_dict: Dict[str, torch.Tensor], crops_u8: torch.Tensor, num_heads: int, img_size=(256, 192), device="cuda",
                  **engine_kw) -> Dict[str, object]:
    """Does a checkpoint stay inside the numeric domain of ``precision="f16x3"`` on these crops? Runs the network once in the fp32 mode
    (fp32 operands, the generic launch plan: every activation passes through HBM) and reads, layer by layer, what the f16x3 kernels would
    be handed as MFMA operands: the residual rows (raw rows are operands since the LayerNorm fold), the LayerNorm output, qkv, the FFN's hidden
    activation, the features. Returns ``{"layers": [...], "max_operand": .., "max_mean_over_std": .., "ok": bool, "advice": str}``.
    A diagnostic for a new checkpoint (run once), not part of the hot path."""
    eng = ProbPoseEngine(state_dict, num_heads, img_size=img_size, precision="f32", device=device, **engine_kw)
    rows: list = []
    crops_u8 = crops_u8.to(eng.device)
    B = crops_u8.shape[0]
    ws = eng._workspace(B, 1, 0)

    def stat(name, t):
        t = t.float()
        return {name + "_absmax": float(t.abs().max())}

    def hook(stage):
        torch.cuda.synchronize(eng.device)
        x = ws["x"].float()
        sd_ = x.std(dim=1, unbiased=False).clamp_min(1e-12)
        r = dict(stage=stage, residual_absmax=float(x.abs().max()), residual_mean_over_std=float((x.mean(dim=1).abs() / sd_).max()))
        r.update(stat("ln_out", ws["h"]))
        r.update(stat("qkv", ws["qkv"]))
        if ws.get("f") is not None and stage != "embed":
            r.update(stat("ffn_hidden", ws["f"]))
        rows.append(r)

    eng.stage_hook = hook
    out = eng.forward(crops_u8, False, None)
    torch.cuda.synchronize(eng.device)
    eng.stage_hook = None
    feat_abs = float(ws["feat"].float().abs().max())
    worst = max(max(v for k, v in r.items() if k.endswith("_absmax")) for r in rows)
    worst = max(worst, feat_abs)
    ratio = max(r["residual_mean_over_std"] for r in rows[1:]) if len(rows) > 1 else 0.0  # (layer 0's ln1 is never folded)
    finite = bool(torch.isfinite(out["keypoints"]).all())
    ok = finite and worst <= F16X3_MAX_OPERAND / 2 and ratio <= F16X3_FOLD_MEAN_OVER_STD
    advice = "inside the f16x3 domain"
    if not finite or worst > F16X3_MAX_OPERAND / 2:
        advice = (f"an operand reaches {worst:.3g} (limit 65504, a factor 2 kept in hand): run precision='f32', or rescale the layer that "
                  "produces it")
    elif ratio > F16X3_FOLD_MEAN_OVER_STD:
        advice = (f"token rows reach |mean| / std = {ratio:.1f}: the folded LayerNorm loses ~log2 of that in bits - build the model with "
                  "plan=dict(ln_fold=False) (LayerNorm applied before the split, no cancellation)")
    return dict(layers=rows, features_absmax=feat_abs, max_operand=worst, max_mean_over_std=ratio, ok=ok, advice=advice)
