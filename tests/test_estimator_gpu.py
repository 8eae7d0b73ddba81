"""GPU: the drop-in surface end to end -- config -> registry build -> load_state_dict (reference key names)
-> test_step -> PoseDataSample.pred_instances -- against the torch-CPU oracle on identical crops.

Tolerances (BASELINE.json north_star: keypoints / probabilities within 1e-3 of the reference CPU path):
  * precision "f16x3" (split-fp16 operands, three fp16 MFMAs per product) and "f32" (exact-fp32 MFMA products): every
    field <= 1e-3, keypoints in image px, no argmax flips - also at the bench's batch size 64;
  * precision "bf16": measured (0.3-0.45 px, 3-8 % flips on these very sparse synthetic maps) and bounded just above
    that; argmax flips are counted, not hidden;
  * the hipGraph replay (what bench.py times) must equal the eager launch sequence bit for bit.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = os.path.join(ROOT, "configs", "td-pm_ProbPose-small_mi355x_coco-256x192.py")
FIELDS = ["keypoints_conf", "keypoints_probs", "keypoints_visible", "keypoints_oks", "keypoints_error", "keypoint_scores"]
B = 6


@pytest.fixture(scope="module")
def setup():
    from oracle import model_ref as M
    from probpose_code_amd import synthetic as S

    torch.set_num_threads(min(16, os.cpu_count()))
    sd = S.synthetic_state_dict("small", seed=3, logit_scale=2.0)
    crops = S.synthetic_crops(B, seed=4)
    rng = np.random.default_rng(5)
    center = np.stack([rng.uniform(80, 400, B), rng.uniform(100, 500, B)], -1).astype(np.float32)
    scale = (np.array([192, 256], np.float32) * rng.uniform(0.8, 2.5, (B, 1)).astype(np.float32) * 1.25).astype(np.float32)
    ref = M.predict(sd, crops, 12, S.IMG_MEAN, S.IMG_STD, input_size=(192, 256), input_center=center, input_scale=scale)
    return sd, crops, center, scale, ref


def _run(sd, crops, center, scale, precision, cfg_options=None):
    from probpose_code_amd import apis

    opts = {"model.precision": precision}
    opts.update(cfg_options or {})
    model = apis.init_model(CFG, {"state_dict": sd}, device="cuda:0", cfg_options=opts)
    batch = apis.pack_crops(crops, center, scale, model.dataset_meta)
    with torch.no_grad():
        return model, model.test_step(batch)


@pytest.mark.parametrize("precision", ["f16x3", "f32"])
def test_parity_modes_pred_instances_within_1e3(setup, precision):
    sd, crops, center, scale, ref = setup
    model, results = _run(sd, crops, center, scale, precision, {"model.test_cfg.output_heatmaps": True})
    assert len(results) == B
    flips = 0
    for b, ds in enumerate(results):
        pi = ds.pred_instances
        assert pi.keypoints.shape == (1, 17, 2) and pi.keypoints.dtype == np.float64
        for f in FIELDS:
            assert getattr(pi, f).shape == (1, 17), f
        d = np.abs(pi.keypoints - ref["keypoints"][b]).max(-1)[0]
        same = d < 0.5 * float(scale[b].min()) / 48  # less than half a heatmap cell: same argmax
        flips += int((~same).sum())
        assert d[same].max() <= 1e-3, f"sample {b}: keypoint L_inf {d[same].max():.2e} image px"
        for f in FIELDS[1:]:
            assert np.abs(getattr(pi, f) - ref[f][b]).max() <= 1e-3, f
        assert np.abs(pi.keypoints_conf - ref["keypoints_conf"][b])[0][same].max() <= 1e-3
        assert np.array_equal(pi.bboxes, ds.gt_instances.bboxes) and np.array_equal(pi.bbox_scores, ds.gt_instances.bbox_scores)
        hm = ds.pred_fields.heatmaps
        assert tuple(hm.shape) == (17, 64, 48)
        assert np.abs(hm.cpu().numpy() - ref["heatmaps"][b]).max() <= 1e-3
    assert flips == 0, f"{flips} argmax flips of {B * 17} keypoints in {precision} mode"
    # keypoint_scores is the OKS branch since freeze_oks=False (probmap_head.py:797-798)
    assert np.array_equal(results[0].pred_instances.keypoint_scores, results[0].pred_instances.keypoints_oks)


def test_bf16_pred_instances_bounded(setup):
    sd, crops, center, scale, ref = setup
    _, results = _run(sd, crops, center, scale, "bf16")
    flips, worst = 0, 0.0
    for b, ds in enumerate(results):
        pi = ds.pred_instances
        d = np.abs(pi.keypoints - ref["keypoints"][b]).max(-1)[0]
        same = d < 0.5 * float(scale[b].min()) / 48
        flips += int((~same).sum())
        worst = max(worst, float(d[same].max()))
        for f in ("keypoints_probs", "keypoints_visible", "keypoints_oks"):
            assert np.abs(getattr(pi, f) - ref[f][b]).max() <= 3e-2, f
    print(f"bf16: keypoint L_inf (same argmax) {worst:.3e} image px, argmax flips {flips}/{B * 17}")
    # bf16 is the throughput mode, not the parity mode (that is f16x3, 1e-3). Measured here: 0.3-0.5 image px on agreeing
    # argmaxes, 3-8 % flips (the synthetic maps have 2-10 px support, so near-ties are common); the bound sits just
    # above the measured band so that a regression in any bf16 kernel shows
    assert worst <= 0.75 and flips <= 0.10 * B * 17


@pytest.mark.parametrize("precision", ["f16x3", "f32"])
def test_full_path_bs64_within_1e3(precision):
    """The bench workload (bs 64, flip test) end to end in the parity modes against the oracle: keypoints <= 1e-3
    input-space px with zero argmax flips, scalar heads <= 1e-3."""
    from oracle import model_ref as M
    from probpose_code_amd import ProbPoseEngine
    from probpose_code_amd import synthetic as S

    torch.set_num_threads(min(16, os.cpu_count()))
    sd = S.synthetic_state_dict("small", seed=0, logit_scale=2.0)
    crops = S.synthetic_crops(64, seed=100)
    ref = M.predict(sd, crops, 12, S.IMG_MEAN, S.IMG_STD)
    eng = ProbPoseEngine(sd, 12, precision=precision)
    out = eng.forward_graph(crops.cuda(), True, S.COCO_FLIP_INDICES)  # the replayed path, as bench.py runs it
    torch.cuda.synchronize()
    d = np.abs(out["keypoints"].cpu().numpy()[:, None] - ref["keypoints_input_space"]).max(-1)
    assert (d < 2.0).all(), f"{int((d >= 2.0).sum())} argmax flips of {d.size}"
    assert d.max() <= 1e-3, f"keypoint L_inf {d.max():.2e} px"
    for i, name in enumerate(("keypoints_probs", "keypoints_visible", "keypoints_oks")):
        assert np.abs(out["scalars"][i].cpu().numpy()[:, None] - ref[name]).max() <= 1e-3, name
    assert np.abs(out["scores"].cpu().numpy()[:, None] - ref["keypoints_conf"]).max() <= 1e-3


def _fields_of(results):
    out = {}
    for f in FIELDS + ["keypoints", "bboxes", "bbox_scores"]:
        out[f] = np.stack([np.asarray(getattr(ds.pred_instances, f)) for ds in results])
    return out


def test_test_step_graph_replay_equals_eager_bit_for_bit(setup):
    """The drop-in call itself (`model.test_step`, mmpose/apis/inference.py:195-196) on the fast path: the first two batches of a size
    are launched kernel by kernel, the third captures the hipGraph (`graph_capture_after` = 3), later ones replay it; the results come back through one
    record copy. Every `pred_instances` field must equal the kernel-by-kernel estimator's (graph_replay=False) bit for bit, on
    DIFFERENT batches in a row, and `test_step_stream` (graph for full batches, two in flight) must deliver the same."""
    from probpose_code_amd import apis
    from probpose_code_amd import synthetic as S

    sd, _, center, scale, _ = setup
    fast = apis.init_model(CFG, {"state_dict": sd}, device="cuda:0")
    slow = apis.init_model(CFG, {"state_dict": sd}, device="cuda:0", cfg_options={"model.graph_replay": False})
    assert fast.graph_replay and not slow.graph_replay
    batches = [S.synthetic_crops(B, seed=40 + i) for i in range(5)]
    got = []
    with torch.no_grad():
        for i, crops in enumerate(batches):
            a = _fields_of(fast.test_step(apis.pack_crops(crops, center, scale, fast.dataset_meta)))
            b = _fields_of(slow.test_step(apis.pack_crops(crops, center, scale, slow.dataset_meta)))
            for f in a:
                assert a[f].dtype == b[f].dtype and np.array_equal(a[f], b[f]), f"{f}: batch {i} differs between replay and eager test_step"
            got.append(a)
        key = (B, True, tuple(S.COCO_FLIP_INDICES), False)
        assert fast._sizes_seen[key] == 5 and any(k[0] == B for k in fast.engine._graphs), "test_step did not reach the graph path"
        assert not slow.engine._graphs
        assert not np.array_equal(got[3]["keypoints"], got[4]["keypoints"]), "a replay repeated the previous batch"
        stream = list(fast.test_step_stream((apis.pack_crops(c, center, scale, fast.dataset_meta) for c in batches), depth=2, max_batch=B))
        assert any(k[0] == B and k[-1] == 1 for k in fast.engine._graphs), "test_step_stream did not capture its second slot's graph"
        for i, res in enumerate(stream):
            c = _fields_of(res)
            for f in c:
                assert c[f].dtype == got[i][f].dtype and np.array_equal(c[f], got[i][f]), f"{f}: streamed batch {i} differs from test_step"


def test_shift_heatmap_test_cfg_matches_the_oracle(setup):
    """``test_cfg.shift_heatmap=True`` (flip_heatmaps(..., shift_heatmap=True), tta.py:64-66) through the drop-in call: keypoints
    against the oracle with the shifted flip-back, on the eager and on the graph-replay path."""
    from oracle import model_ref as M
    from probpose_code_amd import apis
    from probpose_code_amd import synthetic as S

    sd, crops, center, scale, _ = setup
    ref = M.predict(sd, crops, 12, S.IMG_MEAN, S.IMG_STD, input_size=(192, 256), input_center=center, input_scale=scale, shift_heatmap=True)
    model = apis.init_model(CFG, {"state_dict": sd}, device="cuda:0", cfg_options={"model.test_cfg.shift_heatmap": True})
    with torch.no_grad():
        runs = [model.test_step(apis.pack_crops(crops, center, scale, model.dataset_meta)) for _ in range(4)]  # eager, eager, capture, replay
    assert model.engine._graphs, "the fourth batch of a size must have replayed a captured graph"
    kps = [np.stack([ds.pred_instances.keypoints for ds in r]) for r in runs]
    assert np.array_equal(kps[0], kps[3]), "graph replay differs from the eager launches"
    d = np.abs(kps[3] - ref["keypoints"]).max(-1)
    same = d < 2.0
    assert same.mean() > 0.95 and d[same].max() <= 1e-3, f"{d[same].max():.2e} px, {int((~same).sum())} flips"


@pytest.mark.parametrize("precision", ["bf16", "f16x3"])
def test_graph_replay_equals_eager_bit_for_bit(precision):
    """bench.py times forward_graph (hipGraph replay): it must produce exactly what the eager launch sequence does,
    on two DIFFERENT batches in a row (a capture that baked in stale pointers or inputs would repeat batch 1)."""
    from probpose_code_amd import ProbPoseEngine
    from probpose_code_amd import synthetic as S

    sd = S.synthetic_state_dict("small", seed=0, logit_scale=2.0)
    eng = ProbPoseEngine(sd, 12, precision=precision)
    keys = ("keypoints", "scores", "locs", "scalars")
    prev = None
    for seed in (100, 101):
        crops = S.synthetic_crops(64, seed=seed).cuda()
        g = {k: v.clone() for k, v in eng.forward_graph(crops, True, S.COCO_FLIP_INDICES).items() if k in keys}
        e = {k: v.clone() for k, v in eng.forward(crops, True, S.COCO_FLIP_INDICES).items() if k in keys}
        torch.cuda.synchronize()
        for k in keys:
            assert torch.equal(g[k], e[k]), f"{k}: graph replay differs from eager launches (batch seed {seed})"
        if prev is not None:
            assert not torch.equal(prev["keypoints"], g["keypoints"]), "second batch replayed the first batch's results"
        prev = g


def test_module_level_interfaces(setup):
    """backbone(inputs) -> (feat NCHW,), head.forward(feats) -> 5 tensors, head.predict([f, f_flip]) -- the
    reference's own call structure (topdown.py:109-116) -- agree with the fused predict path."""
    from oracle import model_ref as M
    from probpose_code_amd import synthetic as S

    sd, crops, center, scale, ref = setup
    model, fused = _run(sd, crops, center, scale, "f16x3")
    x = M.preprocess(crops, S.IMG_MEAN, S.IMG_STD).cuda()
    with torch.no_grad():
        feats = model.extract_feat(x)
        assert isinstance(feats, tuple) and tuple(feats[0].shape) == (B, 384, 16, 12)
        assert np.abs(feats[0].cpu().numpy() - ref["features"]).max() < 1e-4
        feats_flip = model.extract_feat(x.flip(-1))  # the reference's call pattern as written (topdown.py:109-112)
        assert np.abs(feats[0].cpu().numpy() - ref["features"]).max() < 1e-4, "second call overwrote the first result"
        hm, prob, vis, oks, err = model.head.forward(feats)
        assert tuple(hm.shape) == (B, 17, 64, 48) and tuple(prob.shape) == (B, 17, 1, 1)
        assert torch.allclose(hm.sum((-1, -2)), torch.ones(B, 17, device="cuda"), atol=1e-5)
        from probpose_code_amd import apis

        batch = apis.pack_crops(crops, center, scale, model.dataset_meta)
        preds = model.head.predict([feats, feats_flip], batch["data_samples"], test_cfg=model.test_cfg)
    for b in range(B):
        kp_fused_input = (fused[b].pred_instances.keypoints - center[b] + 0.5 * scale[b]) / scale[b] * (192, 256)
        assert np.abs(preds[b].keypoints - kp_fused_input).max() < 1e-3
        assert np.allclose(preds[b].keypoints_probs, fused[b].pred_instances.keypoints_probs, atol=1e-5)


def test_errors_and_no_cpu_fallback(setup):
    from probpose_code_amd import Config, build_pose_estimator

    cfg = Config.fromfile(CFG)
    model = build_pose_estimator(dict(cfg.model))
    with pytest.raises(RuntimeError, match="no CPU path"):
        model.engine
    with pytest.raises(RuntimeError, match="Invalid mode"):
        model.forward(torch.zeros(1, 3, 256, 192), None, mode="bogus")
    bad = dict(cfg.model)
    bad["head"] = dict(bad["head"], deconv_out_channels=(256, 256), deconv_kernel_sizes=(4,))
    with pytest.raises(ValueError, match="same length"):
        build_pose_estimator(bad)


def test_config4_vit_base_384x288_bf16():
    """BASELINE config 4: ViT-B (768 / 12 heads x 64), 384x288 input, 96x72 heatmaps, bf16 operands. Bounded
    against the fp32 oracle; exercises the 432-token attention, E = 768 LayerNorm, K = 3072 GEMMs and the
    24x18 -> 6x6 -> 3x3 -> 1x1 tower pooling."""
    from oracle import model_ref as M
    from probpose_code_amd import ProbPoseEngine
    from probpose_code_amd import synthetic as S

    torch.set_num_threads(min(16, os.cpu_count()))
    img = (384, 288)
    sd = S.synthetic_state_dict("base", img_size=img, seed=0, logit_scale=2.0)
    x = S.synthetic_crops(3, img_size=img, seed=1)
    ref = M.predict(sd, x, 12, S.IMG_MEAN, S.IMG_STD, input_size=(288, 384))
    eng = ProbPoseEngine(sd, 12, img_size=img, precision="bf16", input_size=(288, 384))
    out = eng.forward(x.cuda(), True, S.COCO_FLIP_INDICES, return_heatmaps=True)
    assert tuple(out["heatmaps"].shape) == (3, 17, 96, 72)
    d = np.abs(out["keypoints"].cpu().numpy()[:, None] - ref["keypoints_input_space"]).max(-1)
    same = d < 2.0
    print(f"config 4 bf16: {float(1 - same.mean()):.3f} argmax flips, {d[same].max():.3f} px on the rest")
    assert same.mean() >= 0.90 and d[same].max() < 0.5  # measured 3-8 % flips, 0.3-0.45 px
    assert np.abs(out["scalars"][0].cpu().numpy()[:, None] - ref["keypoints_probs"]).max() < 3e-2
    with pytest.raises(Exception, match="not instantiated"):  # fp32 at 432 x 64 does not fit one CU's LDS
        ProbPoseEngine(sd, 12, img_size=img, precision="f32", input_size=(288, 384)).forward(
            x.cuda(), True, S.COCO_FLIP_INDICES)
    # the parity mode covers this geometry (attention in key stages): <= 1e-3 px, no flips
    eng = ProbPoseEngine(sd, 12, img_size=img, precision="f16x3", input_size=(288, 384))
    out = eng.forward(x.cuda(), True, S.COCO_FLIP_INDICES)
    d = np.abs(out["keypoints"].cpu().numpy()[:, None] - ref["keypoints_input_space"]).max(-1)
    assert (d < 2.0).all() and d.max() <= 1e-3, f"config 4 f16x3: {int((d >= 2).sum())} flips, L_inf {d[d < 2].max():.2e} px"
    for i, name in enumerate(("keypoints_probs", "keypoints_visible", "keypoints_oks")):
        assert np.abs(out["scalars"][i].cpu().numpy()[:, None] - ref[name]).max() <= 1e-3, name


def test_config4_bs32_graph_replay_default_plan():
    """BASELINE config 4 ON THE LAUNCH PLAN IT IS BENCHMARKED ON: ViT-B 384x288, f16x3, B = 32 crops with the flip
    pass (27 648 token rows: 576 tiles of 192 x 192 for proj / fc2, more for qkv / fc1 - every Linear layer clears the 512-tile threshold of
    pp_linear_dma.hip, as at bs 64), the 24 x 18 first tower stage in its Winograd form, captured as a hipGraph and REPLAYED. The library's
    launch tally (pp_launch_count) proves which kernels the plan ran: since round 5 the layers run with their LayerNorms FOLDED into the
    Linear layers (pp_linear_ln_folded: 48 launches per forward, two LayerNorm launches left - behind the patch embedding and the final
    one); plan switch ln_fold = False is round 4's plan (49 twelve-wave Linear launches + 25 LayerNorm launches) and must agree with it.
    Keypoints / scalars of the replay against oracle.model_ref.predict on the first 8 crops (crops are independent: eval-mode
    BatchNorm, per-crop decode) within 1e-3, no argmax flips (reference workload:
    mmpose/models/heads/hybrid_heads/probmap_head.py:715-804 at heatmap 96 x 72)."""
    from oracle import model_ref as M
    from probpose_code_amd import ProbPoseEngine, _lib
    from probpose_code_amd import synthetic as S

    torch.set_num_threads(min(16, os.cpu_count()))
    img, B, NREF = (384, 288), 32, 8
    sd = S.synthetic_state_dict("base", img_size=img, seed=0, logit_scale=2.0)
    x = S.synthetic_crops(B, img_size=img, seed=3)
    ref = M.predict(sd, x[:NREF], 12, S.IMG_MEAN, S.IMG_STD, input_size=(288, 384))
    xd = x.cuda()
    n_fwd = 3  # two eager warm-ups and the captured one (engine.capture)
    results = {}
    for fold in (True, False):
        eng = ProbPoseEngine(sd, 12, img_size=img, precision="f16x3", input_size=(288, 384), plan=dict(ln_fold=fold))
        assert eng.winograd, "the 24 x 18 tower stage must take the Winograd kernel"
        assert eng.ln_fold == fold and ("LayerNorm folded" in eng.layer_plan) == fold
        _lib.reset_launch_counts()
        eng.forward_graph(xd, True, S.COCO_FLIP_INDICES)  # warm-up launches + the capture
        torch.cuda.synchronize()
        tally = {k: _lib.launch_count(k) for k in ("linear_dma_fold", "linear_dma_tile", "winograd_gemm_pool",
                                                   "pp_attention_dma.hip", "layernorm", "pp_gemm.hip", "pp_panel_split.hip")}
        print(f"config 4 launch tally of three forwards (ln_fold = {fold}):", tally)
        if fold:
            assert tally["linear_dma_fold"] == 48 * n_fwd, "qkv / proj / fc1 / fc2 of every layer through pp_linear_ln_folded"
            assert tally["linear_dma_tile"] == 1 * n_fwd, "the patch embedding on the twelve-wave kernel (one tile per workgroup)"
            assert tally["layernorm"] == 2 * n_fwd, "LayerNorm launches left: ln1 of layer 0 and the final one"
        else:
            assert tally["linear_dma_fold"] == 0 and tally["linear_dma_tile"] == 49 * n_fwd, "patch embed + the four Linear layers of every layer"
            assert tally["layernorm"] == 25 * n_fwd
        assert tally["pp_gemm.hip"] == 0, "no Linear layer of the plan on the 128 x 128 kernel"
        assert _lib.launch_count("winograd_gemm_pool") == n_fwd and _lib.launch_count("winograd_input_transform") == n_fwd
        assert _lib.launch_count("pp_attention_dma.hip") == 12 * n_fwd, "432-token attention on the LDS-DMA kernel"
        _lib.reset_launch_counts()
        out = eng.forward_graph(xd, True, S.COCO_FLIP_INDICES)  # a REPLAY: no host-side launch is tallied
        torch.cuda.synchronize()
        assert _lib.launch_count("linear_dma_fold") == 0 and _lib.launch_count("linear_dma_tile") == 0 and _lib.launch_count("pp_attention_dma.hip") == 0
        kp = out["keypoints"].cpu().numpy()
        assert np.isfinite(kp).all()
        d = np.abs(kp[:NREF, None] - ref["keypoints_input_space"]).max(-1)
        assert (d < 2.0).all() and d.max() <= 1e-3, f"config 4 bs {B} replay (ln_fold = {fold}): {int((d >= 2).sum())} flips, L_inf {d[d < 2].max():.2e} px"
        print(f"config 4 bs {B} replay (ln_fold = {fold}): L_inf {d.max():.2e} px vs the oracle")
        for i, name in enumerate(("keypoints_probs", "keypoints_visible", "keypoints_oks")):
            assert np.abs(out["scalars"][i].cpu().numpy()[:NREF, None] - ref[name]).max() <= 1e-3, name
        # the replay is deterministic and equals the eager plan bit for bit
        eager = eng.forward(xd, True, S.COCO_FLIP_INDICES)
        torch.cuda.synchronize()
        assert torch.equal(eager["keypoints"], out["keypoints"])
        results[fold] = kp
        del eng
        torch.cuda.empty_cache()
    dd = np.abs(results[True] - results[False]).max()
    assert dd <= 1e-3, f"folded and stand-alone LayerNorm plans differ by {dd:.2e} px"


def test_vit_small_folded_chain_is_the_default_plan_and_agrees_with_the_plain_chain():
    """The headline workload's layer plan (round 5): ln1 of layers 1 .. 11 folded into the qkv projection - every projection + FFN launch but the
    last leaves its rows once, in the operand format, with (mean, rstd) per row (pp_proj_ffn_split_folded, tallied "ffn_dma_fold"), the next
    qkv + attention launch applies them (pp_qkv_attention_split_folded). The launch tally names the plan; plan switch ln_fold = False is the plain
    chain (round 4) and must agree within 1e-3 px / 1e-5 on the scalar heads; the hipGraph replay equals the eager launches bit for bit."""
    from probpose_code_amd import ProbPoseEngine, _lib
    from probpose_code_amd import synthetic as S

    sd = S.synthetic_state_dict("small", seed=4, logit_scale=2.0)
    xd = S.synthetic_crops(64, seed=44).cuda()
    outs = {}
    for fold in (True, False):
        eng = ProbPoseEngine(sd, 12, precision="f16x3", plan=dict(ln_fold=fold))
        assert eng.ln_fold_fused == fold and ("folded into the qkv projection" in eng.layer_plan) == fold
        _lib.reset_launch_counts()
        o = eng.forward(xd, True, S.COCO_FLIP_INDICES)
        torch.cuda.synchronize()
        assert _lib.launch_count("ffn_dma_fold") == (12 if fold else 0) and _lib.launch_count("ffn_dma_pair") == (0 if fold else 12)
        assert _lib.launch_count("pp_qkv_attn_split.hip") == 12 and _lib.launch_count("layernorm") == 0
        outs[fold] = {k: o[k].clone() for k in ("keypoints", "scalars")}
        for _ in range(3):
            rep = eng.forward_graph(xd, True, S.COCO_FLIP_INDICES)
        torch.cuda.synchronize()
        assert torch.equal(rep["keypoints"], outs[fold]["keypoints"]) and torch.equal(rep["scalars"], outs[fold]["scalars"])
        del eng
        torch.cuda.empty_cache()
    d = (outs[True]["keypoints"] - outs[False]["keypoints"]).abs().max().item()
    ds = (outs[True]["scalars"] - outs[False]["scalars"]).abs().max().item()
    assert d <= 1e-3 and ds <= 1e-5, f"plans differ: keypoints {d:.2e} px, scalars {ds:.2e}"


def test_config4_folded_plan_ragged_row_count_agrees_with_generic_plan():
    """ViT-B 384x288 at B = 33 with flip: 28 512 token rows = 148.5 row tiles of the twelve-wave Linear kernel - the last tile of every layer
    is half empty (buffer descriptors end at row M: statistics, residual rows and outputs past it are never touched). The folded-LayerNorm
    plan against the generic plan (plan switch ln_fold = False) on the same crops: keypoints within 1e-3 px, scalars within 1e-5."""
    from probpose_code_amd import ProbPoseEngine, _lib
    from probpose_code_amd import synthetic as S

    img, B = (384, 288), 33
    sd = S.synthetic_state_dict("base", img_size=img, seed=2, logit_scale=2.0)
    xd = S.synthetic_crops(B, img_size=img, seed=21).cuda()
    outs = {}
    for fold in (True, False):
        eng = ProbPoseEngine(sd, 12, img_size=img, precision="f16x3", input_size=(288, 384), plan=dict(ln_fold=fold))
        _lib.reset_launch_counts()
        o = eng.forward(xd, True, S.COCO_FLIP_INDICES)
        torch.cuda.synchronize()
        assert (_lib.launch_count("linear_dma_fold") == 48) == fold
        outs[fold] = {k: o[k].clone() for k in ("keypoints", "scalars")}
        assert torch.isfinite(outs[fold]["keypoints"]).all() and torch.isfinite(outs[fold]["scalars"]).all()
        del eng
        torch.cuda.empty_cache()
    d = (outs[True]["keypoints"] - outs[False]["keypoints"]).abs().max().item()
    ds = (outs[True]["scalars"] - outs[False]["scalars"]).abs().max().item()
    assert d <= 1e-3 and ds <= 1e-5, f"plans differ: keypoints {d:.2e} px, scalars {ds:.2e}"


def test_vit_base_256x192_bs64_folded_plan_within_1e3():
    """ViT-B at the reference's own ViTPose-base geometry (configs/body_2d_keypoint/topdown_heatmap/coco/td-hm_ViTPose-base_8xb64-210e_coco-256x192.py:
    46-61: 256x192 crops, 192 tokens of 12 heads x 64) with the ProbPose head, bs 64 + flip = 24 576 token rows: exactly the row count from
    which the folded-LayerNorm plan runs (512 tiles for proj / fc2). The launch tally names the plan; keypoints / scalars of the hipGraph
    replay against oracle.model_ref.predict on the first 6 crops within 1e-3, no argmax flips; one crop fewer (63) falls back to the
    generic plan and must agree."""
    from oracle import model_ref as M
    from probpose_code_amd import ProbPoseEngine, _lib
    from probpose_code_amd import synthetic as S

    torch.set_num_threads(min(16, os.cpu_count()))
    B, NREF = 64, 6
    sd = S.synthetic_state_dict("base", seed=1, logit_scale=2.0)
    x = S.synthetic_crops(B, seed=9)
    ref = M.predict(sd, x[:NREF], 12, S.IMG_MEAN, S.IMG_STD)
    eng = ProbPoseEngine(sd, 12, precision="f16x3")
    xd = x.cuda()
    _lib.reset_launch_counts()
    out = eng.forward(xd, True, S.COCO_FLIP_INDICES)
    torch.cuda.synchronize()
    assert _lib.launch_count("linear_dma_fold") == 48 and _lib.launch_count("layernorm") == 2, "bs 64: the folded-LayerNorm plan"
    rep = eng.forward_graph(xd, True, S.COCO_FLIP_INDICES)
    rep = eng.forward_graph(xd, True, S.COCO_FLIP_INDICES)
    rep = eng.forward_graph(xd, True, S.COCO_FLIP_INDICES)
    torch.cuda.synchronize()
    assert torch.equal(rep["keypoints"], out["keypoints"])
    kp = rep["keypoints"].cpu().numpy()
    d = np.abs(kp[:NREF, None] - ref["keypoints_input_space"]).max(-1)
    assert (d < 2.0).all() and d.max() <= 1e-3, f"ViT-B 256x192 bs 64: {int((d >= 2).sum())} flips, L_inf {d[d < 2].max():.2e} px"
    for i, name in enumerate(("keypoints_probs", "keypoints_visible", "keypoints_oks")):
        assert np.abs(rep["scalars"][i].cpu().numpy()[:NREF, None] - ref[name]).max() <= 1e-3, name
    _lib.reset_launch_counts()
    small = eng.forward(xd[:63].contiguous(), True, S.COCO_FLIP_INDICES)
    torch.cuda.synchronize()
    assert _lib.launch_count("linear_dma_fold") == 0 and _lib.launch_count("layernorm") == 25, "bs 63: below the threshold, the generic plan"
    dd = (small["keypoints"][:NREF] - rep["keypoints"][:NREF]).abs().max().item()
    assert dd <= 1e-3, f"generic and folded plans differ by {dd:.2e} px"


def test_vit_small_384x288_layer_plan_is_named_and_within_1e3():
    """A ViT-S at another input size (432 tokens) misses the fused qkv + attention kernel, which is written for 192-token sequences:
    the engine says so ONCE, by name (RuntimeWarning + `layer_plan`), and runs three launches per layer - the projection + FFN launch
    takes any row count. Parity at that size against the oracle (f16x3: <= 1e-3 px, no flips)."""
    from oracle import model_ref as M
    from probpose_code_amd import ProbPoseEngine, _lib
    from probpose_code_amd import synthetic as S

    torch.set_num_threads(min(16, os.cpu_count()))
    img = (384, 288)
    sd = S.synthetic_state_dict("small", img_size=img, seed=0, logit_scale=2.0)
    x = S.synthetic_crops(3, img_size=img, seed=5)
    ref = M.predict(sd, x, 12, S.IMG_MEAN, S.IMG_STD, input_size=(288, 384))
    with pytest.warns(RuntimeWarning, match="432-token sequences .* miss the fused qkv \\+ attention kernel"):
        eng = ProbPoseEngine(sd, 12, img_size=img, precision="f16x3", input_size=(288, 384), plan=dict(small_plan=False))  # (3 crops: the row-owner plan is what is named here)
    assert eng.layer_plan.startswith("three launches per layer") and not eng.fuse_qkv_attn
    _lib.reset_launch_counts()
    out = eng.forward(x.cuda(), True, S.COCO_FLIP_INDICES)
    torch.cuda.synchronize()
    assert _lib.launch_count("pp_ffn_dma.hip") == 12 and _lib.launch_count("pp_qkv_attn_split.hip") == 0
    d = np.abs(out["keypoints"].cpu().numpy()[:, None] - ref["keypoints_input_space"]).max(-1)
    assert (d < 2.0).all() and d.max() <= 1e-3, f"{int((d >= 2).sum())} flips, L_inf {d[d < 2].max():.2e} px"
    for i, name in enumerate(("keypoints_probs", "keypoints_visible", "keypoints_oks")):
        assert np.abs(out["scalars"][i].cpu().numpy()[:, None] - ref[name]).max() <= 1e-3, name
    # the default geometry reports the two-launch layer and does not warn
    import warnings as _w
    with _w.catch_warnings():
        _w.simplefilter("error")
        eng2 = ProbPoseEngine(S.synthetic_state_dict("small", seed=0), 12, precision="f16x3")
    assert eng2.layer_plan.startswith("two launches per layer")


def test_config4_row_owner_residual_layernorm_plan(monkeypatch):
    """PP_FUSE_RESLN=1 selects the E = 768 form of the fused residual GEMM + LayerNorm kernel (112-row tiles, two column
    halves) for patch embed / projection / fc2 of ViT-B - off by default there (slower with two steps in flight, see
    engine.py). The plan must hold the same parity as the default one: f16x3 <= 1e-3 px, no flips."""
    from oracle import model_ref as M
    from probpose_code_amd import ProbPoseEngine
    from probpose_code_amd import synthetic as S

    torch.set_num_threads(min(16, os.cpu_count()))
    img = (384, 288)
    sd = S.synthetic_state_dict("base", img_size=img, seed=0, logit_scale=2.0)
    x = S.synthetic_crops(3, img_size=img, seed=1)
    ref = M.predict(sd, x, 12, S.IMG_MEAN, S.IMG_STD, input_size=(288, 384))
    monkeypatch.setenv("PP_FUSE_RESLN", "1")
    for precision in ("f16x3", "bf16"):
        eng = ProbPoseEngine(sd, 12, img_size=img, precision=precision, input_size=(288, 384), plan=dict(small_plan=False))
        assert eng._resln_768
        eng.profile = {}
        out = eng.forward(x.cuda(), True, S.COCO_FLIP_INDICES)
        torch.cuda.synchronize()
        assert len(eng.profile["gemm_res_ln"]) == 25 and "layernorm" not in eng.profile
        d = np.abs(out["keypoints"].cpu().numpy()[:, None] - ref["keypoints_input_space"]).max(-1)
        if precision == "f16x3":
            assert (d < 2.0).all() and d.max() <= 1e-3, f"{int((d >= 2).sum())} flips, L_inf {d[d < 2].max():.2e} px"
        else:
            same = d < 2.0
            assert same.mean() >= 0.90 and d[same].max() < 0.5


@pytest.mark.parametrize("switch", ["PP_FUSE_ATTN", "PP_FUSE_QKV", "PP_FUSE_PROJ", "PP_FUSE_MLP", "PP_SPLIT_K", "PP_FUSE_HEAD", "PP_FUSE_POOL", "PP_FUSE_RESLN"])
def test_bf16_fallback_launch_plans_end_to_end(switch, monkeypatch):
    """Every PP_FUSE_* / PP_SPLIT_K switch selects a different launch plan for the bf16 mode (two launches per layer, the
    plain qkv GEMM, separate projection, unfused FFN, unsplit tower convolutions, separate final 1x1 conv): each plan, end
    to end, against the default plan - same network, different rounding points, so the comparison is at bf16 resolution -
    and against the fp32 oracle's argmax."""
    from oracle import model_ref as M
    from probpose_code_amd import ProbPoseEngine
    from probpose_code_amd import synthetic as S

    torch.set_num_threads(min(16, os.cpu_count()))
    sd = S.synthetic_state_dict("small", seed=0, logit_scale=2.0)
    crops = S.synthetic_crops(8, seed=21)
    base = ProbPoseEngine(sd, 12, precision="bf16").forward(crops.cuda(), True, S.COCO_FLIP_INDICES, return_heatmaps=True)
    base = {k: v.clone() for k, v in base.items()}
    monkeypatch.setenv(switch, "0")
    eng = ProbPoseEngine(sd, 12, precision="bf16")
    assert getattr(eng, {"PP_SPLIT_K": "split_k"}.get(switch, switch[3:].lower())) is False
    out = eng.forward(crops.cuda(), True, S.COCO_FLIP_INDICES, return_heatmaps=True)
    torch.cuda.synchronize()
    assert (out["heatmaps"] - base["heatmaps"]).abs().max().item() <= 0.12, "heatmaps drift beyond bf16 noise"
    for i in range(3):
        assert (out["scalars"][i] - base["scalars"][i]).abs().max().item() <= 2e-2
    ref = M.predict(sd, crops, 12, S.IMG_MEAN, S.IMG_STD)
    d = np.abs(out["keypoints"].cpu().numpy()[:, None] - ref["keypoints_input_space"]).max(-1)
    same = d < 2.0
    assert same.mean() >= 0.88 and d[same].max() <= 0.75, f"{switch}=0: {float(1 - same.mean()):.3f} flips, {d[same].max():.3f} px"


def test_normalize_none_head_without_sparsemax():
    """`normalize=None` (probmap_head.py:249,642-646: Identity instead of Sparsemax, then clamp(x / T, 0, 1)) through
    the fused decode, against the oracle; not a ProbPose configuration but part of the head's constructor surface."""
    from oracle import model_ref as M
    from probpose_code_amd import ProbPoseEngine
    from probpose_code_amd import synthetic as S

    torch.set_num_threads(min(16, os.cpu_count()))
    sd = S.synthetic_state_dict("small", seed=0, logit_scale=0.3)  # small logits: the clamp must not saturate everywhere
    crops = S.synthetic_crops(3, seed=31)
    ref = M.predict(sd, crops, 12, S.IMG_MEAN, S.IMG_STD, normalize=None)
    eng = ProbPoseEngine(sd, 12, precision="f16x3", normalize=None)
    out = eng.forward(crops.cuda(), True, S.COCO_FLIP_INDICES, return_heatmaps=True)
    hm = out["heatmaps"].cpu().numpy()
    assert 0.05 < float((ref["heatmaps"] > 0).mean()) < 0.95, "the reference maps must exercise both sides of the clamp"
    assert np.abs(hm - ref["heatmaps"]).max() <= 1e-4
    d = np.abs(out["keypoints"].cpu().numpy()[:, None] - ref["keypoints_input_space"]).max(-1)
    assert (d < 2.0).mean() >= 0.9 and d[d < 2.0].max() <= 1e-2  # dense maps: flat maxima, looser than the Sparsemax case


@pytest.mark.parametrize("precision", ["bf16", "f16x3"])
@pytest.mark.parametrize("B,flip", [(1, True), (3, False), (5, True)])
def test_ragged_batch_sizes_and_single_pass(precision, B, flip):
    """Batch sizes that fill no tile (1, 3, 5 crops: 192 - 960 token rows against 96 / 128 / 256-row tiles, 17 - 85
    decode workgroups) and the single-pass path (`flip_test=False`: no flip copy, no TTA merge) against the oracle."""
    from oracle import model_ref as M
    from probpose_code_amd import ProbPoseEngine
    from probpose_code_amd import synthetic as S

    torch.set_num_threads(min(16, os.cpu_count()))
    sd = S.synthetic_state_dict("small", seed=0, logit_scale=2.0)
    crops = S.synthetic_crops(B, seed=40 + B)
    ref = M.predict(sd, crops, 12, S.IMG_MEAN, S.IMG_STD, flip_test=flip)
    eng = ProbPoseEngine(sd, 12, precision=precision)
    out = eng.forward(crops.cuda(), flip, S.COCO_FLIP_INDICES if flip else None)
    d = np.abs(out["keypoints"].cpu().numpy()[:, None] - ref["keypoints_input_space"]).max(-1)
    same = d < 2.0
    if precision == "f16x3":
        assert same.all() and d.max() <= 1e-3, f"{int((~same).sum())} flips, {d[same].max():.2e} px"
        assert np.abs(out["scalars"][0].cpu().numpy()[:, None] - ref["keypoints_probs"]).max() <= 1e-3
    else:
        assert same.mean() >= 0.85 and d[same].max() <= 0.75
    # the replayed graph of the same shape agrees with the eager launches
    g = eng.forward_graph(crops.cuda(), flip, S.COCO_FLIP_INDICES if flip else None)
    torch.cuda.synchronize()
    assert torch.equal(g["keypoints"], out["keypoints"])


def test_output_keypoint_indices_selects_fields(setup):
    """test_cfg.output_keypoint_indices (topdown.py:172-190): every `keypoint*` field and the heatmaps are sliced."""
    sd, crops, center, scale, ref = setup
    idx = [0, 5, 6, 11, 12]
    _, results = _run(sd, crops, center, scale, "f16x3", {"model.test_cfg.output_keypoint_indices": idx,
                                                          "model.test_cfg.output_heatmaps": True})
    for b, ds in enumerate(results):
        pi = ds.pred_instances
        assert pi.keypoints.shape == (1, 5, 2) and pi.keypoints_probs.shape == (1, 5) and pi.keypoint_scores.shape == (1, 5)
        assert np.abs(pi.keypoints - ref["keypoints"][b][:, idx]).max() <= 1e-3
        assert tuple(ds.pred_fields.heatmaps.shape) == (5, 64, 48)


def test_chaos_soak_of_the_drop_in_calls():
    """Seconds of `test_step` / `test_step_stream(depth 2)` on batches of random size (small-batch plan, its boundary, the row-owner plan; eager
    launches, captures, a full graph cache), every result compared bit for bit with the first one of the same batch - in a subprocess, so that a
    crash inside the HIP runtime (round 6: hipGraphLaunch after a graph eviction) fails this test instead of ending the run."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-X", "faulthandler", os.path.join(root, "scripts", "r06", "chaos_soak.py"), "8"], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0 and "CHAOS SOAK OK" in r.stdout, (r.returncode, r.stdout[-600:], r.stderr[-1200:])
