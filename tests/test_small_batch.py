"""The launch plan of SMALL batches: the reference's callers hand over one crop (demo/image_demo.py:36-61), the boxes of one image
(mmpose/apis/inference.py:161-196) or a handful of persons (demo/topdown_demo_with_mmdet.py:35-41). The row-owner layer kernels of the headline plan
would run 4 workgroups on 256 CUs there; below `engine.small_rows_below` token rows the engine runs every Linear layer column-parallel
(pp_skinny_linear) with the LayerNorm done by the workgroup that completes a row block.

  * GPU, kernel: pp_skinny_linear against torch fp64 - the three tile sizes, ragged row counts, GELU / residual / pos_embed table / split and fp32
    output / weight scale, the LayerNorm tail (counters back at zero, bit-identical repeats whatever the arrival order);
  * GPU, end to end: B = 1, 2, 4, 8 (with and without flip test) against oracle.model_ref.predict <= 1e-3 px, the plan asserted by pp_launch_count,
    the replayed graph bit-identical to the eager launches, the boundary to the headline plan, ViT-B at 384 x 288.
"""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

gpu = pytest.mark.gpu
SPLIT, F32 = 2, 0


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def _lib():
    from probpose_code_amd import _lib

    return _lib


def _sp(x):
    from probpose_code_amd.weights import to_split

    return to_split(x).cuda()


def _unsp(c):
    from probpose_code_amd.weights import from_split

    return from_split(c.cpu()).double()


@gpu
@pytest.mark.parametrize("M,N,K,act,res,split_out", [
    (384, 384, 384, 0, "x", 0),        # proj at B = 1: 32 x 32 tiles, 144 workgroups
    (384, 1536, 384, 1, None, 1),      # fc1 + GELU at B = 1
    (384, 384, 1536, 0, "x", 0),       # fc2: 24 K stages
    (384, 384, 768, 0, "table", 0),    # patch embedding + pos_embed table
    (3072, 1536, 384, 1, None, 1),     # fc1 at B = 8: 96 x 96 tiles
    (3072, 384, 384, 0, "x", 0),       # proj at B = 8: 64 x 64 tiles
    (192 * 5 + 37, 1152, 384, 0, None, 1),  # ragged rows (a last block of 5 / 37 rows)
    (50, 96, 64, 2, None, 0),          # one K stage, ReLU, a single row block
])
def test_skinny_linear_vs_fp64(M, N, K, act, res, split_out):
    L = _lib()
    a, w, b = _rand(M, K, seed=11), _rand(N, K, seed=12, scale=1 / math.sqrt(K)), _rand(N, seed=13, scale=0.3)
    ref = a.double() @ w.double().t() + b.double()
    ref = F.gelu(ref) if act == 1 else (F.relu(ref) if act == 2 else ref)
    r = None
    if res == "x":
        r = _rand(M, N, seed=14)
        ref = ref + r.double()
    elif res == "table":
        r = _rand(192, N, seed=14)
        ref = ref + r.double()[torch.arange(M) % 192]
    e = 15
    ad, wd, bd = _sp(a), _sp(w * 2.0 ** e), b.cuda()
    outs = []
    for _ in range(3):
        out = torch.full((M, N), float("nan"), device="cuda")
        rd = None
        if res == "x":
            out.copy_(r)  # the residual stream updated in place
            rd = out
        elif res == "table":
            rd = r.cuda()
        L.call("pp_skinny_linear", ad.data_ptr(), wd.data_ptr(), bd.data_ptr(), rd.data_ptr() if rd is not None else None, 192 if res == "table" else 0,
               out.data_ptr(), SPLIT if split_out else F32, M, N, K, act, 2.0 ** -e, None, None, 1e-6, None, None, None)
        outs.append(out.cpu())
    got = _unsp(outs[0]) if split_out else outs[0].double()
    assert not torch.isnan(got).any(), "rows or columns left unwritten"
    torch.testing.assert_close(got, ref, rtol=2e-5, atol=2e-5)
    for o in outs[1:]:
        assert torch.equal(o.view(torch.int32), outs[0].view(torch.int32)), "run-to-run difference"
    t = L.lib.pp_skinny_linear_tile(M, N, K, 0)
    assert t // 1000 in (32, 64, 96) and t % 1000 in (32, 64, 96) and N % (t % 1000) == 0


@gpu
@pytest.mark.parametrize("M,K", [(384, 384), (384, 1536), (3072, 384), (777, 768)])
def test_skinny_linear_layernorm_tail(M, K):
    """x <- x + a W^T + b ; h <- LayerNorm(x) in ONE launch: the workgroup that stores a row block's last tile normalises the block. Against fp64;
    the counters are back at zero; thirty repeats are bit-identical (the arrival order changes, the result must not)."""
    L = _lib()
    N, eps = 384, 1e-6
    a, w, b = _rand(M, K, seed=21), _rand(N, K, seed=22, scale=1 / math.sqrt(K)), _rand(N, seed=23, scale=0.3)
    x0 = _rand(M, N, seed=24) * 1.5 + 0.7
    g, be = 1.0 + 0.2 * _rand(N, seed=25), 0.2 * _rand(N, seed=26)
    x_ref = x0.double() + a.double() @ w.double().t() + b.double()
    h_ref = F.layer_norm(x_ref, (N,), g.double(), be.double(), eps)
    ad, wd, bd, gd, bed = _sp(a), _sp(w), b.cuda(), g.cuda(), be.cuda()
    cnt = torch.zeros((M + 31) // 32, dtype=torch.int32, device="cuda")
    first = None
    for it in range(30):
        x = x0.cuda().clone()
        h = torch.full((M, N), float("nan"), device="cuda")
        L.call("pp_skinny_linear", ad.data_ptr(), wd.data_ptr(), bd.data_ptr(), x.data_ptr(), 0, x.data_ptr(), F32, M, N, K, 0, 1.0, gd.data_ptr(),
               bed.data_ptr(), eps, h.data_ptr(), cnt.data_ptr(), None)
        torch.cuda.synchronize()
        assert int(cnt.abs().sum()) == 0, "counters must be left at zero"
        if first is None:
            first = (x.cpu(), h.cpu())
            torch.testing.assert_close(first[0].double(), x_ref, rtol=2e-5, atol=2e-5)
            assert not torch.isnan(_unsp(first[1])).any(), "a row block was never normalised"
            torch.testing.assert_close(_unsp(first[1]), h_ref, rtol=3e-5, atol=3e-5)
        else:
            assert torch.equal(x.cpu(), first[0]) and torch.equal(h.cpu().view(torch.int32), first[1].view(torch.int32)), f"repeat {it} differs"
    with pytest.raises(L.ProbPoseLibraryError):  # the tail normalises fp32 rows
        L.call("pp_skinny_linear", ad.data_ptr(), wd.data_ptr(), bd.data_ptr(), None, 0, x.data_ptr(), SPLIT, M, N, K, 0, 1.0, gd.data_ptr(), bed.data_ptr(),
               eps, h.data_ptr(), cnt.data_ptr(), None)
    with pytest.raises(L.ProbPoseLibraryError):  # ... and needs its counters
        L.call("pp_skinny_linear", ad.data_ptr(), wd.data_ptr(), bd.data_ptr(), None, 0, x.data_ptr(), F32, M, N, K, 0, 1.0, gd.data_ptr(), bed.data_ptr(),
               eps, h.data_ptr(), None, None)
    with pytest.raises(L.ProbPoseLibraryError):  # a scale that is no power of two
        L.call("pp_skinny_linear", ad.data_ptr(), wd.data_ptr(), bd.data_ptr(), None, 0, x.data_ptr(), F32, M, N, K, 0, 0.3, None, None, eps, None, None, None)


# ------------------------------------------------------------------------------------------------- end to end
@gpu
@pytest.mark.parametrize("B,flip", [(1, True), (1, False), (2, True), (4, True), (8, True), (5, False)])
def test_small_batches_take_the_column_parallel_plan_and_match_the_oracle(B, flip):
    from oracle import model_ref as M
    from probpose_code_amd import ProbPoseEngine, _lib
    from probpose_code_amd import synthetic as S

    torch.set_num_threads(min(16, os.cpu_count()))
    sd = S.synthetic_state_dict("small", seed=0, logit_scale=2.0)
    crops = S.synthetic_crops(B, seed=60 + B)
    ref = M.predict(sd, crops, 12, S.IMG_MEAN, S.IMG_STD, flip_test=flip)
    eng = ProbPoseEngine(sd, 12, precision="f16x3")
    fi = S.COCO_FLIP_INDICES if flip else None
    assert eng._small_at(B * (2 if flip else 1) * 192)
    eng.forward(crops.cuda(), flip, fi)  # (first call: workspace, kernel attributes)
    _lib.reset_launch_counts()
    out = eng.forward(crops.cuda(), flip, fi)
    torch.cuda.synchronize()
    # patch embedding + 3 Linear layers per layer on the column-parallel kernel, qkv + attention per (sequence, head), no row-owner layer launch
    n_deconv = _lib.launch_count("skinny_deconv")  # (deconvolutions of at most 1 536 input pixels take the skinny kernel too)
    assert n_deconv == sum(B * (2 if flip else 1) * px <= 1536 for px in (192, 768))
    # the final 1x1 conv of the heatmap branch on 32 x 32 tiles - unless the batch is large enough for the fused deconvolution + 1x1 launch
    n_final = int(B * (2 if flip else 1) * 768 * 4 < 192 * 192)
    assert _lib.launch_count("skinny_conv1x1") == n_final
    assert _lib.launch_count("pp_skinny.hip") - n_deconv - n_final == 1 + 3 * 12, _lib.launch_count("pp_skinny.hip")
    assert _lib.launch_count("pp_qkv_attn_split.hip") == 12
    assert _lib.launch_count("pp_ffn_dma.hip") == 0 and _lib.launch_count("pp_gemm_ln.hip") == 0 and _lib.launch_count("layernorm") == 0
    d = np.abs(out["keypoints"].cpu().numpy()[:, None] - ref["keypoints_input_space"]).max(-1)
    assert (d < 2.0).all() and d.max() <= 1e-3, f"{int((d >= 2).sum())} flips, {d[d < 2].max():.2e} px"
    for i, name in enumerate(("keypoints_probs", "keypoints_visible", "keypoints_oks")):
        assert np.abs(out["scalars"][i].cpu().numpy()[:, None] - ref[name]).max() <= 1e-3, name
    keep = {k: v.clone() for k, v in out.items()}
    g = eng.forward_graph(crops.cuda(), flip, fi)
    torch.cuda.synchronize()
    assert torch.equal(g["keypoints"], keep["keypoints"]) and torch.equal(g["scalars"], keep["scalars"]), "replay differs from the eager launches"
    for _ in range(5):  # replays in a row: the LayerNorm tail's arrival order changes, the results must not
        g = eng.forward_graph(crops.cuda(), flip, fi)
    torch.cuda.synchronize()
    assert torch.equal(g["keypoints"], keep["keypoints"])


@gpu
def test_small_plan_boundary_and_switch():
    """Row counts at the boundary: the last batch of the small plan and the first of the headline plan agree with the oracle and with each other's
    neighbours; `plan=dict(small_plan=False)` keeps the row-owner kernels for every batch size (and releases the plain weight copies)."""
    from oracle import model_ref as M
    from probpose_code_amd import ProbPoseEngine, _lib
    from probpose_code_amd import synthetic as S

    torch.set_num_threads(min(16, os.cpu_count()))
    sd = S.synthetic_state_dict("small", seed=0, logit_scale=2.0)
    eng = ProbPoseEngine(sd, 12, precision="f16x3")
    B_last = (eng.small_rows_below - 1) // (2 * 192)
    assert eng._small_at(B_last * 384) and not eng._small_at((B_last + 1) * 384)
    crops = S.synthetic_crops(B_last + 1, seed=77)
    ref = M.predict(sd, crops, 12, S.IMG_MEAN, S.IMG_STD)
    for n in (B_last, B_last + 1):
        _lib.reset_launch_counts()
        out = eng.forward(crops[:n].cuda(), True, S.COCO_FLIP_INDICES)
        torch.cuda.synchronize()
        assert (_lib.launch_count("pp_skinny.hip") > 0) == (n == B_last)
        d = np.abs(out["keypoints"].cpu().numpy()[:, None] - ref["keypoints_input_space"][:n]).max(-1)
        assert (d < 2.0).all() and d.max() <= 1e-3
    off = ProbPoseEngine(sd, 12, precision="f16x3", plan=dict(small_plan=False))
    assert not off._small_at(384) and not off.w.has("l0.fc1.w")
    _lib.reset_launch_counts()
    out = off.forward(crops[:2].cuda(), True, S.COCO_FLIP_INDICES)
    torch.cuda.synchronize()
    assert _lib.launch_count("pp_skinny.hip") == 0 and _lib.launch_count("pp_ffn_dma.hip") == 12
    d = np.abs(out["keypoints"].cpu().numpy()[:, None] - ref["keypoints_input_space"][:2]).max(-1)
    assert (d < 2.0).all() and d.max() <= 1e-3


@gpu
def test_small_plan_vit_b_384x288():
    """ViT-B at 384 x 288 (432-token sequences, head dim 64: no fused qkv + attention kernel): qkv through pp_skinny_linear as well, pp_attention
    behind it; two crops + flip against the oracle."""
    from oracle import model_ref as M
    from probpose_code_amd import ProbPoseEngine, _lib
    from probpose_code_amd import synthetic as S

    torch.set_num_threads(min(16, os.cpu_count()))
    img = (384, 288)
    sd = S.synthetic_state_dict("base", img_size=img, seed=0, logit_scale=2.0)
    crops = S.synthetic_crops(2, img_size=img, seed=5)
    ref = M.predict(sd, crops, 12, S.IMG_MEAN, S.IMG_STD, input_size=(288, 384))
    eng = ProbPoseEngine(sd, 12, img_size=img, precision="f16x3", input_size=(288, 384))
    eng.forward(crops.cuda(), True, S.COCO_FLIP_INDICES)
    _lib.reset_launch_counts()
    out = eng.forward(crops.cuda(), True, S.COCO_FLIP_INDICES)
    torch.cuda.synchronize()
    assert _lib.launch_count("pp_skinny.hip") - _lib.launch_count("skinny_deconv") - _lib.launch_count("skinny_conv1x1") == 1 + 4 * 12
    d = np.abs(out["keypoints"].cpu().numpy()[:, None] - ref["keypoints_input_space"]).max(-1)
    assert (d < 2.0).all() and d.max() <= 1e-3, f"{int((d >= 2).sum())} flips, {d[d < 2].max():.2e} px"


@gpu
@pytest.mark.parametrize("B", [1, 5])
def test_small_batches_two_steps_in_flight_equal_serial_launches(B):
    """The small-batch plan under `StepPipeline(depth=2)` - what `test_step_stream` drives for the persons of consecutive video frames: two steps
    in flight on two streams, each with its own workspace (arrival counters of the LayerNorm tails included) and captured graph, the towers of
    both on the engine's second head stream. Every pipelined record must equal the serial eager launch of the same batch bit for bit, over forty
    batches (the LayerNorm tail's arrival order changes from replay to replay and between the two concurrent steps)."""
    from probpose_code_amd import ProbPoseEngine
    from probpose_code_amd import synthetic as S
    from probpose_code_amd.dist import pack_records
    from probpose_code_amd.pipeline import StepPipeline

    flip = S.COCO_FLIP_INDICES
    sd = S.synthetic_state_dict("small", seed=0, logit_scale=2.0)
    eng = ProbPoseEngine(sd, 12, precision="f16x3", device="cuda:0")
    assert eng._small_at(B * 2 * 192)
    batches = [S.synthetic_crops(B, seed=700 + i).cuda() for i in range(6)]
    want = []
    for x in batches:
        want.append(pack_records(eng.forward(x, True, flip)).cpu().numpy().copy())
    assert not np.array_equal(want[0], want[1])
    pipe = StepPipeline(eng, B, flip, depth=2)
    tickets = []
    for it in range(40):
        i = (it * 5 + it // 6) % 6
        tickets.append((pipe.submit(batches[i]), i))
        if it >= 1:
            t, j = tickets[it - 1]
            assert np.array_equal(pipe.result(t)[0].numpy(), want[j]), f"step {it - 1} (batch {j}) differs from the serial launch"
    t, j = tickets[-1]
    assert np.array_equal(pipe.result(t)[0].numpy(), want[j])


@gpu
@pytest.mark.parametrize("B,H,W,Cin", [(2, 16, 12, 384), (2, 32, 24, 256), (5, 16, 12, 384), (16, 16, 12, 384), (3, 24, 18, 768)])
def test_skinny_deconv_vs_fp64_and_the_generic_kernel(B, H, W, Cin):
    """pp_skinny_deconv (ConvTranspose2d k4 s2 p1 + folded BatchNorm + ReLU, the four output phases as column-parallel GEMMs with the taps gathered
    by LDS-DMA) against torch fp64 `conv_transpose2d` and against pp_conv_gemm's all-phases launch (same sums in the same order: equal bits)."""
    L = _lib()
    Cout = 256
    x = _rand(B, H, W, Cin, seed=31)
    wt = _rand(Cin, Cout, 4, 4, seed=32, scale=math.sqrt(2.0 / (4 * Cin)))  # ConvTranspose2d weight (Cin, Cout, 4, 4)
    shift = _rand(Cout, seed=33, scale=0.2)
    ref = F.relu(F.conv_transpose2d(x.permute(0, 3, 1, 2).double(), wt.double(), None, stride=2, padding=1) + shift.double().view(1, -1, 1, 1))
    ref = ref.permute(0, 2, 3, 1).contiguous()  # NHWC
    ph = torch.empty((2, 2, Cout, 4 * Cin))     # the four phase matrices, as weights.pack builds them
    for py in range(2):
        for px in range(2):
            for ty in range(2):
                for tx in range(2):
                    tap = ty * 2 + tx
                    ph[py, px, :, tap * Cin:(tap + 1) * Cin] = wt[:, :, 3 - 2 * ty - py, 3 - 2 * tx - px].t()
    xd, wd, bd = _sp(x), _sp(ph), shift.cuda()
    out = torch.full((B, 2 * H, 2 * W, Cout), float("nan"), device="cuda")
    L.call("pp_skinny_deconv", xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), out.data_ptr(), B, H, W, Cin, Cout, None)
    got = _unsp(out)
    assert not torch.isnan(got).any(), "pixels left unwritten"
    torch.testing.assert_close(got, ref, rtol=2e-5, atol=2e-5)
    gen = torch.full((B, 2 * H, 2 * W, Cout), float("nan"), device="cuda")
    L.call("pp_conv_gemm", 2, 2, xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), gen.data_ptr(), B, H, W, Cin, Cout, -1, -1, 1, 0, 0, 0, 0, Cout, 2, 2, None)
    assert torch.equal(out.cpu().view(torch.int32), gen.cpu().view(torch.int32)), "differs from pp_conv_gemm's deconvolution"
    again = torch.full((B, 2 * H, 2 * W, Cout), float("nan"), device="cuda")
    L.call("pp_skinny_deconv", xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), again.data_ptr(), B, H, W, Cin, Cout, None)
    assert torch.equal(out.cpu().view(torch.int32), again.cpu().view(torch.int32))


@gpu
@pytest.mark.parametrize("n_img,P,K,n_valid", [(2, 64 * 48, 256, 17), (5, 64 * 48, 256, 17), (3, 1000, 128, 33), (1, 40, 64, 1)])
def test_skinny_conv1x1_planar_vs_fp64(n_img, P, K, n_valid):
    """pp_skinny_conv1x1_planar (final layer of the heatmap branch at small batches: 1x1 conv to a few channels, planar fp32 out) against fp64; ragged
    pixel counts (rows past the end read as zeros and are not stored), the padded weight rows never reach the output, planes past n_valid untouched."""
    L = _lib()
    x = _rand(n_img * P, K, seed=41)
    wt = _rand(n_valid, K, seed=42, scale=1 / math.sqrt(K))
    b = _rand(n_valid, seed=43, scale=0.3)
    rows = 32 * ((n_valid + 31) // 32)
    wp = torch.zeros(rows, K)
    wp[:n_valid] = wt
    bp = torch.zeros(rows)
    bp[:n_valid] = b
    ref = (x.double() @ wt.double().t() + b.double()).view(n_img, P, n_valid).permute(0, 2, 1).contiguous()
    out = torch.full((n_img * n_valid * P + 64,), float("nan"), device="cuda")
    xd, wd, wd_scaled, bd = _sp(x), _sp(wp), _sp(wp * 4096.0), bp.cuda()
    L.call("pp_skinny_conv1x1_planar", xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), out.data_ptr(), n_img, P, K, n_valid, 1.0, None)
    got = out[: n_img * n_valid * P].view(n_img, n_valid, P).cpu().double()
    assert torch.isnan(out[n_img * n_valid * P:]).all(), "wrote past the planes"
    torch.testing.assert_close(got, ref, rtol=1e-5, atol=1e-5)
    # the weights stored times 2^12: undone exactly
    out2 = torch.empty_like(out)
    L.call("pp_skinny_conv1x1_planar", xd.data_ptr(), wd_scaled.data_ptr(), bd.data_ptr(), out2.data_ptr(), n_img, P, K, n_valid, 1.0 / 4096.0, None)
    torch.testing.assert_close(out2[: n_img * n_valid * P].cpu().double().view(n_img, n_valid, P), ref, rtol=1e-5, atol=1e-5)


@gpu
def test_graph_eviction_and_recapture_with_the_two_stream_head():
    """A graph cache smaller than the set of recurring batch sizes: every call evicts (destroys) the least recently used graph and captures a new
    one - small batches with the towers on a side stream next to sizes on the row-owner plan. With ONE engine-wide side stream the first replay after
    such an eviction crashed inside hipGraphLaunch (scripts/r06/graph_eager_repro.py); every capture now has a side stream of its own. Results equal
    the kernel-by-kernel launches bit for bit."""
    import random

    from probpose_code_amd import ProbPoseEngine
    from probpose_code_amd import synthetic as S

    sd = S.synthetic_state_dict("small", seed=0, logit_scale=2.0)
    eng = ProbPoseEngine(sd, 12, precision="f16x3")
    eng.max_graphs = 3
    fi = S.COCO_FLIP_INDICES
    sizes = (1, 2, 3, 6, 8, 17, 18, 24)
    crops = {B: S.synthetic_crops(B, seed=70 + B).cuda() for B in sizes}
    want = {}
    for B in sizes:
        out = eng.forward(crops[B], True, fi)
        want[B] = (out["keypoints"].cpu().numpy().copy(), out["scalars"].cpu().numpy().copy())
    rng = random.Random(5)
    for it in range(160):
        B = rng.choice(sizes)
        if rng.random() < 0.25:
            out = eng.forward(crops[B], True, fi)
        else:
            out = eng.forward_graph(crops[B], True, fi)
        assert np.array_equal(out["keypoints"].cpu().numpy(), want[B][0]) and np.array_equal(out["scalars"].cpu().numpy(), want[B][1]), (it, B)
    assert eng.graph_captures > 40 and len(eng._graphs) == 3


@gpu
def test_shape_fuzz_of_the_skinny_kernels():
    """tests/fuzz_skinny.py for a few seconds: pp_skinny_linear / pp_skinny_deconv / pp_skinny_conv1x1_planar over random shapes (M ragged against every
    tile edge), every epilogue and every tile shape against fp64, outputs between canaries, LayerNorm counters back at zero."""
    import subprocess
    import sys

    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, os.path.join(here, "fuzz_skinny.py"), "10"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "SKINNY FUZZ OK" in r.stdout, (r.stdout[-800:], r.stderr[-800:])
