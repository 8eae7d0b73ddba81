"""GPU: every HIP kernel of the network, called through the C ABI, against a plain torch fp32/fp64
CPU reference of the same op (asymmetric random data, so transposes cannot hide)."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

BF16, F32 = 0, 1
TOL = {F32: dict(rtol=2e-5, atol=2e-5), BF16: dict(rtol=2e-2, atol=2e-2)}


def _lib():
    from probpose_code_amd import _lib

    return _lib


def _dt(prec):
    return torch.bfloat16 if prec == BF16 else torch.float32


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def _q(x, prec):
    """Operand as the kernel sees it (bf16-rounded in bf16 mode), back in fp64 for the reference."""
    return x.to(_dt(prec)).double()


@pytest.mark.parametrize("prec", [F32, BF16])
@pytest.mark.parametrize("M,N,K,act,res,bias", [(384, 384, 384, 0, True, True), (256, 1152, 384, 1, False, True),
                                                (200, 136, 1536, 2, False, False), (128, 17, 256, 0, False, True)])
def test_gemm_epilogues(prec, M, N, K, act, res, bias):
    L = _lib()
    a, w = _rand(M, K, seed=1), _rand(N, K, seed=2, scale=1 / math.sqrt(K))
    b = _rand(N, seed=3) if bias else None
    r = _rand(M, N, seed=4) if res else None
    ref = _q(a, prec) @ _q(w, prec).t()
    if bias:
        ref = ref + b.double()
    if act == 1:
        ref = F.gelu(ref)
    elif act == 2:
        ref = F.relu(ref)
    if res:
        ref = ref + r.double()
    ad, wd = a.to(_dt(prec)).cuda(), w.to(_dt(prec)).cuda()
    bd = b.cuda() if bias else None
    ldc = N if N % 4 == 0 else (N + 3) // 4 * 4  # row-major fp32 store needs ldc % 4 == 0: pad the row pitch
    out = r.clone().cuda() if res else torch.full((M, ldc), float("nan"), device="cuda")
    L.call("pp_gemm", prec, ad.data_ptr(), wd.data_ptr(), L.ptr(bd), out.data_ptr() if res else None, 0,
           out.data_ptr(), M, N, K, K, K, ldc, act, 0, 0, None)
    torch.testing.assert_close(out.cpu().double()[:, :N], ref, **TOL[prec])
    if ldc != N:
        assert torch.isnan(out[:, N:]).all(), "columns beyond N must stay untouched"


@pytest.mark.parametrize("prec", [F32, BF16])
def test_gemm_planar_and_posembed(prec):
    """N = 17 planar store (final 1x1 conv) and the res_mod broadcast (pos_embed)."""
    L = _lib()
    nb, P, K, N = 3, 96, 256, 17
    a, w, b = _rand(nb * P, K, seed=5), _rand(N, K, seed=6, scale=0.06), _rand(N, seed=7)
    ref = (_q(a, prec) @ _q(w, prec).t() + b.double()).reshape(nb, P, N).permute(0, 2, 1)
    out = torch.full((nb, N, P), float("nan"), device="cuda")
    ad, wd, bd = a.to(_dt(prec)).cuda(), w.to(_dt(prec)).cuda(), b.cuda()  # keep alive across the async launch
    L.call("pp_gemm", prec, ad.data_ptr(), wd.data_ptr(), bd.data_ptr(), None, 0, out.data_ptr(), nb * P, N, K, K, K, N,
           0, 0, P, None)
    torch.testing.assert_close(out.cpu().double(), ref, **TOL[prec])
    # pos_embed broadcast
    M, E, Np = 4 * 48, 128, 48
    a, w, pe = _rand(M, 64, seed=8), _rand(E, 64, seed=9, scale=0.1), _rand(Np, E, seed=10)
    ref = _q(a, prec) @ _q(w, prec).t() + pe.double().repeat(4, 1)
    out = torch.empty((M, E), device="cuda")
    ad, wd, ped = a.to(_dt(prec)).cuda(), w.to(_dt(prec)).cuda(), pe.cuda()
    L.call("pp_gemm", prec, ad.data_ptr(), wd.data_ptr(), None, ped.data_ptr(), Np, out.data_ptr(), M, E, 64, 64, 64, E,
           0, 0, 0, None)
    torch.testing.assert_close(out.cpu().double(), ref, **TOL[prec])


@pytest.mark.parametrize("prec", [F32, BF16])
def test_conv3x3_grouped_vs_torch(prec):
    L = _lib()
    G, B, H, W, C = 4, 3, 8, 6, 128
    x = _rand(G, B, C, H, W, seed=11)
    w = _rand(G, C, C, 3, 3, seed=12, scale=1 / math.sqrt(9 * C))
    b = _rand(G, C, seed=13)
    ref = torch.stack([F.conv2d(_q(x[g], prec), _q(w[g], prec), b[g].double(), padding=1) for g in range(G)])
    xd = x.permute(0, 1, 3, 4, 2).contiguous().to(_dt(prec)).cuda()  # [G][B,H,W,C]
    wd = w.permute(0, 1, 3, 4, 2).reshape(G, C, 9 * C).contiguous().to(_dt(prec)).cuda()
    out = torch.empty((G, B, H, W, C), device="cuda")
    bd = b.cuda()
    L.call("pp_conv_gemm", prec, 1, xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), out.data_ptr(), B, H, W, C, C,
           0, 0, G, B * H * W * C, C * 9 * C, B * H * W * C, C, C, 0, 0, None)
    torch.testing.assert_close(out.cpu().double().permute(0, 1, 4, 2, 3), ref, **TOL[prec])


@pytest.mark.parametrize("prec", [F32, BF16])
def test_deconv_phases_vs_torch(prec):
    """ConvTranspose2d(k4, s2, p1) as four phase GEMMs, weights packed as weights.pack does."""
    L = _lib()
    B, H, W, Cin, Cout = 2, 8, 6, 128, 64
    x = _rand(B, Cin, H, W, seed=14)
    w = _rand(Cin, Cout, 4, 4, seed=15, scale=1 / math.sqrt(4 * Cin))
    ref = F.relu(F.conv_transpose2d(_q(x, prec), _q(w, prec), None, stride=2, padding=1))
    xd = x.permute(0, 2, 3, 1).contiguous().to(_dt(prec)).cuda()
    out = torch.full((B, 2 * H, 2 * W, Cout), float("nan"), device="cuda")
    for py in range(2):
        for px in range(2):
            ph = torch.empty((Cout, 4 * Cin))
            for ty in range(2):
                for tx in range(2):
                    t = ty * 2 + tx
                    ph[:, t * Cin:(t + 1) * Cin] = w[:, :, 3 - 2 * ty - py, 3 - 2 * tx - px].t()
            pd = ph.to(_dt(prec)).cuda()
            L.call("pp_conv_gemm", prec, 2, xd.data_ptr(), pd.data_ptr(), None, out.data_ptr(), B, H, W, Cin, Cout,
                   py, px, 1, 0, 0, 0, 0, Cout, 2, 0, None)
    torch.testing.assert_close(out.cpu().double().permute(0, 3, 1, 2), ref, **TOL[prec])


def test_panel_deconv_all_phases_vs_torch():
    """The wide-tile kernel (pp_panel_gemm.hip) takes bf16 deconvolutions with Cout % 256 == 0 once there are enough
    tiles to fill the chip: ConvTranspose2d(k4, s2, p1) + bias + ReLU, four phases in one launch, batch not a
    multiple of the 192-row tile so that tail rows and image borders are both exercised."""
    L = _lib()
    B, H, W, Cin, Cout = 51, 16, 12, 128, 256          # M = 9792 = 51 tiles of 192 rows, x 4 phases = 204 tiles
    x = _rand(B, Cin, H, W, seed=31)
    w = _rand(Cin, Cout, 4, 4, seed=32, scale=1 / math.sqrt(4 * Cin))
    b = _rand(Cout, seed=33)
    ref = F.relu(F.conv_transpose2d(_q(x, BF16), _q(w, BF16), b.double(), stride=2, padding=1))
    xd = x.permute(0, 2, 3, 1).contiguous().bfloat16().cuda()
    ph = torch.empty((2, 2, Cout, 4 * Cin))
    for py in range(2):
        for px in range(2):
            for ty in range(2):
                for tx in range(2):
                    t = ty * 2 + tx
                    ph[py, px, :, t * Cin:(t + 1) * Cin] = w[:, :, 3 - 2 * ty - py, 3 - 2 * tx - px].t()
    pd = ph.bfloat16().cuda()
    bd = b.cuda()
    out = torch.full((B, 2 * H, 2 * W, Cout), float("nan"), dtype=torch.bfloat16, device="cuda")
    L.call("pp_conv_gemm", BF16, 2, xd.data_ptr(), pd.data_ptr(), bd.data_ptr(), out.data_ptr(), B, H, W, Cin, Cout,
           -1, 0, 1, 0, 0, 0, 0, Cout, 2, 1, None)
    torch.testing.assert_close(out.cpu().double().permute(0, 3, 1, 2), ref, rtol=2e-2, atol=2e-2)


def test_panel_conv3x3_groups_vs_torch():
    """Wide-tile path for the grouped 3x3 convolution (Cout % 192 == 0), shared input, bias, no activation."""
    L = _lib()
    G, B, H, W, C = 4, 124, 8, 6, 384    # M = 5952 -> 24 tiles of 256 rows (tail: 64) x 2 column tiles x 4 groups = 192
    x = _rand(B, C, H, W, seed=34)
    w = _rand(G, C, C, 3, 3, seed=35, scale=1 / math.sqrt(9 * C))
    b = _rand(G, C, seed=36)
    xq, wq = x.bfloat16().float(), w.bfloat16().float()  # fp32 reference on the bf16-rounded operands
    ref = torch.stack([F.conv2d(xq, wq[g], b[g], padding=1) for g in range(G)]).double()
    xd = x.permute(0, 2, 3, 1).contiguous().bfloat16().cuda()
    wd = w.permute(0, 1, 3, 4, 2).reshape(G, C, 9 * C).contiguous().bfloat16().cuda()
    out = torch.full((G, B, H, W, C), float("nan"), dtype=torch.bfloat16, device="cuda")
    bd = b.cuda()
    L.call("pp_conv_gemm", BF16, 1, xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), out.data_ptr(), B, H, W, C, C,
           0, 0, G, 0, C * 9 * C, B * H * W * C, C, C, 0, 1, None)
    torch.testing.assert_close(out.cpu().double().permute(0, 1, 4, 2, 3), ref, rtol=2e-2, atol=2e-2)


@pytest.mark.parametrize("B,act", [(33, 2), (64, 0)])
def test_halo_conv3x3_groups_vs_torch(B, act):
    """Halo-staged 3x3 convolution (pp_conv_halo.hip: 16 x 12 images, two images x 128 channels per tile, activations
    staged once per 64-channel chunk): four towers on a shared input, bias, ReLU or none; B = 33 leaves the second image
    of the last tile past the batch. Border pixels (every tap that leaves the image) are the point of the check."""
    L = _lib()
    G, H, W, C = 4, 16, 12, 384          # 3 column tiles x 17 / 32 image pairs x 4 groups >= 192 tiles
    x = _rand(B, C, H, W, seed=134)
    w = _rand(G, C, C, 3, 3, seed=135, scale=1 / math.sqrt(9 * C))
    b = _rand(G, C, seed=136)
    xq, wq = x.bfloat16().float(), w.bfloat16().float()
    ref = torch.stack([F.conv2d(xq, wq[g], b[g], padding=1) for g in range(G)]).double()
    if act == 2:
        ref = ref.clamp_min(0)
    xd = x.permute(0, 2, 3, 1).contiguous().bfloat16().cuda()
    wd = w.permute(0, 1, 3, 4, 2).reshape(G, C, 9 * C).contiguous().bfloat16().cuda()
    out = torch.full((G, B + 1, H, W, C), float("nan"), dtype=torch.bfloat16, device="cuda")  # one guard image per group
    bd = b.cuda()
    L.call("pp_conv_gemm", BF16, 1, xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), out.data_ptr(), B, H, W, C, C,
           0, 0, G, 0, C * 9 * C, (B + 1) * H * W * C, C, C, act, 1, None)
    got = out.cpu()
    assert torch.isnan(got[:, B].float()).all(), "rows past the batch were written"
    torch.testing.assert_close(got[:, :B].double().permute(0, 1, 4, 2, 3), ref, rtol=2e-2, atol=2e-2)
    # tighter, against the implicit-GEMM arithmetic: same bf16 products, fp32 accumulation in another order
    err = (got[:, :B].double().permute(0, 1, 4, 2, 3) - ref).abs().max().item()
    assert err < 1.6e-2, err


@pytest.mark.parametrize("B,H,W,ph,pw", [(33, 16, 12, 4, 3), (64, 16, 12, 4, 3), (40, 8, 6, 2, 2)])
def test_conv3x3_maxpool_relu_vs_two_launches(B, H, W, ph, pw):
    """pp_conv3x3_maxpool_relu (first tower stage: conv + folded BN -> MaxPool -> ReLU): the one-launch form on 16 x 12 maps
    (pooling in the epilogue of the halo-staged kernel, odd batch = tail image) and the two-launch route through the
    scratch tensor for another shape, both against torch and BIT-EXACT against pp_conv_gemm + pp_maxpool_relu_nhwc."""
    L = _lib()
    G, C = 4, 384
    x = _rand(B, C, H, W, seed=234)
    w = _rand(G, C, C, 3, 3, seed=235, scale=1 / math.sqrt(9 * C))
    b = _rand(G, C, seed=236)
    xq, wq = x.bfloat16().float(), w.bfloat16().float()
    ref = torch.stack([F.max_pool2d(F.conv2d(xq, wq[g], b[g], padding=1), (ph, pw)).clamp_min(0) for g in range(G)]).double()
    xd = x.permute(0, 2, 3, 1).contiguous().bfloat16().cuda()
    wd = w.permute(0, 1, 3, 4, 2).reshape(G, C, 9 * C).contiguous().bfloat16().cuda()
    bd = b.cuda()
    Ho, Wo = H // ph, W // pw
    pooled = torch.full((G, B, Ho, Wo, C), float("nan"), dtype=torch.bfloat16, device="cuda")
    scratch = torch.zeros((G, B, H, W, C), dtype=torch.bfloat16, device="cuda")
    L.call("pp_conv3x3_maxpool_relu", BF16, xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), pooled.data_ptr(), scratch.data_ptr(),
           B, H, W, C, C, ph, pw, G, 0, C * 9 * C, C, 1, None)
    full = torch.empty((G, B, H, W, C), dtype=torch.bfloat16, device="cuda")
    two = torch.empty_like(pooled)
    L.call("pp_conv_gemm", BF16, 1, xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), full.data_ptr(), B, H, W, C, C,
           0, 0, G, 0, C * 9 * C, B * H * W * C, C, C, 0, 1, None)
    L.call("pp_maxpool_relu_nhwc", full.data_ptr(), 1, two.data_ptr(), 1, G * B, H, W, C, ph, pw, None)
    assert torch.equal(pooled.view(torch.int16), two.view(torch.int16))
    if (H, W) == (16, 12):
        assert not scratch.any(), "the one-launch form must not touch the scratch tensor"
    torch.testing.assert_close(pooled.cpu().double().permute(0, 1, 4, 2, 3), ref, rtol=2e-2, atol=2e-2)


@pytest.mark.parametrize("prec,hd,S", [(F32, 32, 192), (BF16, 32, 192), (BF16, 64, 192), (F32, 64, 192), (BF16, 64, 432), (BF16, 32, 432),
                                       (F32, 32, 432)])
def test_attention_vs_torch(prec, hd, S):
    L = _lib()
    n_seq, heads = 3, 4
    E = heads * hd
    qkv = _rand(n_seq * S, 3 * E, seed=16, scale=1.3)
    qd = qkv.to(_dt(prec)).cuda()
    out = torch.empty((n_seq * S, E), dtype=_dt(prec), device="cuda")
    L.call("pp_attention", prec, qd.data_ptr(), out.data_ptr(), n_seq, S, heads, hd, hd ** -0.5, None)
    x = _q(qkv, prec).reshape(n_seq, S, 3, heads, hd).permute(2, 0, 3, 1, 4)
    att = ((x[0] @ x[1].transpose(-2, -1)) * hd ** -0.5).softmax(-1)
    ref = (att @ x[2]).transpose(1, 2).reshape(n_seq * S, E)
    tol = TOL[prec] if prec == F32 else dict(rtol=3e-2, atol=3e-2)
    torch.testing.assert_close(out.cpu().double(), ref, **tol)


@pytest.mark.parametrize("E", [384, 768])
@pytest.mark.parametrize("out_bf16", [0, 1])
def test_layernorm_vs_torch(E, out_bf16):
    L = _lib()
    M = 203
    x, g, b = _rand(M, E, seed=17, scale=3.0) + 0.7, 1 + 0.1 * _rand(E, seed=18), _rand(E, seed=19)
    y = torch.empty((M, E), dtype=torch.bfloat16 if out_bf16 else torch.float32, device="cuda")
    xd, gd, bd = x.cuda(), g.cuda(), b.cuda()
    L.call("pp_layernorm", xd.data_ptr(), gd.data_ptr(), bd.data_ptr(), y.data_ptr(), M, E, 1e-6, out_bf16, None)
    ref = F.layer_norm(x.double(), (E,), g.double(), b.double(), 1e-6)
    torch.testing.assert_close(y.cpu().double(), ref, **(dict(rtol=1e-2, atol=1e-2) if out_bf16 else dict(rtol=1e-5, atol=1e-5)))


@pytest.mark.parametrize("prec", [F32, BF16])
def test_preproc_im2col_vs_torch(prec):
    """BGR->RGB + normalise + flip copy + zero-padded 16x16 patches == Conv2d(k16,s16,p2) input view."""
    from oracle import model_ref as M

    L = _lib()
    B, H, W = 2, 64, 48
    g = torch.Generator().manual_seed(20)
    img = torch.randint(0, 256, (B, 3, H, W), generator=g, dtype=torch.uint8)
    mean = np.array([123.675, 116.28, 103.53], np.float32)
    std = np.array([58.395, 57.12, 57.375], np.float32)
    Hp, Wp = (H + 4 - 16) // 16 + 1, (W + 4 - 16) // 16 + 1
    out = torch.empty((2 * B * Hp * Wp, 768), dtype=_dt(prec), device="cuda")
    imgd = img.cuda()
    L.call("pp_preproc_im2col", prec, imgd.data_ptr(), 0, out.data_ptr(), B, 2, H, W, 16, 2, mean.ctypes.data,
           std.ctypes.data, 1, None)
    x = M.preprocess(img, mean, std)
    both = torch.cat([x, x.flip(-1)])
    ref = F.unfold(F.pad(both, (2, 2, 2, 2))[:, :, : Hp * 16, : Wp * 16], 16, stride=16)  # (2B, 768, Np)
    ref = ref.transpose(1, 2).reshape(-1, 768)
    if prec == F32:
        assert torch.equal(out.cpu(), ref)  # same fp32 ops: bit-exact
    else:
        assert torch.equal(out.cpu(), ref.to(torch.bfloat16))
    # fp32 input that is already preprocessed (the backbone's stand-alone contract): plain im2col
    xd = x.contiguous().cuda()
    out1 = torch.empty((B * Hp * Wp, 768), dtype=_dt(prec), device="cuda")
    L.call("pp_preproc_im2col", prec, xd.data_ptr(), 1, out1.data_ptr(), B, 1, H, W, 16, 2, None, None, 0, None)
    assert torch.equal(out1.cpu(), ref[: B * Hp * Wp].to(_dt(prec)))


def test_maxpool_relu_and_tower_final():
    L = _lib()
    N, H, W, C = 5, 16, 12, 64
    x = _rand(N, H, W, C, seed=21)
    y = torch.empty((N, 4, 4, C), device="cuda")
    xd = x.cuda()
    L.call("pp_maxpool_relu_nhwc", xd.data_ptr(), 0, y.data_ptr(), 0, N, H, W, C, 4, 3, None)
    ref = F.relu(F.max_pool2d(x.permute(0, 3, 1, 2), (4, 3), (4, 3))).permute(0, 2, 3, 1)
    assert torch.equal(y.cpu(), ref)
    B, K, C = 3, 17, 384
    feat, w, b = _rand(4, 2 * B, C, seed=22), _rand(4, K, C, seed=23, scale=0.05), _rand(4, K, seed=24, scale=0.3)
    fi = [0, 2, 1, 4, 3, 6, 5, 8, 7, 10, 9, 12, 11, 14, 13, 16, 15]
    out = torch.empty((4, B, K), device="cuda")
    fd, wd, bd, fid = feat.cuda(), w.cuda(), b.cuda(), torch.tensor(fi, dtype=torch.int32).cuda()
    L.call("pp_tower_final", fd.data_ptr(), 0, wd.data_ptr(), bd.data_ptr(), fid.data_ptr(), out.data_ptr(), B, 2, C, K,
           1.0, None)
    z = torch.einsum("tbc,tkc->tbk", feat.double(), w.double()) + b.double()[:, None]
    a = torch.cat([torch.sigmoid(z[:3]), F.relu(z[3:])])
    ref = (a[:, :B] + a[:, B:][:, :, fi]) * 0.5
    torch.testing.assert_close(out.cpu().double(), ref, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("hw,B", [((64, 48), 16), ((96, 72), 3)])
@pytest.mark.parametrize("flip", [True, False])
def test_fused_sparsemax_decode_vs_oracle(hw, B, flip):
    """pp_probmap_head_decode: logits -> /T -> Sparsemax -> clamp -> flip-average -> decode, vs the oracle's
    sort-based Sparsemax + per-sample decode. Heatmaps to 2e-6 (tau is computed in fp64 here and by an
    fp32 cumulative sum there), keypoints compared where the argmax agrees."""
    from oracle import decode_ref as D
    from oracle import model_ref as M
    from probpose_code_amd import _lib as L
    from probpose_code_amd.codecs import oks_kernel_taps

    H, W = hw
    K = 17
    g = torch.Generator().manual_seed(31)
    smooth = F.interpolate(torch.randn(2 * B, K, H // 4, W // 4, generator=g), size=(H, W), mode="bicubic")
    logits = (2.5 * smooth + 0.3 * torch.randn(2 * B, K, H, W, generator=g)).contiguous()
    probs = torch.clamp(M.sparsemax(logits.reshape(2 * B, K, -1) / 0.5) * 1.0, 0, 1).reshape(2 * B, K, H, W).numpy()
    avg = D.tta_average(probs[:B], probs[B:]) if flip else probs[:B]
    taps, radius = oks_kernel_taps(K, H, W)
    ld = logits.cuda()
    fi = torch.tensor(D.COCO_FLIP_INDICES, dtype=torch.int32).cuda()
    hm = torch.empty((B, K, H, W), device="cuda")
    locs = torch.empty((B, K, 2), device="cuda")
    kp = torch.empty((B, K, 2), dtype=torch.float64, device="cuda")
    sc = torch.empty((B, K), device="cuda")
    in_w, in_h = (192.0, 256.0) if H == 64 else (288.0, 384.0)
    td, rd = torch.from_numpy(taps).cuda(), torch.from_numpy(radius).cuda()
    L.call("pp_probmap_head_decode", ld.data_ptr(), ld[B:].data_ptr() if flip else None, fi.data_ptr() if flip else None,
           td.data_ptr(), rd.data_ptr(), B, K, H, W, in_w, in_h, 0.5, 1.0, hm.data_ptr(), None, locs.data_ptr(),
           kp.data_ptr(), sc.data_ptr(), None)
    hm = hm.cpu().numpy()
    assert np.abs(hm - avg).max() < 2e-6
    assert np.allclose(hm.reshape(B, K, -1).sum(-1), 1.0, atol=1e-5)  # rows stay on the simplex
    n_same = 0
    for b in range(B):
        k_ref, s_ref = D.probmap_decode(avg[b], (int(in_w), int(in_h)), (W, H))
        k_self, s_self = D.probmap_decode(hm[b], (int(in_w), int(in_h)), (W, H))
        # the decode of the kernel's own map must be reproduced exactly (decode stage is bit-exact)
        assert np.array_equal(kp[b].cpu().numpy()[None], k_self) and np.array_equal(sc[b].cpu().numpy()[None], s_self)
        same = np.abs(k_ref - k_self).max(-1) < 0.5
        n_same += same.sum()
        assert np.abs(k_ref - k_self)[same].max() < 1e-3
    assert n_same >= 0.98 * B * K  # argmax flips only on near-ties


def test_fused_sparsemax_decode_shift_heatmap_both_layouts():
    """pp_probmap_decode_flags with PP_DECODE_SHIFT_HEATMAP on logits (tta.py:64-66 inside the fused Sparsemax + flip merge + decode):
    the averaged map against the oracle's shifted flip-back, and the phase-separated layout bit-identical to the planar one."""
    from oracle import decode_ref as D
    from oracle import model_ref as M
    from probpose_code_amd import _lib as L
    from probpose_code_amd.codecs import oks_kernel_taps

    B, K, H, W = 6, 17, 64, 48
    g = torch.Generator().manual_seed(33)
    smooth = F.interpolate(torch.randn(2 * B, K, H // 4, W // 4, generator=g), size=(H, W), mode="bicubic")
    logits = (2.5 * smooth + 0.3 * torch.randn(2 * B, K, H, W, generator=g)).contiguous()
    probs = torch.clamp(M.sparsemax(logits.reshape(2 * B, K, -1) / 0.5) * 1.0, 0, 1).reshape(2 * B, K, H, W).numpy()
    avg = D.tta_average(probs[:B], probs[B:], shift_heatmap=True)
    taps, radius = oks_kernel_taps(K, H, W)
    td, rd = torch.from_numpy(taps).cuda(), torch.from_numpy(radius).cuda()
    fi = torch.tensor(D.COCO_FLIP_INDICES, dtype=torch.int32).cuda()
    planar = logits.cuda()
    phased = planar.reshape(2 * B, K, H // 2, 2, W // 2, 2).permute(0, 1, 3, 5, 2, 4).contiguous()  # (py, px) blocks of (H/2, W/2)
    outs = []
    for flags, src in ((1 | 4, planar), (1 | 2 | 4, phased)):
        hm = torch.empty((B, K, H, W), device="cuda")
        locs = torch.empty((B, K, 2), device="cuda")
        kp = torch.empty((B, K, 2), dtype=torch.float64, device="cuda")
        sc = torch.empty((B, K), device="cuda")
        L.call("pp_probmap_decode_flags", src.data_ptr(), src[B:].data_ptr(), fi.data_ptr(), td.data_ptr(), rd.data_ptr(), B, K, H, W, 192.0, 256.0,
               0.5, 1.0, hm.data_ptr(), None, locs.data_ptr(), kp.data_ptr(), sc.data_ptr(), flags, None)
        outs.append((hm.cpu(), locs.cpu(), kp.cpu(), sc.cpu()))
    assert np.abs(outs[0][0].numpy() - avg).max() < 2e-6
    assert np.abs(outs[0][0].numpy() - D.tta_average(probs[:B], probs[B:])).max() > 1e-3, "the shift did nothing"
    for a, c in zip(*outs):
        assert torch.equal(a, c)
    with pytest.raises(L.ProbPoseLibraryError):  # PP_DECODE_PHASED without PP_DECODE_LOGITS
        L.call("pp_probmap_decode_flags", planar.data_ptr(), None, None, td.data_ptr(), rd.data_ptr(), B, K, H, W, 192.0, 256.0, 0.5, 1.0, None, None,
               locs.data_ptr(), kp.data_ptr(), sc.data_ptr(), 2, None)


@pytest.mark.parametrize("prec", [F32, BF16])
@pytest.mark.parametrize("M,K,res_mod,E", [(192, 384, 0, 384), (480, 1536, 0, 384), (384, 768, 192, 384),
                                           (250, 768, 0, 768), (448, 3072, 0, 768), (864, 768, 432, 768)])
def test_gemm_residual_layernorm_fused(prec, M, K, res_mod, E):
    """x <- x + a W^T + b ; h <- LN(x): fused kernel vs torch, incl. the pos_embed-style broadcast residual,
    a row count that is not a multiple of the row tile, and act aliasing h_out. E = 384 (96-row tiles) and E = 768 (ViT-B:
    112-row tiles, the columns as two halves of 384 through the K-loop)."""
    L = _lib()
    a, w, b = _rand(M, K, seed=41), _rand(E, K, seed=42, scale=1 / math.sqrt(K)), _rand(E, seed=43)
    res = _rand(res_mod if res_mod else M, E, seed=44, scale=2.0)
    g, be = 1 + 0.1 * _rand(E, seed=45), _rand(E, seed=46)
    xref = _q(a, prec) @ _q(w, prec).t() + b.double() + (res.double().repeat(M // res_mod, 1) if res_mod else res.double())
    href = F.layer_norm(xref, (E,), g.double(), be.double(), 1e-6)
    ad, wd, bd, gd, bed = a.to(_dt(prec)).cuda(), w.to(_dt(prec)).cuda(), b.cuda(), g.cuda(), be.cuda()
    if res_mod:
        resd, x = res.cuda(), torch.full((M, E), float("nan"), device="cuda")
    else:
        x = res.clone().cuda()
        resd = x  # in-place residual stream
    alias = (K == E)  # h_out may alias act when shapes agree (proj: act = attention output buffer)
    h = ad if alias else torch.empty((M, E), dtype=_dt(prec), device="cuda")
    L.call("pp_gemm_residual_layernorm", prec, ad.data_ptr(), wd.data_ptr(), bd.data_ptr(), resd.data_ptr(), res_mod,
           x.data_ptr(), gd.data_ptr(), bed.data_ptr(), 1e-6, h.data_ptr(), int(prec == BF16), M, E, K, K, K, None)
    torch.testing.assert_close(x.cpu().double(), xref, **TOL[prec])
    torch.testing.assert_close(h.cpu().double(), href, **(TOL[prec] if prec == F32 else dict(rtol=3e-2, atol=3e-2)))


@pytest.mark.parametrize("M", [96, 480])
def test_fused_mlp_residual_layernorm(M):
    """x <- x + GELU(h W1^T + b1) W2^T + b2 ; h' <- LN(x): fused bf16 kernel vs torch (fp64 on bf16-rounded operands,
    hidden activation rounded to bf16 as the kernel does)."""
    L = _lib()
    E, Fd = 384, 1536
    h, w1, b1 = _rand(M, E, seed=51), _rand(Fd, E, seed=52, scale=1 / math.sqrt(E)), _rand(Fd, seed=53, scale=0.3)
    w2, b2 = _rand(E, Fd, seed=54, scale=1 / math.sqrt(Fd)), _rand(E, seed=55, scale=0.3)
    x0, g, be = _rand(M, E, seed=56, scale=2.0), 1 + 0.1 * _rand(E, seed=57), _rand(E, seed=58)
    hid = F.gelu(_q(h, BF16) @ _q(w1, BF16).t() + b1.double())
    xref = x0.double() + _q(hid.float(), BF16) @ _q(w2, BF16).t() + b2.double()
    href = F.layer_norm(xref, (E,), g.double(), be.double(), 1e-6)
    hd, w1d, w2d = h.bfloat16().cuda(), w1.bfloat16().cuda(), w2.bfloat16().cuda()
    b1d, b2d, gd, bed, x = b1.cuda(), b2.cuda(), g.cuda(), be.cuda(), x0.clone().cuda()
    L.call("pp_mlp_residual_layernorm", hd.data_ptr(), w1d.data_ptr(), b1d.data_ptr(), w2d.data_ptr(), b2d.data_ptr(),
           x.data_ptr(), x.data_ptr(), gd.data_ptr(), bed.data_ptr(), 1e-6, hd.data_ptr(), M, E, Fd, None)
    torch.testing.assert_close(x.cpu().double(), xref, rtol=2e-2, atol=2e-2)
    torch.testing.assert_close(hd.cpu().double(), href, rtol=3e-2, atol=3e-2)


@pytest.mark.parametrize("M,with_qkv", [(96, False), (480, False), (96, True), (480, True)])
def test_fused_proj_mlp_residual_layernorm(M, with_qkv):
    """Second half of a ViT layer in one launch: x1 = x + a Wp^T + bp; h = LN2(x1); x2 = x1 + FFN(h); h' = LN(x2),
    optionally followed by the next layer's qkv = h' Wq^T + bq. Reference in fp64 on the bf16-rounded operands, with h,
    h' and the hidden activation rounded to bf16 as the kernel does."""
    L = _lib()
    E, Fd = 384, 1536
    a, wp, bp = _rand(M, E, seed=61), _rand(E, E, seed=62, scale=1 / math.sqrt(E)), _rand(E, seed=63, scale=0.3)
    w1, b1 = _rand(Fd, E, seed=64, scale=1 / math.sqrt(E)), _rand(Fd, seed=65, scale=0.3)
    w2, b2 = _rand(E, Fd, seed=66, scale=1 / math.sqrt(Fd)), _rand(E, seed=67, scale=0.3)
    wq, bq = _rand(3 * E, E, seed=73, scale=1 / math.sqrt(E)), _rand(3 * E, seed=74, scale=0.3)
    x0 = _rand(M, E, seed=68, scale=2.0)
    g2, be2 = 1 + 0.1 * _rand(E, seed=69), _rand(E, seed=70)
    g, be = 1 + 0.1 * _rand(E, seed=71), _rand(E, seed=72)
    x1 = x0.double() + _q(a, BF16) @ _q(wp, BF16).t() + bp.double()
    h = F.layer_norm(x1, (E,), g2.double(), be2.double(), 1e-6)
    hid = F.gelu(_q(h.float(), BF16) @ _q(w1, BF16).t() + b1.double())
    xref = x1 + _q(hid.float(), BF16) @ _q(w2, BF16).t() + b2.double()
    href = F.layer_norm(xref, (E,), g.double(), be.double(), 1e-6)
    qref = _q(href.float(), BF16) @ _q(wq, BF16).t() + bq.double()
    ad, wpd, w1d, w2d = a.bfloat16().cuda(), wp.bfloat16().cuda(), w1.bfloat16().cuda(), w2.bfloat16().cuda()
    wqd, bqd = wq.bfloat16().cuda(), bq.cuda()
    bpd, b1d, b2d, g2d, be2d, gd, bed = bp.cuda(), b1.cuda(), b2.cuda(), g2.cuda(), be2.cuda(), g.cuda(), be.cuda()
    x = x0.clone().cuda()
    hout = torch.empty(M, E, dtype=torch.bfloat16, device="cuda")
    qout = torch.full((M, 3 * E), float("nan"), dtype=torch.bfloat16, device="cuda")
    L.call("pp_proj_mlp_residual_layernorm", ad.data_ptr(), wpd.data_ptr(), bpd.data_ptr(), x.data_ptr(), g2d.data_ptr(),
           be2d.data_ptr(), w1d.data_ptr(), b1d.data_ptr(), w2d.data_ptr(), b2d.data_ptr(), x.data_ptr(), gd.data_ptr(),
           bed.data_ptr(), 1e-6, hout.data_ptr(), wqd.data_ptr() if with_qkv else None, bqd.data_ptr() if with_qkv else None,
           qout.data_ptr() if with_qkv else None, M, E, Fd, None)
    torch.testing.assert_close(x.cpu().double(), xref, rtol=2e-2, atol=3e-2)
    torch.testing.assert_close(hout.cpu().double(), href, rtol=3e-2, atol=3e-2)
    if with_qkv:
        torch.testing.assert_close(qout.cpu().double(), qref, rtol=3e-2, atol=6e-2)


@pytest.mark.parametrize("prec", [F32, BF16])
def test_conv3x3_splitk_sum_maxpool_vs_torch(prec):
    """Split-K tower convolution + (reduce, bias, MaxPool, ReLU) against torch."""
    L = _lib()
    G, B, H, W, C = 4, 3, 4, 4, 128
    x = _rand(G, B, C, H, W, seed=81)
    w = _rand(G, C, C, 3, 3, seed=82, scale=1 / math.sqrt(9 * C))
    b = _rand(G, C, seed=83)
    ref = torch.stack([F.relu(F.max_pool2d(F.conv2d(_q(x[g], prec), _q(w[g], prec), b[g].double(), padding=1), (2, 2)))
                       for g in range(G)])
    xd = x.permute(0, 1, 3, 4, 2).contiguous().to(_dt(prec)).cuda()
    wd = w.permute(0, 1, 3, 4, 2).reshape(G, C, 9 * C).contiguous().to(_dt(prec)).cuda()
    for ks in (3, 9):
        part = torch.full((ks, G, B, H, W, C), float("nan"), device="cuda")
        L.call("pp_conv3x3_splitk", prec, xd.data_ptr(), wd.data_ptr(), part.data_ptr(), B, H, W, C, C, G, B * H * W * C, C * 9 * C, ks,
               None)
        out = torch.empty((G, B, H // 2, W // 2, C), device="cuda")
        L.call("pp_sum_maxpool_relu_nhwc", part.data_ptr(), ks, G * B * H * W * C, b.cuda().data_ptr(), B, out.data_ptr(), 0, G * B, H, W, C,
               2, 2, None)
        torch.testing.assert_close(out.cpu().double().permute(0, 1, 4, 2, 3), ref, **TOL[prec])


def test_conv3x3_splitk_wide_tiles_vs_torch():
    """The tower stage shape of the bs64 path (4 towers x 128 images x 4 x 4 pixels x 384 channels, three K slices):
    192 tiles of 256 x 192 -> the persistent wide-tile kernel writes the fp32 partial sums; ragged variant (100 images)
    stays on the 128 x 128 kernel. Both against torch."""
    L = _lib()
    G, H, W, C, ks = 4, 4, 4, 384, 3
    for B in (128, 100):
        x = _rand(G, B, C, H, W, seed=84)
        w = _rand(G, C, C, 3, 3, seed=85, scale=1 / math.sqrt(9 * C))
        b = _rand(G, C, seed=86)
        ref = torch.stack([F.relu(F.max_pool2d(F.conv2d(_q(x[g], BF16), _q(w[g], BF16), b[g].double(), padding=1), (2, 2)))
                           for g in range(G)])
        xd = x.permute(0, 1, 3, 4, 2).contiguous().bfloat16().cuda()
        wd = w.permute(0, 1, 3, 4, 2).reshape(G, C, 9 * C).contiguous().bfloat16().cuda()
        part = torch.full((ks, G, B, H, W, C), float("nan"), device="cuda")
        L.call("pp_conv3x3_splitk", BF16, xd.data_ptr(), wd.data_ptr(), part.data_ptr(), B, H, W, C, C, G, B * H * W * C, C * 9 * C, ks, None)
        assert not torch.isnan(part).any()
        bd = b.cuda()
        out = torch.empty((G, B, H // 2, W // 2, C), device="cuda")
        L.call("pp_sum_maxpool_relu_nhwc", part.data_ptr(), ks, G * B * H * W * C, bd.data_ptr(), B, out.data_ptr(), 0, G * B, H, W, C, 2, 2, None)
        torch.testing.assert_close(out.cpu().double().permute(0, 1, 4, 2, 3), ref, **TOL[BF16])


def test_deconv_head_and_phased_decode_vs_unfused():
    """Last deconvolution fused with the 1x1 conv (pp_deconv_head) -> phase-separated logits; decoding them with
    pp_probmap_head_decode_phased gives bit-identical results to decoding the same logits rearranged to planar layout, and
    the logits match ConvTranspose2d + ReLU + Conv1x1 in torch."""
    L = _lib()
    B, H, W, Cin, Cout, K = 6, 32, 24, 128, 256, 17
    x = _rand(B, Cin, H, W, seed=91)
    w = _rand(Cin, Cout, 4, 4, seed=92, scale=1 / math.sqrt(4 * Cin))
    b = _rand(Cout, seed=93, scale=0.2)
    wf, bf = _rand(K, Cout, seed=94, scale=4 / math.sqrt(Cout)), _rand(K, seed=95)
    mid = F.relu(F.conv_transpose2d(_q(x, BF16), _q(w, BF16), b.double(), stride=2, padding=1))
    ref = F.conv2d(_q(mid.float(), BF16), _q(wf, BF16)[:, :, None, None], bf.double())     # (B, K, 2H, 2W)
    xd = x.permute(0, 2, 3, 1).contiguous().bfloat16().cuda()
    ph = torch.empty((2, 2, Cout, 4 * Cin))
    for py in range(2):
        for px in range(2):
            for ty in range(2):
                for tx in range(2):
                    t = ty * 2 + tx
                    ph[py, px, :, t * Cin:(t + 1) * Cin] = w[:, :, 3 - 2 * ty - py, 3 - 2 * tx - px].t()
    wpad = torch.zeros(32, Cout)
    wpad[:K] = wf
    lg = torch.full((B, K, 4, H * W), float("nan"), device="cuda")
    phd, bd, wpd, bfd = ph.bfloat16().cuda(), b.cuda(), wpad.bfloat16().cuda(), bf.cuda()  # (held: the launch is asynchronous)
    L.call("pp_deconv_head", xd.data_ptr(), phd.data_ptr(), bd.data_ptr(), wpd.data_ptr(), bfd.data_ptr(), lg.data_ptr(), B, H, W, Cin,
           Cout, K, None)
    planar = lg.reshape(B, K, 2, 2, H, W).permute(0, 1, 4, 2, 5, 3).reshape(B, K, 2 * H, 2 * W).contiguous()
    torch.testing.assert_close(planar.cpu().double(), ref, rtol=3e-2, atol=5e-2)
    # decode both layouts (first half of the batch un-flipped, second half as its flip partner)
    from probpose_code_amd.codecs import oks_kernel_taps

    taps, radius = oks_kernel_taps(K, 2 * H, 2 * W)
    td, rd = torch.from_numpy(taps).cuda(), torch.from_numpy(radius).cuda()
    fi = torch.tensor([0, 2, 1, 4, 3, 6, 5, 8, 7, 10, 9, 12, 11, 14, 13, 16, 15], dtype=torch.int32, device="cuda")
    outs = []
    for name, src in (("pp_probmap_head_decode_phased", lg), ("pp_probmap_head_decode", planar)):
        hm = torch.empty((3, K, 2 * H, 2 * W), device="cuda")
        locs = torch.empty((3, K, 2), device="cuda")
        kp = torch.empty((3, K, 2), dtype=torch.float64, device="cuda")
        sc = torch.empty((3, K), device="cuda")
        L.call(name, src.data_ptr(), src[3:].data_ptr(), fi.data_ptr(), td.data_ptr(), rd.data_ptr(), 3, K, 2 * H, 2 * W, 192.0, 256.0,
               0.5, 1.0, hm.data_ptr(), None, locs.data_ptr(), kp.data_ptr(), sc.data_ptr(), None)
        outs.append((hm.cpu(), locs.cpu(), kp.cpu(), sc.cpu()))
    for a, c in zip(*outs):
        assert torch.equal(a, c)


@pytest.mark.parametrize("n_seq,with_qkv", [(1, True), (3, False), (3, True)])
def test_vit_layer_one_launch(n_seq, with_qkv):
    """pp_vit_layer = attention + projection + ln2 + FFN + LayerNorm (+ next qkv) in one launch, against the chain of the
    separate kernels' torch reference (fp64 on bf16-rounded operands, intermediates rounded to bf16 where the kernel does)."""
    L = _lib()
    E, Fd, S, heads, hd = 384, 1536, 192, 12, 32
    M = n_seq * S
    qkv = _rand(M, 3 * E, seed=101, scale=1.2)
    wp, bp = _rand(E, E, seed=102, scale=1 / math.sqrt(E)), _rand(E, seed=103, scale=0.3)
    w1, b1 = _rand(Fd, E, seed=104, scale=1 / math.sqrt(E)), _rand(Fd, seed=105, scale=0.3)
    w2, b2 = _rand(E, Fd, seed=106, scale=1 / math.sqrt(Fd)), _rand(E, seed=107, scale=0.3)
    wq, bq = _rand(3 * E, E, seed=108, scale=1 / math.sqrt(E)), _rand(3 * E, seed=109, scale=0.3)
    x0 = _rand(M, E, seed=110, scale=2.0)
    g2, be2 = 1 + 0.1 * _rand(E, seed=111), _rand(E, seed=112)
    g, be = 1 + 0.1 * _rand(E, seed=113), _rand(E, seed=114)
    t = _q(qkv, BF16).reshape(n_seq, S, 3, heads, hd).permute(2, 0, 3, 1, 4)
    att = ((t[0] @ t[1].transpose(-2, -1)) * hd ** -0.5).softmax(-1)
    a = (att @ t[2]).transpose(1, 2).reshape(M, E)
    x1 = x0.double() + _q(a.float(), BF16) @ _q(wp, BF16).t() + bp.double()
    h = F.layer_norm(x1, (E,), g2.double(), be2.double(), 1e-6)
    hid = F.gelu(_q(h.float(), BF16) @ _q(w1, BF16).t() + b1.double())
    xref = x1 + _q(hid.float(), BF16) @ _q(w2, BF16).t() + b2.double()
    href = F.layer_norm(xref, (E,), g.double(), be.double(), 1e-6)
    qref = _q(href.float(), BF16) @ _q(wq, BF16).t() + bq.double()
    d = lambda v, bf=False: (v.bfloat16() if bf else v).cuda()  # noqa: E731
    qd, wpd, w1d, w2d, wqd = d(qkv, True), d(wp, True), d(w1, True), d(w2, True), d(wq, True)
    bpd, b1d, b2d, bqd, g2d, be2d, gd, bed = d(bp), d(b1), d(b2), d(bq), d(g2), d(be2), d(g), d(be)
    x = x0.clone().cuda()
    hout = torch.empty(M, E, dtype=torch.bfloat16, device="cuda")
    qout = torch.full((M, 3 * E), float("nan"), dtype=torch.bfloat16, device="cuda")
    L.call("pp_vit_layer", qd.data_ptr(), S, heads, hd ** -0.5, wpd.data_ptr(), bpd.data_ptr(), x.data_ptr(), g2d.data_ptr(),
           be2d.data_ptr(), w1d.data_ptr(), b1d.data_ptr(), w2d.data_ptr(), b2d.data_ptr(), x.data_ptr(), gd.data_ptr(), bed.data_ptr(),
           1e-6, hout.data_ptr(), wqd.data_ptr() if with_qkv else None, bqd.data_ptr() if with_qkv else None,
           qout.data_ptr() if with_qkv else None, M, E, Fd, None)
    torch.testing.assert_close(x.cpu().double(), xref, rtol=2e-2, atol=4e-2)
    torch.testing.assert_close(hout.cpu().double(), href, rtol=3e-2, atol=4e-2)
    if with_qkv:
        torch.testing.assert_close(qout.cpu().double(), qref, rtol=3e-2, atol=8e-2)


@pytest.mark.parametrize("out_bf16,act,res", [(0, 0, True), (1, 1, False), (0, 2, False)])
def test_wide_tile_bf16_linear_long_k(out_bf16, act, res):
    """bf16 Linear layers with K >= 1536 and enough 256 x 192 tiles go to the wide-tile kernel (pp_panel_split.hip,
    SPLIT = false): fp32 output with an fp32 residual (ViT-B fc2), bf16 output with GELU, tail rows."""
    L = _lib()
    M, N, K = 96 * 256 + 40, 384, 1536
    a, w, b = _rand(M, K, seed=51), _rand(N, K, seed=52, scale=1 / math.sqrt(K)), _rand(N, seed=53)
    r = _rand(M, N, seed=54) if res else None
    ref = _q(a, BF16) @ _q(w, BF16).t() + b.double()
    ref = F.gelu(ref) if act == 1 else (F.relu(ref) if act == 2 else ref)
    if res:
        ref = ref + r.double()
    ad, wd, bd = a.bfloat16().cuda(), w.bfloat16().cuda(), b.cuda()
    out = r.clone().cuda() if res else torch.full((M, N), float("nan"), dtype=torch.bfloat16 if out_bf16 else torch.float32, device="cuda")
    L.call("pp_gemm", BF16, ad.data_ptr(), wd.data_ptr(), bd.data_ptr(), out.data_ptr() if res else None, 0, out.data_ptr(), M, N, K, K, K, N,
           act, out_bf16, 0, None)
    torch.testing.assert_close(out.cpu().double(), ref, **(dict(rtol=2e-2, atol=2e-2) if out_bf16 else dict(rtol=2e-3, atol=2e-3)))


def test_clock_probe_reports_a_plausible_shader_clock():
    """pp_clock_probe (bench hygiene): one sleeping wavefront brackets wall time with s_memtime / s_memrealtime; it stops at the
    host's flag (pinned memory, plain store) or at its time limit, and cycles / ticks * 100 is a shader clock in MHz."""
    import time

    L = _lib()
    words = torch.zeros(3, dtype=torch.int64).pin_memory()
    side = torch.cuda.Stream()
    base = words.data_ptr()
    L.call("pp_clock_probe", base, base + 16, 2_000_000, side.cuda_stream)
    t0 = time.perf_counter()
    time.sleep(0.02)
    words[2] = 1
    side.synchronize()
    assert time.perf_counter() - t0 < 1.0, "the probe ignored the stop flag"
    cyc, ticks = int(words[0]), int(words[1])
    assert 1_500_000 <= ticks <= 60_000_000, ticks          # 15 ms .. 0.6 s of the 100 MHz counter
    assert 50.0 <= cyc / ticks * 100.0 <= 3000.0, cyc / ticks * 100.0
    words.zero_()
    L.call("pp_clock_probe", base, None, 5_000, side.cuda_stream)  # no flag: runs to its limit (5 ms)
    side.synchronize()
    assert 400_000 <= int(words[1]) <= 5_000_000
    with pytest.raises(L.ProbPoseLibraryError):
        L.call("pp_clock_probe", None, None, 1000, None)


def test_shape_fuzz_of_pp_gemm():
    """tests/fuzz_gemm.py for a few seconds: pp_gemm in the three precisions over random shapes (the dispatcher's kernels: 128 x 128, wide tiles,
    twelve-wave Linear), every epilogue and output format against fp64, outputs between canaries."""
    import os
    import subprocess
    import sys

    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, os.path.join(here, "fuzz_gemm.py"), "10"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "GEMM FUZZ OK" in r.stdout, (r.stdout[-800:], r.stderr[-800:])
