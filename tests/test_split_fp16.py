"""PP_PREC_F16X3 (split-fp16 operands, three fp16 MFMAs per product): the host-side format helpers on CPU and every
kernel of the mode through the C ABI on the GPU, against a torch fp64 reference of the UNROUNDED fp32 inputs - the
mode's claim is fp32-class accuracy (operand error ~2^-23), so the tolerances are the fp32 mode's, not bf16's."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

F16X3, SPLIT = 2, 2   # PP_PREC_F16X3, PP_OUT_SPLIT
TOL = dict(rtol=2e-5, atol=2e-5)


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def test_split_format_roundtrip_and_layout():
    """to_split: hi = fp16(x), lo = fp16(x - hi), blocks of 32 elements [32 hi | 32 lo]; hi + lo ~ x to fp32's own step."""
    from probpose_code_amd.weights import from_split, to_split

    x = _rand(3, 5, 96, seed=1) * torch.logspace(-4, 2, 96)  # seven decades: subnormal low halves included
    c = to_split(x)
    assert c.dtype == torch.float32 and c.shape == x.shape
    back = from_split(c)
    assert (back - x).abs().max() <= 2.0 ** -22 * x.abs().max()
    assert ((back - x).abs() <= 2.0 ** -22 * x.abs() + 2.0 ** -25).all()  # relative, down to the fp16 subnormal step
    raw = c.view(torch.float16).reshape(3, 5, 3, 2, 32)
    assert torch.equal(raw[..., 0, :].reshape(3, 5, 96), x.half())
    assert torch.equal(raw[..., 1, :].reshape(3, 5, 96), (x - x.half().float()).half())
    with pytest.raises(AssertionError):
        to_split(torch.zeros(4, 48))


# ------------------------------------------------------------------------------------------------- GPU
gpu = pytest.mark.gpu


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
@pytest.mark.parametrize("M,N,K,act,res,bias,split_out", [(384, 384, 384, 0, True, True, 0), (256, 1152, 384, 1, False, True, 1),
                                                          (200, 160, 1536, 2, False, False, 1), (128, 17, 256, 0, False, True, 0),
                                                          (130, 96, 64, 0, True, True, 1)])
def test_gemm_split(M, N, K, act, res, bias, split_out):
    L = _lib()
    a, w = _rand(M, K, seed=1), _rand(N, K, seed=2, scale=1 / math.sqrt(K))
    b = _rand(N, seed=3) if bias else None
    r = _rand(M, N, seed=4) if res else None
    ref = a.double() @ w.double().t()
    if bias:
        ref = ref + b.double()
    if act == 1:
        ref = F.gelu(ref)
    elif act == 2:
        ref = F.relu(ref)
    if res:
        ref = ref + r.double()
    ad, wd = _sp(a), _sp(w)
    bd = b.cuda() if bias else None
    rd = r.cuda() if res else None
    ldc = N if N % 4 == 0 else (N + 3) // 4 * 4
    out = torch.full((M, ldc), float("nan"), device="cuda")
    L.call("pp_gemm", F16X3, ad.data_ptr(), wd.data_ptr(), L.ptr(bd), L.ptr(rd), 0, out.data_ptr(), M, N, K, K, K, ldc, act,
           SPLIT if split_out else 0, 0, None)
    got = _unsp(out) if split_out else out.cpu().double()[:, :N]
    torch.testing.assert_close(got, ref, **TOL)


@gpu
def test_gemm_split_planar_posembed_and_errors():
    L = _lib()
    nb, P, K, N = 3, 96, 256, 17
    a, w, b = _rand(nb * P, K, seed=5), _rand(N, K, seed=6, scale=0.06), _rand(N, seed=7)
    ref = (a.double() @ w.double().t() + b.double()).reshape(nb, P, N).permute(0, 2, 1)
    out = torch.full((nb, N, P), float("nan"), device="cuda")
    ad, wd, bd = _sp(a), _sp(w), b.cuda()
    L.call("pp_gemm", F16X3, ad.data_ptr(), wd.data_ptr(), bd.data_ptr(), None, 0, out.data_ptr(), nb * P, N, K, K, K, N,
           0, 0, P, None)
    torch.testing.assert_close(out.cpu().double(), ref, **TOL)
    M, E, Np = 4 * 48, 128, 48
    a, w, pe = _rand(M, 64, seed=8), _rand(E, 64, seed=9, scale=0.1), _rand(Np, E, seed=10)
    ref = a.double() @ w.double().t() + pe.double().repeat(4, 1)
    out = torch.empty((M, E), device="cuda")
    ad, wd, ped = _sp(a), _sp(w), pe.cuda()
    L.call("pp_gemm", F16X3, ad.data_ptr(), wd.data_ptr(), None, ped.data_ptr(), Np, out.data_ptr(), M, E, 64, 64, 64, E,
           0, 0, 0, None)
    torch.testing.assert_close(out.cpu().double(), ref, **TOL)
    # split output with N % 32 != 0, and a bf16 output in this mode, are refused - not silently mis-stored
    with pytest.raises(L.ProbPoseLibraryError):
        L.call("pp_gemm", F16X3, ad.data_ptr(), wd.data_ptr(), None, None, 0, out.data_ptr(), M, 72, 64, 64, 64, 72, 0, SPLIT, 0, None)
    with pytest.raises(L.ProbPoseLibraryError):
        L.call("pp_gemm", F16X3, ad.data_ptr(), wd.data_ptr(), None, None, 0, out.data_ptr(), M, E, 64, 64, 64, E, 0, 1, 0, None)


@gpu
@pytest.mark.parametrize("K,res_mod,E", [(768, 48, 384), (384, 0, 384), (1536, 0, 384), (768, 0, 768), (3072, 0, 768), (768, 48, 768)])
def test_gemm_residual_layernorm_split(K, res_mod, E):
    L = _lib()
    M = 96 * 3 - 40
    a, w, b = _rand(M, K, seed=11), _rand(E, K, seed=12, scale=1 / math.sqrt(K)), _rand(E, seed=13, scale=0.1)
    r = _rand(res_mod if res_mod else M, E, seed=14)
    g, be = 1 + 0.1 * _rand(E, seed=15), _rand(E, seed=16, scale=0.1)
    x_ref = a.double() @ w.double().t() + b.double() + (r.double().repeat(M // res_mod + 1, 1)[:M] if res_mod else r.double())
    h_ref = F.layer_norm(x_ref, (E,), g.double(), be.double(), 1e-6)
    ad, wd, bd, rd, gd, bed = _sp(a), _sp(w), b.cuda(), r.cuda(), g.cuda(), be.cuda()
    x_out = torch.empty((M, E), device="cuda")
    h_out = torch.empty((M, E), device="cuda")
    L.call("pp_gemm_residual_layernorm", F16X3, ad.data_ptr(), wd.data_ptr(), bd.data_ptr(), rd.data_ptr(), res_mod,
           x_out.data_ptr(), gd.data_ptr(), bed.data_ptr(), 1e-6, h_out.data_ptr(), SPLIT, M, E, K, K, K, None)
    torch.testing.assert_close(x_out.cpu().double(), x_ref, **TOL)
    torch.testing.assert_close(_unsp(h_out), h_ref, **TOL)


@gpu
@pytest.mark.parametrize("hd,S,dma", [(32, 192, 1), (64, 192, 1), (32, 432, 1), (64, 432, 1), (32, 432, 0), (64, 432, 0)])
def test_attention_split(hd, S, dma):
    """432-token sequences: the LDS-DMA kernel (pp_attention_dma.hip, option attn_dma = 1, the default) and the register-staged
    kernel it replaced (attn_dma = 0), both against fp64."""
    L = _lib()
    n_seq, heads = 3, 4
    E = heads * hd
    qkv = _rand(n_seq * S, 3 * E, seed=16, scale=1.3)
    qd = _sp(qkv)
    out = torch.empty((n_seq * S, E), device="cuda")
    L.set_option("attn_dma", dma)
    try:
        L.call("pp_attention", F16X3, qd.data_ptr(), out.data_ptr(), n_seq, S, heads, hd, hd ** -0.5, None)
    finally:
        L.set_option("attn_dma", 1)
    x = qkv.double().reshape(n_seq, S, 3, heads, hd).permute(2, 0, 3, 1, 4)
    att = ((x[0] @ x[1].transpose(-2, -1)) * hd ** -0.5).softmax(-1)
    ref = (att @ x[2]).transpose(1, 2).reshape(n_seq * S, E)
    torch.testing.assert_close(_unsp(out), ref, **TOL)


@gpu
def test_conv3x3_grouped_and_splitk_split():
    L = _lib()
    G, B, H, W, C = 4, 3, 8, 6, 128
    x = _rand(G, B, C, H, W, seed=11)
    w = _rand(G, C, C, 3, 3, seed=12, scale=1 / math.sqrt(9 * C))
    b = _rand(G, C, seed=13)
    ref = torch.stack([F.conv2d(x[g].double(), w[g].double(), b[g].double(), padding=1) for g in range(G)])
    xd = _sp(x.permute(0, 1, 3, 4, 2).contiguous())
    wd = _sp(w.permute(0, 1, 3, 4, 2).reshape(G, C, 9 * C).contiguous())
    bd = b.cuda()
    for fmt in (0, SPLIT):
        out = torch.empty((G, B, H, W, C), device="cuda")
        L.call("pp_conv_gemm", F16X3, 1, xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), out.data_ptr(), B, H, W, C, C,
               0, 0, G, B * H * W * C, C * 9 * C, B * H * W * C, C, C, 0, fmt, None)
        got = (_unsp(out) if fmt else out.cpu().double()).permute(0, 1, 4, 2, 3)
        torch.testing.assert_close(got, ref, **TOL)
    # split-K partial sums + the pooling kernel that reduces them (shared input: stride 0)
    part = torch.empty((3, G, B, H, W, C), device="cuda")
    x0 = _sp(x[0].permute(0, 2, 3, 1).contiguous())
    L.call("pp_conv3x3_splitk", F16X3, x0.data_ptr(), wd.data_ptr(), part.data_ptr(), B, H, W, C, C, G, 0, C * 9 * C, 3, None)
    pooled = torch.empty((G * B, H // 2, W // 2, C), device="cuda")
    L.call("pp_sum_maxpool_relu_nhwc", part.data_ptr(), 3, G * B * H * W * C, bd.data_ptr(), B, pooled.data_ptr(), SPLIT,
           G * B, H, W, C, 2, 2, None)
    ref2 = torch.stack([F.conv2d(x[0].double(), w[g].double(), b[g].double(), padding=1) for g in range(G)])
    ref2 = F.relu(F.max_pool2d(ref2.reshape(G * B, C, H, W), 2, 2)).permute(0, 2, 3, 1)
    torch.testing.assert_close(_unsp(pooled), ref2, **TOL)


@gpu
@pytest.mark.parametrize("B", [128, 272])
def test_conv3x3_splitk_channel_slices_wide_tiles(B):
    """The 4 x 4 stage of the scalar towers at the bench shape (128 images x 16 pixels, 384 channels, four towers): four
    channel-range K-slices on the wide-tile kernel (one 256 x 192 tile per CU), partial sums reduced by the pooling kernel -
    against torch fp64 on the unrounded operands; and the library's own slice rule picks exactly this form for the shape."""
    L = _lib()
    G, H, W, C = 4, 4, 4, 384  # (B = 272: 17 row tiles, two tiles per workgroup - the 27-stage tiles run on through the two-stage ring - and a ragged last row tile)
    assert L.lib.pp_conv3x3_splitk_slices(F16X3, B, H, W, C, C, G) == 4
    assert L.lib.pp_conv3x3_splitk_slices(F16X3, 8, H, W, C, C, G) in (3, 9), "too few rows for the wide tiles: whole-tap slices"
    x = _rand(G, B, C, H, W, seed=61)
    w = _rand(G, C, C, 3, 3, seed=62, scale=1 / math.sqrt(9 * C))
    b = _rand(G, C, seed=63)
    xd = _sp(x.permute(0, 1, 3, 4, 2).contiguous())
    wd = _sp(w.permute(0, 1, 3, 4, 2).reshape(G, C, 9 * C).contiguous())
    bd = b.cuda()
    part = torch.full((4, G, B, H, W, C), float("nan"), device="cuda")
    L.call("pp_conv3x3_splitk", F16X3, xd.data_ptr(), wd.data_ptr(), part.data_ptr(), B, H, W, C, C, G, B * H * W * C, C * 9 * C, 4, None)
    pooled = torch.empty((G * B, H // 2, W // 2, C), device="cuda")
    L.call("pp_sum_maxpool_relu_nhwc", part.data_ptr(), 4, G * B * H * W * C, bd.data_ptr(), B, pooled.data_ptr(), SPLIT,
           G * B, H, W, C, 2, 2, None)
    torch.cuda.synchronize()
    ref = torch.stack([F.conv2d(x[g].double(), w[g].double(), None, padding=1) for g in range(G)])  # (G, B, C, H, W)
    torch.testing.assert_close(part.sum(0).cpu().double().permute(0, 1, 4, 2, 3), ref, **TOL)
    # every slice is the convolution over ITS channel range
    for s_ in range(4):
        cs = slice(96 * s_, 96 * (s_ + 1))
        ref_s = F.conv2d(x[1][:, cs].double(), w[1][:, cs].double(), None, padding=1)
        torch.testing.assert_close(part[s_, 1].cpu().double().permute(0, 3, 1, 2), ref_s, **TOL)
    ref2 = F.relu(F.max_pool2d((ref + b.double()[:, None, :, None, None]).reshape(G * B, C, H, W), 2, 2)).permute(0, 2, 3, 1)
    torch.testing.assert_close(_unsp(pooled), ref2, **TOL)
    # a slice count the kernel does not take is refused, not mis-computed
    with pytest.raises(L.ProbPoseLibraryError):
        L.call("pp_conv3x3_splitk", F16X3, xd.data_ptr(), wd.data_ptr(), part.data_ptr(), B, H, W, C, C, G, B * H * W * C, C * 9 * C, 5, None)


@gpu
@pytest.mark.parametrize("B,H,W", [(128, 16, 12), (6, 16, 12), (5, 24, 18)])
def test_conv3x3_winograd_maxpool_relu_vs_fp64(B, H, W):
    """First tower stage in its Winograd F(2x2, 3x3) form (pp_winograd.hip: input transform, 16 position GEMMs, output
    transform + MaxPool(4, 3) + bias + ReLU in the epilogue), four towers sharing the input, against torch fp64 on the
    unrounded operands: Conv2d(k3, p1) + bias -> MaxPool2d((4, 3)) -> ReLU (probmap_head.py:261-294). B = 128: the bench shape
    (32 row blocks x 16 column tiles); B = 6: a ragged last row block (two of four images missing); 24 x 18 maps (ViT-B 384x288:
    18 groups of 2 x 3 tiles per image, a workgroup's 32 groups straddle images, ragged last block)."""
    from probpose_code_amd.weights import winograd_weights

    L = _lib()
    G, C = 4, 384
    x = _rand(B, C, H, W, seed=71)
    w = _rand(G, C, C, 3, 3, seed=72, scale=1 / math.sqrt(9 * C))
    b = _rand(G, C, seed=73, scale=0.3)
    ref = torch.stack([F.relu(F.max_pool2d(F.conv2d(x.double(), w[g].double(), b[g].double(), padding=1), (4, 3))) for g in range(G)])
    xd = _sp(x.permute(0, 2, 3, 1).contiguous())
    ud = _sp(torch.stack([winograd_weights(w[g]) for g in range(G)]))
    bd = b.cuda()
    nbytes = L.lib.pp_winograd_scratch_bytes(B, H, W, C)
    assert nbytes == 16 * B * (H // 2) * (W // 2) * C * 4
    scratch = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    out = torch.full((G, B, H // 4, W // 3, C), float("nan"), device="cuda")
    L.call("pp_conv3x3_winograd_maxpool_relu", xd.data_ptr(), ud.data_ptr(), bd.data_ptr(), scratch.data_ptr(), out.data_ptr(), B, H, W, C, C,
           4, 3, G, None)
    torch.cuda.synchronize()
    got = _unsp(out).permute(0, 1, 4, 2, 3)
    err = (got - ref).abs().max().item()
    assert err <= 5e-5, f"max abs error {err:.2e} (ref scale {ref.abs().max().item():.2f})"
    torch.testing.assert_close(got, ref, rtol=5e-5, atol=5e-5)
    # the implicit-GEMM form of the same stage (POOL epilogue) gives the same map to rounding
    wd = _sp(w.permute(0, 1, 3, 4, 2).reshape(G, C, 9 * C).contiguous())
    if B * H * W >= 192 * 8 and (H, W) == (16, 12):
        out2 = torch.empty((G, B, 4, 4, C), device="cuda")
        full = torch.empty((G, B, H, W, C), device="cuda")
        L.call("pp_conv3x3_maxpool_relu", F16X3, xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), out2.data_ptr(), full.data_ptr(), B, H, W, C, C, 4, 3, G,
               0, C * 9 * C, C, SPLIT, None)
        torch.cuda.synchronize()
        torch.testing.assert_close(_unsp(out2).permute(0, 1, 4, 2, 3), got, rtol=5e-5, atol=5e-5)
    # shapes the kernel is not built for are refused
    with pytest.raises(L.ProbPoseLibraryError):
        L.call("pp_conv3x3_winograd_maxpool_relu", xd.data_ptr(), ud.data_ptr(), bd.data_ptr(), scratch.data_ptr(), out.data_ptr(), B, 14, 12, C, C,
               4, 3, G, None)


@gpu
def test_deconv_all_phases_split():
    L = _lib()
    B, H, W, Cin, Cout = 2, 8, 6, 128, 64
    x = _rand(B, Cin, H, W, seed=14)
    w = _rand(Cin, Cout, 4, 4, seed=15, scale=1 / math.sqrt(4 * Cin))
    b = _rand(Cout, seed=17)
    ref = F.relu(F.conv_transpose2d(x.double(), w.double(), b.double(), stride=2, padding=1))
    xd = _sp(x.permute(0, 2, 3, 1).contiguous())
    ph = torch.empty((2, 2, Cout, 4 * Cin))
    for py in range(2):
        for px in range(2):
            for ty in range(2):
                for tx in range(2):
                    t = ty * 2 + tx
                    ph[py, px, :, t * Cin:(t + 1) * Cin] = w[:, :, 3 - 2 * ty - py, 3 - 2 * tx - px].t()
    pd, bd = _sp(ph), b.cuda()
    out = torch.full((B, 2 * H, 2 * W, Cout), float("nan"), device="cuda")
    L.call("pp_conv_gemm", F16X3, 2, xd.data_ptr(), pd.data_ptr(), bd.data_ptr(), out.data_ptr(), B, H, W, Cin, Cout,
           -1, 0, 1, 0, 0, 0, 0, Cout, 2, SPLIT, None)
    torch.testing.assert_close(_unsp(out).permute(0, 3, 1, 2), ref, **TOL)


@gpu
def test_elementwise_kernels_split():
    """im2col (+ preprocessing), LayerNorm, MaxPool + ReLU, tower_final with split-fp16 inputs / outputs."""
    from oracle import model_ref as Mref

    L = _lib()
    # --- im2col: hi + lo reproduces the fp32 patch matrix to the format's resolution; hi alone is its fp16 rounding
    B, H, W = 2, 64, 48
    g = torch.Generator().manual_seed(20)
    img = torch.randint(0, 256, (B, 3, H, W), generator=g, dtype=torch.uint8)
    mean = np.array([123.675, 116.28, 103.53], np.float32)
    std = np.array([58.395, 57.12, 57.375], np.float32)
    Hp, Wp = (H + 4 - 16) // 16 + 1, (W + 4 - 16) // 16 + 1
    out = torch.empty((2 * B * Hp * Wp, 768), device="cuda")
    imgd = img.cuda()
    L.call("pp_preproc_im2col", F16X3, imgd.data_ptr(), 0, out.data_ptr(), B, 2, H, W, 16, 2, mean.ctypes.data,
           std.ctypes.data, 1, None)
    x = Mref.preprocess(img, mean, std)
    both = torch.cat([x, x.flip(-1)])
    ref = F.unfold(F.pad(both, (2, 2, 2, 2))[:, :, : Hp * 16, : Wp * 16], 16, stride=16).transpose(1, 2).reshape(-1, 768)
    from probpose_code_amd.weights import to_split

    assert torch.equal(out.cpu(), to_split(ref)), "the device split of the fp32 patch matrix is the host's to_split, bit for bit"
    # --- LayerNorm
    M, E = 203, 768
    xx, gg, bb = _rand(M, E, seed=17, scale=3.0) + 0.7, 1 + 0.1 * _rand(E, seed=18), _rand(E, seed=19)
    y = torch.empty((M, E), device="cuda")
    xd, gd, bd = xx.cuda(), gg.cuda(), bb.cuda()
    L.call("pp_layernorm", xd.data_ptr(), gd.data_ptr(), bd.data_ptr(), y.data_ptr(), M, E, 1e-6, SPLIT, None)
    torch.testing.assert_close(_unsp(y), F.layer_norm(xx.double(), (E,), gg.double(), bb.double(), 1e-6), rtol=1e-5, atol=1e-5)
    # --- MaxPool + ReLU: split -> split, fp32 -> split, split -> fp32
    N, H2, W2, C = 5, 16, 12, 64
    t = _rand(N, H2, W2, C, seed=21)
    refp = F.relu(F.max_pool2d(t.permute(0, 3, 1, 2), (4, 3), (4, 3))).permute(0, 2, 3, 1).double()
    for fi, fo in ((SPLIT, SPLIT), (0, SPLIT), (SPLIT, 0)):
        src = _sp(t) if fi else t.cuda()
        dst = torch.empty((N, 4, 4, C), device="cuda")
        L.call("pp_maxpool_relu_nhwc", src.data_ptr(), fi, dst.data_ptr(), fo, N, H2, W2, C, 4, 3, None)
        torch.testing.assert_close(_unsp(dst) if fo else dst.cpu().double(), refp, rtol=1e-6, atol=1e-7)
    # --- tower_final on split features
    Bt, K, Ct = 3, 17, 384
    feat, w, b = _rand(4, 2 * Bt, Ct, seed=22), _rand(4, K, Ct, seed=23, scale=0.05), _rand(4, K, seed=24, scale=0.3)
    fi = [0, 2, 1, 4, 3, 6, 5, 8, 7, 10, 9, 12, 11, 14, 13, 16, 15]
    o = torch.empty((4, Bt, K), device="cuda")
    fd, wd, bd2, fid = _sp(feat), w.cuda(), b.cuda(), torch.tensor(fi, dtype=torch.int32).cuda()
    L.call("pp_tower_final", fd.data_ptr(), SPLIT, wd.data_ptr(), bd2.data_ptr(), fid.data_ptr(), o.data_ptr(), Bt, 2, Ct, K, 1.0, None)
    z = torch.einsum("tbc,tkc->tbk", feat.double(), w.double()) + b.double()[:, None]
    a = torch.cat([torch.sigmoid(z[:3]), F.relu(z[3:])])
    torch.testing.assert_close(o.cpu().double(), (a[:, :Bt] + a[:, Bt:][:, :, fi]) * 0.5, rtol=1e-5, atol=1e-6)


# ---- the wide-tile kernel (pp_panel_split.hip) takes these problems once there are enough 256 x 192 / 192 x 256 tiles to
# fill the chip; shapes with tail rows and image borders
@gpu
@pytest.mark.parametrize("act,split_out,K", [(1, 1, 384), (0, 0, 128), (2, 1, 64)])
def test_panel_split_linear(act, split_out, K):
    L = _lib()
    M, N = 96 * 256 + 40, 384   # 97 row tiles x 2 column tiles = 194 tiles, last row tile 40 rows
    a, w, b = _rand(M, K, seed=41), _rand(N, K, seed=42, scale=1 / math.sqrt(K)), _rand(N, seed=43)
    ref = a.double() @ w.double().t() + b.double()
    ref = F.gelu(ref) if act == 1 else (F.relu(ref) if act == 2 else ref)
    ad, wd, bd = _sp(a), _sp(w), b.cuda()
    out = torch.full((M, N), float("nan"), device="cuda")
    L.call("pp_gemm", F16X3, ad.data_ptr(), wd.data_ptr(), bd.data_ptr(), None, 0, out.data_ptr(), M, N, K, K, K, N, act,
           SPLIT if split_out else 0, 0, None)
    torch.testing.assert_close(_unsp(out) if split_out else out.cpu().double(), ref, **TOL)


@gpu
@pytest.mark.parametrize("res,split_out", [(True, 0), (False, 1)])
def test_panel_split_linear_ragged_round_tail(res, split_out):
    """A Linear layer whose 256 x 192 tiles make whole rounds of the chip plus a short last one (79 row tiles x 4 = 316 tiles: one
    round of 256 + 60): the rows of the whole rounds run on the wide tiles, the tail rows (3 616, the last tile 32 rows) in a
    second launch on 128 x 192 tiles - fp32 output with residual (the ViT-B projection / fc2 form) and split output - against
    torch fp64; and the same result with the tail launch switched off."""
    L = _lib()
    M, N, K = 20000, 768, 768
    a, w, b = _rand(M, K, seed=81), _rand(N, K, seed=82, scale=1 / math.sqrt(K)), _rand(N, seed=83)
    r = _rand(M, N, seed=84)
    ref = a.double() @ w.double().t() + b.double() + (r.double() if res else 0)
    ad, wd, bd, rd = _sp(a), _sp(w), b.cuda(), r.cuda()
    outs = []
    for tail in (1, 0):
        L.set_option("psplit_tail", tail)
        out = torch.full((M, N), float("nan"), device="cuda")
        L.call("pp_gemm", F16X3, ad.data_ptr(), wd.data_ptr(), bd.data_ptr(), rd.data_ptr() if res else None, 0, out.data_ptr(), M, N, K, K, K, N, 0,
               SPLIT if split_out else 0, 0, None)
        got = _unsp(out) if split_out else out.cpu().double()
        assert not torch.isnan(got).any(), "rows left unwritten"
        torch.testing.assert_close(got, ref, **TOL)
        outs.append(out.clone())
    L.set_option("psplit_tail", 1)
    assert torch.equal(outs[0], outs[1]), "the tail launch changes the bits (same K order per output: it must not)"


@gpu
@pytest.mark.parametrize("act,split_out,res,res_mod,bias,K", [(1, 1, False, 0, True, 96), (0, 0, True, 0, True, 768), (2, 1, False, 0, False, 64),
                                                           (0, 0, True, 100, False, 128)])
def test_linear_dma_twelve_wave_tiles(act, split_out, res, res_mod, bias, K):
    """pp_gemm on the twelve-wave 192 x 192 kernels (pp_linear_dma.hip: >= 512 tiles, N % 192 == 0 - the Linear layers of ViT-B at bs 64)
    against torch fp64 on the unrounded inputs: the three activations, both output formats, residual rows and a broadcast residual
    table (res_mod), no bias, a ragged last row tile (77 rows), K = 64 (two stages) .. 768; repeated launches bit-identical; and the
    wide-tile kernel (option linear_dma = 0) gives the same numbers to rounding."""
    L = _lib()
    M, N = 192 * 128 + 77, 768  # 129 x 4 = 516 tiles
    a, w = _rand(M, K, seed=91), _rand(N, K, seed=92, scale=1 / math.sqrt(K))
    b = _rand(N, seed=93) if bias else None
    r = _rand(res_mod if res_mod else M, N, seed=94) if res else None
    ref = a.double() @ w.double().t() + (b.double() if bias else 0)
    ref = F.gelu(ref) if act == 1 else (F.relu(ref) if act == 2 else ref)
    if res:
        ref = ref + (r.double()[torch.arange(M) % res_mod] if res_mod else r.double())
    ad, wd = _sp(a), _sp(w)
    bd, rd = (b.cuda() if bias else None), (r.cuda() if res else None)
    outs = []
    try:
        for opt in (1, 1, 0):
            L.set_option("linear_dma", opt)
            out = torch.full((M, N), float("nan"), device="cuda")
            L.call("pp_gemm", F16X3, ad.data_ptr(), wd.data_ptr(), L.ptr(bd), L.ptr(rd), res_mod, out.data_ptr(), M, N, K, K, K, N, act,
                   SPLIT if split_out else 0, 0, None)
            got = _unsp(out) if split_out else out.cpu().double()
            assert not torch.isnan(got).any(), "rows left unwritten"
            torch.testing.assert_close(got, ref, **TOL)
            outs.append((out.clone(), got))
    finally:
        L.set_option("linear_dma", 1)
    assert torch.equal(outs[0][0].view(torch.int32), outs[1][0].view(torch.int32)), "run-to-run difference"
    torch.testing.assert_close(outs[0][1], outs[2][1], rtol=3e-6, atol=3e-6)


# ---- large split Linear layers without residual (qkv / fc1 of the ViT at bs 64: the twelve-wave tile kernel since round 5; the overlapped-epilogue
# kernel these shapes were written for was retired in round 6): tail rows, the three activations, both output formats, uneven tile counts per
# workgroup - and a WEIGHT SCALE: the weights stored times 2^e, pp_gemm_ws handed 2^-e, must give the same bits as the unscaled launch
@gpu
@pytest.mark.parametrize("act,split_out,N,K,M", [(1, 1, 1536, 384, 192 * 86 + 77), (0, 1, 1152, 384, 192 * 86 + 77), (2, 0, 1152, 192, 192 * 86 + 77),
                                                  (0, 0, 1536, 768, 192 * 86 + 77), (1, 1, 1536, 384, 24576), (0, 1, 1152, 384, 24576)])
def test_linear_large_tiles_and_weight_scale(act, split_out, N, K, M):
    L = _lib()
    a, w, b = _rand(M, K, seed=61), _rand(N, K, seed=62, scale=1 / math.sqrt(K)), _rand(N, seed=63)
    ref = a.double() @ w.double().t() + b.double()
    ref = F.gelu(ref) if act == 1 else (F.relu(ref) if act == 2 else ref)
    ad, wd, bd = _sp(a), _sp(w), b.cuda()
    out = torch.full((M, N), float("nan"), device="cuda")
    L.call("pp_gemm", F16X3, ad.data_ptr(), wd.data_ptr(), bd.data_ptr(), None, 0, out.data_ptr(), M, N, K, K, K, N, act,
           SPLIT if split_out else 0, 0, None)
    got = _unsp(out) if split_out else out.cpu().double()
    assert not torch.isnan(got).any(), "rows or columns left unwritten"
    torch.testing.assert_close(got, ref, **TOL)
    first = out.clone()
    for _ in range(3):  # the same launch again: bit-identical every time (no wave reads a staging buffer another one is rewriting)
        out.fill_(float("nan"))
        L.call("pp_gemm", F16X3, ad.data_ptr(), wd.data_ptr(), bd.data_ptr(), None, 0, out.data_ptr(), M, N, K, K, K, N, act,
               SPLIT if split_out else 0, 0, None)
        assert torch.equal(out, first)
    # weights * 2^17 (their low halves leave the fp16 subnormals), sums * 2^-17: at least as close to fp64, and close to the unscaled result
    ws = _sp(w * 2.0 ** 17)
    out.fill_(float("nan"))
    L.call("pp_gemm_ws", F16X3, ad.data_ptr(), ws.data_ptr(), bd.data_ptr(), None, 0, out.data_ptr(), M, N, K, K, K, N, act,
           SPLIT if split_out else 0, 0, 2.0 ** -17, None)
    got_s = _unsp(out) if split_out else out.cpu().double()
    torch.testing.assert_close(got_s, ref, **TOL)
    assert (got_s - ref).abs().max() <= (got - ref).abs().max() * 1.5 + 1e-7
    with pytest.raises(L.ProbPoseLibraryError):  # not a power of two
        L.call("pp_gemm_ws", F16X3, ad.data_ptr(), ws.data_ptr(), bd.data_ptr(), None, 0, out.data_ptr(), M, N, K, K, K, N, act,
               SPLIT if split_out else 0, 0, 0.3, None)


@gpu
def test_panel_split_conv3x3_groups():
    L = _lib()
    G, B, H, W, C = 4, 124, 8, 6, 384    # M = 5952 -> 24 tiles of 256 rows (tail: 64) x 2 column tiles x 4 groups = 192
    x = _rand(B, C, H, W, seed=34)
    w = _rand(G, C, C, 3, 3, seed=35, scale=1 / math.sqrt(9 * C))
    b = _rand(G, C, seed=36)
    ref = torch.stack([F.conv2d(x, w[g], b[g], padding=1) for g in range(G)]).double()  # fp32 reference (K = 3456: ~1e-6)
    xd = _sp(x.permute(0, 2, 3, 1).contiguous())
    wd = _sp(w.permute(0, 1, 3, 4, 2).reshape(G, C, 9 * C).contiguous())
    out = torch.full((G, B, H, W, C), float("nan"), device="cuda")
    bd = b.cuda()
    L.call("pp_conv_gemm", F16X3, 1, xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), out.data_ptr(), B, H, W, C, C,
           0, 0, G, 0, C * 9 * C, B * H * W * C, C, C, 0, SPLIT, None)
    torch.testing.assert_close(_unsp(out).permute(0, 1, 4, 2, 3), ref, rtol=3e-5, atol=3e-5)


@gpu
def test_panel_split_deconv_all_phases():
    L = _lib()
    B, H, W, Cin, Cout = 51, 16, 12, 128, 256          # M = 9792 = 51 tiles of 192 rows, x 4 phases = 204 tiles
    x = _rand(B, Cin, H, W, seed=31)
    w = _rand(Cin, Cout, 4, 4, seed=32, scale=1 / math.sqrt(4 * Cin))
    b = _rand(Cout, seed=33)
    ref = F.relu(F.conv_transpose2d(x.double(), w.double(), b.double(), stride=2, padding=1))
    xd = _sp(x.permute(0, 2, 3, 1).contiguous())
    ph = torch.empty((2, 2, Cout, 4 * Cin))
    for py in range(2):
        for px in range(2):
            for ty in range(2):
                for tx in range(2):
                    t = ty * 2 + tx
                    ph[py, px, :, t * Cin:(t + 1) * Cin] = w[:, :, 3 - 2 * ty - py, 3 - 2 * tx - px].t()
    pd, bd = _sp(ph), b.cuda()
    out = torch.full((B, 2 * H, 2 * W, Cout), float("nan"), device="cuda")
    L.call("pp_conv_gemm", F16X3, 2, xd.data_ptr(), pd.data_ptr(), bd.data_ptr(), out.data_ptr(), B, H, W, Cin, Cout,
           -1, 0, 1, 0, 0, 0, 0, Cout, 2, SPLIT, None)
    torch.testing.assert_close(_unsp(out).permute(0, 3, 1, 2), ref, **TOL)


def _ffn_inputs(M, F_, E=384, seed=40):
    h, r = _rand(M, E, seed=seed), _rand(M, E, seed=seed + 1)
    w1, b1 = _rand(F_, E, seed=seed + 2, scale=1 / math.sqrt(E)), _rand(F_, seed=seed + 3, scale=0.2)
    w2, b2 = _rand(E, F_, seed=seed + 4, scale=1 / math.sqrt(F_)), _rand(E, seed=seed + 5, scale=0.2)
    g, be = 1 + 0.1 * _rand(E, seed=seed + 6), _rand(E, seed=seed + 7, scale=0.1)
    return h, r, w1, b1, w2, b2, g, be


def _ffn_pack(L, w1, w2, E, F_):
    nbytes = L.lib.pp_ffn_split_packed_bytes(E, F_)
    assert nbytes == (F_ // 128) * 384 * 1024
    w1d, w2d = _sp(w1), _sp(w2)
    packed = torch.empty(nbytes // 4, dtype=torch.float32, device="cuda")
    L.call("pp_ffn_split_pack_weights", w1d.data_ptr(), w2d.data_ptr(), packed.data_ptr(), E, F_, None)
    return packed


@pytest.fixture
def ffn_form(request):
    """the forms of the fused feed-forward launch (pp_ffn_dma.hip, twelve waves): 2 = two hidden chunks per streamed x block (the pair kernels,
    shipped for an even chunk count), 1 = one chunk at a time (the eight-wave kernel of round 3 - form 0 - was retired in round 6)"""
    L = _lib()
    L.set_option("ffn_pair", int(request.param == 2))
    yield request.param
    L.set_option("ffn_pair", 1)


@gpu
@pytest.mark.parametrize("ffn_form", [2, 1], indirect=True)
@pytest.mark.parametrize("M,F_", [(96 * 3, 1536), (96 * 2 - 40, 256), (24576, 1536)])
def test_ffn_split_fused_vs_fp64(M, F_, ffn_form):
    """pp_ffn_split_residual_layernorm (fc1 - GELU - fc2 + residual + LayerNorm in one launch, hidden activation on the CU)
    against torch fp64 on the unrounded fp32 inputs; the bs 64 shape of the bench included; repeated launches bit-identical
    (a ring / barrier race shows up as run-to-run differences long before it shows up as a tolerance failure)."""
    L = _lib()
    E = 384
    h, r, w1, b1, w2, b2, g, be = _ffn_inputs(M, F_)
    x_ref = r.double() + F.gelu(h.double() @ w1.double().t() + b1.double()) @ w2.double().t() + b2.double()
    h_ref = F.layer_norm(x_ref, (E,), g.double(), be.double(), 1e-6)
    packed = _ffn_pack(L, w1, w2, E, F_)
    hd, rd = _sp(h), r.cuda()
    dev = [t.cuda() for t in (b1, b2, g, be)]
    outs = []
    for _ in range(3):
        x_out = torch.full((M, E), float("nan"), device="cuda")
        h_out = torch.full((M, E), float("nan"), device="cuda")
        L.call("pp_ffn_split_residual_layernorm", hd.data_ptr(), packed.data_ptr(), dev[0].data_ptr(), dev[1].data_ptr(),
               rd.data_ptr(), x_out.data_ptr(), dev[2].data_ptr(), dev[3].data_ptr(), 1e-6, h_out.data_ptr(), M, E, F_, None)
        outs.append((x_out.cpu(), h_out.cpu()))
    torch.testing.assert_close(outs[0][0].double(), x_ref, **TOL)
    torch.testing.assert_close(_unsp(outs[0][1]), h_ref, **TOL)
    for x_o, h_o in outs[1:]:
        assert torch.equal(x_o, outs[0][0]) and torch.equal(h_o.view(torch.int32), outs[0][1].view(torch.int32))


@gpu
@pytest.mark.parametrize("ffn_form", [2, 1], indirect=True)
def test_ffn_split_fused_in_place_and_errors(ffn_form):
    """residual aliasing x_out and h_in aliasing h_out (how the engine calls it), and the argument checks."""
    L = _lib()
    M, E, F_ = 96 * 5 + 17, 384, 512
    h, r, w1, b1, w2, b2, g, be = _ffn_inputs(M, F_, seed=60)
    x_ref = r.double() + F.gelu(h.double() @ w1.double().t() + b1.double()) @ w2.double().t() + b2.double()
    h_ref = F.layer_norm(x_ref, (E,), g.double(), be.double(), 1e-6)
    packed = _ffn_pack(L, w1, w2, E, F_)
    hd, xd = _sp(h), r.cuda()
    dev = [t.cuda() for t in (b1, b2, g, be)]
    L.call("pp_ffn_split_residual_layernorm", hd.data_ptr(), packed.data_ptr(), dev[0].data_ptr(), dev[1].data_ptr(),
           xd.data_ptr(), xd.data_ptr(), dev[2].data_ptr(), dev[3].data_ptr(), 1e-6, hd.data_ptr(), M, E, F_, None)
    torch.testing.assert_close(xd.cpu().double(), x_ref, **TOL)
    torch.testing.assert_close(_unsp(hd), h_ref, **TOL)
    assert L.lib.pp_ffn_split_packed_bytes(768, 3072) == -1 and L.lib.pp_ffn_split_packed_bytes(384, 100) == -1
    with pytest.raises(L.ProbPoseLibraryError):
        L.call("pp_ffn_split_residual_layernorm", hd.data_ptr(), packed.data_ptr(), dev[0].data_ptr(), dev[1].data_ptr(),
               xd.data_ptr(), xd.data_ptr(), dev[2].data_ptr(), dev[3].data_ptr(), 1e-6, hd.data_ptr(), M, 768, F_, None)
    with pytest.raises(L.ProbPoseLibraryError):
        L.call("pp_ffn_split_residual_layernorm", None, packed.data_ptr(), dev[0].data_ptr(), dev[1].data_ptr(),
               xd.data_ptr(), xd.data_ptr(), dev[2].data_ptr(), dev[3].data_ptr(), 1e-6, hd.data_ptr(), M, E, F_, None)


def _proj_inputs(M, E=384, seed=80):
    att = _rand(M, E, seed=seed)
    wp, bp = _rand(E, E, seed=seed + 1, scale=1 / math.sqrt(E)), _rand(E, seed=seed + 2, scale=0.2)
    g2, be2 = 1 + 0.1 * _rand(E, seed=seed + 3), _rand(E, seed=seed + 4, scale=0.1)
    return att, wp, bp, g2, be2


@gpu
@pytest.mark.parametrize("ffn_form", [2, 1], indirect=True)
@pytest.mark.parametrize("M,F_", [(96 * 3, 1536), (96 * 2 - 40, 256), (24576, 1536)])
def test_proj_ffn_split_fused_vs_fp64(M, F_, ffn_form):
    """pp_proj_ffn_split_residual_layernorm (projection + residual + ln2 + FFN + residual + LayerNorm in one launch) against
    torch fp64 on the unrounded fp32 inputs, called the way the engine calls it (residual aliases x_out, the attention rows
    alias h_out); repeated launches bit-identical; a ragged last tile; the bs 64 shape."""
    L = _lib()
    E = 384
    _, r, w1, b1, w2, b2, g, be = _ffn_inputs(M, F_)
    att, wp, bp, g2, be2 = _proj_inputs(M)
    x_mid = r.double() + att.double() @ wp.double().t() + bp.double()
    h_mid = F.layer_norm(x_mid, (E,), g2.double(), be2.double(), 1e-6)
    x_ref = x_mid + F.gelu(h_mid @ w1.double().t() + b1.double()) @ w2.double().t() + b2.double()
    h_ref = F.layer_norm(x_ref, (E,), g.double(), be.double(), 1e-6)
    packed = _ffn_pack(L, w1, w2, E, F_)
    nb = L.lib.pp_proj_split_packed_bytes(E)
    assert nb == E * E * 4 and L.lib.pp_proj_split_packed_bytes(768) == -1
    wpp = torch.empty(nb // 4, dtype=torch.float32, device="cuda")
    L.call("pp_proj_split_pack_weights", _sp(wp).data_ptr(), wpp.data_ptr(), E, None)
    dev = [t.cuda() for t in (bp, g2, be2, b1, b2, g, be)]
    outs = []
    for _ in range(3):
        ad, xd = _sp(att), r.cuda()
        scratch = torch.full((M, E), float("nan"), device="cuda")
        L.call("pp_proj_ffn_split_residual_layernorm", ad.data_ptr(), wpp.data_ptr(), dev[0].data_ptr(), dev[1].data_ptr(),
               dev[2].data_ptr(), scratch.data_ptr(), packed.data_ptr(), dev[3].data_ptr(), dev[4].data_ptr(), xd.data_ptr(),
               xd.data_ptr(), dev[5].data_ptr(), dev[6].data_ptr(), 1e-6, ad.data_ptr(), M, E, F_, None)
        outs.append((xd.cpu(), ad.cpu(), scratch.cpu()))
    torch.testing.assert_close(_unsp(outs[0][2]), h_mid, **TOL)
    torch.testing.assert_close(outs[0][0].double(), x_ref, rtol=3e-5, atol=3e-5)
    torch.testing.assert_close(_unsp(outs[0][1]), h_ref, rtol=3e-5, atol=3e-5)
    for x_o, h_o, _ in outs[1:]:
        assert torch.equal(x_o, outs[0][0]) and torch.equal(h_o.view(torch.int32), outs[0][1].view(torch.int32))
    with pytest.raises(L.ProbPoseLibraryError):  # scratch aliasing the attention rows
        L.call("pp_proj_ffn_split_residual_layernorm", ad.data_ptr(), wpp.data_ptr(), dev[0].data_ptr(), dev[1].data_ptr(),
               dev[2].data_ptr(), ad.data_ptr(), packed.data_ptr(), dev[3].data_ptr(), dev[4].data_ptr(), xd.data_ptr(),
               xd.data_ptr(), dev[5].data_ptr(), dev[6].data_ptr(), 1e-6, ad.data_ptr(), M, E, F_, None)


@gpu
@pytest.mark.parametrize("deep", [1, 0])
@pytest.mark.parametrize("n_seq,bias", [(3, True), (16, False), (128, True)])
def test_qkv_attention_split_fused_vs_fp64(n_seq, bias, deep):
    """pp_qkv_attention_split (qkv Linear + attention of a (sequence, head) per workgroup, qkv never in HBM) against torch fp64
    on the unrounded fp32 inputs, with mmpretrain's packing of the qkv rows; an odd number of sequences (no XCD remap), the
    bs 64 shape; repeated launches bit-identical. `deep`: small launches (at most two workgroups per CU: 3 and 16 sequences here) run the projection on a
    ring of four stages (option "qkv_attn_deep", shipped) or of two."""
    L = _lib()
    L.set_option("qkv_attn_deep", deep)
    L.reset_launch_counts()
    S, E, H, hd = 192, 384, 12, 32
    M = n_seq * S
    h = _rand(M, E, seed=90)
    w, b = _rand(3 * E, E, seed=91, scale=1 / math.sqrt(E)), _rand(3 * E, seed=92, scale=0.3)
    qkv = h.double() @ w.double().t() + (b.double() if bias else 0.0)
    q, k, v = qkv.reshape(n_seq, S, 3, H, hd).permute(2, 0, 3, 1, 4)
    att = torch.softmax(q @ k.transpose(-1, -2) * hd ** -0.5, dim=-1) @ v
    ref = att.transpose(1, 2).reshape(M, E)
    hd_, wd = _sp(h), _sp(w)
    bd = b.cuda() if bias else None
    outs = []
    for _ in range(3):
        out = torch.full((M, E), float("nan"), device="cuda")
        L.call("pp_qkv_attention_split", hd_.data_ptr(), wd.data_ptr(), bd.data_ptr() if bias else None, out.data_ptr(), n_seq, S, H,
               hd, hd ** -0.5, None)
        outs.append(out.cpu())
    torch.testing.assert_close(_unsp(outs[0]), ref, **TOL)
    for o in outs[1:]:
        assert torch.equal(o.view(torch.int32), outs[0].view(torch.int32))
    for bad in ((n_seq, 432, H, hd), (n_seq, S, 6, 64)):
        with pytest.raises(L.ProbPoseLibraryError):
            L.call("pp_qkv_attention_split", hd_.data_ptr(), wd.data_ptr(), None, out.data_ptr(), bad[0], bad[1], bad[2], bad[3], 0.1, None)
    with pytest.raises(L.ProbPoseLibraryError):
        L.call("pp_qkv_attention_split", hd_.data_ptr(), wd.data_ptr(), None, hd_.data_ptr(), n_seq, S, H, hd, 0.1, None)
    assert (L.launch_count("qkv_attn_deep") > 0) == (deep == 1 and n_seq * H <= 512)
    # small launches of the deep-ring form use two workgroups per (sequence, head), half of the query tiles each (option "qkv_attn_qsplit"): same bits
    L.set_option("qkv_attn_qsplit", 0)
    out = torch.full((M, E), float("nan"), device="cuda")
    L.call("pp_qkv_attention_split", hd_.data_ptr(), wd.data_ptr(), bd.data_ptr() if bias else None, out.data_ptr(), n_seq, S, H, hd, hd ** -0.5, None)
    L.set_option("qkv_attn_qsplit", 1)
    assert torch.equal(out.cpu().view(torch.int32), outs[0].view(torch.int32))
    L.set_option("qkv_attn_deep", 1)


@gpu
@pytest.mark.parametrize("n_seq", [3, 128])
def test_qkv_attention_split_folded_vs_fp64(n_seq):
    """pp_qkv_attention_split_folded: the same launch on CENTERED residual rows (x - mean, what pp_proj_ffn_split_folded leaves) with ln1 folded into
    the projection (gamma into the weights, beta into the bias, rstd per row applied where the bias is added) and the weights stored with a power-of-two
    scale, against torch fp64 with an explicit LayerNorm; rows with an offset (mean / std ~ 2: it must not matter any more); the unfolded launch on the
    normalised rows must agree; repeated launches bit-identical."""
    from probpose_code_amd.weights import fold_layernorm, weight_scale_exponent

    L = _lib()
    S, E, H, hd, eps = 192, 384, 12, 32, 1e-6
    M = n_seq * S
    x = _rand(M, E, seed=190) * 1.3 + 2.5 * _rand(M, 1, seed=191)
    g, be = 1.0 + 0.2 * _rand(E, seed=192), 0.2 * _rand(E, seed=193)
    w, b = _rand(3 * E, E, seed=194, scale=1 / math.sqrt(E)), _rand(3 * E, seed=195, scale=0.3)
    mean = x.mean(dim=1, keepdim=True)
    xs = _sp(x - mean)          # the rows as the producer leaves them
    xq = _unsp(xs) + mean.double()  # ... and the values they stand for
    hn = F.layer_norm(xq, (E,), g.double(), be.double(), eps)
    qkv = hn @ w.double().t() + b.double()
    q, k, v = qkv.reshape(n_seq, S, 3, H, hd).permute(2, 0, 3, 1, 4)
    ref = (torch.softmax(q @ k.transpose(-1, -2) * hd ** -0.5, dim=-1) @ v).transpose(1, 2).reshape(M, E)
    stats = torch.stack([xq.mean(dim=1), 1.0 / torch.sqrt(xq.var(dim=1, unbiased=False) + eps)], dim=1).float().cuda()
    e = weight_scale_exponent(w.double() * g.double()[None, :])
    assert e >= 10
    wf, _, bf = [t.cuda() for t in fold_layernorm(w, b, g, be, scale_exp=e)]
    outs = []
    for _ in range(3):
        out = torch.full((M, E), float("nan"), device="cuda")
        L.call("pp_qkv_attention_split_folded", xs.data_ptr(), wf.data_ptr(), bf.data_ptr(), stats.data_ptr(), out.data_ptr(), n_seq, S, H,
               hd, hd ** -0.5, 2.0 ** -e, None)
        outs.append(out.cpu())
    torch.testing.assert_close(_unsp(outs[0]), ref, rtol=1e-5, atol=1e-5)
    for o in outs[1:]:
        assert torch.equal(o.view(torch.int32), outs[0].view(torch.int32))
    plain = torch.full((M, E), float("nan"), device="cuda")
    hs, wd, bd = _sp(hn.float()), _sp(w), b.cuda()
    L.call("pp_qkv_attention_split", hs.data_ptr(), wd.data_ptr(), bd.data_ptr(), plain.data_ptr(), n_seq, S, H, hd, hd ** -0.5, None)
    torch.testing.assert_close(_unsp(outs[0]), _unsp(plain.cpu()), rtol=1e-5, atol=1e-5)
    # the unfolded launch with a weight scale
    e2 = weight_scale_exponent(w)
    plain2 = torch.full((M, E), float("nan"), device="cuda")
    L.call("pp_qkv_attention_split_ws", hs.data_ptr(), _sp(w * 2.0 ** e2).data_ptr(), bd.data_ptr(), plain2.data_ptr(), n_seq, S, H, hd, hd ** -0.5,
           2.0 ** -e2, None)
    torch.testing.assert_close(_unsp(plain2.cpu()), ref, rtol=1e-5, atol=1e-5)
    with pytest.raises(L.ProbPoseLibraryError):
        L.call("pp_qkv_attention_split_folded", xs.data_ptr(), wf.data_ptr(), bf.data_ptr(), None, out.data_ptr(), n_seq, S, H, hd, 0.1, 1.0, None)
    with pytest.raises(L.ProbPoseLibraryError):  # a scale that is no power of two
        L.call("pp_qkv_attention_split_folded", xs.data_ptr(), wf.data_ptr(), bf.data_ptr(), stats.data_ptr(), out.data_ptr(), n_seq, S, H, hd, 0.1, 0.7, None)


@gpu
@pytest.mark.parametrize("ffn_form", [2, 1], indirect=True)
def test_proj_ffn_split_two_streams_under_contention(ffn_form):
    """Two independent problems through pp_proj_ffn_split_residual_layernorm on two streams at once must each give the result
    they give alone, bit for bit. This is the condition bench.py's two steps in flight create; it caught counted vmcnt waits
    that allowed plain loads issued between LDS-DMA pieces to be outstanding (the two kinds do not retire in order with
    respect to each other): correct alone, a few 48- / 96-row blocks wrong in 5 - 26 of 80 contended launches."""
    L = _lib()
    M, E, F_ = 24576, 384, 1536
    probs = []
    for seed in (100, 200):
        _, r, w1, b1, w2, b2, g, be = _ffn_inputs(M, F_, seed=seed)
        att, wp, bp, g2, be2 = _proj_inputs(M, seed=seed + 20)
        wpp = torch.empty(E * E, dtype=torch.float32, device="cuda")
        L.call("pp_proj_split_pack_weights", _sp(wp).data_ptr(), wpp.data_ptr(), E, None)
        d = dict(att=_sp(att), x=r.cuda(), wpp=wpp, packed=_ffn_pack(L, w1, w2, E, F_), dev=[t.cuda() for t in (bp, g2, be2, b1, b2, g, be)],
                 xo=torch.empty(M, E, device="cuda"), ho=torch.empty(M, E, device="cuda"), hs=torch.empty(M, E, device="cuda"))
        probs.append(d)

    def run(d, stream):
        v = d["dev"]
        L.call("pp_proj_ffn_split_residual_layernorm", d["att"].data_ptr(), d["wpp"].data_ptr(), v[0].data_ptr(), v[1].data_ptr(),
               v[2].data_ptr(), d["hs"].data_ptr(), d["packed"].data_ptr(), v[3].data_ptr(), v[4].data_ptr(), d["x"].data_ptr(),
               d["xo"].data_ptr(), v[5].data_ptr(), v[6].data_ptr(), 1e-6, d["ho"].data_ptr(), M, E, F_,
               None if stream is None else stream.cuda_stream)

    want = []
    for d in probs:
        run(d, None)
        torch.cuda.synchronize()
        want.append((d["xo"].clone(), d["ho"].clone()))
    s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
    for it in range(25):
        for d in probs:
            d["xo"].fill_(float("nan")), d["ho"].fill_(float("nan")), d["hs"].fill_(float("nan"))
        torch.cuda.synchronize()
        for _ in range(3):
            run(probs[0], s0), run(probs[1], s1)
        torch.cuda.synchronize()
        for k, (d, (xo, ho)) in enumerate(zip(probs, want)):
            assert torch.equal(d["xo"], xo), f"iteration {it}, stream {k}: x_out differs from the solo launch"
            assert torch.equal(d["ho"].view(torch.int32), ho.view(torch.int32)), f"iteration {it}, stream {k}: h_out differs"


@gpu
@pytest.mark.parametrize("pair", [1, 0])
@pytest.mark.parametrize("M,F_", [(96 * 3, 1536), (96 * 2 - 40, 256), (96 * 2 + 5, 384), (24576, 1536)])
def test_ffn_dma_waves_form_vs_fp64_and_weight_scales(M, F_, pair):
    """Both fused feed-forward entry points on the twelve-wave kernel (pp_ffn_dma.hip: eight computing waves + four DMA waves);
    pp_set_option("ffn_pair", 1): its paired-chunk form when the hidden width is an even number of 128-column chunks (F = 384 - three
    chunks - runs the single-chunk kernel under either setting). Against fp64, in place as the engine calls them, a ragged last tile
    included; repeated launches bit-identical, also when two launches share the chip (what two steps in flight create); and with WEIGHT
    SCALES: Wp / W1 / W2 stored times three different powers of two, the *_ws launches handed the inverses - the same numbers to rounding
    (power-of-two factors commute with every rounding; only the weights' low halves gain bits)."""
    L = _lib()
    E = 384
    h, r, w1, b1, w2, b2, g, be = _ffn_inputs(M, F_, seed=300)
    att, wp, bp, g2, be2 = _proj_inputs(M, seed=320)
    x_ref = r.double() + F.gelu(h.double() @ w1.double().t() + b1.double()) @ w2.double().t() + b2.double()
    h_ref = F.layer_norm(x_ref, (E,), g.double(), be.double(), 1e-6)
    x_mid = r.double() + att.double() @ wp.double().t() + bp.double()
    h_mid = F.layer_norm(x_mid, (E,), g2.double(), be2.double(), 1e-6)
    xp_ref = x_mid + F.gelu(h_mid @ w1.double().t() + b1.double()) @ w2.double().t() + b2.double()
    hp_ref = F.layer_norm(xp_ref, (E,), g.double(), be.double(), 1e-6)
    packed = _ffn_pack(L, w1, w2, E, F_)
    wpp = torch.empty(E * E, dtype=torch.float32, device="cuda")
    L.call("pp_proj_split_pack_weights", _sp(wp).data_ptr(), wpp.data_ptr(), E, None)
    dev = [t.cuda() for t in (bp, g2, be2, b1, b2, g, be)]

    def ffn(stream=None):
        hd, xd = _sp(h), r.cuda()
        L.call("pp_ffn_split_residual_layernorm", hd.data_ptr(), packed.data_ptr(), dev[3].data_ptr(), dev[4].data_ptr(), xd.data_ptr(),
               xd.data_ptr(), dev[5].data_ptr(), dev[6].data_ptr(), 1e-6, hd.data_ptr(), M, E, F_, stream)
        return xd, hd

    def proj(stream=None):
        ad, xd = _sp(att), r.cuda()
        scratch = torch.full((M, E), float("nan"), device="cuda")
        L.call("pp_proj_ffn_split_residual_layernorm", ad.data_ptr(), wpp.data_ptr(), dev[0].data_ptr(), dev[1].data_ptr(),
               dev[2].data_ptr(), scratch.data_ptr(), packed.data_ptr(), dev[3].data_ptr(), dev[4].data_ptr(), xd.data_ptr(),
               xd.data_ptr(), dev[5].data_ptr(), dev[6].data_ptr(), 1e-6, ad.data_ptr(), M, E, F_, stream)
        return xd, ad, scratch

    def same(a, b):
        return all(torch.equal(u.view(torch.int32), v.view(torch.int32)) for u, v in zip(a, b))

    try:
        L.set_option("ffn_pair", pair)
        L.reset_launch_counts()
        want_f, want_p = [t.cpu() for t in ffn()], [t.cpu() for t in proj()]
        assert L.launch_count("pp_ffn_dma.hip") == 2 and L.launch_count("ffn_dma_pair") == (2 if pair and (F_ // 128) % 2 == 0 else 0)
        torch.testing.assert_close(want_f[0].double(), x_ref, **TOL)
        torch.testing.assert_close(_unsp(want_f[1]), h_ref, **TOL)
        torch.testing.assert_close(_unsp(want_p[2]), h_mid, **TOL)
        torch.testing.assert_close(want_p[0].double(), xp_ref, rtol=3e-5, atol=3e-5)
        torch.testing.assert_close(_unsp(want_p[1]), hp_ref, rtol=3e-5, atol=3e-5)
        # weight scales: 2^15 / 2^17 / 2^16 on Wp / W1 / W2
        packed_s = _ffn_pack(L, w1 * 2.0 ** 17, w2 * 2.0 ** 16, E, F_)
        wpp_s = torch.empty(E * E, dtype=torch.float32, device="cuda")
        L.call("pp_proj_split_pack_weights", _sp(wp * 2.0 ** 15).data_ptr(), wpp_s.data_ptr(), E, None)
        hd, xd = _sp(h), r.cuda()
        L.call("pp_ffn_split_residual_layernorm_ws", hd.data_ptr(), packed_s.data_ptr(), dev[3].data_ptr(), dev[4].data_ptr(), xd.data_ptr(),
               xd.data_ptr(), dev[5].data_ptr(), dev[6].data_ptr(), 1e-6, hd.data_ptr(), M, E, F_, 2.0 ** -17, 2.0 ** -16, None)
        torch.testing.assert_close(xd.cpu().double(), x_ref, **TOL)
        torch.testing.assert_close(xd.cpu(), want_f[0], rtol=1e-5, atol=1e-5)
        assert (xd.cpu().double() - x_ref).abs().max() <= (want_f[0].double() - x_ref).abs().max() * 1.5 + 1e-7
        ad, xd = _sp(att), r.cuda()
        scratch = torch.full((M, E), float("nan"), device="cuda")
        L.call("pp_proj_ffn_split_residual_layernorm_ws", ad.data_ptr(), wpp_s.data_ptr(), dev[0].data_ptr(), dev[1].data_ptr(),
               dev[2].data_ptr(), scratch.data_ptr(), packed_s.data_ptr(), dev[3].data_ptr(), dev[4].data_ptr(), xd.data_ptr(),
               xd.data_ptr(), dev[5].data_ptr(), dev[6].data_ptr(), 1e-6, ad.data_ptr(), M, E, F_, 2.0 ** -15, 2.0 ** -17, 2.0 ** -16, None)
        torch.testing.assert_close(xd.cpu().double(), xp_ref, rtol=3e-5, atol=3e-5)
        torch.testing.assert_close(_unsp(ad), hp_ref, rtol=3e-5, atol=3e-5)
        torch.testing.assert_close(xd.cpu(), want_p[0], rtol=1e-5, atol=1e-5)
        for _ in range(3):
            assert same([t.cpu() for t in ffn()], want_f), "FFN form: run-to-run difference"
            assert same([t.cpu() for t in proj()], want_p), "projection + FFN form: run-to-run difference"
        if M >= 24576:
            s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
            for it in range(10):
                torch.cuda.synchronize()
                with torch.cuda.stream(s0):
                    a = proj(s0.cuda_stream)
                with torch.cuda.stream(s1):
                    b = ffn(s1.cuda_stream)
                torch.cuda.synchronize()
                assert same([t.cpu() for t in a], want_p) and same([t.cpu() for t in b], want_f), f"contended launch {it}"
    finally:
        L.set_option("ffn_pair", 1)


@gpu
@pytest.mark.parametrize("B,K", [(12, 17), (13, 28)])
def test_deconv_head_split_vs_fp64(B, K):
    """pp_deconv_head_split (last deconvolution + BN + ReLU with the 1x1 conv in its epilogue, split-fp16 operands) against
    ConvTranspose2d + ReLU + Conv1x1 in torch fp64 on the unrounded inputs; phase-separated logits rearranged to planar; a
    ragged last tile (13 x 768 pixels), the largest map count; too few tiles are refused."""
    from probpose_code_amd.weights import pack_head_split

    L = _lib()
    H, W, Cin, Cout = 32, 24, 256, 256
    x = _rand(B, Cin, H, W, seed=120)
    w = _rand(Cin, Cout, 4, 4, seed=121, scale=1 / math.sqrt(4 * Cin))
    b = _rand(Cout, seed=122, scale=0.2)
    wf, bf = _rand(K, Cout, seed=123, scale=4 / math.sqrt(Cout)), _rand(K, seed=124)
    mid = F.relu(F.conv_transpose2d(x.double(), w.double(), b.double(), stride=2, padding=1))
    ref = F.conv2d(mid, wf.double()[:, :, None, None], bf.double())  # (B, K, 2H, 2W)
    ph = torch.empty((2, 2, Cout, 4 * Cin))
    for py in range(2):
        for px in range(2):
            for ty in range(2):
                for tx in range(2):
                    t = ty * 2 + tx
                    ph[py, px, :, t * Cin:(t + 1) * Cin] = w[:, :, 3 - 2 * ty - py, 3 - 2 * tx - px].t()
    wpad = torch.zeros(32, Cout)
    wpad[:K] = wf
    xd, phd, bd, bfd = _sp(x.permute(0, 2, 3, 1).contiguous()), _sp(ph), b.cuda(), bf.cuda()
    hwd = pack_head_split(wpad).cuda()
    outs = []
    for _ in range(2):
        lg = torch.full((B, K, 4, H * W), float("nan"), device="cuda")
        L.call("pp_deconv_head_split", xd.data_ptr(), phd.data_ptr(), bd.data_ptr(), hwd.data_ptr(), bfd.data_ptr(), lg.data_ptr(), B, H, W,
               Cin, Cout, K, None)
        outs.append(lg.cpu())
    planar = outs[0].reshape(B, K, 2, 2, H, W).permute(0, 1, 4, 2, 5, 3).reshape(B, K, 2 * H, 2 * W)
    torch.testing.assert_close(planar.double(), ref, rtol=3e-5, atol=3e-5)
    assert torch.equal(outs[0], outs[1])
    with pytest.raises(L.ProbPoseLibraryError):
        L.call("pp_deconv_head_split", xd.data_ptr(), phd.data_ptr(), bd.data_ptr(), hwd.data_ptr(), bfd.data_ptr(), lg.data_ptr(), 2, H, W,
               Cin, Cout, K, None)


@gpu
@pytest.mark.parametrize("B", [24, 33])
def test_conv3x3_maxpool_relu_split_fused_vs_two_launches(B):
    """pp_conv3x3_maxpool_relu in the split-fp16 mode: MaxPool(4, 3) + ReLU in the epilogue of the wide-tile convolution (one
    16 x 12 image per 192-row tile) against torch fp64 on the unrounded inputs, and against the two-launch route
    (pp_set_option("conv_pool_split", 0): pp_conv_gemm + pp_maxpool_relu_nhwc - another tile shape, so another fp32 summation
    order: equal to fp32 accumulation noise, not bit for bit); the scratch tensor stays untouched. (Below 24
    images the launch has fewer than 192 tiles and takes the two-launch route by itself.)"""
    L = _lib()
    G, C, H, W, ph, pw = 4, 384, 16, 12, 4, 3
    x = _rand(B, C, H, W, seed=234)
    w = _rand(G, C, C, 3, 3, seed=235, scale=1 / math.sqrt(9 * C))
    b = _rand(G, C, seed=236)
    ref = torch.stack([F.max_pool2d(F.conv2d(x.double(), w[g].double(), b[g].double(), padding=1), (ph, pw)).clamp_min(0) for g in range(G)])
    xd = _sp(x.permute(0, 2, 3, 1).contiguous())
    wd = _sp(w.permute(0, 1, 3, 4, 2).reshape(G, C, 9 * C).contiguous())
    bd = b.cuda()
    outs = []
    for fused in (1, 0):
        L.set_option("conv_pool_split", fused)
        pooled = torch.full((G, B, H // ph, W // pw, C), float("nan"), device="cuda")
        scratch = torch.zeros((G, B, H, W, C), device="cuda")
        L.call("pp_conv3x3_maxpool_relu", F16X3, xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), pooled.data_ptr(), scratch.data_ptr(),
               B, H, W, C, C, ph, pw, G, 0, C * 9 * C, C, SPLIT, None)
        torch.cuda.synchronize()
        if fused:
            assert not scratch.any(), "the one-launch form must not touch the scratch tensor"
        outs.append(pooled.cpu())
    L.set_option("conv_pool_split", 1)
    torch.testing.assert_close(_unsp(outs[0]).permute(0, 1, 4, 2, 3), ref, **TOL)
    torch.testing.assert_close(_unsp(outs[0]), _unsp(outs[1]), **TOL)


def _row_part_stats(y):
    """(M, N) fp64 -> (M, N / 96, 2): per row and 96-column part (mean, sum of squared deviations from it) - pp_linear_ln_folded's stats_out."""
    p = y.reshape(y.shape[0], -1, 96)
    mean = p.mean(dim=2)
    return torch.stack([mean, ((p - mean[..., None]) ** 2).sum(dim=2)], dim=2)


@gpu
@pytest.mark.parametrize("M", [192 * 3, 192 * 2 + 77, 24576])
def test_linear_ln_folded_chain_vs_fp64(M):
    """pp_linear_ln_folded: a ViT block's Linear layers with the LayerNorms folded in (mmpretrain TransformerEncoderLayer [3P]:
    x = x + attn(ln1(x)); x = ffn(ln2(x)) + x). The chain the engine runs - proj (fp32 residual in, split rows + statistics out), fc1 on
    the RAW rows with ln2 folded in (gamma into the weights, beta into the bias, mean / rstd from the statistics) + GELU, fc2 (split
    residual in place, statistics out), and an fp32-output fc2 - against torch fp64 on the unrounded inputs with an explicit LayerNorm;
    the rows carry an offset (mean / std ~ 3) so that the mean term is not small; a ragged last tile; the statistics themselves checked;
    repeated launches bit-identical; shapes / aliasing it does not serve are refused."""
    from probpose_code_amd.weights import fold_layernorm

    L = _lib()
    E, Fd, F32, eps = 768, 1536, 0, 1e-6
    att = _rand(M, E, seed=700)
    x0 = _rand(M, E, seed=701) + 3.0 * _rand(M, 1, seed=702)
    wp, bp = _rand(E, E, seed=703, scale=1 / math.sqrt(E)), _rand(E, seed=704, scale=0.1)
    g2, be2 = 1.0 + 0.2 * _rand(E, seed=705), 0.2 * _rand(E, seed=706)
    w1, b1 = _rand(Fd, E, seed=707, scale=1 / math.sqrt(E)), _rand(Fd, seed=708, scale=0.1)
    w2, b2 = _rand(E, Fd, seed=709, scale=1 / math.sqrt(Fd)), _rand(E, seed=710, scale=0.1)
    # fp64 reference
    x1 = x0.double() + att.double() @ wp.double().t() + bp.double()
    h2 = F.layer_norm(x1, (E,), g2.double(), be2.double(), eps)
    f = F.gelu(h2 @ w1.double().t() + b1.double())
    x2 = x1 + f @ w2.double().t() + b2.double()
    # device
    w1f, c1, b1f = fold_layernorm(w1, b1, g2, be2)
    d = dict(att=_sp(att), x0=x0.cuda(), wp=_sp(wp), bp=bp.cuda(), w1f=w1f.cuda(), c1=c1.cuda(), b1f=b1f.cuda(), w2=_sp(w2), b2=b2.cuda())

    def chain():
        xs = torch.full((M, E), float("nan"), device="cuda")
        st = torch.full((M, E // 96, 2), float("nan"), device="cuda")
        fbuf = torch.full((M, Fd), float("nan"), device="cuda")
        xo = torch.full((M, E), float("nan"), device="cuda")
        L.call("pp_linear_ln_folded", d["att"].data_ptr(), d["wp"].data_ptr(), d["bp"].data_ptr(), d["x0"].data_ptr(), F32, xs.data_ptr(), SPLIT,
               M, E, E, 0, None, None, eps, st.data_ptr(), None)
        x1_dev, st1 = _unsp(xs), st.cpu().double()
        L.call("pp_linear_ln_folded", xs.data_ptr(), d["w1f"].data_ptr(), d["b1f"].data_ptr(), None, F32, fbuf.data_ptr(), SPLIT,
               M, Fd, E, 1, st.data_ptr(), d["c1"].data_ptr(), eps, None, None)
        L.call("pp_linear_ln_folded", fbuf.data_ptr(), d["w2"].data_ptr(), d["b2"].data_ptr(), xs.data_ptr(), SPLIT, xo.data_ptr(), F32,
               M, E, Fd, 0, None, None, eps, None, None)   # the last layer's form: fp32 rows for the final LayerNorm
        L.call("pp_linear_ln_folded", fbuf.data_ptr(), d["w2"].data_ptr(), d["b2"].data_ptr(), xs.data_ptr(), SPLIT, xs.data_ptr(), SPLIT,
               M, E, Fd, 0, None, None, eps, st.data_ptr(), None)  # in place, statistics for the next ln1
        return x1_dev, st1, _unsp(fbuf), xo.cpu().double(), _unsp(xs), st.cpu().double()

    x1_dev, st1, f_dev, x2_f32, x2_split, st2 = chain()
    torch.testing.assert_close(x1_dev, x1, **TOL)
    torch.testing.assert_close(st1, _row_part_stats(x1), rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(f_dev, f, rtol=3e-5, atol=3e-5)
    torch.testing.assert_close(x2_f32, x2, rtol=3e-5, atol=3e-5)
    torch.testing.assert_close(x2_split, x2, rtol=3e-5, atol=3e-5)
    torch.testing.assert_close(st2, _row_part_stats(x2), rtol=1e-4, atol=1e-4)
    again = chain()
    for a, b in zip((x1_dev, st1, f_dev, x2_f32, x2_split, st2), again):
        assert torch.equal(a, b), "run-to-run difference"
    if M < 1000:  # rows wider than eight 96-column parts take the two-pass form of the statistics: fc2's rows (K = 1536) normalised in front of a layer
        g3, be3 = 1.0 + 0.2 * _rand(Fd, seed=711), 0.2 * _rand(Fd, seed=712)
        w3, b3 = _rand(192, Fd, seed=713, scale=1 / math.sqrt(Fd)), _rand(192, seed=714, scale=0.1)
        fr = f + 2.0  # (an offset: the mean term must matter)
        w3f, c3, b3f = [t.cuda() for t in fold_layernorm(w3, b3, g3, be3)]
        frs, st3 = _sp(fr.float()), _row_part_stats(_unsp(_sp(fr.float()))).float().cuda()
        o3 = torch.full((M, 192), float("nan"), device="cuda")
        L.call("pp_linear_ln_folded", frs.data_ptr(), w3f.data_ptr(), b3f.data_ptr(), None, F32, o3.data_ptr(), F32, M, 192, Fd, 0, st3.data_ptr(),
               c3.data_ptr(), eps, None, None)
        ref3 = F.layer_norm(fr, (Fd,), g3.double(), be3.double(), eps) @ w3.double().t() + b3.double()
        torch.testing.assert_close(o3.cpu().double(), ref3, rtol=3e-5, atol=3e-5)
    assert L.lib.pp_linear_ln_folded_supported(M, E, E, 1) == (2 if M >= 24576 else 1)
    assert L.lib.pp_linear_ln_folded_supported(M, 100, E, 0) == 0 and L.lib.pp_linear_ln_folded_supported(M, E, 800, 1) == 0
    xs = torch.zeros((M, E), device="cuda")
    with pytest.raises(L.ProbPoseLibraryError):  # act aliasing out
        L.call("pp_linear_ln_folded", xs.data_ptr(), d["wp"].data_ptr(), d["bp"].data_ptr(), None, F32, xs.data_ptr(), SPLIT, M, E, E, 0, None, None,
               eps, None, None)
    with pytest.raises(L.ProbPoseLibraryError):  # statistics without the column sums
        L.call("pp_linear_ln_folded", xs.data_ptr(), d["w1f"].data_ptr(), d["b1f"].data_ptr(), None, F32, d["x0"].data_ptr(), SPLIT, M, Fd, E, 1,
               xs.data_ptr(), None, eps, None, None)
    with pytest.raises(L.ProbPoseLibraryError):  # a layer with statistics in takes no residual
        L.call("pp_linear_ln_folded", xs.data_ptr(), d["w1f"].data_ptr(), d["b1f"].data_ptr(), d["x0"].data_ptr(), F32, d["x0"].data_ptr(), SPLIT, M, E, E, 0,
               xs.data_ptr(), d["c1"].data_ptr(), eps, None, None)


@gpu
@pytest.mark.parametrize("M,N,K,ln", [(1, 192, 64, False), (300, 384, 192, True), (50000, 192, 64, False), (50000, 192, 192, True), (777, 576, 1536, True)])
def test_linear_ln_folded_edge_shapes_vs_fp64(M, N, K, ln):
    """pp_linear_ln_folded at the edges of what it serves: a single row; K = 64 (two K-steps: the shortest ring walk); 261 tiles on 256 CUs (five
    workgroups of the tile loop take a second tile, with K = 64 the next tile's stages are requested at the first barriers of this one); statistics
    of 2 / 16 parts; with GELU, an fp32 residual, statistics out. Against torch fp64."""
    from probpose_code_amd.weights import fold_layernorm

    L = _lib()
    eps = 1e-6
    x = _rand(M, K, seed=800) + 1.5 * _rand(M, 1, seed=801)
    w, b = _rand(N, K, seed=802, scale=1 / math.sqrt(K)), _rand(N, seed=803, scale=0.1)
    res = _rand(M, N, seed=804)
    xs = _sp(x)
    if ln:
        g, be = 1.0 + 0.2 * _rand(K, seed=805), 0.2 * _rand(K, seed=806)
        wf, cs, bf = [t.cuda() for t in fold_layernorm(w, b, g, be)]
        st = _row_part_stats(_unsp(xs)).float().cuda()
        out = torch.full((M, N), float("nan"), device="cuda")
        L.call("pp_linear_ln_folded", xs.data_ptr(), wf.data_ptr(), bf.data_ptr(), None, 0, out.data_ptr(), SPLIT, M, N, K, 1, st.data_ptr(), cs.data_ptr(),
               eps, None, None)
        ref = F.gelu(F.layer_norm(x.double(), (K,), g.double(), be.double(), eps) @ w.double().t() + b.double())
        torch.testing.assert_close(_unsp(out), ref, rtol=3e-5, atol=3e-5)
    else:
        wd, bd, rd = _sp(w), b.cuda(), res.cuda()
        out = torch.full((M, N), float("nan"), device="cuda")
        so = torch.full((M, N // 96, 2), float("nan"), device="cuda")
        L.call("pp_linear_ln_folded", xs.data_ptr(), wd.data_ptr(), bd.data_ptr(), rd.data_ptr(), 0, out.data_ptr(), 0, M, N, K, 0, None, None, eps,
               so.data_ptr(), None)
        ref = x.double() @ w.double().t() + b.double() + res.double()
        torch.testing.assert_close(out.cpu().double(), ref, **TOL)
        torch.testing.assert_close(so.cpu().double(), _row_part_stats(ref), rtol=1e-4, atol=1e-4)


@gpu
@pytest.mark.parametrize("M", [96 * 3, 96 * 2 - 40, 24576])
def test_proj_ffn_split_folded_vs_plain_launch(M):
    """pp_proj_ffn_split_folded (the projection + FFN launch inside the folded chain: CENTERED operand-format rows in and / or out, the final LayerNorm
    left to the next layer's pp_qkv_attention_split_folded) against the plain launch on the same numbers: centered residual rows + their means in
    ((hi + lo) + mean: the plain launch's results on those fp32 rows, bit for bit); folded out: h_out holds x - mean in the operand format, stats_out
    (mean, rstd) per row - together the plain launch's x_out to 2^-22; both at once, in place, statistics in place; weight scales; a ragged last
    block; the bs 64 shape; repeated launches bit-identical; shapes it does not serve are refused."""
    from probpose_code_amd.weights import from_split

    L = _lib()
    E, F_, F32, eps = 384, 1536, 0, 1e-6
    h, r, w1, b1, w2, b2, g, be = _ffn_inputs(M, F_, seed=600)
    r = r + 3.0 * _rand(M, 1, seed=601)  # rows with an offset: centering must really happen
    att, wp, bp, g2, be2 = _proj_inputs(M, seed=620)
    packed = _ffn_pack(L, w1, w2, E, F_)
    wpp = torch.empty(E * E, dtype=torch.float32, device="cuda")
    wps = _sp(wp)
    L.call("pp_proj_split_pack_weights", wps.data_ptr(), wpp.data_ptr(), E, None)
    dev = [t.cuda() for t in (bp, g2, be2, b1, b2, g, be)]
    rmean = r.mean(dim=1, keepdim=True)
    rs = _sp(r - rmean)                                        # centered residual rows in the operand format
    rst = torch.cat([rmean, torch.ones_like(rmean)], dim=1).contiguous().cuda()  # (mean, rstd): only the mean is read back
    rq = (from_split(rs.cpu()) + rmean).cuda()                 # ... and the fp32 rows they stand for: (hi + lo) + mean, one rounding
    ad = _sp(att)

    def plain(res32):
        xo, ho, sc = (torch.full((M, E), float("nan"), device="cuda") for _ in range(3))
        L.call("pp_proj_ffn_split_residual_layernorm", ad.data_ptr(), wpp.data_ptr(), dev[0].data_ptr(), dev[1].data_ptr(), dev[2].data_ptr(), sc.data_ptr(),
               packed.data_ptr(), dev[3].data_ptr(), dev[4].data_ptr(), res32.data_ptr(), xo.data_ptr(), dev[5].data_ptr(), dev[6].data_ptr(), eps, ho.data_ptr(),
               M, E, F_, None)
        return xo.cpu(), ho.cpu()

    def folded(res, res_fmt, fold_out, in_place=False, res_stats=None, wpk=None, pk=None, scales=(1.0, 1.0, 1.0)):
        xo, sc = (torch.full((M, E), float("nan"), device="cuda") for _ in range(2))
        ho = res if in_place else torch.full((M, E), float("nan"), device="cuda")
        st = res_stats if (in_place and res_stats is not None) else torch.full((M, 2), float("nan"), device="cuda")
        L.call("pp_proj_ffn_split_folded", ad.data_ptr(), (wpk if wpk is not None else wpp).data_ptr(), dev[0].data_ptr(), dev[1].data_ptr(), dev[2].data_ptr(),
               sc.data_ptr(), (pk if pk is not None else packed).data_ptr(), dev[3].data_ptr(), dev[4].data_ptr(), res.data_ptr(), res_fmt,
               res_stats.data_ptr() if res_stats is not None else None, int(fold_out), None if fold_out else xo.data_ptr(),
               None if fold_out else dev[5].data_ptr(), None if fold_out else dev[6].data_ptr(), eps, ho.data_ptr(), st.data_ptr() if fold_out else None,
               M, E, F_, scales[0], scales[1], scales[2], None)
        return xo.cpu(), ho.cpu(), st.cpu()

    want_x, want_h = plain(rq)
    # 1: centered operand-format residual in, LayerNorm out (the last layer of a folded chain)
    x1, h1, _ = folded(rs, SPLIT, False, res_stats=rst)
    assert torch.equal(x1, want_x) and torch.equal(h1.view(torch.int32), want_h.view(torch.int32)), "centered residual in: must equal the plain launch on (hi + lo) + mean"
    # 2: fp32 residual in, folded out (the first layer): centered rows + statistics
    _, h2, st2 = folded(rq, F32, True)
    ref_stats = torch.stack([want_x.double().mean(dim=1), 1.0 / torch.sqrt(want_x.double().var(dim=1, unbiased=False) + eps)], dim=1)
    torch.testing.assert_close(st2.double(), ref_stats, rtol=2e-5, atol=2e-5)
    torch.testing.assert_close(_unsp(h2) + st2[:, :1].double(), want_x.double(), rtol=1e-6, atol=1e-6)  # (x - mean rounds once more than x)
    assert _unsp(h2).mean(dim=1).abs().max() < 1e-5, "the rows that leave must be centered"
    # 3: both, in place (the layers in between): the residual buffer becomes the new rows, the statistics buffer the new statistics
    rs3, rst3 = rs.clone(), rst.clone()
    _, h3, st3 = folded(rs3, SPLIT, True, in_place=True, res_stats=rst3)
    assert torch.equal(h3.view(torch.int32), h2.view(torch.int32)) and torch.equal(st3, st2)
    rs4, rst4 = rs.clone(), rst.clone()
    _, h4, st4 = folded(rs4, SPLIT, True, in_place=True, res_stats=rst4)
    assert torch.equal(h4.view(torch.int32), h3.view(torch.int32)) and torch.equal(st4, st3), "run-to-run difference"
    # 4: weight scales (Wp 2^15, W1 2^17, W2 2^16): the same rows to rounding
    pk_s = _ffn_pack(L, w1 * 2.0 ** 17, w2 * 2.0 ** 16, E, F_)
    wpp_s = torch.empty(E * E, dtype=torch.float32, device="cuda")
    L.call("pp_proj_split_pack_weights", _sp(wp * 2.0 ** 15).data_ptr(), wpp_s.data_ptr(), E, None)
    _, h5, st5 = folded(rs, SPLIT, True, res_stats=rst, wpk=wpp_s, pk=pk_s, scales=(2.0 ** -15, 2.0 ** -17, 2.0 ** -16))
    torch.testing.assert_close(_unsp(h5) + st5[:, :1].double(), _unsp(h2) + st2[:, :1].double(), rtol=5e-6, atol=5e-6)
    assert L.launch_count("ffn_dma_fold") >= 5
    with pytest.raises(L.ProbPoseLibraryError):  # an odd number of hidden chunks: no paired kernel
        L.call("pp_proj_ffn_split_folded", ad.data_ptr(), wpp.data_ptr(), dev[0].data_ptr(), dev[1].data_ptr(), dev[2].data_ptr(), rs4.data_ptr(),
               packed.data_ptr(), dev[3].data_ptr(), dev[4].data_ptr(), rs.data_ptr(), SPLIT, rst.data_ptr(), 1, None, None, None, eps, rs3.data_ptr(),
               rst4.data_ptr(), M, E, 384, 1.0, 1.0, 1.0, None)
    with pytest.raises(L.ProbPoseLibraryError):  # folded out without a statistics buffer
        L.call("pp_proj_ffn_split_folded", ad.data_ptr(), wpp.data_ptr(), dev[0].data_ptr(), dev[1].data_ptr(), dev[2].data_ptr(), rs4.data_ptr(),
               packed.data_ptr(), dev[3].data_ptr(), dev[4].data_ptr(), rs.data_ptr(), SPLIT, rst.data_ptr(), 1, None, None, None, eps, rs3.data_ptr(), None,
               M, E, F_, 1.0, 1.0, 1.0, None)
    with pytest.raises(L.ProbPoseLibraryError):  # centered rows without their means
        L.call("pp_proj_ffn_split_folded", ad.data_ptr(), wpp.data_ptr(), dev[0].data_ptr(), dev[1].data_ptr(), dev[2].data_ptr(), rs4.data_ptr(),
               packed.data_ptr(), dev[3].data_ptr(), dev[4].data_ptr(), rs.data_ptr(), SPLIT, None, 1, None, None, None, eps, rs3.data_ptr(), rst4.data_ptr(),
               M, E, F_, 1.0, 1.0, 1.0, None)
