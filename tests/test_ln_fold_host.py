"""CPU: the host side of the folded-LayerNorm layer plan (weights.fold_layernorm) - the algebra pp_linear_ln_folded's epilogue relies on.

mmpretrain TransformerEncoderLayer [3P]: ``x + attn(ln1(x))``, ``ffn(ln2(x)) + x``. The plan runs ``Linear(LayerNorm(x))`` as
``rstd * (x @ W'^T - mean * colsum(W')) + b'`` with ``W' = W * gamma``, ``b' = b + W @ beta`` (include/probpose_mi355x.h), the row statistics
combined from 96-column (mean, M2) parts as the producing launch leaves them. No GPU: the split-fp16 container is emulated with
``from_split`` (hi + lo), sums in fp64."""
import math

import torch
import torch.nn.functional as F

from probpose_code_amd.weights import fold_layernorm, from_split, to_split


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def test_fold_layernorm_matches_linear_of_layernorm():
    M, K, N, eps = 37, 768, 192, 1e-6
    x = _rand(M, K, seed=1) * 1.7 + 3.0 * _rand(M, 1, seed=2)  # rows with an offset: the mean term must carry weight
    w, b = _rand(N, K, seed=3, scale=1 / math.sqrt(K)), _rand(N, seed=4, scale=0.1)
    gamma, beta = 1.0 + 0.3 * _rand(K, seed=5), 0.3 * _rand(K, seed=6)
    ref = F.layer_norm(x.double(), (K,), gamma.double(), beta.double(), eps) @ w.double().t() + b.double()
    wf, colsum, bias = fold_layernorm(w, b, gamma, beta)
    assert wf.shape == (N, K) and wf.dtype == torch.float32 and colsum.shape == (N,) and bias.shape == (N,)
    wq = from_split(wf).double()                      # what the MFMAs multiply with: the split-ROUNDED folded weights
    xq = from_split(to_split(x)).double()             # ... and the split-rounded raw rows
    assert torch.equal(colsum, wq.sum(dim=1).float()), "column sums are taken of the rounded weights"
    # the statistics as the producer leaves them: per row and 96-column part (mean, sum of squared deviations), combined with Chan's formula
    parts = xq.reshape(M, K // 96, 96)
    mean_p, m2_p = parts.mean(dim=2), ((parts - parts.mean(dim=2, keepdim=True)) ** 2).sum(dim=2)
    mean = mean_p.mean(dim=1)
    m2 = (m2_p + 96.0 * (mean_p - mean[:, None]) ** 2).sum(dim=1)
    torch.testing.assert_close(mean, xq.mean(dim=1), rtol=1e-12, atol=1e-12)
    torch.testing.assert_close(m2 / K, xq.var(dim=1, unbiased=False), rtol=1e-11, atol=1e-12)
    rstd = 1.0 / torch.sqrt(m2 / K + eps)
    got = rstd[:, None] * (xq @ wq.t() - mean[:, None] * colsum.double()[None, :]) + bias.double()[None, :]
    # differences left: the split rounding of x and W' (2^-22 relative each), nothing from the algebra
    torch.testing.assert_close(got, ref, rtol=2e-6, atol=2e-6)


def test_fold_layernorm_identity_affine_is_plain_linear():
    K, N = 192, 192
    w, b = _rand(N, K, seed=7), _rand(N, seed=8)
    wf, colsum, bias = fold_layernorm(w, b, torch.ones(K), torch.zeros(K))
    assert torch.equal(wf, to_split(w)) and torch.equal(bias, b)
    torch.testing.assert_close(colsum.double(), from_split(to_split(w)).double().sum(dim=1), rtol=1e-7, atol=1e-6)


def test_weight_scale_exponent_and_pack_scales():
    """weights.weight_scale_exponent: the tensor's largest element lands in [2^12, 2^13); pack(split=True) stores the backbone's Linear weights that
    way and remembers 2^-e; the stored values times 2^-e are the weights to the split format's 2^-22 - also for rows of 1e-3 of the width, whose low
    halves would be fp16 subnormals unscaled."""
    from probpose_code_amd import synthetic as S
    from probpose_code_amd import weights as W

    for m in (1e-4, 0.03, 0.9, 1.0, 4096.0, 7000.0, 9000.0):
        e = W.weight_scale_exponent(torch.tensor([m, -m / 3]))
        assert 2.0 ** 12 <= m * 2.0 ** e < 2.0 ** 13, (m, e)
    assert W.weight_scale_exponent(torch.zeros(4)) == 0
    arch = dict(embed_dims=384, num_layers=2, num_heads=12, feedforward_channels=1536)
    sd = S.synthetic_state_dict(arch, seed=0, stats="trained")
    pw = W.pack(sd, torch.float32, "cpu", split=True)
    plain = W.pack(sd, torch.float32, "cpu", split=True, scale_linear=False)
    assert all(v == 1.0 for v in plain.inv_scale.values()) and plain.inv("l1.fc1.w") == 1.0
    for name, key in (("l1.qkv.w", "attn.qkv.weight"), ("l1.proj.w", "attn.proj.weight"), ("l1.fc1.w", "ffn.layers.0.0.weight"),
                      ("l1.fc2.w", "ffn.layers.1.weight")):
        w = sd["backbone.layers.1." + key].double()
        inv = pw.inv(name)
        assert inv < 1.0 and math.log2(inv) == round(math.log2(inv))
        got = from_split(pw[name]).double() * inv
        got_plain = from_split(plain[name]).double()
        small = w.abs() < 0.125 * w.abs().max() * 2.0 ** -10  # elements whose unscaled low halves are deep in the subnormals
        err_s, err_p = (got - w).abs(), (got_plain - w).abs()
        assert (err_s <= w.abs() * 2.0 ** -21 + 2.0 ** -25 * inv).all()
        assert err_s[small].max() <= err_p[small].max() / 8, "scaling must buy the small elements their low halves back"
    assert "l1.qkv.wf" in pw.inv_scale and "l1.qkv.cf" not in pw.t  # the ViT-S chain's folded projection: centered rows, no column sums


def test_centered_fold_is_linear_of_layernorm_without_a_mean_term():
    """The ViT-S chain's form (pp_proj_ffn_split_folded -> pp_qkv_attention_split_folded): rows travel as x - mean with (mean, rstd) beside them, the
    projection is rstd * ((x - mean) @ W'^T) * 2^-e + b'. Against Linear(LayerNorm(x)) in fp64, rows with |mean| / std = 100."""
    from probpose_code_amd.weights import weight_scale_exponent

    M, K, N, eps = 29, 384, 96, 1e-6
    x = _rand(M, K, seed=11) * 1.3 + 130.0
    w, b = _rand(N, K, seed=13, scale=1 / math.sqrt(K)), _rand(N, seed=14, scale=0.1)
    gamma, beta = 1.0 + 0.3 * _rand(K, seed=15), 0.3 * _rand(K, seed=16)
    ref = F.layer_norm(x.double(), (K,), gamma.double(), beta.double(), eps) @ w.double().t() + b.double()
    e = weight_scale_exponent(w.double() * gamma.double()[None, :])
    wf, _, bias = fold_layernorm(w, b, gamma, beta, scale_exp=e)
    mean = x.double().mean(dim=1, keepdim=True)
    xc = from_split(to_split((x.double() - mean).float())).double()
    rstd = 1.0 / torch.sqrt(x.double().var(dim=1, unbiased=False, keepdim=True) + eps)
    got = rstd * (xc @ from_split(wf).double().t()) * 2.0 ** -e + bias.double()
    torch.testing.assert_close(got, ref, rtol=1e-6, atol=1e-6)
