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
