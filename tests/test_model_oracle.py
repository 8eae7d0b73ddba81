"""CPU: known-answer tests that pin the (parity-unpinned) restatements in oracle/model_ref.py to their
published definitions -- Sparsemax (Martins & Astudillo 2016), the head's layer order, the flip merge."""
import numpy as np
import torch
import torch.nn.functional as F

from oracle import decode_ref as D
from oracle import model_ref as M


def test_sparsemax_known_answers():
    # rows on the simplex, support shrinks with the gap, constant shift invariance
    z = torch.tensor([[0.1, 1.1, 0.2, 0.3], [3.0, 0.0, 0.0, 0.0], [0.5, 0.5, 0.5, 0.5], [2.0, 1.5, -1.0, 0.0]])
    p = M.sparsemax(z)
    assert torch.allclose(p.sum(-1), torch.ones(4), atol=1e-6)
    assert torch.equal(p[1], torch.tensor([1.0, 0.0, 0.0, 0.0]))           # one-hot limit for large gaps
    assert torch.allclose(p[2], torch.full((4,), 0.25))                     # uniform for equal logits
    assert torch.allclose(p[3], torch.tensor([0.75, 0.25, 0.0, 0.0]))       # tau = (3.5 - 1) / 2 = 1.25
    assert torch.allclose(M.sparsemax(z + 7.5), p, atol=1e-6)               # shift invariance
    # closed form for two logits: p = clip((1 + z0 - z1) / 2, 0, 1)
    z2 = torch.tensor([[0.3, -0.1]])
    assert torch.allclose(M.sparsemax(z2), torch.tensor([[0.7, 0.3]]), atol=1e-6)


def test_sparsemax_is_euclidean_projection():
    g = torch.Generator().manual_seed(0)
    z = torch.randn(5, 300, generator=g) * 2
    p = M.sparsemax(z)
    assert (p >= 0).all() and torch.allclose(p.sum(-1), torch.ones(5), atol=1e-5)
    # KKT: on the support z - p is a constant (tau); off the support z <= tau
    for r in range(5):
        s = p[r] > 0
        tau = (z[r][s] - p[r][s])
        assert (tau.max() - tau.min()) < 1e-5
        assert (z[r][~s] <= tau.mean() + 1e-6).all()


def test_head_order_temperature_then_sparsemax_then_clamp():
    """probmap_head.py:637-646: x / 0.5 -> Sparsemax(dim=-1) over H*W -> * normalize -> clamp(0, 1)."""
    g = torch.Generator().manual_seed(1)
    sd = {
        "head.deconv_layers.0.weight": torch.randn(8, 4, 4, 4, generator=g) * 0.3,
        "head.deconv_layers.1.weight": torch.rand(4, generator=g) + 0.5, "head.deconv_layers.1.bias": torch.randn(4, generator=g),
        "head.deconv_layers.1.running_mean": torch.randn(4, generator=g), "head.deconv_layers.1.running_var": torch.rand(4, generator=g) + 0.5,
        "head.final_layer.weight": torch.randn(3, 4, 1, 1, generator=g), "head.final_layer.bias": torch.randn(3, generator=g),
    }
    feat = torch.randn(2, 8, 4, 3, generator=g)
    hm, logits = M.head_heatmap(sd, feat, return_logits=True)
    assert hm.shape == (2, 3, 8, 6)
    manual = M.sparsemax((logits / 0.5).reshape(2, 3, -1)).clamp(0, 1).reshape(2, 3, 8, 6)
    assert torch.equal(hm, manual)
    assert torch.allclose(hm.sum((-1, -2)), torch.ones(2, 3), atol=1e-5)
    # deconv spec: k4 s2 p1 doubles the size; BN uses running stats (eval)
    x = F.conv_transpose2d(feat, sd["head.deconv_layers.0.weight"], stride=2, padding=1)
    assert x.shape[-2:] == (8, 6)


def test_flip_merge_matches_tta():
    """(htm + flip_heatmaps(htm_flip)) * 0.5 with flip_indices (probmap_head.py:757-763)."""
    rng = np.random.default_rng(2)
    a, b = rng.random((2, 17, 4, 6), dtype=np.float32), rng.random((2, 17, 4, 6), dtype=np.float32)
    avg = D.tta_average(a, b)
    fi = list(D.COCO_FLIP_INDICES)
    assert np.array_equal(avg[1, 5], ((a[1, 5] + b[1, fi[5], :, ::-1]) * np.float32(0.5)))
    assert fi[5] == 6 and fi[0] == 0


def test_vit_shapes_and_token_order():
    from probpose_code_amd import synthetic as S

    sd = S.synthetic_state_dict(dict(embed_dims=64, num_layers=2, num_heads=2, feedforward_channels=128),
                                img_size=(64, 48), seed=0)
    x = torch.randn(2, 3, 64, 48)
    f = M.vit_forward(sd, x, num_heads=2)
    assert f.shape == (2, 64, 4, 3)  # (64 + 4 - 16)//16 + 1 = 4, (48 + 4 - 16)//16 + 1 = 3
    # per-token LayerNorm at the end: every spatial position is normalised over channels
    g, b = sd["backbone.ln1.weight"], sd["backbone.ln1.bias"]
    tok = ((f.permute(0, 2, 3, 1) - b) / g)
    assert torch.allclose(tok.mean(-1), torch.zeros(2, 4, 3), atol=1e-4)
