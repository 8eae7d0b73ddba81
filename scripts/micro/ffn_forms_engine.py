"""dev: the whole engine at several batch sizes, eight-wave against twelve-wave feed-forward launch: backbone features compared."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from probpose_code_amd import _lib as L, synthetic as S
from probpose_code_amd.engine import ProbPoseEngine

eng = ProbPoseEngine(S.synthetic_state_dict("small", seed=0, logit_scale=2.0), 12, precision="f16x3", device="cuda:0")
for B in [int(a) for a in sys.argv[1:]] or [1, 2, 3, 4, 8, 64]:
    crops = S.synthetic_crops(B, seed=5).cuda()
    outs = {}
    for form in (0, 1):
        L.set_option("ffn_dma_waves", form)
        f = eng.export_features(eng.run_backbone(crops, True)).float().clone()
        torch.cuda.synchronize()
        outs[form] = f
    d = (outs[0] - outs[1]).abs()
    bad = torch.isnan(outs[1]).sum().item()
    rows = d.reshape(-1, d.shape[-1]).amax(1)
    print(f"B={B}: features {tuple(outs[0].shape)} max |diff| {d.max().item():.3e}, NaN in the twelve-wave form {bad}, rows off by > 1e-3: {(rows > 1e-3).sum().item()} of {rows.numel()}: {(rows > 1e-3).nonzero().flatten()[:16].tolist()}")
L.set_option("ffn_dma_waves", 1)
