"""dev only: first tower stage (conv 3x3 + pool, split fp16) at bs 64: tile order (pp_set_option psplit_conv_weight_major) x K walk
(psplit_tap_inner) A/B; launches go in blocks of 13 per combination, in the order of COMBOS (scripts/micro/conv1_fetch.sh relies on it)."""
import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from probpose_code_amd import _lib as L
from probpose_code_amd.weights import to_split
G, C, H, W, B = 4, 384, 16, 12, 128
x = to_split(torch.randn(B, H, W, C)).cuda(); w = to_split(torch.randn(G, C, 9 * C) / math.sqrt(9 * C)).cuda(); b = torch.randn(G, C).cuda()
pooled = torch.empty((G, B, 4, 4, C), device="cuda"); scratch = torch.empty((G, B, H, W, C), device="cuda")
def run():
    L.call("pp_conv3x3_maxpool_relu", 2, x.data_ptr(), w.data_ptr(), b.data_ptr(), pooled.data_ptr(), scratch.data_ptr(), B, H, W, C, C, 4, 3, G, 0, C * 9 * C, C, 2, None)
COMBOS = [(0, 0), (1, 0), (1, 1), (0, 1)]  # (tap_inner, weight_major)
res = {}
for rep in range(4):
    for wm in COMBOS:
        L.set_option("psplit_tap_inner", wm[0])
        L.set_option("psplit_conv_weight_major", wm[1])
        for _ in range(3): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): run()
        e1.record(); torch.cuda.synchronize()
        res.setdefault(wm, []).append(e0.elapsed_time(e1) / 10 * 1e3)
        if rep == 0: res.setdefault(("out", wm), pooled.clone())
L.set_option("psplit_conv_weight_major", 0); L.set_option("psplit_tap_inner", 0)
for wm in COMBOS: print(f"tap_inner={wm[0]} weight_major={wm[1]}: min {min(res[wm]):7.1f} us   max |out - out(0,0)| = {(res[('out', wm)] - res[('out', (0, 0))]).abs().max().item():.3e}")
