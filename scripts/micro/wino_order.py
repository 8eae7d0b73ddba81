"""dev only: pp_conv3x3_winograd_maxpool_relu at the bench shape (128 images of 16 x 12 x 384, four towers) per tile order
(option wino_order: 0 = column tiles fastest, k = k column tiles x 32 / k row blocks per 32-workgroup super tile)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from probpose_code_amd import _lib
from probpose_code_amd.weights import to_split, winograd_weights
B, H, W, C, G = int(os.environ.get("WB", 128)), int(os.environ.get("WH", 16)), int(os.environ.get("WW", 12)), int(os.environ.get("WC", 384)), 4
g = torch.Generator().manual_seed(0)
x = to_split(torch.randn(B, H, W, C, generator=g)).cuda()
u = to_split(torch.stack([winograd_weights(torch.randn(C, C, 3, 3, generator=g) / (9 * C) ** 0.5) for _ in range(G)])).cuda()
b = torch.randn(G, C, generator=g).cuda()
scratch = torch.empty(_lib.lib.pp_winograd_scratch_bytes(B, H, W, C), dtype=torch.uint8, device="cuda")
out = torch.empty((G, B, H // 4, W // 3, C), device="cuda")
def go():
    _lib.call("pp_conv3x3_winograd_maxpool_relu", x.data_ptr(), u.data_ptr(), b.data_ptr(), scratch.data_ptr(), out.data_ptr(), B, H, W, C, C, 4, 3, G, None)
ref = None
for order in [int(v) for v in os.environ.get("ORDERS", "0 2 4 8 16 0 4").split()]:
    _lib.set_option("wino_order", order)
    for _ in range(3): go()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): go()
    e1.record(); torch.cuda.synchronize()
    if ref is None: ref = out.clone()
    print(f"wino_order {order:2d}: {e0.elapsed_time(e1) / 10 * 1e3:7.1f} us (transform + GEMM)   same bits as order 0: {torch.equal(out, ref)}")
