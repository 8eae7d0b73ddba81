import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from probpose_code_amd import _lib as L
from probpose_code_amd.codecs import oks_kernel_taps
K, H, W = 17, 64, 48
taps, radius = oks_kernel_taps(K, H, W)
td, rd = torch.from_numpy(taps).cuda(), torch.from_numpy(radius).cuda()
fi = torch.tensor([0, 2, 1, 4, 3, 6, 5, 8, 7, 10, 9, 12, 11, 14, 13, 16, 15], dtype=torch.int32).cuda()
for B in (64, 512):
    g = torch.Generator().manual_seed(0)
    sm = torch.nn.functional.interpolate(torch.randn(2 * B, K, 16, 12, generator=g), size=(H, W), mode="bicubic")
    logits = (2.0 * sm + 0.3 * torch.randn(2 * B, K, H, W, generator=g)).cuda().contiguous()
    probs = torch.rand(2 * B, K, H, W, device="cuda") ** 8
    locs = torch.empty(B, K, 2, device="cuda"); kp = torch.empty(B, K, 2, dtype=torch.float64, device="cuda"); sc = torch.empty(B, K, device="cuda")
    def run_head():
        L.call("pp_probmap_head_decode", logits.data_ptr(), logits[B:].data_ptr(), fi.data_ptr(), td.data_ptr(), rd.data_ptr(), B, K, H, W, 192.0, 256.0, 0.5, 1.0, None, None, locs.data_ptr(), kp.data_ptr(), sc.data_ptr(), None)
    def run_dec():
        L.call("pp_probmap_decode", probs.data_ptr(), probs[B:].data_ptr(), fi.data_ptr(), td.data_ptr(), rd.data_ptr(), B, K, H, W, 192.0, 256.0, None, None, locs.data_ptr(), kp.data_ptr(), sc.data_ptr(), None)
    for name, fn in (("sparsemax+decode", run_head), ("decode only", run_dec)):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        byts = 2 * B * K * H * W * 4
        print(f"B={B} {name:18s}: {ms*1e3:7.1f} us  {byts/ms/1e6:7.1f} GB/s ({byts/ms/1e6/8000*100:.1f}% of 8 TB/s)")
