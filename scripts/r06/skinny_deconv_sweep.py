#!/usr/bin/env python
"""pp_skinny_deconv against pp_conv_gemm's all-phases deconvolution for the two deconvolutions of the head (384 -> 256 on 16 x 12, 256 -> 256 on 32 x 24)
at small batches, per tile shape (option "skinny_tile"); us per launch inside a replayed graph of 20 launches."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from probpose_code_amd import _lib as L  # noqa: E402
from probpose_code_amd.weights import to_split  # noqa: E402


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / 100


for B in (1, 2, 4, 8, 16):
    nb = 2 * B
    for name, H, W, Cin in (("deconv1", 16, 12, 384), ("deconv2", 32, 24, 256)):
        x = to_split(torch.randn(nb, H, W, Cin)).cuda()
        w = to_split(torch.randn(4, 256, 4 * Cin) * 0.03).cuda()
        b = torch.randn(256).cuda()
        out = torch.zeros(nb, 2 * H, 2 * W, 256, device="cuda")
        res = []
        for code in (11, 22, 12, 32):
            L.set_option("skinny_tile", code)
            res.append(f"{timed(lambda: L.call('pp_skinny_deconv', x.data_ptr(), w.data_ptr(), b.data_ptr(), out.data_ptr(), nb, H, W, Cin, 256, L.stream_ptr())):6.1f}")
        L.set_option("skinny_tile", 0)
        auto = timed(lambda: L.call('pp_skinny_deconv', x.data_ptr(), w.data_ptr(), b.data_ptr(), out.data_ptr(), nb, H, W, Cin, 256, L.stream_ptr()))
        gen = timed(lambda: L.call("pp_conv_gemm", 2, 2, x.data_ptr(), w.data_ptr(), b.data_ptr(), out.data_ptr(), nb, H, W, Cin, 256, -1, -1, 1, 0, 0, 0, 0, 256, 2, 2, L.stream_ptr()))
        print(f"B {B:2d} {name} rows {nb * H * W:6d}: skinny 32x32 / 64x64 / 32x64 / 96x64: {' | '.join(res)}  rule {auto:6.1f}   pp_conv_gemm {gen:6.1f} us", flush=True)
