#!/usr/bin/env python
"""pp_skinny_linear: time per launch for the Linear shapes of the small-batch plan under each tile edge (option "skinny_tile"), B = 1 .. 24 crops +
flip - the measurements behind pick_tile's cost rule. 50 back-to-back launches between two events (same stream: every launch waits for the one
before, as in the replayed step)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from probpose_code_amd import _lib as L  # noqa: E402
from probpose_code_amd.weights import to_split  # noqa: E402

shapes = [("proj+ln", 384, 384, True), ("fc1", 1536, 384, False), ("fc2+ln", 384, 1536, True), ("qkv", 1152, 384, False)]
for B in (1, 2, 4, 8, 16, 24):
    M = B * 2 * 192
    for name, N, K, ln in shapes:
        a, w = to_split(torch.randn(M, K)).cuda(), to_split(torch.randn(N, K) * 0.05).cuda()
        bias, g = torch.randn(N).cuda(), torch.ones(N).cuda()
        x = torch.zeros(M, N, device="cuda")
        h = torch.zeros(M, N, device="cuda")
        cnt = torch.zeros((M + 31) // 32, dtype=torch.int32, device="cuda")
        res = []
        for code in (11, 22, 33, 13, 12, 23):
            t, tn = 32 * (code // 10), 32 * (code % 10)
            if N % tn:
                res.append("    -  ")
                continue
            L.set_option("skinny_tile", code)

            def go():
                if ln:
                    L.call("pp_skinny_linear", a.data_ptr(), w.data_ptr(), bias.data_ptr(), x.data_ptr(), 0, x.data_ptr(), 0, M, N, K, 0, 1.0, g.data_ptr(),
                           bias.data_ptr(), 1e-6, h.data_ptr(), cnt.data_ptr(), L.stream_ptr())
                else:
                    L.call("pp_skinny_linear", a.data_ptr(), w.data_ptr(), bias.data_ptr(), None, 0, h.data_ptr(), 2, M, N, K, 1, 1.0, None, None, 1e-6, None,
                           None, L.stream_ptr())
            for _ in range(5):
                go()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            g_ = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g_):
                for _ in range(20):
                    go()
            g_.replay()
            torch.cuda.synchronize()
            e0.record()
            for _ in range(5):
                g_.replay()
            e1.record()
            torch.cuda.synchronize()
            wgs = ((M + t - 1) // t) * (N // tn)
            res.append(f"{e0.elapsed_time(e1) * 1e3 / 100:6.1f} ({wgs:4d})")
        L.set_option("skinny_tile", 0)
        auto = L.lib.pp_skinny_linear_tile(M, N, K, int(ln))
        print(f"B {B:2d} M {M:5d} {name:8s} N {N:4d} K {K:4d}: us per launch (workgroups) at 32x32 / 64x64 / 96x96 / 32x96 / 32x64 / 64x96: {' | '.join(res)}   rule picks {auto}", flush=True)
