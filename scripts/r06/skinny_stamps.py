#!/usr/bin/env python
"""dev: where a pp_skinny_linear launch with a LayerNorm tail spends its time - s_memtime stamps (core-clock counts, ~2 GHz: divided by 2 000 for "us") of wave 0 of the workgroup that normalises
row block 0. Library built with -DSK_STAMP=1 (scripts/micro/build/lib_skstamp.so, copied over the in-tree library ON THE GPU BOX)."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from probpose_code_amd import _lib as L  # noqa: E402
from probpose_code_amd.weights import to_split  # noqa: E402

names = ["start -> K loop done", "epilogue: residual load, stores issued", "stores acknowledged (vmcnt 0)", "barrier, counter atomic, barrier", "tail: loads, LayerNorm, stores issued",
         "tail stores acknowledged"]
for B, N, K, code in ((1, 384, 384, 11), (1, 384, 384, 33), (1, 384, 1536, 11), (8, 384, 384, 23), (8, 384, 1536, 23), (8, 384, 384, 33)):
    M = B * 384
    a, w = to_split(torch.randn(M, K)).cuda(), to_split(torch.randn(N, K) * 0.05).cuda()
    bias, g = torch.randn(N).cuda(), torch.ones(N).cuda()
    x = torch.zeros(M, N, device="cuda")
    h = torch.zeros(M, N, device="cuda")
    cnt = torch.zeros((M + 31) // 32, dtype=torch.int32, device="cuda")
    L.set_option("skinny_tile", code)
    rows = []
    for rep in range(6):
        for _ in range(3):
            L.call("pp_skinny_linear", a.data_ptr(), w.data_ptr(), bias.data_ptr(), x.data_ptr(), 0, x.data_ptr(), 0, M, N, K, 0, 1.0, g.data_ptr(), bias.data_ptr(),
                   1e-6, h.data_ptr(), cnt.data_ptr(), L.stream_ptr())
        torch.cuda.synchronize()
        buf = (ctypes.c_ulonglong * 16)()
        L.lib.pp_dev_sk_stamps.restype = ctypes.c_int
        assert L.lib.pp_dev_sk_stamps(buf) == 0
        t = [buf[i] for i in range(7)]
        rows.append([(t[i + 1] - t[i]) / 2000.0 for i in range(6)])
    med = [sorted(r[i] for r in rows)[len(rows) // 2] for i in range(6)]
    print(f"B {B} N {N} K {K} tile {code}: total {sum(med):6.2f} us (median of 6 back-to-back launches)")
    for nm, v in zip(names, med):
        print(f"    {nm:42s} {v:6.2f} us")
L.set_option("skinny_tile", 0)
