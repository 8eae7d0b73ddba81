#!/usr/bin/env python
"""Fuzz of pp_gemm (all three precisions: the dispatcher picks among the 128 x 128 kernel, the wide-tile kernels and the twelve-wave Linear kernel by
shape) over random shapes against torch fp64: ragged M, N from 17 to 3 072, K from 64 to 3 072, bias / GELU / ReLU / fp32 residual (in place or
broadcast table) / every output format / planar planes; outputs between canaries.   python tests/fuzz_gemm.py [seconds]"""
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from probpose_code_amd import _lib as L  # noqa: E402
from probpose_code_amd.weights import from_split, to_split  # noqa: E402

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
CAN = 8192


def gelu64(x):
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


n_cases, bad, seed = 0, 0, 0
t_end = time.time() + seconds
while time.time() < t_end:
    g = torch.Generator().manual_seed(40000 + seed)
    rng = np.random.default_rng(40000 + seed)
    seed += 1
    prec = int(rng.integers(0, 3))  # bf16, f32, f16x3
    M = int(rng.choice([1, 17, 127, 128, 129, 191, 192, 193, 383, 384, 1000, 3071, 6144, 24576, int(rng.integers(1, 30000))]))
    K = int(rng.choice([64, 128, 256, 384, 768, 1536, 3072]))
    planar = rng.random() < 0.15
    if planar:
        N, P = 17, int(rng.choice([64, 3072]))
        M = P * max(1, M // P)
    elif prec == 2:
        N = int(rng.choice([32, 64, 192, 384, 768, 1152, 1536, 2304, 3072]))
    else:
        N = int(rng.choice([17, 64, 100, 192, 384, 768, 1152, 1536, 3072]))
    if M * max(N, K) > 60e6:
        M = max(1, int(60e6 // max(N, K)))
        if planar:
            M = P * max(1, M // P)
    a, w = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / math.sqrt(K)
    bias = torch.randn(N, generator=g) * 0.3 if rng.random() < 0.8 else None
    act = int(rng.integers(0, 3))
    out_fmt = 0 if (planar or rng.random() < 0.5) else (1 if prec == 0 else (2 if prec == 2 else 0))
    if out_fmt == 2 and N % 32:
        out_fmt = 0
    res_kind = 0 if planar else int(rng.integers(0, 3))
    res_mod = int(rng.choice([1, 7, 192])) if res_kind == 2 else 0
    res = torch.randn(res_mod if res_kind == 2 else M, N, generator=g) if res_kind else None
    if prec == 0:
        a_q, w_q = a.bfloat16(), w.bfloat16()
        ad, wd = a_q.cuda(), w_q.cuda()
        ref = a_q.double() @ w_q.double().t()
        tol = 2e-2 if out_fmt == 1 else 2e-3
    elif prec == 1:
        ad, wd = a.cuda(), w.cuda()
        ref = a.double() @ w.double().t()
        tol = 2e-4
    else:
        ad, wd = to_split(a).cuda(), to_split(w).cuda()
        ref = a.double() @ w.double().t()
        tol = 3e-5
    if bias is not None:
        ref = ref + bias.double()
    if act == 1:
        ref = gelu64(ref)
    elif act == 2:
        ref = torch.relu(ref)
    if res is not None:
        ref = ref + (res.double()[torch.arange(M) % res_mod] if res_kind == 2 else res.double())
    n_out = M * N
    odt = torch.bfloat16 if out_fmt == 1 else torch.float32
    buf = torch.full((n_out + 2 * CAN,), float("nan"), dtype=odt, device="cuda")
    out = buf[CAN:CAN + n_out]
    rd = None
    if res_kind == 1 and out_fmt == 0:
        out.copy_(res.reshape(-1).cuda())
        rd = out
    elif res_kind:
        rd = res.cuda()
    bd = bias.cuda() if bias is not None else None
    try:
        L.call("pp_gemm", prec, ad.data_ptr(), wd.data_ptr(), L.ptr(bd), L.ptr(rd), res_mod, out.data_ptr(), M, N, K, K, K, N, act, out_fmt,
               P if planar else 0, None)
    except L.ProbPoseLibraryError as exc:  # a refusal is fine (documented constraints); a wrong result is not
        if "UNSUPPORTED" in str(exc) or "INVALID" in str(exc):
            continue
        raise
    torch.cuda.synchronize()
    if planar:
        got = out.view(M // P, N, P).permute(0, 2, 1).reshape(M, N).cpu().double()
    elif out_fmt == 2:
        got = from_split(out.view(M, N).cpu()).double()
    else:
        got = out.view(M, N).cpu().double()
    ok = bool(torch.isnan(buf[:CAN].float()).all() and torch.isnan(buf[CAN + n_out:].float()).all()) and torch.allclose(got, ref, rtol=tol, atol=tol)
    n_cases += 1
    if not ok:
        bad += 1
        print(f"MISMATCH seed {seed - 1} prec {prec} M {M} N {N} K {K} act {act} out {out_fmt} res {res_kind}/{res_mod} planar {planar}: "
              f"max err {float((got - ref).abs().max()):.2e}", flush=True)
print(f"{n_cases} launches in {seconds:.0f} s, {bad} mismatches")
print("GEMM FUZZ", "FAILED" if bad else "OK")
