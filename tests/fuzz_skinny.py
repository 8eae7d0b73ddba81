#!/usr/bin/env python
"""Fuzz of pp_skinny_linear / pp_skinny_deconv / pp_skinny_conv1x1_planar over random shapes against torch fp64: M from 1 to ~7 000 (ragged against every
tile edge), N and K over the multiples the entry points admit, every epilogue (bias, GELU / ReLU, fp32 residual in place or broadcast table,
LayerNorm tail, split or fp32 rows out), every tile shape forced in turn; the outputs sit between canaries (a write outside the tensor is caught),
the LayerNorm counters must be back at zero.   python tests/fuzz_skinny.py [seconds]"""
import math
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from probpose_code_amd import _lib as L  # noqa: E402
from probpose_code_amd.weights import from_split, to_split  # noqa: E402

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
CAN = 4096  # canary elements on each side


def guarded(n, dtype=torch.float32):
    buf = torch.full((n + 2 * CAN,), float("nan"), dtype=dtype, device="cuda")
    return buf, buf[CAN:CAN + n]


def intact(buf, n):
    return bool(torch.isnan(buf[:CAN]).all() and torch.isnan(buf[CAN + n:]).all())


def gelu64(x):
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


n_cases, bad, seed = 0, 0, 0
t_end = time.time() + seconds
while time.time() < t_end:
    g = torch.Generator().manual_seed(20000 + seed)
    rng = np.random.default_rng(20000 + seed)
    seed += 1
    kind = rng.integers(0, 10)
    M = int(rng.choice([1, 2, 31, 32, 33, 63, 95, 96, 97, 191, 384, 385, 767, 1536, 2047, 2048, 3071, 3072, 6911, int(rng.integers(1, 7000))]))
    if kind <= 6:  # ---- Linear
        ln = kind in (0, 1, 2)
        N = int(rng.choice([384, 768] if (ln and rng.random() < 0.8) else ([64, 128, 320, 1024] if ln else [32, 64, 96, 384, 1152, 1536])))
        K = int(rng.choice([64, 128, 384, 768, 1536]))
        a, w = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / math.sqrt(K)
        bias = torch.randn(N, generator=g) * 0.3 if rng.random() < 0.8 else None
        act = 0 if ln else int(rng.integers(0, 3))
        res_kind = int(rng.integers(0, 3)) if (ln or rng.random() < 0.5) else 0  # 0 none, 1 fp32 (in place when out is fp32), 2 table
        res_mod = int(rng.choice([1, 7, 192])) if res_kind == 2 else 0
        out_split = (not ln) and (res_kind == 0 or rng.random() < 0.5) and rng.random() < 0.6
        scale = float(rng.choice([1.0, 4096.0]))
        res = torch.randn(res_mod if res_kind == 2 else M, N, generator=g) if res_kind else None
        ref = a.double() @ w.double().t()
        if bias is not None:
            ref = ref + bias.double()
        if act == 1:
            ref = gelu64(ref)
        elif act == 2:
            ref = torch.relu(ref)
        if res is not None:
            ref = ref + (res.double()[torch.arange(M) % res_mod] if res_kind == 2 else res.double())
        gam, bet = torch.rand(N, generator=g) + 0.5, torch.randn(N, generator=g) * 0.1
        ref_h = F.layer_norm(ref, (N,), gam.double(), bet.double(), 1e-6) if ln else None
        ad, wd = to_split(a).cuda(), to_split(w * scale).cuda()
        obuf, out = guarded(M * N)
        hbuf, hout = guarded(M * N)
        cnt = torch.zeros((M + 31) // 32, dtype=torch.int32, device="cuda")
        bd, gd, btd = (bias.cuda() if bias is not None else None), gam.cuda(), bet.cuda()
        rd = None
        if res_kind == 1 and not out_split:
            out.copy_(res.reshape(-1).cuda())  # the residual stream updated in place
            rd = out
        elif res_kind:
            rd = res.cuda()
        for code in (0, 11, 22, 33, 13, 12, 23):
            if code and N % (32 * (code % 10)) != 0:
                continue
            if res_kind == 1 and not out_split:
                out.copy_(res.reshape(-1).cuda())
            L.set_option("skinny_tile", code)
            L.set_option("skinny_xcd_order", int(rng.integers(0, 2)))
            L.call("pp_skinny_linear", ad.data_ptr(), wd.data_ptr(), L.ptr(bd), L.ptr(rd), res_mod, out.data_ptr(), 2 if out_split else 0, M, N, K, act,
                   1.0 / scale, gd.data_ptr() if ln else None, btd.data_ptr() if ln else None, 1e-6, hout.data_ptr() if ln else None,
                   cnt.data_ptr() if ln else None, None)
            torch.cuda.synchronize()
            got = (from_split(out.view(M, N).cpu()) if out_split else out.view(M, N).cpu()).double()
            ok = intact(obuf, M * N) and intact(hbuf, M * N) and int(cnt.abs().sum()) == 0 and torch.allclose(got, ref, rtol=3e-5, atol=3e-5)
            if ln:
                ok = ok and torch.allclose(from_split(hout.view(M, N).cpu()).double(), ref_h, rtol=1e-4, atol=1e-4)
            n_cases += 1
            if not ok:
                bad += 1
                print(f"MISMATCH linear seed {seed - 1} M {M} N {N} K {K} ln {ln} act {act} res {res_kind}/{res_mod} split {out_split} tile {code}: "
                      f"canaries {intact(obuf, M * N)} {intact(hbuf, M * N)} counters {int(cnt.abs().sum())} max err {float((got - ref).abs().max()):.2e}", flush=True)
        L.set_option("skinny_tile", 0)
        L.set_option("skinny_xcd_order", 1)
    elif kind <= 8:  # ---- deconvolution
        B, H, W = int(rng.integers(1, 7)), int(rng.choice([2, 3, 8, 16, 32])), int(rng.choice([2, 5, 12, 24]))
        Cin, Cout = int(rng.choice([64, 256, 384])), int(rng.choice([64, 256]))
        x = torch.randn(B, H, W, Cin, generator=g)
        wt = torch.randn(Cin, Cout, 4, 4, generator=g) * math.sqrt(2.0 / (4 * Cin))
        shift = torch.randn(Cout, generator=g) * 0.2
        ref = F.relu(F.conv_transpose2d(x.permute(0, 3, 1, 2).double(), wt.double(), None, stride=2, padding=1) + shift.double().view(1, -1, 1, 1))
        ref = ref.permute(0, 2, 3, 1).contiguous()
        ph = torch.empty((2, 2, Cout, 4 * Cin))
        for py in range(2):
            for px in range(2):
                for ty in range(2):
                    for tx in range(2):
                        ph[py, px, :, (ty * 2 + tx) * Cin:(ty * 2 + tx + 1) * Cin] = wt[:, :, 3 - 2 * ty - py, 3 - 2 * tx - px].t()
        xd, wd, bd = to_split(x).cuda(), to_split(ph).cuda(), shift.cuda()
        n_out = B * 2 * H * 2 * W * Cout
        obuf, out = guarded(n_out)
        for code in (0, 11, 22, 12, 32):
            if code and Cout % (32 * (code % 10)) != 0:
                continue
            L.set_option("skinny_tile", code)
            L.call("pp_skinny_deconv", xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), out.data_ptr(), B, H, W, Cin, Cout, None)
            torch.cuda.synchronize()
            got = from_split(out.view(B, 2 * H, 2 * W, Cout).cpu()).double()
            ok = intact(obuf, n_out) and torch.allclose(got, ref, rtol=3e-5, atol=3e-5)
            n_cases += 1
            if not ok:
                bad += 1
                print(f"MISMATCH deconv seed {seed - 1} B {B} {H}x{W} Cin {Cin} Cout {Cout} tile {code}: canaries {intact(obuf, n_out)}", flush=True)
        L.set_option("skinny_tile", 0)
    else:  # ---- 1x1 conv, planar out
        n_img, P, K, nv = int(rng.integers(1, 6)), int(rng.choice([1, 40, 1000, 3072])), int(rng.choice([64, 128, 256])), int(rng.choice([1, 17, 32, 33]))
        x, wt, b = torch.randn(n_img * P, K, generator=g), torch.randn(nv, K, generator=g) / math.sqrt(K), torch.randn(nv, generator=g) * 0.3
        rows = 32 * ((nv + 31) // 32)
        wp, bp = torch.zeros(rows, K), torch.zeros(rows)
        wp[:nv], bp[:nv] = wt, b
        ref = (x.double() @ wt.double().t() + b.double()).view(n_img, P, nv).permute(0, 2, 1).contiguous()
        xd, wd, bd = to_split(x).cuda(), to_split(wp).cuda(), bp.cuda()
        obuf, out = guarded(n_img * nv * P)
        L.call("pp_skinny_conv1x1_planar", xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), out.data_ptr(), n_img, P, K, nv, 1.0, None)
        torch.cuda.synchronize()
        ok = intact(obuf, n_img * nv * P) and torch.allclose(out.view(n_img, nv, P).cpu().double(), ref, rtol=3e-5, atol=3e-5)
        n_cases += 1
        if not ok:
            bad += 1
            print(f"MISMATCH conv1x1 seed {seed - 1} n_img {n_img} P {P} K {K} n_valid {nv}", flush=True)
print(f"{n_cases} launches over {seed} random problems in {seconds:.0f} s, {bad} mismatches")
print("SKINNY FUZZ", "FAILED" if bad else "OK")
