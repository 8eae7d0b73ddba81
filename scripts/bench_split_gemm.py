"""dev only: the Linear layers of the f16x3 mode at the bs64 shapes; PP_PANEL=0 forces the 128 x 128 kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from probpose_code_amd import _lib as L
import ctypes
if os.environ.get("LIB"):
    _alt = ctypes.CDLL(os.environ["LIB"]); _alt.pp_gemm.restype = ctypes.c_int; _alt.pp_gemm.argtypes = L.SIGNATURES["pp_gemm"][1]
from probpose_code_amd.weights import to_split
M = 24576
for name, N, K, act in (("qkv", 1152, 384, 0), ("fc1", 1536, 384, 1), ("fc1 no gelu", 1536, 384, 0)):
    a = to_split(torch.randn(M, K)).cuda(); w = to_split(torch.randn(N, K) / K ** 0.5).cuda(); b = torch.randn(N).cuda()
    out = torch.empty(M, N, device="cuda")
    run = (lambda: _alt.pp_gemm(2, a.data_ptr(), w.data_ptr(), b.data_ptr(), None, 0, out.data_ptr(), M, N, K, K, K, N, act, 2, 0, None)) if os.environ.get("LIB") else lambda: L.call("pp_gemm", 2, a.data_ptr(), w.data_ptr(), b.data_ptr(), None, 0, out.data_ptr(), M, N, K, K, K, N, act, 2, 0, None)
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"PP_PANEL={os.environ.get('PP_PANEL', '1')} {name:12s} {us:7.1f} us  {2 * M * N * K / us / 1e6:6.0f} TF algorithmic")
# ViT-B (bs 64, flip: 55296 rows) bf16 Linear layers with fp32 output + residual: proj (K 768) and fc2 (K 3072)
Mb = 55296
for name, N, K in (("vit-b proj", 768, 768), ("vit-b fc2", 768, 3072)):
    a = torch.randn(Mb, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16(); b = torch.randn(N).cuda()
    res = torch.randn(Mb, N, device="cuda"); out = torch.empty(Mb, N, device="cuda")
    args = (0, a.data_ptr(), w.data_ptr(), b.data_ptr(), res.data_ptr(), 0, out.data_ptr(), Mb, N, K, K, K, N, 0, 0, 0, None)
    run = (lambda: _alt.pp_gemm(*args)) if os.environ.get("LIB") else (lambda: L.call("pp_gemm", *args))
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"{name:12s} {us:7.1f} us  {2 * Mb * N * K / us / 1e6:6.0f} TF")
