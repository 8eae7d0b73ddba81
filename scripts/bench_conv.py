"""dev only: time the head convolutions at the bs64 shapes. LIB=path of an alternative library (scripts/micro/panel_ablate.sh)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from probpose_code_amd import _lib as L
libpath = os.environ.get("LIB")
if libpath:
    lib = ctypes.CDLL(libpath)
    fn = lib.pp_conv_gemm
    fn.restype = ctypes.c_int
    P, I, LL = ctypes.c_void_p, ctypes.c_int, ctypes.c_longlong
    fn.argtypes = [I, I, P, P, P, P, I, I, I, I, I, I, I, I, LL, LL, LL, LL, I, I, I, P]
    call = lambda *a: fn(*a)
else:
    call = lambda *a: L.call("pp_conv_gemm", *a)
dt = torch.bfloat16
B = 128
cases = [("conv1 (4 towers)", 1, 16, 12, 384, 384, 4), ("deconv1", 2, 16, 12, 384, 256, 1), ("deconv2", 2, 32, 24, 256, 256, 1)]
if os.environ.get("LIB"):
    fh = lib.pp_deconv_head
    fh.restype = ctypes.c_int
    fh.argtypes = [ctypes.c_void_p] * 6 + [ctypes.c_int] * 6 + [ctypes.c_void_p]
    call_head = lambda *a: fh(*a)
else:
    call_head = lambda *a: L.call("pp_deconv_head", *a)
# deconv2 fused with the final 1x1 convolution (what the engine runs)
x = torch.randn(B, 32, 24, 256, device="cuda").to(dt)
w = (torch.randn(4, 256, 4 * 256, device="cuda") / 32).to(dt)
b = torch.randn(256, device="cuda")
hw = torch.zeros(32, 256, device="cuda", dtype=dt); hw[:17] = (torch.randn(17, 256, device="cuda") / 16).to(dt)
hb = torch.randn(17, device="cuda")
logits = torch.empty(B, 17, 4, 32 * 24, device="cuda")
hargs = (x.data_ptr(), w.data_ptr(), b.data_ptr(), hw.data_ptr(), hb.data_ptr(), logits.data_ptr(), B, 32, 24, 256, 256, 17, None)
for _ in range(3): assert call_head(*hargs) in (0, None)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): call_head(*hargs)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
print(f"{'deconv2 + final 1x1':18s} {ms*1e3:8.1f} us  {2 * B * 768 * 256 * 1024 * 4 / ms / 1e9:7.0f} TF (deconvolution FLOPs only)")
for name, kind, H, W, Cin, Cout, G in cases:
    x = torch.randn(B, H, W, Cin, device="cuda").to(dt)
    taps = 9 if kind == 1 else 4
    ng = G if kind == 1 else 4
    w = (torch.randn(ng, Cout, taps * Cin, device="cuda") / (taps * Cin) ** 0.5).to(dt)
    b = torch.randn(ng, Cout, device="cuda")
    if kind == 1:
        out = torch.empty(G, B, H, W, Cout, device="cuda", dtype=dt)
        args = (0, 1, x.data_ptr(), w.data_ptr(), b.data_ptr(), out.data_ptr(), B, H, W, Cin, Cout, 0, 0, G, 0, Cout * taps * Cin, B * H * W * Cout, Cout, Cout, 0, 1, None)
        flops = 2 * B * H * W * Cout * taps * Cin * G
    else:
        out = torch.empty(B, 2 * H, 2 * W, Cout, device="cuda", dtype=dt)
        args = (0, 2, x.data_ptr(), w.data_ptr(), b.data_ptr(), out.data_ptr(), B, H, W, Cin, Cout, -1, 0, 1, 0, 0, 0, 0, Cout, 2, 1, None)
        flops = 2 * B * H * W * Cout * taps * Cin * 4
    for _ in range(3): assert call(*args) in (0, None)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): call(*args)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    extra = ""
    if os.environ.get("CLOCK_PROBE") and (kind == 1 or os.environ["CLOCK_PROBE"] == "all"):  # library built with HALO_DBG & 256
        t = out.view(torch.int64).flatten()[:2].tolist()
        extra = f"  shader clock {t[0] / max(t[1], 1) * 100:.0f} MHz over {t[1] / 100:.1f} us"
    print(f"{name:18s} {ms*1e3:8.1f} us  {flops/ms/1e9:7.0f} TF{extra}")
