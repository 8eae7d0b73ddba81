import sys, os, ctypes
sys.path.insert(0, "/root/repo")
import torch
n_seq, S, heads, hd = 128, 192, 12, 32
E = heads * hd
qkv = torch.randn(n_seq * S, 3 * E, device="cuda").to(torch.bfloat16)
out = torch.empty(n_seq * S, E, device="cuda", dtype=torch.bfloat16)
P = ctypes.c_void_p
for name in ("probpose_code_amd/libprobpose_mi355x.so", "scripts/micro/build/libatt_abl1.so", "scripts/micro/build/libatt_abl2.so", "scripts/micro/build/libatt_abl3.so"):
    lib = ctypes.CDLL(os.path.join("/root/repo", name))
    fn = lib.pp_attention; fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_int, P, P, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float, P]
    run = lambda: fn(0, qkv.data_ptr(), out.data_ptr(), n_seq, S, heads, hd, hd ** -0.5, None)
    for _ in range(5): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): run()
    e1.record(); torch.cuda.synchronize()
    print(f"{name}: {e0.elapsed_time(e1)/30*1e3:.1f} us")
