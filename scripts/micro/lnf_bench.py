# dev only: pp_linear_ln_folded's epilogue variants on the proj / fc2 shapes of BASELINE config 4 (M = 55 296), back to back, HIP events
import sys, os, math, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import test_split_fp16 as T
L = T._lib()
M, E = 55296, 768
F32, SPLIT = 0, 2
for K in (768, 3072):
    a = torch.randn(M, K, device="cuda").contiguous(); a_s = a.clone()  # (container: values irrelevant for timing, but keep them finite fp16 pairs)
    a_s = T._sp(torch.randn(M, K)); w = T._sp(torch.randn(E, K) / math.sqrt(K)); b = torch.randn(E, device="cuda")
    x32 = torch.randn(M, E, device="cuda"); xs = T._sp(torch.randn(M, E)); st = torch.zeros(M, 8, 2, device="cuda")
    o32 = torch.empty(M, E, device="cuda"); osp = torch.empty(M, E, device="cuda")
    variants = {
        "res f32 -> out f32            ": (x32, F32, o32, F32, None),
        "res f32 -> out split          ": (x32, F32, osp, SPLIT, None),
        "res split -> out f32          ": (xs, SPLIT, o32, F32, None),
        "res split -> out split        ": (xs, SPLIT, osp, SPLIT, None),
        "res split -> out split + stats": (xs, SPLIT, osp, SPLIT, st),
        "res split in place + stats    ": (xs, SPLIT, xs, SPLIT, st),
        "no residual -> out split      ": (None, F32, osp, SPLIT, None),
    }
    def run(v):
        r, rf, o, of, s_ = v
        L.call("pp_linear_ln_folded", a_s.data_ptr(), w.data_ptr(), b.data_ptr(), None if r is None else r.data_ptr(), rf, o.data_ptr(), of, M, E, K, 0,
               None, None, 1e-6, None if s_ is None else s_.data_ptr(), None)
    for rep in range(2):
        for name, v in variants.items():
            for _ in range(3): run(v)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): run(v)
            e1.record(); torch.cuda.synchronize()
            if rep: print(f"K {K:5d}  {name}: {e0.elapsed_time(e1) / 20 * 1e3:7.1f} us")
