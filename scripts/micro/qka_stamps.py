# dev only: phase breakdown of the fused qkv + attention kernel from s_memtime stamps (library built with -DQKA_STAMP=1 into scripts/micro/build/lib_qstamp.so;
# run on the GPU box: cp scripts/micro/build/lib_qstamp.so probpose_code_amd/libprobpose_mi355x.so; python scripts/micro/qka_stamps.py)
import sys, os, ctypes, math, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import test_split_fp16 as T
L = T._lib()
n_seq, S, H, hd = int(os.environ.get("NSEQ", 128)), 192, 12, 32
E = H * hd
h = T._sp(T._rand(n_seq * S, E, seed=1))
w = T._sp(T._rand(3 * E, E, seed=2, scale=1 / math.sqrt(E)))
b = T._rand(3 * E, seed=3, scale=0.1).cuda()
out = torch.empty(n_seq * S, E, device="cuda")
for _ in range(40):
    L.call("pp_qkv_attention_split", h.data_ptr(), w.data_ptr(), b.data_ptr(), out.data_ptr(), n_seq, S, H, hd, hd ** -0.5, None)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 64)()
L.lib.pp_dev_qka_stamps.restype = ctypes.c_int
assert L.lib.pp_dev_qka_stamps(buf) == 0
for k, tag in enumerate(("workgroup 0 wave 0 (two query tiles)", "workgroup 0 wave 7 (one query tile)", "workgroup 777 wave 0", "workgroup 777 wave 7")):
    t = [buf[k * 16 + i] for i in range(16)]
    two = "wave 0" in tag
    names = ["start -> first stage landed", "qkv projection: 12 K-steps", "+ bias, split, q / k / V^T to LDS", "wait for every wave", "query tile 1"] + (["query tile 2"] if two else []) + ["end"]
    n = len(names) + 1
    tot = t[n - 1] - t[0]
    print(f"--- {tag}: {tot} cycles")
    for i, nm in enumerate(names):
        print(f"  {nm:38s} {t[i + 1] - t[i]:7d}  {100 * (t[i + 1] - t[i]) / tot:5.1f} %")
