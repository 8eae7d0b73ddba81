# dev only: phase breakdown of the paired proj + FFN kernel from s_memtime stamps (library built with -DFFD_STAMP=1, scripts/micro/ffd_stamps.sh)
import sys, os, ctypes, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import test_split_fp16 as T
L = T._lib()
M, E, F_ = 24576, 384, 1536
h, r, w1, b1, w2, b2, g, be = T._ffn_inputs(M, F_, seed=500)
att, wp, bp, g2, be2 = T._proj_inputs(M, seed=520)
packed = T._ffn_pack(L, w1, w2, E, F_)
wpp = torch.empty(E * E, dtype=torch.float32, device="cuda")
wps = T._sp(wp)
L.call("pp_proj_split_pack_weights", wps.data_ptr(), wpp.data_ptr(), E, None)
dev = [t.cuda() for t in (bp, g2, be2, b1, b2, g, be)]
ad, xd = T._sp(att), r.cuda()
scratch = torch.empty((M, E), device="cuda")
def run():
    L.call("pp_proj_ffn_split_residual_layernorm", ad.data_ptr(), wpp.data_ptr(), dev[0].data_ptr(), dev[1].data_ptr(), dev[2].data_ptr(), scratch.data_ptr(), packed.data_ptr(),
           dev[3].data_ptr(), dev[4].data_ptr(), xd.data_ptr(), xd.data_ptr(), dev[5].data_ptr(), dev[6].data_ptr(), 1e-6, ad.data_ptr(), M, E, F_, None)
for _ in range(30): run()   # (back to back: the clock settles where the step's does)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 128)()
L.lib.pp_dev_ffd_stamps.restype = ctypes.c_int
assert L.lib.pp_dev_ffd_stamps(buf) == 0
fine = len(sys.argv) > 1 and sys.argv[1] == "fine"   # library built with -DFFD_STAMP=2: the LayerNorm phases in pieces
names = ["start -> residual rows requested", "projection: 24 steps"] + (["ln2: wait for every wave (P1)", "ln2: row statistics", "ln2: normalise, split, stores issued",
         "ln2: stores acknowledged by the L2", "ln2: wait for every wave (P2)"] if fine else ["ln2 + rows to L2"])
for cp in range(6):
    names += [f"pair {cp}: 24 A-steps", f"pair {cp}: GELU chunk 0", f"pair {cp}: 8 B-steps chunk 0", f"pair {cp}: GELU chunk 1", f"pair {cp}: 8 B-steps chunk 1"]
names += ["wait for every wave (E1)"] + (["final LayerNorm: row statistics", "final LayerNorm: normalise, split, stores issued"] if fine else ["final LayerNorm + stores"])
for wg, tag in ((0, "workgroup 0"), (1, "workgroup 131")):
    t = [buf[wg * 64 + i] for i in range(64)]
    n = len(names) + 1
    d = [t[i + 1] - t[i] for i in range(n - 1)]
    tot = t[n - 1] - t[0]
    print(f"--- {tag}: {tot} cycles from the first to the last stamp")
    agg = {"A-steps (144)": 0, "B-steps (96)": 0, "GELU (12 chunks)": 0}
    for nm, c in zip(names, d):
        if "A-steps" in nm: agg["A-steps (144)"] += c
        elif "B-steps" in nm: agg["B-steps (96)"] += c
        elif "GELU" in nm: agg["GELU (12 chunks)"] += c
        else: print(f"  {nm:36s} {c:8d}  {100 * c / tot:5.1f} %")
    for k, c in agg.items(): print(f"  {k:36s} {c:8d}  {100 * c / tot:5.1f} %")
    print("  per A-step", agg["A-steps (144)"] // 144, " per B-step", agg["B-steps (96)"] // 96, " per chunk GELU", agg["GELU (12 chunks)"] // 12,
          " (MFMA issue per SIMD: 576 per A-step, 864 per B-step and projection step)")
