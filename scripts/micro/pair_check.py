import sys, torch
import os; R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import test_split_fp16 as T
L = T._lib()
M, E, F_ = 24576, 384, 1536
h, r, w1, b1, w2, b2, g, be = T._ffn_inputs(M, F_, seed=500)
att, wp, bp, g2, be2 = T._proj_inputs(M, seed=520)
packed = T._ffn_pack(L, w1, w2, E, F_)
wpp = torch.empty(E * E, dtype=torch.float32, device="cuda")
L.call("pp_proj_split_pack_weights", T._sp(wp).data_ptr(), wpp.data_ptr(), E, None)
dev = [t.cuda() for t in (bp, g2, be2, b1, b2, g, be)]
def proj():
    ad, xd = T._sp(att), r.cuda()
    scratch = torch.full((M, E), float("nan"), device="cuda")
    L.call("pp_proj_ffn_split_residual_layernorm", ad.data_ptr(), wpp.data_ptr(), dev[0].data_ptr(), dev[1].data_ptr(), dev[2].data_ptr(), scratch.data_ptr(), packed.data_ptr(), dev[3].data_ptr(), dev[4].data_ptr(), xd.data_ptr(), xd.data_ptr(), dev[5].data_ptr(), dev[6].data_ptr(), 1e-6, ad.data_ptr(), M, E, F_, None)
    return xd.cpu(), T._unsp(ad)
L.set_option("ffn_pair", 0); a = proj()
L.set_option("ffn_pair", 1); b = proj(); c = proj()
print("pair vs single: max |dx|", (a[0] - b[0]).abs().max().item(), "max |dh|", (a[1] - b[1]).abs().max().item(), "run-to-run equal", torch.equal(b[0], c[0]))
