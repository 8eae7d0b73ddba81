"""dev: both forms of the fused feed-forward launch against torch fp64, and against each other: where do they differ?"""
import os, sys
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from probpose_code_amd import _lib as L
from probpose_code_amd.weights import to_split, from_split

M, E, F_ = int(sys.argv[1]) if len(sys.argv) > 1 else 288, 384, int(sys.argv[2]) if len(sys.argv) > 2 else 1536
g = torch.Generator().manual_seed(1)
rnd = lambda *s, scale=1.0: torch.randn(*s, generator=g) * scale
h, r, att = rnd(M, E), rnd(M, E), rnd(M, E)
w1, w2, wp = rnd(F_, E, scale=E ** -0.5), rnd(E, F_, scale=F_ ** -0.5), rnd(E, E, scale=E ** -0.5)
bp, g2, be2, b1, b2, ga, be = rnd(E, scale=0.2), 1 + 0.1 * rnd(E), rnd(E, scale=0.1), rnd(F_, scale=0.2), rnd(E, scale=0.2), 1 + 0.1 * rnd(E), rnd(E, scale=0.1)
d = lambda t: t.double()
x_ref = d(r) + F.gelu(d(h) @ d(w1).t() + d(b1)) @ d(w2).t() + d(b2)
x_mid = d(r) + d(att) @ d(wp).t() + d(bp)
h_mid = F.layer_norm(x_mid, (E,), d(g2), d(be2), 1e-6)
xp_ref = x_mid + F.gelu(h_mid @ d(w1).t() + d(b1)) @ d(w2).t() + d(b2)
vec = [t.cuda() for t in (bp, g2, be2, b1, b2, ga, be)]
packed = torch.empty(L.lib.pp_ffn_split_packed_bytes(E, F_) // 4, device="cuda")
w1d, w2d, wpd, rd = to_split(w1).cuda(), to_split(w2).cuda(), to_split(wp).cuda(), r.cuda()
L.call("pp_ffn_split_pack_weights", w1d.data_ptr(), w2d.data_ptr(), packed.data_ptr(), E, F_, None)
wpp = torch.empty(E * E, device="cuda")
L.call("pp_proj_split_pack_weights", wpd.data_ptr(), wpp.data_ptr(), E, None)

def ffn():
    hd, xo, ho = to_split(h).cuda(), torch.full((M, E), float("nan"), device="cuda"), torch.full((M, E), float("nan"), device="cuda")
    L.call("pp_ffn_split_residual_layernorm", hd.data_ptr(), packed.data_ptr(), vec[3].data_ptr(), vec[4].data_ptr(), rd.data_ptr(), xo.data_ptr(),
           vec[5].data_ptr(), vec[6].data_ptr(), 1e-6, ho.data_ptr(), M, E, F_, None)
    return xo.cpu(), ho.cpu()

def proj():
    ad, xo, ho, hs = to_split(att).cuda(), torch.full((M, E), float("nan"), device="cuda"), torch.full((M, E), float("nan"), device="cuda"), torch.full((M, E), float("nan"), device="cuda")
    L.call("pp_proj_ffn_split_residual_layernorm", ad.data_ptr(), wpp.data_ptr(), vec[0].data_ptr(), vec[1].data_ptr(), vec[2].data_ptr(), hs.data_ptr(),
           packed.data_ptr(), vec[3].data_ptr(), vec[4].data_ptr(), rd.data_ptr(), xo.data_ptr(), vec[5].data_ptr(), vec[6].data_ptr(), 1e-6, ho.data_ptr(), M, E, F_, None)
    return xo.cpu(), ho.cpu(), hs.cpu()

res = {}
for form in (0, 1):
    L.set_option("ffn_dma_waves", form)
    res[form] = (ffn(), proj())
    for name, out, ref in (("ffn", res[form][0][0], x_ref), ("proj_ffn", res[form][1][0], xp_ref)):
        e = (out.double() - ref).abs()
        print(f"form {form} {name}: x_out max err {e.max():.3e}, > 2e-5: {(e > 2e-5).sum().item()}, nan {torch.isnan(out).sum().item()}")
    hs_ = from_split(res[form][1][2]).double()
    bad = torch.isnan(hs_).any(1).nonzero().flatten()
    if bad.numel():
        print(f"form {form}: {bad.numel()} ln2 rows hold NaN: {bad[:24].tolist()} ... per row NaN columns of the first: {torch.isnan(hs_[bad[0]]).nonzero().flatten()[:40].tolist()}")
    badx = torch.isnan(res[form][1][0]).any(1).nonzero().flatten()
    if badx.numel():
        print(f"form {form}: x rows with NaN: {badx[:24].tolist()}")
    e = (hs_ - h_mid).abs()
    big = (e > 1e-4).any(1).nonzero().flatten()
    print(f"form {form}: ln2 rows off by > 1e-4: {big.numel()}: {big[:32].tolist()}; columns of the first: {(e[big[0]] > 1e-4).nonzero().flatten()[:48].tolist() if big.numel() else []}")
    print(f"form {form} ln2 rows: max err {e.max():.3e}")
L.set_option("ffn_dma_waves", 0)
for name, a, b in (("ffn x", res[0][0][0], res[1][0][0]), ("proj x", res[0][1][0], res[1][1][0]), ("ffn h", res[0][0][1], res[1][0][1]), ("proj ln2", res[0][1][2], res[1][1][2])):
    ne = (a.view(torch.int32) != b.view(torch.int32))
    if name.endswith("x"):
        ea, eb = (a.double() - (x_ref if name[0] == "f" else xp_ref)).abs(), (b.double() - (x_ref if name[0] == "f" else xp_ref)).abs()
        worst = (a - b).abs().flatten().argmax().item()
        print(f"   at the largest difference (row {worst // E}, col {worst % E}): form 0 err {ea.flatten()[worst]:.3e}, form 1 err {eb.flatten()[worst]:.3e}")
    print(f"{name}: {ne.sum().item()} words differ; rows {ne.any(1).nonzero().flatten()[:12].tolist()}... cols {ne.any(0).nonzero().flatten()[:16].tolist()}; max |diff| {(a - b).abs().max():.3e}")

# raw words of the ln2 rows where the forms differ (split layout: per 32 columns 16 words of hi pairs, then 16 words of lo pairs)
a, b = res[0][1][2].view(torch.int32), res[1][1][2].view(torch.int32)
ne = (a != b).nonzero()
big = [(q_, c) for q_, c in ne.tolist() if abs(int(a[q_, c]) - int(b[q_, c])) > 4][:24]
for rr, c in big:
    r_ = rr
    print(f"ln2 raw row {r_} (block row {r_ % 96}) word {c} (k-block {c // 32}, {'hi' if c % 32 < 16 else 'lo'} pair {c % 16}): form 0 {int(a[rr, c]) & 0xffffffff:08x}  form 1 {int(b[rr, c]) & 0xffffffff:08x}")

cands = {"residual r": r, "x_out (form 1)": res[1][1][0], "x_mid (fp64 ref)": x_mid.float(), "b2": b2, "bp": bp}
for rr, c in big[:8]:
    val = b[rr, c].view(torch.float32)
    for name, t in cands.items():
        hit = (t.view(torch.int32) == b[rr, c]).nonzero()
        if hit.numel():
            print(f"   word ({rr}, {c}) = {val.item():.6f} is {name}{hit[:3].tolist()}")
        else:
            near = (t - val).abs().flatten().argmin().item()
            if (t.flatten()[near] - val).abs() < 1e-4:
                print(f"   word ({rr}, {c}) = {val.item():.6f} ~ {name}[{near // t.shape[-1] if t.dim() > 1 else 0}, {near % t.shape[-1]}] = {t.flatten()[near].item():.6f}")
