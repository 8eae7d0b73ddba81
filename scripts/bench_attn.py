import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from probpose_code_amd import _lib as L
n_seq, S, heads, hd = 128, 192, 12, 32
E = heads * hd
qkv = torch.randn(n_seq * S, 3 * E, device="cuda").to(torch.bfloat16)
out = torch.empty(n_seq * S, E, device="cuda", dtype=torch.bfloat16)
def run(): L.call("pp_attention", 0, qkv.data_ptr(), out.data_ptr(), n_seq, S, heads, hd, hd ** -0.5, None)
for _ in range(5): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(30): run()
e1.record(); torch.cuda.synchronize()
print(f"attention bs128x192x12x32: {e0.elapsed_time(e1)/30*1e3:.1f} us")
