"""dev only: does running two half-batches on two streams (kernels of one half overlapping the other's) beat one
full batch? Both variants under hipGraph replay, bs64 total."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from probpose_code_amd import synthetic as S
from probpose_code_amd.engine import ProbPoseEngine
dev = torch.device("cuda", 0)
sd = S.synthetic_state_dict("small", seed=0, logit_scale=2.0)
flip = S.COCO_FLIP_INDICES
def timeit(fn, n=40):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
nsplit = int(sys.argv[1]) if len(sys.argv) > 1 else 2
B = 64
crops = S.synthetic_crops(B, seed=100).to(dev)
eng = ProbPoseEngine(sd, 12, precision="bf16", device=dev)
eng.capture(B, True, flip).copy_(crops)
print(f"one batch of {B}: {timeit(lambda: eng.forward_graph(crops, True, flip)):.3f} ms")
engs = [ProbPoseEngine(sd, 12, precision="bf16", device=dev) for _ in range(nsplit)]
Bh = B // nsplit
ins = [crops[i * Bh:(i + 1) * Bh].clone() for i in range(nsplit)]
streams = [torch.cuda.Stream(device=dev) for _ in range(nsplit)]
side = torch.cuda.Stream(device=dev)
side.wait_stream(torch.cuda.current_stream(dev))
with torch.cuda.stream(side):
    for e, x in zip(engs, ins):
        for _ in range(2): e.forward(x, True, flip)
torch.cuda.current_stream(dev).wait_stream(side)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    cur = torch.cuda.current_stream(dev)
    for s in streams: s.wait_stream(cur)
    for e, x, s in zip(engs, ins, streams):
        with torch.cuda.stream(s):
            e.forward(x, True, flip)
    for s in streams: cur.wait_stream(s)
print(f"{nsplit} x {Bh} on {nsplit} streams: {timeit(lambda: g.replay()):.3f} ms")
