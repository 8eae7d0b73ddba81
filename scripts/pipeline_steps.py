"""dev only: whole bs64 steps alternating over S streams (each stream its own engine workspace + captured graph), so that
the low-occupancy tail of step n (small tower convolutions, pooling, decode, record pack) and the HBM-bound head of step
n + 1 (im2col, patch embed) can share the chip. Compared with S = 1 (what bench.py times). Usage: pipeline_steps.py [S] [prec]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from probpose_code_amd import synthetic as S
from probpose_code_amd.engine import ProbPoseEngine
from probpose_code_amd.dist import ResultGather

dev = torch.device("cuda", 0)
nstream = int(sys.argv[1]) if len(sys.argv) > 1 else 2
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16"
B = int(sys.argv[3]) if len(sys.argv) > 3 else 64
sd = S.synthetic_state_dict("small", seed=0, logit_scale=2.0)
flip = S.COCO_FLIP_INDICES
crops = S.synthetic_crops(B, seed=100).to(dev)
engs = [ProbPoseEngine(sd, 12, precision=prec, device=dev) for _ in range(nstream)]
gathers = [ResultGather(B, 17, dev, 1) for _ in range(nstream)]
for e in engs:
    e.capture(B, True, flip).copy_(crops)
streams = [torch.cuda.Stream(device=dev) for _ in range(nstream)]
torch.cuda.synchronize()


def run(n, ns):
    for i in range(n):
        j = i % ns
        with torch.cuda.stream(streams[j]):
            gathers[j](engs[j].forward_graph(crops, True, flip))


for ns in (1, nstream, 1, nstream):
    run(10, ns)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(100, ns)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 100
    print(f"[{prec}] {ns} stream(s): {dt * 1e3:.3f} ms/step -> {B / dt:.0f} crops/s")
a = gathers[0].wait().clone()
if nstream > 1:
    b = gathers[1].wait()
    print("records of the two pipelines identical:", bool(torch.equal(a, b)))
