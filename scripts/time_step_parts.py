"""dev: where a bench step's wall time goes - graph replay alone vs replay + result record pack / pinned-host copy."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from probpose_code_amd import synthetic as S
from probpose_code_amd.dist import ResultGather, pack_records
from probpose_code_amd.engine import ProbPoseEngine

B = 64
dev = torch.device("cuda", 0)
sd = S.synthetic_state_dict("small", seed=0, logit_scale=2.0)
crops = S.synthetic_crops(B, seed=100).to(dev)
eng = ProbPoseEngine(sd, 12, precision="bf16", device=dev)
flip = S.COCO_FLIP_INDICES
eng.capture(B, True, flip).copy_(crops)
gather = ResultGather(B, eng.K, dev, 1)

def timed(fn, n=50):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

def host_only(fn, n=50):  # host time to enqueue, GPU idle-ish: sync every call
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return sum(ts) / n * 1e3

print(f"graph replay only                : {timed(lambda: eng.forward_graph(crops, True, flip)):.3f} ms/step")
print(f"graph replay + pack + host copy  : {timed(lambda: gather(eng.forward_graph(crops, True, flip))):.3f} ms/step")
out = eng.forward_graph(crops, True, flip)
print(f"pack + host copy alone           : {timed(lambda: gather(out)):.3f} ms/step")
print(f"host enqueue time: replay {host_only(lambda: eng.forward_graph(crops, True, flip)):.3f} ms, pack+copy {host_only(lambda: gather(out)):.3f} ms")
