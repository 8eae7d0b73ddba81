"""dev: BASELINE configs[3] - ProbPose-base (ViT-B 12 x 768, 12 heads x 64) at 384x288, bf16, bs B, flip test - through the
engine (generic kernels: the fused layer kernel is E = 384 only). Prints ms/step, crops/s and the path TFLOP/s."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from probpose_code_amd import ProbPoseEngine
from probpose_code_amd import synthetic as S

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
PREC = sys.argv[2] if len(sys.argv) > 2 else "bf16"
img = (384, 288)
sd = S.synthetic_state_dict("base", img_size=img, seed=0, logit_scale=2.0)
x = S.synthetic_crops(B, img_size=img, seed=1).cuda()
eng = ProbPoseEngine(sd, 12, img_size=img, precision=PREC, input_size=(288, 384))
for _ in range(3): eng.forward(x, True, S.COCO_FLIP_INDICES)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
n = 10
for _ in range(n): eng.forward(x, True, S.COCO_FLIP_INDICES)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / n
Np, E, Fd, L = 24 * 18, 768, 3072, 12
M = 2 * B * Np
fl = L * (2.0 * M * E * 3 * E + 2.0 * M * E * E + 4.0 * M * E * Fd + 4.0 * 2 * B * 12 * Np * Np * 64) + 2.0 * M * E * 768
print(f"ViT-B 384x288 bs{B} flip: {ms:.2f} ms/step, {B / ms * 1e3:.0f} crops/s, backbone {fl / ms / 1e9:.0f} TFLOP/s")
eng.profile = {}
eng.forward(x, True, S.COCO_FLIP_INDICES)
torch.cuda.synchronize()
per = {k: sum(a.elapsed_time(b) for a, b in v) for k, v in eng.profile.items()}
print({k: round(v, 3) for k, v in sorted(per.items(), key=lambda kv: -kv[1])})
# the same through the captured graph, one and two steps in flight (pipeline.StepPipeline, what bench.py times for config 2)
from probpose_code_amd.pipeline import StepPipeline
for depth in (1, 2):
    pipe = StepPipeline(eng, B, S.COCO_FLIP_INDICES, depth=depth)
    for _ in range(4): pipe.submit(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): pipe.submit(x)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 20 * 1e3
    print(f"hipGraph replay, {depth} step(s) in flight: {ms:.2f} ms/step, {B / ms * 1e3:.0f} crops/s")
