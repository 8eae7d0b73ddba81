"""Dev script: time the engine only (no oracle)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from probpose_code_amd import synthetic as S
from probpose_code_amd.engine import ProbPoseEngine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16"
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 20
sd = S.synthetic_state_dict("small", seed=0)
x = S.synthetic_crops(B, seed=1).cuda()
eng = ProbPoseEngine(sd, 12, precision=prec)
for _ in range(3): eng.forward(x, True, S.COCO_FLIP_INDICES)
torch.cuda.synchronize(); t = time.time()
for _ in range(iters): eng.forward(x, True, S.COCO_FLIP_INDICES)
torch.cuda.synchronize(); dt = (time.time() - t) / iters
print(f"[{prec}] B={B}: {dt*1e3:.3f} ms/batch -> {B/dt:.0f} crops/s")
