import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from probpose_code_amd import synthetic as S
from oracle import model_ref as M
sd = S.synthetic_state_dict("small", seed=0)
x = S.synthetic_crops(8, seed=1)
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for nt in (8, 16, 32, 64, 128):
    torch.set_num_threads(nt)
    t = time.time(); M.predict(sd, x[:2], 12, S.IMG_MEAN, S.IMG_STD); w = time.time() - t
    t = time.time(); M.predict(sd, x, 12, S.IMG_MEAN, S.IMG_STD); d = time.time() - t
    print(f"threads {nt}: warm {w:.2f}s, 8 crops {d:.2f}s -> {8/d:.1f} crops/s")
