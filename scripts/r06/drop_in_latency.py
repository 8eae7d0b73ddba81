import os, sys, time, torch, numpy as np
sys.path.insert(0, "/root/repo")
from probpose_code_amd import apis
from probpose_code_amd import synthetic as S
cfg = "/root/repo/configs/td-pm_ProbPose-small_mi355x_coco-256x192.py"
sd = S.synthetic_state_dict("small", seed=0, logit_scale=2.0)
model = apis.init_model(cfg, {"state_dict": sd}, device="cuda:0")
for B in (1, 2, 4, 8):
    crops = S.synthetic_crops(B, seed=B).cuda(); c, s = S.whole_image_bbox_meta(B)
    batch = apis.pack_crops(crops, c, s, model.dataset_meta)
    with torch.no_grad():
        for _ in range(10): model.test_step(batch)
        t0 = time.perf_counter()
        for _ in range(300): model.test_step(batch)
        dt = (time.perf_counter() - t0) / 300
    print(f"test_step B {B}: {dt*1e3:.3f} ms per call (host packaging included)")
img = np.random.default_rng(0).integers(0, 256, (480, 640, 3), dtype=np.uint8)
for n in (1, 4):
    bb = np.array([[50 + 30 * i, 40, 250 + 30 * i, 440] for i in range(n)], np.float32)
    for _ in range(10): apis.inference_topdown(model, img, bb)
    t0 = time.perf_counter()
    for _ in range(200): apis.inference_topdown(model, img, bb)
    print(f"inference_topdown 640x480, {n} boxes: {(time.perf_counter() - t0) / 200 * 1e3:.3f} ms per image")
