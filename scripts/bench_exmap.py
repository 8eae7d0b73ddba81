"""dev: the Ex-mAP evaluator at COCO-val2017 scale (5000 images, ~10.8k person instances, ~100k detections) - host
parsing vs the three GPU launches, and the CPU oracle timed on a 250-image subset. Also times heatmap revert + merge."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oracle import exmap_ref, warp_ref
from probpose_code_amd.evaluation import COCO_SIGMAS, COCOeval
from probpose_code_amd.structures import revert_heatmaps_max

K = 17
rng = np.random.default_rng(0)
gts, dts = [], []
for img in range(5000):
    G = int(rng.choice([0, 0, 1, 1, 2, 3, 4, 8, 14], 1)[0])
    here = []
    for _ in range(G):
        w, h = rng.uniform(30, 220), rng.uniform(40, 320)
        x0, y0 = rng.uniform(0, 640 - w), rng.uniform(0, 480 - h)
        kp = np.zeros((K, 3)); kp[:, 0] = rng.uniform(x0, x0 + w, K); kp[:, 1] = rng.uniform(y0, y0 + h, K)
        kp[:, 2] = rng.choice([0, 1, 2, 3], K, p=[0.25, 0.15, 0.5, 0.1]); kp[kp[:, 2] == 0, :2] = 0
        g = dict(id=len(gts) + 1, image_id=img, category_id=1, keypoints=kp.flatten().tolist(), bbox=[x0, y0, w, h], area=float(w * h * 0.5), iscrowd=int(rng.random() < 0.05))
        gts.append(g); here.append(g)
    for _ in range(int(rng.integers(5, 36))):
        if here and rng.random() < 0.7:
            g = here[rng.integers(0, len(here))]
            kp = np.array(g["keypoints"]).reshape(K, 3).copy()
            kp[:, :2] += rng.normal(0, rng.choice([0.01, 0.03, 0.08]) * np.sqrt(g["bbox"][2] * g["bbox"][3]), (K, 2))
            kp[:, 2] = np.where(kp[:, 2] == 3, rng.beta(1.2, 4, K), rng.beta(5, 1.2, K)); bbox = list(g["bbox"])
        else:
            kp = np.stack([rng.uniform(0, 640, K), rng.uniform(0, 480, K), rng.uniform(0, 1, K)], 1)
            bbox = [float(kp[:, 0].min()), float(kp[:, 1].min()), float(np.ptp(kp[:, 0])), float(np.ptp(kp[:, 1]))]
        dts.append(dict(id=len(dts) + 1, image_id=img, category_id=1, keypoints=kp.flatten().tolist(), score=float(rng.uniform(0.05, 1)), bbox=bbox, area=float(bbox[2] * bbox[3])))
print(f"{len(gts)} instances, {len(dts)} detections")
img_ids = list(range(5000))
torch.zeros(1, device="cuda")
for rep in range(2):
    e = COCOeval(gts, dts, "keypoints", sigmas=COCO_SIGMAS, extended_oks=True)
    e.params.imgIds = img_ids
    t0 = time.perf_counter(); h = e._prepare(); t1 = time.perf_counter()
    e = COCOeval(gts, dts, "keypoints", sigmas=COCO_SIGMAS, extended_oks=True); e.params.imgIds = img_ids
    t2 = time.perf_counter(); e.evaluate(); torch.cuda.synchronize(); t3 = time.perf_counter(); e.accumulate(); e.summarize(); torch.cuda.synchronize(); t4 = time.perf_counter()
    print(f"host parse {t1 - t0:.3f} s | evaluate (parse + upload + cells + match) {t3 - t2:.3f} s | accumulate + summarize {t4 - t3:.4f} s | AP {e.stats[0]:.4f}")
# kernels alone
d, o, m = e._d, e._o, e._meta
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
from probpose_code_amd import _lib
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize(); ev0.record()
    for _ in range(n): fn()
    ev1.record(); torch.cuda.synchronize(); return ev0.elapsed_time(ev1) / n
L, A, T = m["L"], 3, 10
s = torch.cuda.current_stream().cuda_stream
t_cells = timeit(lambda: _lib.call("pp_exoks_cells", d["gt_kpts"].data_ptr(), d["gt_bbox"].data_ptr(), d["gt_area_oks"].data_ptr(), d["dt_kpts"].data_ptr(), d["sigmas"].data_ptr(), d["gt_vis"].data_ptr(), d["cell_gt_off"].data_ptr(), d["cell_dt_off"].data_ptr(), d["cell_iou_off"].data_ptr(), m["n_cells"], K, L - 1, 0.5, 1.25, 1, 0, e.ious.data_ptr(), s))
t_match = timeit(lambda: _lib.call("pp_exoks_match", e.ious.data_ptr(), d["cell_gt_off"].data_ptr(), d["cell_dt_off"].data_ptr(), d["cell_iou_off"].data_ptr(), d["gt_ignore"].data_ptr(), d["gt_iscrowd"].data_ptr(), d["gt_area_rng"].data_ptr(), d["gt_bbox"].data_ptr(), d["dt_area"].data_ptr(), d["dt_bbox"].data_ptr(), d["area_rng"].data_ptr(), d["iou_thrs"].data_ptr(), m["n_cells"], m["max_g"], m["N_gt"], m["N_dt"], L, A, T, 0, o["dt_match"].data_ptr(), o["dt_ignore"].data_ptr(), o["gt_match"].data_ptr(), o["gt_ignore"].data_ptr(), o["sim_sum"].data_ptr(), o["sim_cnt"].data_ptr(), s))
t_acc = timeit(lambda: e.accumulate())
print(f"kernels: cells {t_cells:.3f} ms ({m['iou_total']} similarities), match {t_match:.3f} ms, accumulate (incl. table copies) {t_acc:.3f} ms; N_dt kept {m['N_dt']}")
sub = 250
g_s, d_s = [g for g in gts if g["image_id"] < sub], [x for x in dts if x["image_id"] < sub]
t0 = time.perf_counter(); r = exmap_ref.evaluate(g_s, d_s, COCO_SIGMAS, img_ids=list(range(sub))); t1 = time.perf_counter()
print(f"oracle (numpy / python, 1 core) on {sub} images: {t1 - t0:.2f} s -> {(t1 - t0) * 5000 / sub:.0f} s for 5000 images")

# heatmap revert + merge: 8 persons on a 480x640 image
hms = torch.rand(8, 17, 64, 48, device="cuda") ** 8
centers = np.stack([rng.uniform(50, 590, 8), rng.uniform(50, 430, 8)], 1); hh = rng.uniform(100, 400, 8); scales = np.stack([hh * 0.75, hh], 1)
revert_heatmaps_max(hms, centers, scales, (480, 640)); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): out = revert_heatmaps_max(hms, centers, scales, (480, 640))
torch.cuda.synchronize(); t1 = time.perf_counter()
from probpose_code_amd.transforms import get_warp_matrix, invert_affine
inv = torch.from_numpy(np.stack([invert_affine(get_warp_matrix(centers[i], scales[i], 0, (48, 64), inv=True)) for i in range(8)])).cuda()
t_k = timeit(lambda: _lib.call("pp_revert_heatmaps_max", hms.data_ptr(), inv.data_ptr(), out.data_ptr(), 8, 17, 64, 48, 480, 640, s), 20)
hn = hms.cpu().numpy()
t2 = time.perf_counter(); ref = np.max([warp_ref.revert_heatmap(hn[i], centers[i], scales[i], (480, 640)) for i in range(8)], axis=0); t3 = time.perf_counter()
print(f"revert+merge 8 persons -> 17x480x640: call {1e3 * (t1 - t0) / 20:.3f} ms, kernel {t_k * 1e3:.1f} us ({17 * 480 * 640 * 4 / t_k / 1e6:.0f} GB/s of output), numpy oracle {t3 - t2:.2f} s, max diff {np.abs(out.cpu().numpy() - ref).max():.2e}")
