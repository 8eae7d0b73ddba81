"""dev: which ingredient of `test_step` + `test_step_stream` makes a later graph replay crash? (variants run as subprocesses)"""
import os
import random
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) < 2:
    for v in ("step+pinned_churn", "step+zeros_churn", "step+sync", "step+gather_ctor", "step+pipeline_ctor"):
        r = subprocess.run([sys.executable, "-X", "faulthandler", __file__, v], capture_output=True, text=True)
        tail = [l for l in (r.stdout + r.stderr).splitlines() if l.strip() and not l.startswith("  File") and not l.startswith("Extension")][-3:]
        print(f"== {v}: rc {r.returncode}: " + " | ".join(tail), flush=True)
    sys.exit(0)

import numpy as np  # noqa: E402
import torch  # noqa: E402

sys.path.insert(0, ROOT)
from probpose_code_amd import apis  # noqa: E402
from probpose_code_amd import synthetic as S  # noqa: E402
from probpose_code_amd.dist import ResultGather  # noqa: E402
from probpose_code_amd.pipeline import StepPipeline  # noqa: E402

variant = sys.argv[1]
cfg = os.path.join(ROOT, "configs", "td-pm_ProbPose-small_mi355x_coco-256x192.py")
sd = S.synthetic_state_dict("small", seed=0, logit_scale=2.0)
model = apis.init_model(cfg, {"state_dict": sd}, device="cuda:0")
eng = model.engine
fi = S.COCO_FLIP_INDICES
sizes = [1, 2, 3, 4, 5, 6, 8, 12, 16, 17, 18, 24]
crops = {B: S.synthetic_crops(B, seed=B).cuda() for B in sizes}
metas = {B: S.whole_image_bbox_meta(B) for B in sizes}
rng = random.Random(3)
gather = ResultGather(64, eng.K, eng.device, 1)
with torch.no_grad():
    for it in range(1500):
        B = rng.choice(sizes)
        if rng.random() < 0.6:
            if os.environ.get("REPRO_VERBOSE") == "1":
                print("it", it, "B", B, "graph", eng.has_graph(B, True, fi), "keys", [k[0] for k in eng._graphs], "captures", eng.graph_captures, flush=True)
            model.test_step(apis.pack_crops(crops[B], *metas[B], model.dataset_meta))
        elif variant == "step+engine_eager":
            eng.forward(crops[B], True, fi)
        elif variant == "step+pipeline_ctor":
            StepPipeline(eng, 64, fi, flip_test=True, depth=1, use_graph="full")
        elif variant == "step+pipeline_submit":
            p = StepPipeline(eng, 64, fi, flip_test=True, depth=1, use_graph="full")
            p.result(p.submit(crops[B]))
        elif variant == "step+pinned_churn":
            torch.empty((1, 65, 17, 7), dtype=torch.float64, pin_memory=True)
        elif variant == "step+zeros_churn":
            torch.zeros((65, 17, 7), dtype=torch.float64, device="cuda")
            torch.empty((1, 65, 17, 7), dtype=torch.float64, device="cuda")
        elif variant == "step+sync":
            torch.cuda.synchronize()
        elif variant == "step+gather_ctor":
            ResultGather(64, eng.K, eng.device, 1)
        elif variant == "step+gather_only":
            out = eng.forward(crops[B], True, fi)
            gather(out)
            gather.wait()
torch.cuda.synchronize()
print("ok", variant, "captures", eng.graph_captures)
