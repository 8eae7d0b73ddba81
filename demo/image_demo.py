"""Counterpart of the reference's ``demo/image_demo.py`` (:12-100) without the visualizer: pose estimation of the people
in one image - the whole image as one box, or ``--bboxes x0,y0,x1,y1;...`` - printed / saved as JSON.

    python demo/image_demo.py IMG configs/td-pm_ProbPose-small_mi355x_coco-256x192.py CHECKPOINT --out-file out.json

CHECKPOINT may be "synthetic" (seeded random weights: plumbing check, BASELINE config 1)."""
import json
import os
import sys
from argparse import ArgumentParser

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = ArgumentParser()
    ap.add_argument("img")
    ap.add_argument("config")
    ap.add_argument("checkpoint")
    ap.add_argument("--out-file", default=None)
    ap.add_argument("--device", default="cuda:0")
    ap.add_argument("--bboxes", default=None, help="x0,y0,x1,y1;x0,y0,x1,y1;... (default: the whole image)")
    ap.add_argument("--precision", default=None, choices=[None, "f16x3", "bf16", "f32"],
                    help="overrides model.precision of the config (default there: f16x3, the mode within 1e-3 of the fp32 reference)")
    args = ap.parse_args()

    from probpose_code_amd import apis, synthetic
    from probpose_code_amd.structures import merge_data_samples

    ckpt = dict(state_dict=synthetic.synthetic_state_dict("small", seed=0, logit_scale=2.0)) if args.checkpoint == "synthetic" else args.checkpoint
    opts = {"model.precision": args.precision} if args.precision else None
    model = apis.init_model(args.config, ckpt, device=args.device, cfg_options=opts)
    boxes = None
    if args.bboxes:
        boxes = np.array([[float(v) for v in b.split(",")] for b in args.bboxes.split(";")], np.float32)
    results = merge_data_samples(apis.inference_topdown(model, args.img, boxes))
    pi = results.pred_instances
    out = [dict(bbox=pi.bboxes[i].tolist(), keypoints=pi.keypoints[i].tolist(), keypoint_scores=pi.keypoint_scores[i].tolist(),
                keypoints_probs=pi.keypoints_probs[i].tolist(), keypoints_visible=pi.keypoints_visible[i].tolist())
           for i in range(len(pi.keypoints))]
    text = json.dumps(out, indent=1)
    if args.out_file:
        open(args.out_file, "w").write(text)
    else:
        print(text)


if __name__ == "__main__":
    main()
