# dev only: soak of a launch plan - BASELINE config 4 (ViT-B 384x288, bs 32 + flip: pp_linear_ln_folded with the tile loop, 432-token attention,
# 24 x 18 Winograd): two engines replaying their hipGraphs on two streams at once, every result compared bit for bit with the first
import sys, os, time, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R)
from probpose_code_amd import ProbPoseEngine
from probpose_code_amd import synthetic as S
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
arch = sys.argv[2] if len(sys.argv) > 2 else "base"   # "small": the headline workload (ProbPose-small 256x192, bs 64 + flip)
img, B = ((384, 288), 32) if arch == "base" else ((256, 192), 64)
sd = S.synthetic_state_dict(arch, img_size=img, seed=0, logit_scale=2.0)
engs, xs, want, streams = [], [], [], [torch.cuda.Stream(), torch.cuda.Stream()]
for k in range(2):
    e = ProbPoseEngine(sd, 12, img_size=img, precision="f16x3", input_size=(img[1], img[0]))
    x = S.synthetic_crops(B, img_size=img, seed=30 + k).cuda()
    with torch.cuda.stream(streams[k]):
        for _ in range(4):
            o = e.forward_graph(x, True, S.COCO_FLIP_INDICES)
    torch.cuda.synchronize()
    engs.append(e); xs.append(x); want.append({n: o[n].clone() for n in ("keypoints", "scalars", "scores")})
bad, t0 = 0, time.time()
for it in range(iters):
    outs = []
    for k in range(2):
        with torch.cuda.stream(streams[k]):
            outs.append(engs[k].forward_graph(xs[k], True, S.COCO_FLIP_INDICES))
    torch.cuda.synchronize()
    for k in range(2):
        if not all(torch.equal(outs[k][n], want[k][n]) for n in want[k]):
            bad += 1
            print("MISMATCH iteration", it, "engine", k, flush=True)
print(f"{iters} iterations x 2 engines in {time.time() - t0:.1f} s: {bad} mismatching results")
sys.exit(1 if bad else 0)
