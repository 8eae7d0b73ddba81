"""dev only: two half-batches on two streams with a FORCED phase offset between them. The fused layer kernel runs all
256 workgroups through the same phases at the same time (HBM-heavy attention / store phases, HBM-free FFN phase); two
half-batch chains half a layer apart would let one chain's HBM bursts fall into the other's FFN. Stream B starts its
chain when stream A has reached stage `--after` (embed, layer0, layer1 ...). Both under hipGraph replay, bs64 total."""
import os, sys, time, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from probpose_code_amd import synthetic as S
from probpose_code_amd.engine import ProbPoseEngine
ap = argparse.ArgumentParser(); ap.add_argument("--after", nargs="*", default=["none", "embed", "layer0", "layer1"]); args = ap.parse_args()
dev = torch.device("cuda", 0)
sd = S.synthetic_state_dict("small", seed=0, logit_scale=2.0)
flip = S.COCO_FLIP_INDICES
def timeit(fn, n=40):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
B = 64
crops = S.synthetic_crops(B, seed=100).to(dev)
eng = ProbPoseEngine(sd, 12, precision="bf16", device=dev)
eng.capture(B, True, flip).copy_(crops)
print(f"one batch of {B}: {timeit(lambda: eng.forward_graph(crops, True, flip)):.3f} ms")
engs = [ProbPoseEngine(sd, 12, precision="bf16", device=dev) for _ in range(2)]
Bh = B // 2
ins = [crops[i * Bh:(i + 1) * Bh].clone() for i in range(2)]
streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
side = torch.cuda.Stream(device=dev)
side.wait_stream(torch.cuda.current_stream(dev))
with torch.cuda.stream(side):
    for e, x in zip(engs, ins):
        for _ in range(2): e.forward(x, True, flip)
torch.cuda.current_stream(dev).wait_stream(side)
torch.cuda.synchronize()
for after in args.after:
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        cur = torch.cuda.current_stream(dev)
        sA, sB = streams
        sA.wait_stream(cur)
        ev = torch.cuda.Event()
        def hook(name, ev=ev, sA=sA):
            if name == after: ev.record(sA)
        engs[0].stage_hook = hook if after != "none" else None
        with torch.cuda.stream(sA):
            if after == "none": ev.record(sA)
            # A's chain; the hook records `ev` on sA when stage `after` has been enqueued
            pass
        # B must be enqueued AFTER the event record exists in capture order: run A first (enqueue only), then B waits on ev
        with torch.cuda.stream(sA):
            engs[0].forward(ins[0], True, flip)
        sB.wait_event(ev)
        with torch.cuda.stream(sB):
            engs[1].forward(ins[1], True, flip)
        cur.wait_stream(sA); cur.wait_stream(sB)
        engs[0].stage_hook = None
    print(f"2 x {Bh} on 2 streams, B starts after A's {after:7s}: {timeit(lambda: g.replay()):.3f} ms")
