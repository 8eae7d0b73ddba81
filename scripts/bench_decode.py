"""dev only: time the fused Sparsemax + flip-average + decode kernel (pp_probmap_head_decode) at the bs64 shape on logits
of the bench's synthetic model (sparse maps: 2-10 px support) and on dense noise logits (low temperature -> wide support)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from probpose_code_amd import synthetic as S
from probpose_code_amd.engine import ProbPoseEngine
from probpose_code_amd import _lib
B = 64
sd = S.synthetic_state_dict("small", seed=0, logit_scale=2.0)
eng = ProbPoseEngine(sd, 12, precision="bf16")
crops = S.synthetic_crops(B, seed=100).cuda()
eng.fuse_head = False  # planar logits (row-major), the layout pp_probmap_head_decode reads
out = eng.forward(crops, True, S.COCO_FLIP_INDICES)
ws = eng._workspace(B, 2)
logits = ws["logits"].clone()
fi = eng._flip_indices(S.COCO_FLIP_INDICES)
def run(lg, T, nb=B):
    _lib.call("pp_probmap_head_decode", lg.data_ptr(), lg[B:].data_ptr(), fi.data_ptr(), eng.taps.data_ptr(), eng.radius.data_ptr(), nb, 17, 64, 48,
              192.0, 256.0, T, 1.0, None, None, ws["locs"].data_ptr(), ws["keypoints"].data_ptr(), ws["scores"].data_ptr(), None)
for wgs in [int(x) for x in os.environ.get('DECODE_WGS', '5 4 3 2').split()]:
  _lib.set_option("decode_wgs_per_cu", wgs)
  print("decode_wgs_per_cu", wgs)
  for name, lg, T in (("model logits, T=0.5 (sparse)", logits, 0.5), ("noise logits, T=50 (dense support)", torch.randn_like(logits), 50.0)):
    for _ in range(3): run(lg, T)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run(lg, T)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"{name:36s} {us:7.1f} us   {2 * B * 17 * 3072 * 4 / us / 1e3:7.1f} GB/s of logits")
_lib.set_option("decode_wgs_per_cu", 5)
# the dependent chain of ONE workgroup (17 workgroups on 256 CUs: nothing queues): loads -> max -> threshold iterations -> map ->
# box -> row pass -> column pass -> argmax, a dozen barriers; the bs64 launch is 1088 workgroups on 768 slots (3 per CU: 49 KiB LDS)
for _ in range(3): run(logits, 0.5, 1)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): run(logits, 0.5, 1)
e1.record(); torch.cuda.synchronize()
print(f"one crop (17 workgroups): {e0.elapsed_time(e1) / 50 * 1e3:.1f} us per launch = launch overhead + one workgroup's chain")
