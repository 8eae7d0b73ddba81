"""dev only: a few launches of the fused decode kernel at bs 64 on the synthetic model's logits (sparse maps), the band buffer
sized for argv[1] workgroups per CU - the command scripts/micro/decode_pmc.sh counts."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from probpose_code_amd import synthetic as S
from probpose_code_amd.engine import ProbPoseEngine
from probpose_code_amd import _lib
B = 64
sd = S.synthetic_state_dict("small", seed=0, logit_scale=2.0)
eng = ProbPoseEngine(sd, 12, precision="bf16")
crops = S.synthetic_crops(B, seed=100).cuda()
eng.fuse_head = False
eng.forward(crops, True, S.COCO_FLIP_INDICES)
ws = eng._workspace(B, 2)
logits = ws["logits"].clone()
fi = eng._flip_indices(S.COCO_FLIP_INDICES)
_lib.set_option("decode_wgs_per_cu", int(sys.argv[1]))
for _ in range(5):
    _lib.call("pp_probmap_head_decode", logits.data_ptr(), logits[B:].data_ptr(), fi.data_ptr(), eng.taps.data_ptr(), eng.radius.data_ptr(), B, 17, 64, 48,
              192.0, 256.0, 0.5, 1.0, None, None, ws["locs"].data_ptr(), ws["keypoints"].data_ptr(), ws["scores"].data_ptr(), None)
torch.cuda.synchronize()
for _ in range(5):  # one crop: 17 workgroups on 256 CUs - a workgroup's dependent chain with the CU to itself
    _lib.call("pp_probmap_head_decode", logits.data_ptr(), logits[B:].data_ptr(), fi.data_ptr(), eng.taps.data_ptr(), eng.radius.data_ptr(), 1, 17, 64, 48,
              192.0, 256.0, 0.5, 1.0, None, None, ws["locs"].data_ptr(), ws["keypoints"].data_ptr(), ws["scores"].data_ptr(), None)
torch.cuda.synchronize()
print("radius", eng.radius.tolist())
