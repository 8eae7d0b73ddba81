"""dev only: the fused decode kernel on the PRODUCT's input - the phase-separated logits the f16x3 engine's fused deconvolution head
writes - at bs 64: `decode_phased.py time` prints us per launch for 3 / 4 / 5 workgroups per CU (and bs 512); `decode_phased.py <wgs>`
runs a few launches for scripts/micro/decode_pmc.sh-style counter passes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from probpose_code_amd import synthetic as S
from probpose_code_amd.engine import ProbPoseEngine
from probpose_code_amd import _lib
B = 64
sd = S.synthetic_state_dict("small", seed=0, logit_scale=2.0)
eng = ProbPoseEngine(sd, 12, precision="f16x3")
crops = S.synthetic_crops(B, seed=100).cuda()
eng.forward(crops, True, S.COCO_FLIP_INDICES)
assert eng._logits_phased
ws = eng._workspace(B, 2)
fi = eng._flip_indices(S.COCO_FLIP_INDICES)


def run(lg, nb):
    kp = torch.empty((nb, 17, 2), dtype=torch.float64, device="cuda")
    lo, sc = torch.empty((nb, 17, 2), device="cuda"), torch.empty((nb, 17), device="cuda")
    def go():
        _lib.call("pp_probmap_head_decode_phased", lg.data_ptr(), lg[nb:].data_ptr(), fi.data_ptr(), eng.taps.data_ptr(), eng.radius.data_ptr(), nb, 17, 64, 48,
                  192.0, 256.0, 0.5, 1.0, None, None, lo.data_ptr(), kp.data_ptr(), sc.data_ptr(), None)
    return go


logits = ws["logits"].clone()
if sys.argv[1] == "time":
    lg512 = torch.cat([logits[:B].repeat(8, 1, 1), logits[B:].repeat(8, 1, 1)]).contiguous()
    for wgs in (3, 4, 5, 3):
        _lib.set_option("decode_wgs_per_cu", wgs)
        for name, lg, nb in (("bs 64", logits, B), ("bs 512", lg512, 512)):
            go = run(lg, nb)
            for _ in range(3): go()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): go()
            e1.record(); torch.cuda.synchronize()
            print(f"decode_wgs_per_cu {wgs} {name}: {e0.elapsed_time(e1) / 20 * 1e3:7.1f} us")
else:
    _lib.set_option("decode_wgs_per_cu", int(sys.argv[1]))
    go = run(logits, B)
    for _ in range(5): go()
    torch.cuda.synchronize()
