#!/usr/bin/env python
"""bench.py -- person-crops/sec of the ProbPose hot path on N MI355X (one process per GPU).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch 64]

N>1 is launched by the driver as `python -m torch.distributed.run --nproc-per-node N ... bench.py
--gpus N ...`; ranks shard the crops (64 per GPU, weak scaling), there is no data-path collective
except the final all_gather of the fixed-layout keypoint results.

One "step" = one pass of the hot path over one batch of synthetic crops already resident in HBM.
Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` for the dominant
kernel (measured live with HIP events on the launch stream) and `cpu_baseline` (the oracle timed on
this box's host cores, rank 0, N=1 only, bounded sample).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=64, help="crops per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    return ap.parse_args()


def synthetic_maps(B, K, H, W, seed, device):
    """Sparsemax-like probability maps (sparse, rows sum to 1), seeded."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    x = torch.rand((B, K, H * W), generator=g) ** 6
    x = torch.clamp(x - 0.35, min=0)
    x = x / x.sum(-1, keepdim=True).clamp_min(1e-12)
    return x.reshape(B, K, H, W).to(device)


def cpu_baseline_decode(hm, hmf, n_crops):
    """Oracle (port of the reference's per-sample scipy decode loop, base_head.py:69-77) on the host."""
    from oracle import decode_ref as D

    hm = hm[:n_crops].cpu().numpy()
    hmf = hmf[:n_crops].cpu().numpy()
    t0 = time.perf_counter()
    avg = D.tta_average(hm, hmf)
    for b in range(n_crops):
        D.probmap_decode(avg[b], backend="scipy")
    dt = time.perf_counter() - t0
    return n_crops / dt, dt


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if distributed:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    import probpose_code_amd as pp
    from probpose_code_amd.codecs import ProbMap  # noqa: F401

    B, K, H, W = args.batch, 17, 64, 48
    flip = [0, 2, 1, 4, 3, 6, 5, 8, 7, 10, 9, 12, 11, 14, 13, 16, 15]
    codec = pp.KEYPOINT_CODECS.build(dict(type="ProbMap", input_size=(192, 256), heatmap_size=(48, 64), sigma=-1))
    hm = synthetic_maps(B, K, H, W, 1000 + rank, dev)
    hmf = synthetic_maps(B, K, H, W, 2000 + rank, dev)

    def step():
        return codec.decode_device(hm, hmf, flip)

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        out = step()
    barrier()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        ev[i][0].record()
        out = step()
        ev[i][1].record()
    barrier()
    dt = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        # the one exchange the path has: fixed-layout results gathered over RCCL
        res = torch.cat([out["keypoints"].float(), out["scores"][..., None]], -1)
        gathered = torch.empty((world,) + tuple(res.shape), dtype=res.dtype, device=dev)
        dist.all_gather_into_tensor(gathered, res)
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))

    if rank == 0:
        crops = B * world * args.steps
        alg_bytes = B * K * H * W * 4 * 2  # two f32 maps per crop read once (417 792 B/crop)
        achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
        line = {
            "metric": "person-crops/sec @ 256x192 bs64",
            "value": crops / dt,
            "unit": "crops/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32 maps, f64 accumulate",
            "data": "synthetic",
            "config": {"workload": "PARTIAL PATH (round-1 first slice): fused flip-average + ProbMap decode, "
                       f"bs{B}x17x64x48 probability maps per GPU; backbone/head not yet in the timed region",
                       "crops_per_gpu": B, "parallelism": f"dp{world}"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "kernel": "probmap_decode_kernel<true>", "kernel_ms": kern_ms,
                         "algorithmic_bytes_per_launch": alg_bytes},
        }
        if world == 1 and not args.no_cpu_baseline:
            n = min(B, 32)
            v, secs = cpu_baseline_decode(hm, hmf, n)
            line["cpu_baseline"] = {"value": v, "unit": "crops/s", "cores": 1, "kind": "port",
                                    "sample": f"{n} crops of the same batch, decode stage only, {secs:.1f} s, "
                                              f"1 Python thread as the reference runs it ({os.cpu_count()} host cores present)"}
        print(json.dumps(line))
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
