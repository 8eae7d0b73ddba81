#!/usr/bin/env python
"""bench.py -- person-crops/sec of the ProbPose top-down inference hot path on N MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch 64] [--precision bf16|f32] [--no-graph]

One process per GPU. N>1 is launched by the driver as
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...`;
the crops shard across ranks (64 per GPU per step, weak scaling, seeds offset by rank), there is no
data-path collective except the all_gather of the fixed-layout keypoint results.

One "step" = the whole hot path over one batch of synthetic uint8 crops ALREADY RESIDENT IN HBM:
preprocess + flip copy -> ViT-S backbone (both flip-test passes) -> ProbMapHead (deconv heatmap branch +
Sparsemax, 4 scalar towers, flip average) -> ProbMap decode -> result tensors copied to pinned host memory
(+ RCCL all_gather of the results when N>1). Random-init (seeded) weights, synthetic crops.

Rank 0 prints ONE JSON line with `roofline` for the dominant kernel -- its launches timed live with HIP events
on the launch stream in an instrumented pass of the same step -- and `cpu_baseline`: the oracle (torch-CPU
model + per-sample scipy decode loop, as the reference runs) timed on this box's host cores on a bounded sample.
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

# /opt/skills/guides/MI355X_MICROARCH.md: dense MFMA peaks, HBM3E spec bandwidth
PEAK_TFLOPS = {"bf16": 2500.0, "f32": 157.3}
HBM_PEAK_GBS = 8000.0
# BASELINE.md 3 / SURVEY 8d: algorithmic FLOPs of ProbPose-S @256x192, MAC = 2
GFLOP_PER_CROP_FLIP = 26.877


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=64, help="crops per GPU per step")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-graph", action="store_true", help="launch kernel by kernel instead of replaying the HIP graph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    return ap.parse_args()


def kernel_work_per_step(eng, B, passes, tag):
    """Algorithmic FLOPs (MAC = 2), algorithmic HBM bytes (operands read once, outputs written once; weights < 1 %
    ignored), launch count and mangled name of one kernel of the launch plan, per step. Tags are the ones the engine
    uses (probpose_code_amd/engine.py)."""
    M = B * passes * eng.Np
    E, Fd, L = eng.E, eng.w.ffn_dims, eng.w.num_layers
    P = eng.Hh * eng.Wh
    f32 = eng.precision == "f32"
    esz = 4 if f32 else 2
    t = "f" if f32 else "DF16b"
    fused_ln = E == 384
    fused_mlp = fused_ln and not f32 and eng.fuse_mlp and Fd % 128 == 0
    fused_proj = fused_mlp and eng.fuse_proj
    c_last = eng.w.deconv_channels[-1]
    final_fl, final_b = 2.0 * (B * passes * P) * eng.K * c_last, B * passes * P * (c_last * esz + eng.K * 4)
    if tag == "vit_layer":  # attention + proj + residual + ln2 + fc1 + GELU + fc2 + residual + LN (+ the next layer's qkv)
        att_fl = 4.0 * (B * passes) * eng.heads * eng.Np * eng.Np * eng.hd  # QK^T + PV
        fl = L * (att_fl + 4.0 * M * E * Fd + 2.0 * M * E * E)
        by = L * (M * 3 * E * 2 + 2 * M * E * 4) + M * E * 2  # qkv in, residual stream in + out; features out once
        if eng.fuse_qkv:
            fl += (L - 1) * 2.0 * M * E * 3 * E
            by += (L - 1) * M * 3 * E * 2
        else:
            by += (L - 1) * M * E * 2
        return fl, by, L, "_ZN2pp3mlp17mlp_res_ln_kernelILb1ELb%dELb1EEEvNS0_6ParamsE" % int(eng.fuse_qkv)
    if tag == "proj_mlp_res_ln":  # proj + residual + ln2 + fc1 + GELU + fc2 + residual + LN (+ the next layer's qkv)
        fl = L * (4.0 * M * E * Fd + 2.0 * M * E * E)
        by = L * (M * E * 2 + 2 * M * E * 4) + M * E * 2  # attention rows in, residual stream in + out; features out once
        if eng.fuse_qkv:  # L - 1 launches also apply the next qkv Linear and write qkv instead of the LayerNorm output
            fl += (L - 1) * 2.0 * M * E * 3 * E
            by += (L - 1) * M * 3 * E * 2
        else:
            by += (L - 1) * M * E * 2
        return fl, by, L, "_ZN2pp3mlp17mlp_res_ln_kernelILb1ELb%dELb0EEEvNS0_6ParamsE" % int(eng.fuse_qkv)
    if tag == "mlp_res_ln":  # fc1 + GELU + fc2 + residual + LN per layer
        return L * 4.0 * M * E * Fd, L * (M * E * 2 + 2 * M * E * 4 + M * E * 2), L, "_ZN2pp3mlp17mlp_res_ln_kernelILb0ELb0ELb0EEEvNS0_6ParamsE"
    if tag == "gemm_res_ln":  # patch embed, proj (and fc2 when the FFN is not fused) + residual + LN
        fl = 2.0 * M * E * 768
        by = M * 768 * esz + M * E * (4 + esz)
        n = 1
        if not fused_proj:
            fl += L * 2.0 * M * E * E
            by += L * (M * E * esz + 2 * M * E * 4 + M * E * esz)
            n += L
        if not fused_mlp:
            fl += L * 2.0 * M * E * Fd
            by += L * (M * Fd * esz + 2 * M * E * 4 + M * E * esz)
            n += L
        return fl, by, n, f"_ZN2pp2rl18gemm_res_ln_kernelI{t}EEvNS0_6ParamsE"
    # plain dense layers: qkv (+ fc1 when the FFN is not fused) write the operand dtype; the final 1x1 conv (+ the
    # residual GEMMs when E != 384) write fp32
    nq = 1 if (fused_proj and eng.fuse_qkv) else L  # qkv Linears left to the plain GEMM
    act_fl, act_by, act_n = nq * 2.0 * M * 3 * E * E, nq * (M * E * esz + M * 3 * E * esz), nq
    if not fused_mlp:
        act_fl += L * 2.0 * M * E * Fd
        act_by += L * (M * E * esz + M * Fd * esz)
        act_n += L
    res_fl, res_by, res_n = final_fl, final_b, 1
    if not fused_ln:
        res_fl += 2.0 * M * E * 768 + L * 2.0 * M * (E * E + E * Fd)
        res_by += M * 768 * esz + 2 * M * E * 4 + L * (M * E * esz + M * Fd * esz + 4 * M * E * 4)
        res_n += 1 + 2 * L
    if f32:  # every output is fp32: one instantiation
        return act_fl + res_fl, act_by + res_by, act_n + res_n, "_ZN2pp11gemm_kernelIfLi0ELb0EEEvNS_10GemmParamsE"
    if tag == "gemm_bf16out":
        return act_fl, act_by, act_n, "_ZN2pp11gemm_kernelIDF16bLi0ELb1EEEvNS_10GemmParamsE"
    return res_fl, res_by, res_n, "_ZN2pp11gemm_kernelIDF16bLi0ELb0EEEvNS_10GemmParamsE"


def pmc_traffic(kernel_mangled):
    """HBM bytes per launch of `kernel_mangled` from the committed rocprofv3 --pmc passes (profiles/), or None."""
    path = os.path.join(ROOT, "profiles", "r01_bf16_bs64_hbm_traffic.json")
    try:
        ks = json.load(open(path))["kernels"]
        if kernel_mangled in ks:
            return ks[kernel_mangled]["hbm_bytes_per_launch"]
        # rocprofv3 reports some names demangled: match on the template arguments of the fused layer kernel
        if "mlp_res_ln_kernel" in kernel_mangled:
            import re as _re
            bits = _re.search(r"ILb(\d)ELb(\d)ELb(\d)E", kernel_mangled)
            want = "<" + ", ".join("true" if b == "1" else "false" for b in bits.groups()) + ">" if bits else "<"
            for k, v in ks.items():
                if "mlp_res_ln_kernel" in k and want in k:
                    return v["hbm_bytes_per_launch"]
        return None
    except Exception:  # noqa: BLE001
        return None


def cpu_baseline(sd, crops_cpu, n_crops, threads):
    from oracle import model_ref as M
    from probpose_code_amd import synthetic as S

    torch.set_num_threads(threads)
    M.predict(sd, crops_cpu[:2], 12, S.IMG_MEAN, S.IMG_STD)  # warm-up (thread pool, allocator)
    done, t0 = 0, time.perf_counter()
    ref = None
    while done < n_crops:
        n = min(crops_cpu.shape[0], n_crops - done)
        r = M.predict(sd, crops_cpu[:n], 12, S.IMG_MEAN, S.IMG_STD)
        ref = ref or r
        done += n
    dt = time.perf_counter() - t0
    return done / dt, dt, ref


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
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", device_id=dev)

    from probpose_code_amd import synthetic as S
    from probpose_code_amd.dist import ResultGather
    from probpose_code_amd.engine import ProbPoseEngine

    B = args.batch
    sd = S.synthetic_state_dict("small", seed=0, logit_scale=2.0)  # same weights on every rank
    crops_cpu = S.synthetic_crops(B, seed=100 + rank)              # a different shard per rank
    crops = crops_cpu.to(dev)
    eng = ProbPoseEngine(sd, 12, precision=args.precision, device=dev)
    flip = S.COCO_FLIP_INDICES
    gather = ResultGather(B, eng.K, dev, world)  # fixed-layout result record, pinned host copy, RCCL all_gather
    use_graph = not args.no_graph
    if use_graph:
        eng.capture(B, True, flip).copy_(crops)

    def step():
        out = eng.forward_graph(crops, True, flip) if use_graph else eng.forward(crops, True, flip)
        return gather(out)

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # ---- instrumented pass: HIP events around every launch, on the launch stream (not inside the timed region)
    eng.profile = {}
    for _ in range(5):
        eng.forward(crops, True, flip)
    torch.cuda.synchronize()
    prof = {k: [a.elapsed_time(b) for a, b in v] for k, v in eng.profile.items()}
    eng.profile = None

    if rank == 0:
        per_tag = {k: (float(np.sum(v)) / 5, len(v) // 5) for k, v in prof.items()}  # ms per step, launches per step
        dom = max(per_tag, key=lambda k: per_tag[k][0])
        dom_ms, dom_n = per_tag[dom]
        line = {
            "metric": "person-crops/sec @ 256x192 bs64",
            "value": B * world * args.steps / dt,
            "unit": "crops/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.precision,
            "dtype_detail": "bf16 MFMA operands, fp32 accumulate; fp32 LayerNorm/softmax/residual stream/Sparsemax; f64 decode"
            if args.precision == "bf16" else "exact-fp32 MFMA products (v_mfma_f32_16x16x4_f32), fp32 accumulate; f64 decode",
            "data": "synthetic",
            "config": {
                "workload": f"ProbPose-small (ViT-S 12x384, 12 heads x 32) bs{B} random 256x192 uint8 crops per GPU, "
                            "flip_test=True, seeded random-init weights: preprocess -> backbone x2 passes -> ProbMapHead "
                            "(heatmap branch + Sparsemax + 4 towers) -> ProbMap decode -> results in pinned host memory",
                "crops_per_gpu": B, "global_batch": B * world, "parallelism": f"dp{world}",
                "launch": "hipGraph replay" if use_graph else "eager launches",
                "gflop_per_crop": GFLOP_PER_CROP_FLIP,
            },
            "path_tflops": B * world * args.steps * GFLOP_PER_CROP_FLIP / dt / 1e3,
            "kernel_ms_per_step": {k: round(v[0], 4) for k, v in sorted(per_tag.items(), key=lambda kv: -kv[1][0])},
        }
        if dom in ("vit_layer", "proj_mlp_res_ln", "mlp_res_ln", "gemm_res_ln", "gemm_bf16out", "gemm_f32out"):
            fl, alg_bytes, n, mangled = kernel_work_per_step(eng, B, 2, dom)
            assert n == dom_n, (dom, n, dom_n)
            secs = dom_ms / n * 1e-3
            peak = PEAK_TFLOPS[args.precision]
            ridge = peak * 1e12 / (HBM_PEAK_GBS * 1e9)
            intensity = fl / alg_bytes
            mfma_view = {"achieved_TFLOPs": fl / n / secs / 1e12, "peak_TFLOPs": peak, "frac": fl / n / secs / 1e12 / peak}
            hbm_view = {"achieved_GBps": alg_bytes / n / secs / 1e9, "peak_GBps": HBM_PEAK_GBS,
                        "frac": alg_bytes / n / secs / 1e9 / HBM_PEAK_GBS}
            bound = "mfma" if intensity >= ridge else "hbm"  # which side of the ridge the kernel's algorithm sits on
            traffic = pmc_traffic(mangled) if (args.precision == "bf16" and B == 64) else None
            line["roofline"] = {
                "bound": bound,
                "achieved": mfma_view["achieved_TFLOPs"] if bound == "mfma" else hbm_view["achieved_GBps"],
                "peak": peak if bound == "mfma" else HBM_PEAK_GBS,
                "unit": "TFLOP/s" if bound == "mfma" else "GB/s",
                "frac": mfma_view["frac"] if bound == "mfma" else hbm_view["frac"],
                "traffic": traffic,
                "traffic_source": "profiles/r01_bf16_bs64_hbm_traffic.json (separate rocprofv3 --pmc passes)" if traffic else None,
                "kernel": dom, "kernel_mangled": mangled, "launches_per_step": n, "avg_launch_ms": dom_ms / n,
                "algorithmic_gflop_per_launch": fl / n / 1e9, "algorithmic_mbytes_per_launch": alg_bytes / n / 1e6,
                "arithmetic_intensity_flop_per_byte": intensity, "ridge_flop_per_byte": ridge,
                "mfma_view": mfma_view, "hbm_view": hbm_view,
            }
        else:  # pragma: no cover - a different kernel dominates: report its time, flag the roofline as undefined
            line["roofline"] = {"bound": "mfma", "achieved": None, "peak": PEAK_TFLOPS[args.precision], "unit": "TFLOP/s",
                                "frac": None, "traffic": None, "kernel": dom, "avg_launch_ms": dom_ms / dom_n}
        if world == 1 and not args.no_cpu_baseline:
            threads = min(16, len(os.sched_getaffinity(0)))
            try:
                q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
                if q != "max":
                    threads = max(1, min(threads, int(int(q) / int(p))))
            except Exception:  # noqa: BLE001
                pass
            v, secs, ref = cpu_baseline(sd, crops_cpu, 4 * B, threads)
            line["cpu_baseline"] = {
                "value": v, "unit": "crops/s", "cores": threads, "kind": "port",
                "sample": f"{4 * B} crops (the same bs{B} batch x4) through the oracle: torch-CPU fp32 model on {threads} "
                          f"threads + per-sample scipy decode loop on 1 thread, {secs:.1f} s; host has "
                          f"{os.cpu_count()} logical CPUs, cgroup quota {threads}",
            }
            if not args.no_parity:
                out = eng.forward(crops, True, flip)
                kp = out["keypoints"].cpu().numpy()[:, None]
                d = np.abs(kp - ref["keypoints_input_space"]).max(-1)
                same = d < 2.0
                line["parity_vs_oracle"] = {
                    "precision": args.precision, "crops": B,
                    "keypoint_linf_px_input_space_same_argmax": float(d[same].max()),
                    "argmax_flips": int((~same).sum()), "keypoints": int(same.size),
                    "probs_linf": float(np.abs(out["scalars"][0].cpu().numpy()[:, None] - ref["keypoints_probs"]).max()),
                }
        print(json.dumps(line))
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
