#!/usr/bin/env python
"""bench.py -- person-crops/sec of the ProbPose top-down inference hot path on N MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch 64] [--precision f16x3|bf16|f32] [--no-graph] [--in-flight 2]

One process per GPU. The driver launches N>1 as
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...`;
started WITHOUT that wrapper (`python bench.py --gpus N`, WORLD_SIZE unset) the script re-executes itself under
`torch.distributed.run` (the reference's tools/dist_test.sh:13-23 does the same around tools/test.py). Either way the
world size must equal --gpus: anything else is refused with a non-zero exit code, never silently run on fewer GPUs.
The crops shard across ranks (64 per GPU per step, weak scaling, seeds offset by rank); there is no data-path
collective except the all_gather of the fixed-layout keypoint results.

One "step" = the whole hot path over one batch of synthetic uint8 crops ALREADY RESIDENT IN HBM:
preprocess + flip copy -> ViT-S backbone (both flip-test passes) -> ProbMapHead (deconv heatmap branch +
Sparsemax, 4 scalar towers, flip average) -> ProbMap decode -> result tensors copied to pinned host memory
(+ RCCL all_gather of the results when N>1). Random-init (seeded) weights, synthetic crops.
Steps are submitted through probpose_code_amd.pipeline.StepPipeline: by default TWO steps are in flight (each slot its own
HIP stream, workspace, captured graph and pinned record buffer), so the low-occupancy tail of one batch shares the chip with
the head of the next; each step is the complete launch sequence and delivers its own record; all K have completed at the
closing synchronize. `one_step_in_flight` in the JSON line is the same engine strictly one batch at a time.

Rank 0 prints ONE JSON line:
  * `value` etc. for --precision. The default is `f16x3` (split-fp16 operands, three fp16 MFMAs per product): the fastest
    mode whose keypoints / probabilities are within the path's 1e-3 tolerance of the fp32 reference - the headline is the
    number the tolerance allows, not the fastest arithmetic the chip has;
  * `roofline` for that mode's dominant kernel -- its launches timed live with HIP events on the launch stream in an
    instrumented pass of the same step;
  * `parity_vs_oracle`: the outputs of the LAST TIMED STEP (the hipGraph replay that was measured, not a separate
    eager run) against the CPU oracle on the same crops;
  * `throughput_mode`: the same bench (timing, roofline, parity of the replayed output) in bf16 - 3x the rate, 0.3 - 0.45 px
    and a few % argmax flips away from the reference: reported with its parity numbers, never as `value`;
  * `config4`: a bounded run of BASELINE config 4 (ProbPose-base = ViT-B 12 x 768, 384x288 crops, bs 32, flip test) in both
    modes, with its dominant kernel's roofline and its parity against the oracle on a few crops;
  * `cpu_baseline`: the oracle (torch-CPU model + per-sample scipy decode loop, as the reference runs) timed on this
    box's host cores on a bounded sample.

`--stub` (tests only): a fake engine on CPU tensors with the gloo backend -- exercises the launcher / sharding /
gather / reduction plumbing of this file where there is no GPU (tests/test_bench_launcher.py).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# /opt/skills/guides/MI355X_MICROARCH.md: dense MFMA peaks, HBM3E spec bandwidth. f16x3 runs on the fp16 MFMA pipe
# (same 2.5 PF dense peak as bf16) and spends THREE MFMAs per algorithmic product (hi*hi + hi*lo + lo*hi): the
# roofline prices algorithmic FLOPs against the datasheet peak, `derived_ceiling` states the 1/3 the format allows.
PEAK_TFLOPS = {"bf16": 2500.0, "f16x3": 2500.0, "f32": 157.3}
MFMA_PER_PRODUCT = {"bf16": 1, "f16x3": 3, "f32": 1}
HBM_PEAK_GBS = 8000.0
# BASELINE.md 3 / SURVEY 8d: algorithmic FLOPs of ProbPose-S @256x192, MAC = 2
GFLOP_PER_CROP_FLIP = 26.877
PARITY_PRECISION = "f16x3"  # the fastest mode that meets north_star's 1e-3: the default and the headline
THROUGHPUT_PRECISION = "bf16"  # secondary record (fails 1e-3: 0.3 - 0.45 px)
DTYPE_DETAIL = {
    "bf16": "bf16 MFMA operands, fp32 accumulate; fp32 LayerNorm/softmax/residual stream/Sparsemax; f64 decode",
    "f16x3": "split-fp16 MFMA operands (x = hi + lo, hi*hi + hi*lo + lo*hi on v_mfma_f32_16x16x32_f16: 22+ operand bits), "
             "fp32 accumulate; fp32 LayerNorm/softmax/residual stream/Sparsemax; f64 decode",
    "f32": "exact-fp32 MFMA products (v_mfma_f32_16x16x4_f32), fp32 accumulate; f64 decode",
}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=64, help="crops per GPU per step")
    ap.add_argument("--precision", default=PARITY_PRECISION, choices=sorted(PEAK_TFLOPS))
    ap.add_argument("--no-graph", action="store_true", help="launch kernel by kernel instead of replaying the HIP graph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--no-second-mode", "--no-parity-mode", dest="no_second_mode", action="store_true",
                    help="skip the second bench in the other precision (bf16 `throughput_mode` beside an f16x3 headline and vice versa)")
    ap.add_argument("--no-config4", action="store_true", help="skip the bounded ViT-B 384x288 record")
    ap.add_argument("--config4-batch", type=int, default=64, help="batch size of the `config4` record (SURVEY 8d: 64)")
    ap.add_argument("--config4-quick", action="store_true", help="with --config4-only: f16x3 at --config4-batch only, 3 timed steps (counter passes)")
    ap.add_argument("--config4-only", action="store_true",
                    help="run ONLY BASELINE config 4 (ViT-B 384x288) - for the profile passes of scripts/collect_profiles.sh; prints its record as the JSON line")
    ap.add_argument("--no-clock-probe", action="store_true", help="skip the probed second pass that records the shader clock (profile passes)")
    ap.add_argument("--no-drop-in", action="store_true", help="skip the `drop_in` record (model.test_step / test_step_stream timed)")
    ap.add_argument("--no-small-batch", action="store_true", help="skip the `small_batch` record (latency of B = 1, 2, 4, 8 one step in flight)")
    ap.add_argument("--no-bs512-decode", action="store_true",
                    help="skip roofline_targets.head_decode_bs512 (counter passes: its launches would mix into the step's decode kernel)")
    ap.add_argument("--in-flight", type=int, default=2,
                    help="steps in flight (StepPipeline slots: own HIP stream, workspace and graph each); 1 = strictly one batch "
                         "at a time on one stream")
    ap.add_argument("--stub", action="store_true", help="tests only: fake engine on CPU + gloo (no GPU needed)")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------ launcher
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def relaunch_under_torchrun(args, argv):
    """`python bench.py --gpus N` without a launcher: become `torch.distributed.run --nproc-per-node N bench.py ...`."""
    if not args.stub:
        have = torch.cuda.device_count()
        if have < args.gpus:
            print(f"bench.py: --gpus {args.gpus} but this node has {have} visible GPU(s); refusing to run on fewer",
                  file=sys.stderr)
            return 2
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL needs it on this host driver
    return subprocess.call(cmd, env=env)


# ------------------------------------------------------------------------------------------ accounting
def kernel_work_per_step(eng, B, passes, tag):
    """Algorithmic FLOPs (MAC = 2), algorithmic HBM bytes (operands read once, outputs written once; weights < 1 %
    ignored), launch count and the kernel's name as rocprofv3 lists it, of one kernel tag of the launch plan, per step.
    Tags are the ones the engine uses (probpose_code_amd/engine.py). Returns None for a tag without a formula."""
    M = B * passes * eng.Np
    E, Fd, L = eng.E, eng.w.ffn_dims, eng.w.num_layers
    P = eng.Hh * eng.Wh
    wide = eng.precision != "bf16"  # f32 and split-fp16 operands are 4 bytes per element
    esz = 4 if wide else 2
    t = {"bf16": "DF16b", "f32": "f", "f16x3": "NS_6SplitHE"}[eng.precision]
    op_fmt = {"bf16": 1, "f32": 0, "f16x3": 2}[eng.precision]  # template argument OUT of an operand-format output
    fused_ln = E == 384
    fused_mlp = fused_ln and not wide and eng.fuse_mlp and Fd % 128 == 0
    fused_proj = fused_mlp and eng.fuse_proj
    ffn_split = bool(getattr(eng, "_ffn_packed", None))  # f16x3: fc1 + GELU + fc2 + residual + LN in one launch (pp_ffn_split.hip)
    proj_split = bool(getattr(eng, "_proj_packed", None))  # f16x3: ... with projection + residual + ln2 in front, same launch
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
    if tag == "ffn_split":  # f16x3: fc1 + GELU + fc2 + residual + LN per layer; h in, residual in, x out, h out (4 bytes each)
        from probpose_code_amd import _lib
        pair = _lib.get_option("ffn_pair") != 0 and (Fd // 128) % 2 == 0  # (launch_dma_form's choice: hidden chunks in pairs)
        name = "_ZN2pp3ffd19ffn_dma_pair_kernelENS_3ffs6ParamsE" if pair else "_ZN2pp3ffd14ffn_dma_kernelENS_3ffs6ParamsE"
        return L * 4.0 * M * E * Fd, L * 4 * M * E * 4, L, name
    if tag == "proj_ffn_split":  # f16x3: proj + residual + ln2 + fc1 + GELU + fc2 + residual + LN per layer; attention rows in,
        # residual in, x out, h out (4 bytes each); the ln2 rows a workgroup parks in L2 and streams back are not algorithmic bytes
        from probpose_code_amd import _lib
        pair = _lib.get_option("ffn_pair") != 0 and (Fd // 128) % 2 == 0
        name = "_ZN2pp3ffd24proj_ffn_dma_pair_kernelENS_3ffs6ParamsE" if pair else "_ZN2pp3ffd19proj_ffn_dma_kernelENS_3ffs6ParamsE"
        by = L * 4 * M * E * 4
        if getattr(eng, "ln_fold_fused", False):  # the folded chain: fold2 (first layer), fold3 (layers between), fold1 (last) of the paired kernel;
            name = "_ZN2pp3ffd30proj_ffn_dma_pair_fold3_kernelENS_3ffs6ParamsE"  # every layer but the last writes its rows ONCE (operand format, + 8 B of statistics)
            by = (3 * (L - 1) + 4) * M * E * 4 + (L - 1) * M * 8
        return L * (4.0 * M * E * Fd + 2.0 * M * E * E), by, L, name
    if tag == "qkv_attention":  # f16x3: qkv Linear + attention per (sequence, head); LayerNorm rows in, attention rows out
        att_fl = 4.0 * (B * passes) * eng.heads * eng.Np * eng.Np * eng.hd
        return L * (2.0 * M * 3 * E * E + att_fl), L * 2 * M * E * 4, L, "_ZN2pp3qka26qkv_attention_split_kernelENS0_6ParamsE"
    if tag == "gemm_res_ln":  # patch embed, proj (and fc2 when the FFN is not fused) + residual + LN
        fl = 2.0 * M * E * 768
        by = M * 768 * esz + M * E * (4 + esz)
        n = 1
        if not fused_proj and not proj_split:
            fl += L * 2.0 * M * E * E
            by += L * (M * E * esz + 2 * M * E * 4 + M * E * esz)
            n += L
        if not fused_mlp and not ffn_split:
            fl += L * 2.0 * M * E * Fd
            by += L * (M * Fd * esz + 2 * M * E * 4 + M * E * esz)
            n += L
        return fl, by, n, f"_ZN2pp2rl18gemm_res_ln_kernelI{t}NS0_3CfgILi96ELi1ELi3ELi4ELi3EEEEEvNS0_6ParamsE"
    if tag == "linear_fold":  # f16x3 without a fused layer kernel (ViT-B): qkv, proj, fc1, fc2 of every layer through pp_linear_ln_folded (LayerNorms
        # folded in): operand-format rows in and out (4 bytes per element), the residual stream read and written by proj / fc2
        fl = L * 2.0 * M * (3 * E * E + E * E + 2 * E * Fd)
        by = L * 4 * M * ((E + 3 * E) + (E + 2 * E) + (E + Fd) + (Fd + 2 * E))
        return fl, by, 4 * L, "_ZN2pp3ldm17linear_dma_kernelILi1EEEvNS0_6ParamsE|_ZN2pp3ldm17linear_dma_kernelILi2EEEvNS0_6ParamsE"
    if tag not in ("gemm_bf16out", "gemm_f32out"):
        return None
    # plain dense layers: qkv (+ fc1 when the FFN is not fused) write the operand dtype; the final 1x1 conv (+ the
    # residual GEMMs when E != 384) write fp32
    nq = 0 if getattr(eng, "fuse_qkv_attn", False) else (1 if (fused_proj and eng.fuse_qkv) else L)  # qkv Linears left to the plain GEMM
    act_fl, act_by, act_n = nq * 2.0 * M * 3 * E * E, nq * (M * E * esz + M * 3 * E * esz), nq
    if not fused_mlp and not ffn_split:
        act_fl += L * 2.0 * M * E * Fd
        act_by += L * (M * E * esz + M * Fd * esz)
        act_n += L
    res_fl, res_by, res_n = final_fl, final_b, 1
    if not fused_ln:
        res_fl += 2.0 * M * E * 768 + L * 2.0 * M * (E * E + E * Fd)
        res_by += M * 768 * esz + 2 * M * E * 4 + L * (M * E * esz + M * Fd * esz + 4 * M * E * 4)
        res_n += 1 + 2 * L
    if eng.precision == "f32":  # every output is fp32: one instantiation
        return act_fl + res_fl, act_by + res_by, act_n + res_n, "_ZN2pp11gemm_kernelIfLi0ELi0EEEvNS_10GemmParamsE"
    if tag == "gemm_bf16out":  # "operand-dtype output": bf16 or split-fp16
        if eng.precision == "f16x3" and (3 * E) % 192 == 0 and Fd % 192 == 0:
            from probpose_code_amd import _lib
            if _lib.get_option("linear_dma") != 0 and ((M + 191) // 192) * ((3 * E) // 192) >= 512:
                # the twelve-wave kernel (pp_linear_dma.hip: pp_gemm tries it FIRST, at every K >= 64)
                return act_fl, act_by, act_n, "_ZN2pp3ldm17linear_dma_kernelILi0EEEvNS0_6ParamsE"
        if eng.precision == "f16x3" and E >= 768 and (3 * E) % 192 == 0 and Fd % 192 == 0:
            # K >= 768: the wide-tile split kernel (256 x 192 tiles; the fp32-output Linear layers run on the same instantiation)
            return act_fl, act_by, act_n, "_ZN2pp6psplit18panel_split_kernelILi0ELi8ELi3ELb1ELi2ELb0ELb0EEEvNS_10GemmParamsE"
        return act_fl, act_by, act_n, f"_ZN2pp11gemm_kernelI{t}Li0ELi{op_fmt}EEEvNS_10GemmParamsE"
    return res_fl, res_by, res_n, f"_ZN2pp11gemm_kernelI{t}Li0ELi0EEEvNS_10GemmParamsE"


def pmc_traffic(kernel_mangled, precision, B, prefix=""):
    """HBM bytes per launch of `kernel_mangled` from the committed rocprofv3 --pmc passes (profiles/), or None."""
    if B != 64:
        return None, None
    short = {"_ZN2pp3ffd19proj_ffn_dma_kernelENS_3ffs6ParamsE": "pp::ffd::proj_ffn_dma_kernel(",
             "_ZN2pp3ffd24proj_ffn_dma_pair_kernelENS_3ffs6ParamsE": "pp::ffd::proj_ffn_dma_pair_kernel(",
             "_ZN2pp3qka26qkv_attention_split_kernelENS0_6ParamsE": "pp::qka::qkv_attention_split_kernel("}.get(kernel_mangled)
    short = short or {"_ZN2pp6psplit18panel_split_kernelILi0ELi8ELi3ELb1ELi2ELb0ELb0EEEvNS_10GemmParamsE": "void pp::psplit::panel_split_kernel<0, 8, 3, true, 2, false, false>("}.get(kernel_mangled)
    names = (f"r06_{prefix}{precision}_bs64_hbm_traffic.json", f"r05_{prefix}{precision}_bs64_hbm_traffic.json", f"r04_{prefix}{precision}_bs64_hbm_traffic.json") if prefix else \
        (f"r06_{precision}_bs64_hbm_traffic.json", f"r05_{precision}_bs64_hbm_traffic.json", f"r04_{precision}_bs64_hbm_traffic.json", f"r03_{precision}_bs64_hbm_traffic.json",
         f"r02_{precision}_bs64_hbm_traffic.json", f"r01_{precision}_bs64_hbm_traffic.json")
    for name in names:
        path = os.path.join(ROOT, "profiles", name)
        try:
            ks = json.load(open(path))["kernels"]
        except Exception:  # noqa: BLE001
            continue
        if "proj_ffn_dma_pair_fold" in kernel_mangled:  # the three instantiations of the folded chain share the tag: launch-weighted mean
            hit = [v for k, v in ks.items() if "proj_ffn_dma_pair_fold" in k]
            if hit:
                n = sum(v["launches"] for v in hit)
                return int(sum(v["hbm_bytes_per_launch"] * v["launches"] for v in hit) / n), "profiles/" + name
            continue
        if "linear_dma_kernelILi" in kernel_mangled:  # the twelve-wave Linear kernel's instantiations that share a tag (MODE 1 + MODE 2 of the
            # folded-LayerNorm plan, each with its compile-time epilogue switches: rocprofv3 lists them as linear_dma_kernel<1, 0, 1, 0, 0> ...)
            modes = [m[m.index("linear_dma_kernelILi") + len("linear_dma_kernelILi")] for m in kernel_mangled.split("|")]
            hit = [v for k, v in ks.items() if any(("linear_dma_kernel<%s," % md) in k or ("linear_dma_kernel<%s>" % md) in k for md in modes)]
            if hit:
                n = sum(v["launches"] for v in hit)
                return int(sum(v["hbm_bytes_per_launch"] * v["launches"] for v in hit) / n), "profiles/" + name
            continue
        if kernel_mangled in ks:
            return ks[kernel_mangled]["hbm_bytes_per_launch"], "profiles/" + name
        if short:  # rocprofv3 lists kernels without template arguments demangled
            for k, v in ks.items():
                if k.startswith(short):
                    return v["hbm_bytes_per_launch"], "profiles/" + name
        # rocprofv3 reports some names demangled: match on the template arguments of the fused layer kernel
        if "mlp_res_ln_kernel" in kernel_mangled:
            import re as _re
            bits = _re.search(r"ILb(\d)ELb(\d)ELb(\d)E", kernel_mangled)
            want = "<" + ", ".join("true" if b == "1" else "false" for b in bits.groups()) + ">" if bits else "<"
            for k, v in ks.items():
                if "mlp_res_ln_kernel" in k and want in k:
                    return v["hbm_bytes_per_launch"], "profiles/" + name
    return None, None


def profiled_duration(kernel_mangled, precision, B, prefix=""):
    """Average duration (ms) of `kernel_mangled` in the committed rocprofv3 kernel trace of the one-step-in-flight run
    (profiles/r06_<prefix><precision>_bs64_kernel_stats_one_in_flight.csv), with the file it came from - what a reader who recomputes `frac` from
    profiles/ divides by. rocprofv3's tracing itself slows the kernels by a few per cent (the *_under_rocprof*.json runs carry the clock)."""
    if B != 64:
        return None, None
    import csv
    import re as _re

    def norm(name):  # Itanium-mangled nested name -> its last identifier (the kernel's own name, as rocprofv3 prints it)
        i, last = (3 if name.startswith("_ZN") else 0), name
        while i < len(name) and name[i].isdigit():
            j = i
            while name[j].isdigit():
                j += 1
            n_ = int(name[i:j])
            last = name[j:j + n_]
            i = j + n_
        return last

    wants = [norm(k) for k in kernel_mangled.split("|")]
    modes = [m[m.index("kernelILi") + len("kernelILi")] for m in kernel_mangled.split("|") if "kernelILi" in m]
    for rnd in ("r06", "r05"):
        name = f"{rnd}_{prefix}{precision}_bs64_kernel_stats_one_in_flight.csv"
        try:
            rows = list(csv.DictReader(open(os.path.join(ROOT, "profiles", name))))
        except OSError:
            continue
        hit = []
        for r in rows:
            kn = r["Name"]
            if not any(w in kn for w in wants):
                continue
            if modes and not any((f"_kernel<{md}," in kn or f"_kernel<{md}>" in kn) for md in modes):
                continue
            if "fold3" in kernel_mangled and "_fold" not in kn:
                continue
            hit.append((float(r["TotalDurationNs"]), int(r["Calls"])))
        if hit:
            return sum(h[0] for h in hit) / sum(h[1] for h in hit) / 1e6, "profiles/" + name
    return None, None


def sustained_mfma_rate():
    """What the MFMA pipes sustain on realistic operands, measured (scripts/micro/mfma_power.hip, committed output): the
    chip is power-limited there - N(0,1) bf16 fragments re-read from LDS, no other traffic, run the 16x16x32 MFMA stream at
    ~1.78 PFLOP/s with the shader clock at ~1.9 GHz (all-zero operands: 2.33 PFLOP/s at 2.39 GHz). Instruction rate, so an
    f16x3 product counts three times."""
    path = os.path.join(ROOT, "profiles", "r02_mfma_power.txt")
    try:
        for line in open(path):
            if line.startswith("LDS-fed") and "N(0,1)" in line:
                tok = line.split()
                return {"TFLOPs": float(tok[tok.index("TFLOP/s") - 1]), "shader_clock_MHz": float(tok[tok.index("MHz") - 1]),
                        "source": "profiles/r02_mfma_power.txt (register double-buffered MFMA stream, 10 ds_read_b128 per 24 MFMAs, "
                                  "N(0,1) bf16 operands, 256 CUs x 8 waves; power-limited clock)"}
    except (OSError, ValueError):
        pass
    return None



class ClockProbe:
    """Average shader clock over the timed loop: pp_clock_probe (one sleeping wavefront that brackets wall time with s_memtime and
    the 100 MHz s_memrealtime) on a side stream of its own, started with the loop and stopped - through a word in pinned host
    memory - when the loop's last step has been submitted (the device is then at most `steps in flight` steps from the end)."""

    def __init__(self, dev):
        self.dev = dev
        self.words = torch.zeros(3, dtype=torch.int64).pin_memory()  # [cycles, ticks, stop]
        self.stream = torch.cuda.Stream(device=dev)
        self.running = False

    def start(self, max_us=4_000_000):
        from probpose_code_amd import _lib

        self.words.zero_()
        base = self.words.data_ptr()
        _lib.call("pp_clock_probe", base, base + 16, int(max_us), self.stream.cuda_stream)
        self.running = True

    def stop(self):
        self.words[2] = 1  # plain host store; the wavefront polls it with a system-scope load every few microseconds

    def read(self):
        if not self.running:
            return None
        self.stop()
        self.stream.synchronize()
        self.running = False
        cyc, ticks = int(self.words[0]), int(self.words[1])
        if ticks <= 0:
            return None
        return {"shader_clock_MHz": cyc / ticks * 100.0, "window_ms": ticks / 1e5,
                "how": "pp_clock_probe: s_memtime cycles / s_memrealtime (100 MHz) ticks of one sleeping wavefront on a side stream, from the "
                       "start of the probed loop to the submission of its last step"}


def device_limits(index=0):
    """Power cap / clock limits of the GPU the bench ran on, best effort (sysfs first, rocm-smi second; absent on a box without
    them): lets a driver-side number be normalised for the box-to-box clock spread."""
    import glob

    out = {}
    try:
        out["max_clock_MHz"] = torch.cuda.get_device_properties(index).clock_rate / 1e3
    except Exception:  # noqa: BLE001
        pass
    try:
        cards = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_cap"))
        if cards:
            out["power_cap_W"] = [int(open(c).read()) / 1e6 for c in cards][min(index, len(cards) - 1)]
            avg = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_average"))
            if avg:
                out["power_now_W"] = int(open(avg[min(index, len(avg) - 1)]).read()) / 1e6
    except Exception:  # noqa: BLE001
        pass
    if "power_cap_W" not in out:
        try:
            r = subprocess.run(["rocm-smi", "-d", str(index), "--showmaxpower", "--showpower", "--json"], capture_output=True, text=True, timeout=20)
            js = json.loads(r.stdout)
            card = next(iter(js.values()))
            for k, v in card.items():
                if "Max Graphics Package Power" in k:
                    out["power_cap_W"] = float(v)
                elif "Package Power" in k and "Max" not in k:
                    out["power_now_W"] = float(v)
        except Exception:  # noqa: BLE001
            pass
    return out


def roofline_record(eng, B, prof, reps):
    """`roofline` of the kernel tag with the largest time per step in the instrumented pass."""
    per_tag = {k: (float(np.sum(v)) / reps, len(v) // reps) for k, v in prof.items()}  # ms per step, launches per step
    dom = max(per_tag, key=lambda k: per_tag[k][0])
    dom_ms, dom_n = per_tag[dom]
    kernel_ms = {k: round(v[0], 4) for k, v in sorted(per_tag.items(), key=lambda kv: -kv[1][0])}
    prec = eng.precision
    peak = PEAK_TFLOPS[prec]
    ceil = peak / MFMA_PER_PRODUCT[prec]  # what the arithmetic format can reach on the MFMA pipe
    work = kernel_work_per_step(eng, B, 2, dom)
    if work is None:  # pragma: no cover - a kernel without a formula dominates: report its time, roofline undefined
        return kernel_ms, {"bound": "mfma", "achieved": None, "peak": peak, "unit": "TFLOP/s", "frac": None,
                           "traffic": None, "kernel": dom, "avg_launch_ms": dom_ms / dom_n}
    fl, alg_bytes, n, mangled = work
    assert n == dom_n, (dom, n, dom_n)
    secs = dom_ms / n * 1e-3
    ridge = ceil * 1e12 / (HBM_PEAK_GBS * 1e9)  # FLOP/B where the format's MFMA ceiling meets the HBM roof
    intensity = fl / alg_bytes
    mfma_view = {"achieved_TFLOPs": fl / n / secs / 1e12, "peak_TFLOPs": peak, "frac": fl / n / secs / 1e12 / peak}
    hbm_view = {"achieved_GBps": alg_bytes / n / secs / 1e9, "peak_GBps": HBM_PEAK_GBS,
                "frac": alg_bytes / n / secs / 1e9 / HBM_PEAK_GBS}
    bound = "mfma" if intensity >= ridge else "hbm"  # which side of the ridge the kernel's algorithm sits on
    traffic, src = pmc_traffic(mangled, prec, B)
    rec = {
        "bound": bound,
        "achieved": mfma_view["achieved_TFLOPs"] if bound == "mfma" else hbm_view["achieved_GBps"],
        "peak": peak if bound == "mfma" else HBM_PEAK_GBS,
        "unit": "TFLOP/s" if bound == "mfma" else "GB/s",
        "frac": mfma_view["frac"] if bound == "mfma" else hbm_view["frac"],
        "traffic": traffic,
        "traffic_source": (src + " (separate rocprofv3 --pmc passes)") if traffic else None,
        "kernel": dom, "kernel_mangled": mangled, "launches_per_step": n, "avg_launch_ms": dom_ms / n,
        "algorithmic_gflop_per_launch": fl / n / 1e9, "algorithmic_mbytes_per_launch": alg_bytes / n / 1e6,
        "arithmetic_intensity_flop_per_byte": intensity, "ridge_flop_per_byte": ridge,
        "mfma_view": mfma_view, "hbm_view": hbm_view,
        "durations_from": "HIP events around every launch of an instrumented pass of the same step, one step at a time on the launch "
                          "stream (the kernel with the chip to itself). The rocprofv3 trace of that mode is profiles/*_kernel_stats_one_in_flight.csv "
                          "(`bench.py --in-flight 1`); in the trace of the default run (profiles/*_kernel_stats.csv, two steps in flight) a "
                          "kernel's duration includes the time it shares CUs with the other step's kernels",
    }
    # the same fraction from the COMMITTED rocprofv3 trace (what a reader recomputing it from profiles/ gets): tracing slows kernels by a few per cent
    rec["frac_unprofiled"] = rec["frac"]
    rec["avg_launch_us_unprofiled"] = dom_ms / n * 1e3
    pms, psrc = profiled_duration(mangled, prec, B)
    if pms:
        per = (fl / n / (pms * 1e-3) / 1e12 / peak) if bound == "mfma" else (alg_bytes / n / (pms * 1e-3) / 1e9 / HBM_PEAK_GBS)
        rec["frac_profiled"] = per
        rec["avg_launch_us_profiled"] = pms * 1e3
        rec["profiled_from"] = psrc + " (TotalDurationNs / Calls of the kernel's rows; bench.py --in-flight 1 under rocprofv3 --kernel-trace --stats)"
    if bound == "mfma":
        power = sustained_mfma_rate()
        if power:
            rec["sustained_mfma_rate"] = dict(power, frac_of_sustained=mfma_view["achieved_TFLOPs"] * MFMA_PER_PRODUCT[prec] / power["TFLOPs"])
        # the ceiling this kernel's arithmetic format can reach: the MFMA pipe issues MFMA_PER_PRODUCT instructions per
        # algorithmic product, and sustains ~1.93 PF of the 2.5 PF datasheet peak under load (the clock drops to ~1.85 GHz)
        rec["derived_ceiling"] = {"TFLOPs": ceil, "why": f"{MFMA_PER_PRODUCT[prec]} MFMA per algorithmic product on the {peak:.0f} TF pipe",
                                  "frac_of_ceiling": mfma_view["achieved_TFLOPs"] / ceil}
    return kernel_ms, rec


def secondary_rooflines(eng, B, prof, reps, bs512=True):
    """The two kernels BASELINE.json's `north_star` sets targets for - ViT attention (>= 70 % MFMA) and the ProbMap decode
    (>= 60 % HBM) - with the datasheet fraction AND the ceiling their arithmetic allows at this shape, with the
    instruction-count derivation. Measured live from the same instrumented pass."""
    per_tag = {k: (float(np.sum(v)) / reps, len(v) // reps) for k, v in prof.items()}
    out = {}
    S, hd, heads, nseq = eng.Np, eng.hd, eng.heads, 2 * B
    if "head_decode" in per_tag:
        ms, n = per_tag["head_decode"]
        by = 2 * B * eng.K * eng.Hh * eng.Wh * 4  # both logit maps of every (crop, keypoint), read once (417 792 B / crop)
        wgs, slots = B * eng.K, 3 * 256
        rounds = -(-wgs // slots)
        chain_us = 17.0  # one workgroup's dependent chain: the 17-workgroup launch (one crop) in a rocprofv3 kernel trace, scripts/micro/decode_pmc.sh
        ceil_gbs = by / (rounds * chain_us * 1e-6) / 1e9
        out["head_decode"] = {
            "bound": "hbm", "achieved": by / (ms / n * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": by / (ms / n * 1e-3) / 1e9 / HBM_PEAK_GBS, "avg_launch_ms": ms / n, "algorithmic_mbytes_per_launch": by / 1e6,
            "derived_ceiling": {
                "GBps": ceil_gbs, "frac_of_ceiling": by / (ms / n * 1e-3) / 1e9 / ceil_gbs,
                "why": f"instruction- and latency-bound, not bandwidth-bound: the {by / 1e6:.1f} MB are L2 / MALL resident; each (crop, keypoint) "
                       f"workgroup issues ~2 550 VALU instructions per wave (Sparsemax of two 3072-pixel rows in registers, flip merge, "
                       f"support box, fp64 row and column passes over the box dilated by a radius of up to 9, argmax) behind a dozen "
                       f"barriers: ~{chain_us:.0f} us for one workgroup alone, 40 % of it issuing; {wgs} workgroups = {wgs / 256:.2f} per CU x 4 waves "
                       f"x 2 550 VALU x 4 cycles = 23 us of VALU issue per SIMD at best balance; the band buffer is sized for 3 workgroups per "
                       f"CU ({slots} slots, {rounds} rounds -> >= {rounds * chain_us:.0f} us) - 4 and 5 per CU measured SLOWER (40 / 46 us vs 36: "
                       f"4.25 workgroups per CU are balanced by the dispatcher handing out work, not by residency)",
            },
        }
        # the same kernel at bs 512 (BASELINE config 3's global batch on one GPU; SURVEY H6: where the HBM fraction becomes meaningful):
        # the step's own logits tiled eight times (same sparse maps), timed alone
        try:
            from probpose_code_amd import _lib
            from probpose_code_amd import synthetic as S_

            ws = eng._workspace(B, 2)
            if bs512 and getattr(eng, "_logits_phased", False) and B * 8 <= 512:
                rep8 = 512 // B
                lg = ws["logits"][:B].repeat(rep8, 1, 1).contiguous()
                lgf = ws["logits"][B:].repeat(rep8, 1, 1).contiguous()
                nb = rep8 * B
                kp = torch.empty((nb, eng.K, 2), dtype=torch.float64, device=eng.device)
                lo, sc = torch.empty((nb, eng.K, 2), device=eng.device), torch.empty((nb, eng.K), device=eng.device)
                fi = eng._flip_indices(S_.COCO_FLIP_INDICES)

                def run512():
                    _lib.call("pp_probmap_head_decode_phased", lg.data_ptr(), lgf.data_ptr(), fi.data_ptr(), eng.taps.data_ptr(), eng.radius.data_ptr(),
                              nb, eng.K, eng.Hh, eng.Wh, float(eng.input_size[0]), float(eng.input_size[1]), eng.temperature,
                              -1.0 if eng.normalize is None else float(eng.normalize), None, None, lo.data_ptr(), kp.data_ptr(), sc.data_ptr(),
                              _lib.stream_ptr(eng.device))

                for _ in range(3):
                    run512()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    run512()
                e1.record()
                torch.cuda.synchronize()
                ms512 = e0.elapsed_time(e1) / 10
                by512 = 2 * nb * eng.K * eng.Hh * eng.Wh * 4
                out["head_decode_bs512"] = {"bound": "hbm", "crops": nb, "avg_launch_ms": ms512, "algorithmic_mbytes_per_launch": by512 / 1e6,
                                            "achieved": by512 / (ms512 * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                            "frac": by512 / (ms512 * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                            "why": f"{nb * eng.K} workgroups = {nb * eng.K / 256:.0f} per CU x 4 waves x ~2 550 VALU instructions: the launch is "
                                                   "instruction-bound at every batch size; the logits are read exactly once"}
        except Exception as exc:  # noqa: BLE001 -- a secondary record must not take the bench line down
            out["head_decode_bs512"] = {"error": str(exc)[:200]}
    att_fl = 4.0 * nseq * heads * S * S * hd  # QK^T + PV per layer
    # Two ceilings for softmax(q k^T) v at 192 tokens x head dim 32, both far under the MFMA peak (128 FLOP per v_exp_f32):
    #  * VALU: per (16-query tile, head) task 48 v_exp_f32 + ~50 v_fma + 24 v_max3 + 28 v_cvt_pk + ~30 others; SIMD retire rates
    #    measured on gfx950 (scripts/micro/valu_rate.hip, 4 waves per SIMD): v_exp_f32 8.35, v_fma 2.75, v_max3 / v_cvt_pk 4.6
    #    cycles per wave64 instruction -> ~870 cycles per task; 72 tasks per 96-row workgroup over 4 SIMDs (round-robin);
    #  * HBM: the bytes the phase has to pull (q, k, v; in the fused layer kernel also the fp32 residual rows that load under
    #    it) at the ~5 TB/s the chip sustains on reads when all 256 CUs ask at once (ATT_DBG ablations).
    if "qkv_attention" in per_tag:
        # f16x3: the attention of a (sequence, head) runs in the workgroup that has just computed that head's q, k, v rows
        # (pp_qkv_attn_split.hip); there is no attention-only kernel to time. Priced as one kernel against the matrix pipe:
        # three fp16 MFMAs per algorithmic product.
        ms, n = per_tag["qkv_attention"]
        M_ = nseq * S
        qkv_fl = 2.0 * M_ * 3 * heads * hd * heads * hd
        ach = (qkv_fl + att_fl) / (ms / n * 1e-3) / 1e12
        ceil_tf = PEAK_TFLOPS["f16x3"] / MFMA_PER_PRODUCT["f16x3"]
        out["attention"] = {
            "bound": "mfma", "kernel": "qkv_attention (qkv Linear + attention, one workgroup per (sequence, head))", "avg_launch_ms": ms / n,
            "algorithmic_gflop_per_layer": (qkv_fl + att_fl) / 1e9, "attention_share_of_flops": att_fl / (qkv_fl + att_fl),
            "achieved": ach, "peak": PEAK_TFLOPS["f16x3"], "unit": "TFLOP/s", "frac": ach / PEAK_TFLOPS["f16x3"],
            "algorithmic_mbytes_per_launch": 2 * M_ * heads * hd * 4 / 1e6,
            "derived_ceiling": {"TFLOPs": ceil_tf, "frac_of_ceiling": ach / ceil_tf,
                                "why": "3 fp16 MFMAs per algorithmic product; q, k, v never leave the CU, so the phase's HBM floor of the "
                                       "stand-alone kernels (113 MB of qkv written and read back per layer) is gone"}}
    elif S == 192 and hd == 32:
        valu_cycles, clk, hbm_read_bps = 870.0, 2.1e9, 5.0e12
        rounds = max(1, -(-(nseq * S // 96) // 256))  # 96-row workgroups over 256 CUs (bs 64 + flip test: exactly one round)
        t_valu = heads * 6 * valu_cycles / 4 / clk * rounds
        fused = "attention" not in per_tag
        qkv_bytes = nseq * S * 3 * heads * hd * 2
        phase_bytes = qkv_bytes + (nseq * S * heads * hd * 4 if fused else nseq * S * heads * hd * 2)  # + residual rows in / + output rows out
        t_hbm = phase_bytes / hbm_read_bps
        t_min = max(t_valu, t_hbm)
        ceil_tf = att_fl / t_min / 1e12
        rec = {"bound": "hbm" if t_hbm >= t_valu else "valu", "algorithmic_gflop_per_layer": att_fl / 1e9, "peak": PEAK_TFLOPS["bf16"], "unit": "TFLOP/s",
               "derived_ceiling": {"TFLOPs": ceil_tf, "frac_of_datasheet_peak": ceil_tf / PEAK_TFLOPS["bf16"],
                                   "valu_floor_us": t_valu * 1e6, "hbm_floor_us": t_hbm * 1e6, "phase_mbytes": phase_bytes / 1e6,
                                   "why": f"at 192 tokens x 32 dims the phase moves {phase_bytes / 1e6:.0f} MB for {att_fl / 1e9:.2f} GFLOP: at the ~5 TB/s "
                                          f"the chip sustains on reads that is {t_hbm * 1e6:.1f} us; the softmax VALU work (48 v_exp_f32 at 8.35 cycles, "
                                          f"fma / max3 / cvt_pk at 2.75 - 4.6, measured retire rates) is {t_valu * 1e6:.1f} us with the 72 tasks of a "
                                          "workgroup dealt evenly to its four SIMDs; all arithmetic and LDS reads compiled out, a round of the "
                                          "fused phase still takes 2 800 of its 4 000 cycles (ATT_DBG). The 70 % MFMA target of BASELINE.json "
                                          "is not reachable at this shape."}}
        if not fused:  # stand-alone kernel (PP_FUSE_ATTN=0 and the f16x3 / f32 modes)
            ms, n = per_tag["attention"]
            rec.update(achieved=att_fl / (ms / n * 1e-3) / 1e12, avg_launch_ms=ms / n, kernel="attention")
            rec["frac"] = rec["achieved"] / PEAK_TFLOPS["bf16"]
            if eng.precision == "bf16":
                rec["derived_ceiling"]["frac_of_ceiling"] = rec["achieved"] / ceil_tf
        else:  # inside the fused layer kernel: phase time from the kernel's own stamps (scripts/micro/layer_trace.py):
            # start -> round 0 ~3 us (cold q, k, v of two heads), then nine rounds of ~1.9 us
            rec.update(kernel="vit_layer (attention phase)", phase_us_from_stamps=20.5, achieved=att_fl / 20.5e-6 / 1e12)
            rec["frac"] = rec["achieved"] / PEAK_TFLOPS["bf16"]
            rec["derived_ceiling"]["frac_of_ceiling"] = rec["achieved"] / ceil_tf
        out["attention"] = rec
    return out


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


def parity_record(precision, B, snap, ref, source):
    """Keypoints / probabilities of a timed step's output (host copies in `snap`) against the oracle's `ref`."""
    d = np.abs(snap["keypoints"][:, None] - ref["keypoints_input_space"]).max(-1)
    same = d < 2.0
    return {
        "precision": precision, "crops": B, "output_of": source,
        "keypoint_linf_px_input_space_same_argmax": float(d[same].max()),
        "argmax_flips": int((~same).sum()), "keypoints": int(same.size),
        "probs_linf": float(np.abs(snap["scalars"][0][:, None] - ref["keypoints_probs"]).max()),
        "visible_linf": float(np.abs(snap["scalars"][1][:, None] - ref["keypoints_visible"]).max()),
        "oks_linf": float(np.abs(snap["scalars"][2][:, None] - ref["keypoints_oks"]).max()),
        "within_1e-3": bool(d[same].max() <= 1e-3 and int((~same).sum()) == 0),
    }


# ------------------------------------------------------------------------------------------ stub (tests only)
class StubEngine:
    """Stands in for ProbPoseEngine where there is no GPU: outputs encode (rank, crop index) so that the test can check
    the gather; a fixed sleep stands in for the kernels."""
    precision, K, device = "stub", 17, torch.device("cpu")

    def __init__(self, rank):
        self.rank = rank

    def forward(self, crops, flip_test=True, flip_indices=None, slot=0):
        B = crops.shape[0]
        time.sleep(0.002)
        ids = (self.rank * 1000 + torch.arange(B, dtype=torch.float64))
        return dict(keypoints=ids[:, None, None].expand(B, self.K, 2).clone(), scores=torch.zeros(B, self.K),
                    scalars=torch.zeros(4, B, self.K))

    forward_graph = forward


# ------------------------------------------------------------------------------------------ one timed run
def timed_run(eng, crops, gather, flip, steps, warmup, use_graph, dist_mod, dev, depth=1, world=1, probe=None):
    """W untimed + exactly K timed steps bracketed by barrier + synchronize; returns this rank's seconds and host
    copies of the LAST TIMED step's outputs (what the graph replay left in the engine's output buffers).
    ``depth`` > 1: consecutive steps go to consecutive slots of a StepPipeline (own stream, workspace, graph and pinned
    record buffer per slot), so up to ``depth`` steps are in flight; every step still runs the whole captured launch
    sequence and delivers its own record, and all K of them have completed at the closing synchronize."""
    is_cuda = dev.type == "cuda"
    pipe = None
    if depth > 1:  # (also with the CPU stub engine of the gloo tests: slots without streams)
        from probpose_code_amd.pipeline import StepPipeline

        pipe = StepPipeline(eng, crops.shape[0], flip, True, depth, world, use_graph=use_graph)
    elif use_graph and is_cuda:
        eng.capture(crops.shape[0], True, flip).copy_(crops)
    out, ticket = None, -1

    def step():
        nonlocal out, ticket
        if pipe is not None:
            ticket = pipe.submit(crops)
            return
        out = eng.forward_graph(crops, True, flip) if use_graph else eng.forward(crops, True, flip)
        gather(out)

    def barrier():
        if is_cuda:
            torch.cuda.synchronize()  # every stream of the device
        if dist_mod is not None:
            dist_mod.barrier()
        if is_cuda:
            torch.cuda.synchronize()

    for _ in range(warmup):
        step()
    barrier()
    if probe is not None:
        probe.start()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    if probe is not None:
        probe.stop()  # (a host store: the probe's wavefront has left the chip long before the closing synchronize returns)
    barrier()
    dt = time.perf_counter() - t0
    if pipe is not None:
        out = pipe.device_outputs(ticket)
        records = pipe.result(ticket)
    else:
        records = gather.wait()
    snap = {k: out[k].detach().cpu().numpy().copy() for k in ("keypoints", "scores", "scalars")}
    snap["records"] = records.numpy().copy()  # what the step delivered to the host (world, B, K, 7)
    return dt, snap


def reduce_times(dt, dist_mod, dev, world, local_dev_index):
    """max over ranks (the job's step time) + every rank's own seconds and device index."""
    if dist_mod is None:
        return dt, [dt], [local_dev_index]
    t = torch.tensor([dt, float(local_dev_index)], dtype=torch.float64, device=dev)
    allt = torch.empty(world * 2, dtype=torch.float64, device=dev)  # (flat: gloo wants the concatenated form)
    dist_mod.all_gather_into_tensor(allt, t)
    allt = allt.cpu().view(world, 2)
    return float(allt[:, 0].max()), [float(x) for x in allt[:, 0]], [int(x) for x in allt[:, 1]]


def instrumented_pass(eng, crops, flip, reps=5):
    """HIP events around every launch, on the launch stream (outside the timed region)."""
    eng.profile = {}
    for _ in range(reps):
        eng.forward(crops, True, flip)
    torch.cuda.synchronize()
    prof = {k: [a.elapsed_time(b) for a, b in v] for k, v in eng.profile.items()}
    eng.profile = None
    return prof, reps


def config4_record(dev, args):
    """BASELINE config 4 - ProbPose-base (ViT-B 12 x 768, 12 heads x 64), 384x288 crops, 17-keypoint head, bs 64, flip
    test, one GPU - as a bounded, driver-timed record: both precisions through the same two-deep pipeline as the headline (10 timed
    steps), the dominant kernel's roofline from an instrumented pass (+ its counter traffic from the committed profile), and parity
    of the LAST TIMED REPLAY's output against the oracle on its first 16 crops (the CPU model takes ~1 s per 384x288 crop). A
    second, shorter entry repeats the f16x3 run at bs 32 (the batch size of the round-3 record)."""
    from oracle import model_ref as M
    from probpose_code_amd import synthetic as S
    from probpose_code_amd.dist import ResultGather
    from probpose_code_amd.engine import ProbPoseEngine

    B4, img, steps, warm, NPAR = args.config4_batch, (384, 288), 10, 3, 16
    quick = getattr(args, "config4_quick", False)
    if quick:
        steps, warm = 3, 1
    sd4 = S.synthetic_state_dict("base", img_size=img, seed=0, logit_scale=2.0)
    crops_cpu = S.synthetic_crops(B4, img_size=img, seed=7)
    crops = crops_cpu.to(dev)
    flip = S.COCO_FLIP_INDICES
    rec = {"workload": f"ProbPose-base (ViT-B 12x768, 12 heads x 64) bs{B4} random 384x288 uint8 crops, flip_test=True, 96x72 heatmaps, "
                       "seeded random-init weights; same step definition and two-deep pipeline as the headline",
           "batch": B4, "gflop_per_crop": None, "steps": steps, "warmup": warm}
    ref = None
    npar = min(NPAR, B4)
    if not args.no_parity:
        torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
        ref = M.predict(sd4, crops_cpu[:npar], 12, S.IMG_MEAN, S.IMG_STD, input_size=(288, 384))
    depth = max(1, args.in_flight)

    def run(prec, B, k, w):
        eng = ProbPoseEngine(sd4, 12, img_size=img, precision=prec, input_size=(288, 384), device=dev)
        gather = ResultGather(B, eng.K, dev, 1)
        dt, snap = timed_run(eng, crops[:B].contiguous(), gather, flip, k, w, not args.no_graph, None, dev, depth, 1)
        prof, reps = instrumented_pass(eng, crops[:B].contiguous(), flip, reps=2)
        per_tag = {kk: (float(np.sum(v)) / reps, len(v) // reps) for kk, v in prof.items()}
        dom = max(per_tag, key=lambda kk: per_tag[kk][0])
        r = {"value": B * k / dt, "unit": "crops/s", "ms_per_step": dt / k * 1e3, "batch": B, "dtype_detail": DTYPE_DETAIL[prec],
             "kernel_ms_per_step": {kk: round(v[0], 4) for kk, v in sorted(per_tag.items(), key=lambda kv: -kv[1][0])}}
        work = kernel_work_per_step(eng, B, 2, dom)
        if work is not None and work[2] == per_tag[dom][1]:
            fl, by, n, mangled = work
            secs = per_tag[dom][0] / n * 1e-3
            traffic, src = pmc_traffic(mangled, prec, B, prefix="config4_")
            r["roofline"] = {"bound": "mfma", "kernel": dom, "kernel_mangled": mangled, "launches_per_step": n, "avg_launch_ms": secs * 1e3,
                             "achieved": fl / n / secs / 1e12, "peak": PEAK_TFLOPS[prec], "unit": "TFLOP/s",
                             "frac": fl / n / secs / 1e12 / PEAK_TFLOPS[prec], "traffic": traffic, "traffic_source": src,
                             "algorithmic_mbytes_per_launch": by / n / 1e6,
                             "derived_ceiling_TFLOPs": PEAK_TFLOPS[prec] / MFMA_PER_PRODUCT[prec]}
            r["roofline"]["frac_unprofiled"] = r["roofline"]["frac"]
            r["roofline"]["avg_launch_us_unprofiled"] = secs * 1e6
            pms, psrc = profiled_duration(mangled, prec, B, prefix="config4_")
            if pms:
                r["roofline"].update(frac_profiled=fl / n / (pms * 1e-3) / 1e12 / PEAK_TFLOPS[prec], avg_launch_us_profiled=pms * 1e3,
                                     profiled_from=psrc + " (TotalDurationNs / Calls; bench.py --config4-only --in-flight 1 under rocprofv3)")
        del eng
        torch.cuda.empty_cache()
        return r, snap

    precs = (PARITY_PRECISION,) if quick else (PARITY_PRECISION, THROUGHPUT_PRECISION)
    for prec in precs:
        r, snap = run(prec, B4, steps, warm)
        if ref is not None:
            sub = {"keypoints": snap["keypoints"][:npar], "scalars": snap["scalars"][:, :npar]}
            r["parity_vs_oracle"] = parity_record(prec, npar, sub, ref, f"last timed step (hipGraph replay of the bs {B4} pipeline), its first {npar} crops")
        rec[prec] = r
    if B4 != 32 and not quick:
        rec["f16x3_bs32"] = run(PARITY_PRECISION, 32, 6, 2)[0]
    # algorithmic FLOPs per crop with flip test (MAC = 2): backbone Linear layers + attention + patch embed + the head
    Np, E, Fd, L = 24 * 18, 768, 3072, 12
    fl = 2 * (L * (2.0 * Np * E * 3 * E + 2.0 * Np * E * E + 4.0 * Np * E * Fd + 4.0 * 12 * Np * Np * 64) + 2.0 * Np * E * 768)
    fl += 2 * (2.0 * (4 * Np) * 256 * 4 * E + 2.0 * (16 * Np) * 256 * 4 * 256 + 2.0 * (16 * Np) * 17 * 256)  # deconv x2 + final 1x1
    fl += 2 * 4 * (2.0 * Np * E * 9 * E + 2.0 * 36 * E * 9 * E + 2.0 * 9 * E * 9 * E)                      # four towers x three 3x3 stages
    rec["gflop_per_crop"] = fl / 1e9
    for prec in precs:
        rec[prec]["path_tflops"] = rec[prec]["value"] * fl / 1e12
    return rec


def drop_in_record(dev, args, sd, B):
    """The drop-in call timed: `model.test_step(batch)` - what the reference's callers drive (mmpose/apis/inference.py:195-196;
    tools/test.py:115-136 through mmengine's `Runner.test()`) - on the estimator built from the config through the registry,
    batches as `apis.pack_crops` makes them (`pseudo_collate` layout: a list of B uint8 CHW crops + B PoseDataSample), the host
    side included: preprocessor stacking, the engine's graph replay, ONE record copy to pinned host memory, B `InstanceData` /
    `PoseDataSample` built and mapped to image space. `test_step`: strictly one batch at a time, the host packaging after the
    device has finished; `test_step_stream(depth=2)`: the packaging of batch n under the device's work on batch n + 1."""
    from probpose_code_amd import apis
    from probpose_code_amd import synthetic as S

    cfg = os.path.join(ROOT, "configs", "td-pm_ProbPose-small_mi355x_coco-256x192.py")
    model = apis.init_model(cfg, {"state_dict": sd}, device=str(dev), cfg_options={"model.precision": args.precision})
    rng = np.random.default_rng(11)
    center = np.stack([rng.uniform(80, 400, B), rng.uniform(100, 500, B)], -1).astype(np.float32)
    scale = (np.array([192, 256], np.float32) * rng.uniform(0.8, 2.5, (B, 1)).astype(np.float32) * 1.25).astype(np.float32)
    crops = [S.synthetic_crops(B, seed=200 + i).to(dev) for i in range(3)]
    batches = [apis.pack_crops(c, center, scale, model.dataset_meta) for c in crops]  # three batch dicts: none is in flight twice
    steps, warm = min(args.steps, 30), 4
    rec = {"what": "crops/s of the drop-in entry points on the registry-built estimator, host packaging included; inputs = pack_crops batches "
                   f"(list of {B} device-resident uint8 crops + {B} PoseDataSample); ProbPose-small, flip_test=True, {args.precision}",
           "steps": steps, "warmup": warm, "batch": B}
    with torch.no_grad():
        for i in range(warm):  # (first call of a size: launches one by one; second: graph capture; then replays)
            model.test_step(batches[i % 3])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            res = model.test_step(batches[i % 3])
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        rec["test_step"] = {"value": B * steps / dt, "unit": "crops/s", "ms_per_batch": dt / steps * 1e3}
        # host-side share of a test_step: the same call with the device work already done is not separable; time the packaging alone
        t1 = time.perf_counter()
        for i in range(steps):
            data = model.data_preprocessor(batches[i % 3], False)
        torch.cuda.synchronize()
        rec["test_step"]["preprocessor_ms"] = (time.perf_counter() - t1) / steps * 1e3
        for depth in (2,):
            gen = model.test_step_stream((batches[i % 3] for i in range(warm + steps)), depth=depth, max_batch=B)
            for _ in range(warm):
                next(gen)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            last = None
            for last in gen:
                pass
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            rec[f"test_step_stream_depth{depth}"] = {"value": B * steps / dt, "unit": "crops/s", "ms_per_batch": dt / steps * 1e3}
            kp_stream = np.stack([r.pred_instances.keypoints for r in last])
            same_batch = model.test_step(batches[(warm + steps - 1) % 3])  # (the batch the stream delivered last)
            rec[f"test_step_stream_depth{depth}"]["identical_to_test_step"] = bool(np.array_equal(
                kp_stream, np.stack([r.pred_instances.keypoints for r in same_batch])))
    del model
    torch.cuda.empty_cache()
    return rec


def small_batch_record(dev, args, sd, ref_fn=None):
    """Latency of the batch sizes the reference's own callers produce (demo/image_demo.py:36-61: one crop; mmpose/apis/inference.py:161-196: the
    boxes of one image; demo/topdown_demo_with_mmdet.py:35-41: a handful of persons): B = 1, 2, 4, 8 crops + flip test, the replayed hipGraph,
    strictly ONE step in flight (what a caller who waits for the result sees), with the launch plan the engine picks there (pp_skinny_linear:
    column-parallel Linear layers, `small_plan`) and with the headline's row-owner plan for comparison. Timed after inputs are resident."""
    from probpose_code_amd import ProbPoseEngine
    from probpose_code_amd import synthetic as S

    rec = {"what": "ms per step, one step in flight (replayed hipGraph), ProbPose-small 256x192 flip_test=True, " + args.precision,
           "reps": 100, "latency_ms": {}, "crops_per_s": {}, "row_owner_plan_latency_ms": {}}
    engines = {"small": ProbPoseEngine(sd, 12, precision=args.precision, device=dev),
               "row_owner": ProbPoseEngine(sd, 12, precision=args.precision, device=dev, plan=dict(small_plan=False))}
    rec["plan_below_rows"] = engines["small"].small_rows_below if engines["small"].small_plan else 0
    for B in (1, 2, 4, 8):
        crops = S.synthetic_crops(B, seed=300 + B).to(dev)
        for name, eng in engines.items():
            for _ in range(3):
                out = eng.forward_graph(crops, True, S.COCO_FLIP_INDICES)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(rec["reps"]):
                out = eng.forward_graph(crops, True, S.COCO_FLIP_INDICES)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / rec["reps"]
            if name == "small":
                rec["latency_ms"][str(B)] = round(dt * 1e3, 4)
                rec["crops_per_s"][str(B)] = round(B / dt, 1)
                if ref_fn is not None:
                    ref = ref_fn(crops.cpu())
                    d = np.abs(out["keypoints"].cpu().numpy()[:, None] - ref["keypoints_input_space"]).max(-1)
                    rec.setdefault("parity_vs_oracle", {})[str(B)] = {"keypoint_linf_px": float(d[d < 2.0].max()), "argmax_flips": int((d >= 2.0).sum()),
                                                                      "within_1e-3": bool((d < 2.0).all() and d.max() <= 1e-3)}
            else:
                rec["row_owner_plan_latency_ms"][str(B)] = round(dt * 1e3, 4)
    del engines
    torch.cuda.empty_cache()
    return rec


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse(argv)
    if args.gpus < 1:
        print("bench.py: --gpus must be >= 1", file=sys.stderr)
        return 2
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return relaunch_under_torchrun(args, argv)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print(f"bench.py: launched with WORLD_SIZE={world} but --gpus {args.gpus}; refusing (the JSON line must describe "
              "the job that ran)", file=sys.stderr)
        return 2
    distributed = world > 1
    dist = None
    if args.stub:
        dev = torch.device("cpu")
    else:
        if torch.cuda.device_count() <= local_rank:
            print(f"bench.py: rank {rank} wants cuda:{local_rank}, node has {torch.cuda.device_count()} GPU(s)", file=sys.stderr)
            return 2
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    if distributed:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if args.stub:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)  # "nccl" IS RCCL on ROCm
        assert dist.get_world_size() == args.gpus, (dist.get_world_size(), args.gpus)

    from probpose_code_amd import synthetic as S
    from probpose_code_amd.dist import ResultGather

    if args.config4_only:
        if world != 1 or args.stub:
            print("bench.py: --config4-only is a single-GPU run", file=sys.stderr)
            return 2
        print(json.dumps({"metric": "person-crops/sec @ 384x288 (BASELINE config 4)", "n_gpus": 1, "config4": config4_record(dev, args)}))
        return 0
    B = args.batch
    flip = S.COCO_FLIP_INDICES
    use_graph = not args.no_graph
    crops_cpu = S.synthetic_crops(B, seed=100 + rank)  # a different shard per rank
    crops = crops_cpu.to(dev)
    sd = None

    def make_engine(precision):
        nonlocal sd
        if args.stub:
            return StubEngine(rank)
        from probpose_code_amd.engine import ProbPoseEngine

        if sd is None:
            sd = S.synthetic_state_dict("small", seed=0, logit_scale=2.0)  # same weights on every rank
        return ProbPoseEngine(sd, 12, precision=precision, device=dev)

    eng = make_engine(args.precision)
    gather = ResultGather(B, eng.K, dev, world)  # fixed-layout result record, pinned host copy, RCCL all_gather
    depth = max(1, args.in_flight)
    dt_rank, snap = timed_run(eng, crops, gather, flip, args.steps, args.warmup, use_graph, dist, dev, depth, world)
    dt, rank_secs, rank_devs = reduce_times(dt_rank, dist, dev, world, local_rank)
    # The shader clock of that loop, from a SECOND pass of the same loop with the probe beside it - never from the timed pass itself:
    # the probe's sleeping wavefront holds a few registers of one SIMD, and a workgroup of the 256-register layer kernel (one per CU,
    # all 256 CUs) then cannot be placed on that CU until the probe has left (seen in the rocprofv3 trace: some launches of
    # proj_ffn_split at twice their duration while the probe ran)
    clock, dt_probe = None, None
    if not args.stub and not args.no_clock_probe:
        probe = ClockProbe(dev)
        kp = min(args.steps, 30)
        dtp_rank, _ = timed_run(eng, crops, gather, flip, kp, min(args.warmup, 3), use_graph, dist, dev, depth, world, probe=probe)
        clock = probe.read()
        dt_probe = reduce_times(dtp_rank, dist, dev, world, local_rank)[0] / kp
    # the PCIe-inclusive rate: every step's crops start in pinned HOST memory (never `value`: the boundary hands over device
    # buffers); the host-to-device copy rides on a copy stream under the kernels in flight (pipeline.StepPipeline.submit)
    dth = None
    if depth > 1 and not args.stub:
        kh, wh = min(args.steps, 20), min(args.warmup, 5)
        dth_rank, _ = timed_run(eng, crops_cpu.pin_memory(), gather, flip, kh, wh, use_graph, dist, dev, depth, world)
        dth = reduce_times(dth_rank, dist, dev, world, local_rank)[0] / kh
    # the same engine strictly one batch at a time (one stream, one graph replay after the other): the latency figure
    dt1 = None
    if depth > 1:
        k1, w1 = min(args.steps, 20), min(args.warmup, 5)
        dt1_rank, _ = timed_run(eng, crops, gather, flip, k1, w1, use_graph, dist, dev, 1, world)
        dt1 = reduce_times(dt1_rank, dist, dev, world, local_rank)[0] / k1
    prof = instrumented_pass(eng, crops, flip) if not args.stub else None

    # ---- the same bench in the other precision: bf16 (`throughput_mode`) beside the f16x3 headline; `parity_mode` (f16x3) when
    # the bench was asked for another precision (every rank takes part: barriers inside)
    pm = None
    second = THROUGHPUT_PRECISION if args.precision == PARITY_PRECISION else PARITY_PRECISION
    second_key = "throughput_mode" if second == THROUGHPUT_PRECISION else "parity_mode"
    if not args.stub and not args.no_second_mode:
        eng_p = make_engine(second)
        k_p, w_p = min(args.steps, 20), min(args.warmup, 5)
        dt_p_rank, snap_p = timed_run(eng_p, crops, gather, flip, k_p, w_p, use_graph, dist, dev, depth, world)
        dt_p, _, _ = reduce_times(dt_p_rank, dist, dev, world, local_rank)
        dt1_p = None
        if depth > 1:
            dt1_p_rank, _ = timed_run(eng_p, crops, gather, flip, k_p, w_p, use_graph, dist, dev, 1, world)
            dt1_p = reduce_times(dt1_p_rank, dist, dev, world, local_rank)[0] / k_p
        pm = (eng_p, k_p, w_p, dt_p, snap_p, instrumented_pass(eng_p, crops, flip), dt1_p)

    if rank == 0:
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
            "dtype": args.precision if not args.stub else "stub",
            "dtype_detail": DTYPE_DETAIL.get(args.precision) if not args.stub else "stub engine (tests)",
            "data": "synthetic",
            "config": {
                "workload": f"ProbPose-small (ViT-S 12x384, 12 heads x 32) bs{B} random 256x192 uint8 crops per GPU, "
                            "flip_test=True, seeded random-init weights: preprocess -> backbone x2 passes -> ProbMapHead "
                            "(heatmap branch + Sparsemax + 4 towers) -> ProbMap decode -> results in pinned host memory",
                "crops_per_gpu": B, "global_batch": B * world, "parallelism": f"dp{world}",
                "launch": ("hipGraph replay" if use_graph else "eager launches")
                          + (f", {depth} steps in flight (one HIP stream, workspace and captured graph per slot)" if depth > 1 else ""),
                "steps_in_flight": depth,
                "gflop_per_crop": GFLOP_PER_CROP_FLIP,
            },
            "path_tflops": B * world * args.steps * GFLOP_PER_CROP_FLIP / dt / 1e3,
            # one process per GPU: who ran where, and how fast each rank was on its own clock
            "rccl_ranks": world if (distributed and not args.stub) else (0 if not distributed else world),
            "collective_backend": ("gloo" if args.stub else "nccl(RCCL)") if distributed else None,
            "rank_devices": [("cpu" if args.stub else f"cuda:{d}") for d in rank_devs],
            "rank_crops_per_s": [B * args.steps / s for s in rank_secs],
            "max_rank_ms_per_step": dt / args.steps * 1e3,
        }
        if dt1 is not None:
            line["one_step_in_flight"] = {"ms_per_step": dt1 * 1e3, "value": B * world / dt1, "unit": "crops/s",
                                          "what": "the same engine, one hipGraph replay after the other on one stream (--in-flight 1)"}
        if dth is not None:
            line["host_input"] = {"ms_per_step": dth * 1e3, "value": B * world / dth, "unit": "crops/s",
                                  "what": "PCIe-inclusive: every step's crops start in pinned host memory (9.4 MB per bs64 batch); the H2D copy goes "
                                          "on a copy stream ahead of the wait for the slot, i.e. under the kernels in flight, then 9.4 MB inside HBM "
                                          "into the graph's input; not the headline (inputs resident in HBM)"}
        if args.stub:
            rec = snap["records"]  # (world, B, K, 7): x of crop i on rank r is r * 1000 + i
            line["stub_gather_ok"] = bool(all(rec[r, i, 0, 0] == r * 1000 + i for r in range(world) for i in (0, B - 1)))
            print(json.dumps(line))
        else:
            # the clock the timed loop ran at on this box (rank 0's GPU) and the limits it ran under: the box-to-box spread of
            # `value` is clock spread, `value / shader_clock_MHz` is the box-independent figure
            line["clock"] = dict(clock or {}, **device_limits(local_rank))
            if clock:
                line["clock"]["crops_per_s_per_GHz"] = line["value"] / (clock["shader_clock_MHz"] / 1e3)
                line["clock"]["measured_in"] = "a second pass of the timed loop with the probe beside it (the timed pass itself runs without it)"
                line["clock"]["ms_per_step_with_probe"] = dt_probe * 1e3
            line["kernel_ms_per_step"], line["roofline"] = roofline_record(eng, B, *prof)
            line["roofline_targets"] = secondary_rooflines(eng, B, *prof, bs512=not args.no_bs512_decode)
            ref = None
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
                    "sample": f"{4 * B} crops (the same bs{B} batch x4) through the oracle - the repo's CPU restatement of "
                              f"the reference path, not the reference itself (it cannot travel): torch-CPU fp32 model on "
                              f"{threads} threads + per-sample scipy decode loop on 1 thread, {secs:.1f} s; host has "
                              f"{os.cpu_count()} logical CPUs, cgroup quota {threads}",
                }
            src = ("last timed step (hipGraph replay)" if use_graph else "last timed step (eager launches)") + \
                  (f" of the {depth}-deep pipeline" if depth > 1 else "")
            if ref is not None and not args.no_parity:
                line["parity_vs_oracle"] = parity_record(args.precision, B, snap, ref, src)
            if pm is not None:
                eng_p, k_p, w_p, dt_p, snap_p, prof_p, dt1_p = pm
                kms, roof = roofline_record(eng_p, B, *prof_p)
                line[second_key] = {
                    "precision": second, "dtype_detail": DTYPE_DETAIL[second],
                    "value": B * world * k_p / dt_p, "unit": "crops/s", "ms_per_step": dt_p / k_p * 1e3,
                    "steps": k_p, "warmup": w_p, "launch": line["config"]["launch"],
                    "path_tflops": B * world * k_p * GFLOP_PER_CROP_FLIP / dt_p / 1e3,
                    "kernel_ms_per_step": kms, "roofline": roof,
                }
                if dt1_p is not None:
                    line[second_key]["one_step_in_flight"] = {"ms_per_step": dt1_p * 1e3, "value": B * world / dt1_p, "unit": "crops/s"}
                if ref is not None and not args.no_parity:
                    line[second_key]["parity_vs_oracle"] = parity_record(second, B, snap_p, ref, src)
                if second == THROUGHPUT_PRECISION:
                    line[second_key]["note"] = ("bf16 operands are narrower arithmetic than the fp32 reference: outside the path's 1e-3 "
                                                "tolerance (see parity_vs_oracle of this record); reported for comparison, never as `value`")
            if world == 1 and not args.no_drop_in:
                try:
                    line["drop_in"] = drop_in_record(dev, args, sd, B)
                    for k in ("test_step", "test_step_stream_depth2"):
                        if k in line["drop_in"]:
                            line["drop_in"][k]["frac_of_headline"] = line["drop_in"][k]["value"] / line["value"]
                except Exception as exc:  # noqa: BLE001 -- a secondary record must not take the bench line down
                    line["drop_in"] = {"error": repr(exc)[:300]}
            if world == 1 and not args.no_small_batch:
                try:
                    ref_fn = None
                    if not args.no_parity:
                        from oracle import model_ref as M_

                        ref_fn = lambda c: M_.predict(sd, c, 12, S.IMG_MEAN, S.IMG_STD)  # noqa: E731  (the checker, never the thing timed)
                    line["small_batch"] = small_batch_record(dev, args, sd, ref_fn)
                except Exception as exc:  # noqa: BLE001 -- a secondary record must not take the bench line down
                    line["small_batch"] = {"error": repr(exc)[:300]}
            if world == 1 and not args.no_config4:
                line["config4"] = config4_record(dev, args)
            print(json.dumps(line))
    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
