"""CPU (build check, no GPU): the stage loops of the hot kernels must not touch scratch memory.

Round 4 found the wide-tile split kernel loading its DMA cursor from scratch in every K-step (a local the compiler could not
promote: two scratch loads + `s_waitcnt vmcnt(0)` in front of each stage's DMA issue, since round 2) and the first generalised
Winograd kernel keeping six accumulators in scratch for the whole launch - neither shows up in any numerics test, both cost
8 - 27 % of their kernels. This test compiles the sources to gfx950 ISA (hipcc cross-compiles without a GPU) and asserts that no
loop that contains MFMAs contains a scratch instruction (scripts/scratch_in_loops.py is the dev-side tool with the same logic).
"""
import os
import re
import subprocess
import tempfile
from concurrent.futures import ThreadPoolExecutor

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "probpose_code_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"
# kernels allowed to keep scratch accesses in an OUTER (per-tile) loop: the fused 1x1-head deconvolution reloads six registers per tile
OUTER_OK = ("panel_split_kernelILi2ELi6ELi4ELb1ELi2ELb1",)


# per-source flags of probpose_code_amd/csrc/Makefile
EXTRA_FLAGS = {"pp_ffn_split.hip": ["-fno-slp-vectorize"], "pp_ffn_dma.hip": ["-fno-slp-vectorize"], "pp_mlp.hip": ["-fno-slp-vectorize"]}


def _isa(src):
    with tempfile.TemporaryDirectory() as td:
        subprocess.run([HIPCC, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", f"-I{CSRC}", f"-I{ROOT}/include"] + EXTRA_FLAGS.get(src, []) +
                       ["-c", os.path.join(CSRC, src), "-o", "x.o", "--save-temps"], cwd=td, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
        asm = [f for f in os.listdir(td) if f.endswith("gfx950.s")][0]
        return open(os.path.join(td, asm)).read().split("\n")


def _loops_with_scratch(src):
    lines = _isa(src)
    funcs, cur = {}, None
    for l in lines:
        m = re.match(r"^(_Z\w+):", l)
        if m:
            cur = m.group(1)
            funcs[cur] = []
        elif l.startswith(".Lfunc_end"):
            cur = None
        elif cur:
            funcs[cur].append(l)
    bad = []
    for name, body in funcs.items():
        labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"(\.LBB\d+_\d+):", l)] if m}
        loops = []
        for i, l in enumerate(body):
            m = re.search(r"s_c?branch\w* (\.LBB\d+_\d+)", l)
            if m and m.group(1) in labels and labels[m.group(1)] < i:
                loops.append((labels[m.group(1)], i))
        inner = [(a, b) for a, b in loops if not any(a2 >= a and b2 <= b and (a2, b2) != (a, b) for a2, b2 in loops)]
        for a, b in loops:
            n_scr = sum("scratch_" in x for x in body[a:b])
            n_mfma = sum("v_mfma" in x for x in body[a:b])
            if n_scr and n_mfma and ((a, b) in inner or not any(k in name for k in OUTER_OK)):
                bad.append((name, a, b, n_scr, n_mfma))
    return bad


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
def test_hot_kernels_keep_scratch_out_of_their_mfma_loops():
    srcs = ["pp_panel_split.hip", "pp_winograd.hip", "pp_qkv_attn_split.hip", "pp_ffn_split.hip", "pp_ffn_dma.hip"]
    with ThreadPoolExecutor(max_workers=4) as ex:
        results = list(ex.map(_loops_with_scratch, srcs))
    bad = [(s,) + b for s, r in zip(srcs, results) for b in r]
    assert not bad, "scratch memory inside an MFMA loop:\n" + "\n".join(f"  {s}: {n[:80]} lines {a}-{b}: {k} scratch ops / {m} MFMAs" for s, n, a, b, k, m in bad)


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
def test_twelve_wave_feed_forward_kernels_fit_three_waves_per_simd_without_scratch():
    """pp_ffn_dma.hip runs twelve waves per workgroup = three per SIMD: every wave has 512 / 3 -> 168 registers, and the kernel was
    built on the condition that nothing spills (a spill inside the LayerNorm phase was the first suspect when stale words showed up
    in its output). The kernel descriptors of both entry kernels must say so."""
    text = "\n".join(_isa("pp_ffn_dma.hip"))
    kernels = re.findall(r"\.amdhsa_kernel (\w+)(.*?)\.end_amdhsa_kernel", text, flags=re.S)
    seen = {}
    for name, body in kernels:
        vgpr = int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", body).group(1))
        scratch = int(re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", body).group(1))
        seen[name] = (vgpr, scratch)
    assert any("ffn_dma_kernel" in n for n in seen) and any("proj_ffn_dma_kernel" in n for n in seen), seen
    for name, (vgpr, scratch) in seen.items():
        assert vgpr <= 168 and scratch == 0, f"{name}: {vgpr} registers, {scratch} bytes of scratch per lane"


def _regs(tok):
    """VGPR numbers named by an operand token: v12 or v[12:15]"""
    m = re.fullmatch(r"v\[(\d+):(\d+)\],?", tok) or re.fullmatch(r"v(\d+),?", tok)
    if not m:
        return set()
    lo = int(m.group(1))
    return set(range(lo, int(m.group(m.lastindex)) + 1))


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
def test_sixteen_byte_buffer_stores_keep_their_data_registers_for_a_wait_state():
    """A buffer_store_dwordx4 reads its data registers a cycle late for lanes 12 - 15 of every row (scripts/micro/mubuf_store_hazard.hip:
    one wait state with an SGPR soffset, two with an immediate one), and the compiler inserts none for the SGPR form - a dead value's
    register reused directly behind the store changes what those lanes write ("Three traps"). The sources guard every such
    store with an `s_nop 3` that depends on its data; in the ISA at least two instructions must separate the store from the first
    instruction that writes one of its data registers."""
    srcs = [f for f in sorted(os.listdir(CSRC)) if f.endswith(".hip")]
    with ThreadPoolExecutor(max_workers=4) as ex:
        isas = list(ex.map(_isa, srcs))
    bad, seen = [], 0
    for src, lines in zip(srcs, isas):
        code = [l.split(";")[0].strip() for l in lines]
        code = [l for l in code if l and not l.startswith((".", "//")) and not l.endswith(":")]
        for i, l in enumerate(code):
            if not l.startswith(("buffer_store_dwordx4", "buffer_store_dwordx3")):
                continue
            seen += 1
            data = _regs(l.split()[1])
            for d, nxt in enumerate(code[i + 1:i + 4]):
                tok = nxt.split()
                writes = _regs(tok[1]) if len(tok) > 1 and tok[0].startswith(("v_", "ds_read", "buffer_load", "global_load", "flat_load", "scratch_load")) else set()
                wait = sum(int(c.split()[1]) + 1 if c.startswith("s_nop") else 1 for c in code[i + 1:i + 1 + d])
                if writes & data and wait < 2:
                    bad.append(f"{src}: `{l}` then after {wait} wait state(s) `{nxt}`")
    assert seen >= 18, f"expected the twelve-wave kernels' 16-byte buffer stores in the ISA, found {seen}"
    assert not bad, "\n".join(bad[:10])
