"""dev only: per kernel of a source file, the s_waitcnt instructions (and barriers, scratch, MFMAs) inside its hottest loop (the
innermost loop with the most MFMAs) - a compiler-inserted `vmcnt(0)` in a stage loop drains the whole DMA ring.
    python scripts/loop_waitcnt.py probpose_code_amd/csrc/pp_winograd.hip [kernel-name-substring]"""
import collections, os, re, subprocess, sys, tempfile
src = os.path.abspath(sys.argv[1]); pat = sys.argv[2] if len(sys.argv) > 2 else ""
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
with tempfile.TemporaryDirectory() as td:
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", f"-I{root}/probpose_code_amd/csrc", f"-I{root}/include",
                    "-c", src, "-o", "x.o", "--save-temps"], cwd=td, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
    asm = [f for f in os.listdir(td) if f.endswith("gfx950.s")][0]
    lines = open(os.path.join(td, asm)).read().split("\n")
funcs, cur = {}, None
for l in lines:
    m = re.match(r"^(_Z\w+):", l)
    if m: cur = m.group(1); funcs[cur] = []
    elif l.startswith(".Lfunc_end"): cur = None
    elif cur: funcs[cur].append(l)
for name, body in funcs.items():
    if pat not in name: continue
    labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"(\.LBB\d+_\d+):", l)] if m}
    loops = []
    for i, l in enumerate(body):
        m = re.search(r"s_cbranch_\w+ (\.LBB\d+_\d+)|s_branch (\.LBB\d+_\d+)", l)
        if m:
            t = m.group(1) or m.group(2)
            if t in labels and labels[t] < i: loops.append((labels[t], i))
    if not loops: continue
    inner = [(a, b) for a, b in loops if not any(a2 >= a and b2 <= b and (a2, b2) != (a, b) for a2, b2 in loops)]
    a, b = max(inner, key=lambda ab: sum("v_mfma" in x for x in body[ab[0]:ab[1]]))
    nm = sum("v_mfma" in x for x in body[a:b])
    if nm == 0: continue
    wc = collections.Counter(l.strip() for l in body[a:b] if "s_waitcnt" in l)
    print(f"{name[:90]}\n   hottest inner loop: {b - a} instructions, {nm} MFMAs, {sum('s_barrier' in x for x in body[a:b])} barriers, "
          f"{sum('scratch_' in x for x in body[a:b])} scratch, {sum('buffer_load' in x for x in body[a:b])} buffer loads, {sum('ds_read' in x for x in body[a:b])} ds_read")
    for k, v in wc.most_common(): print(f"      {v:3d} x {k}")
