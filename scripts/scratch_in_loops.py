"""dev only: compile a kernel source to gfx950 ISA and list, per kernel with scratch memory, the scratch loads / stores that sit
INSIDE loops (a spilled register or an un-promoted local in a stage loop costs a memory round trip and an s_waitcnt per pass).
    python scripts/scratch_in_loops.py probpose_code_amd/csrc/pp_panel_split.hip [extra hipcc flags]"""
import os, re, subprocess, sys, tempfile
src = os.path.abspath(sys.argv[1])
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
with tempfile.TemporaryDirectory() as td:
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", f"-I{root}/probpose_code_amd/csrc", f"-I{root}/include",
                    "-c", src, "-o", "x.o", "--save-temps"] + sys.argv[2:], cwd=td, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
    asm = [f for f in os.listdir(td) if f.endswith("gfx950.s")][0]
    lines = open(os.path.join(td, asm)).read().split("\n")
funcs, cur = {}, None
for l in lines:
    m = re.match(r"^(_Z\w+):", l)
    if m:
        cur = m.group(1); funcs[cur] = []
    elif l.startswith(".Lfunc_end"):
        cur = None
    elif cur:
        funcs[cur].append(l)
for name, body in funcs.items():
    scr = [i for i, l in enumerate(body) if "scratch_" in l]
    if not scr:
        continue
    labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"(\.LBB\d+_\d+):", l)] if m}
    loops = []
    for i, l in enumerate(body):
        m = re.search(r"s_cbranch_\w+ (\.LBB\d+_\d+)|s_branch (\.LBB\d+_\d+)", l)
        if m:
            t = m.group(1) or m.group(2)
            if t in labels and labels[t] < i:
                loops.append((labels[t], i))
    inner = [(a, b) for a, b in loops if not any(a2 >= a and b2 <= b and (a2, b2) != (a, b) for a2, b2 in loops)]
    print(f"{name[:100]}: {len(scr)} scratch instructions")
    for a, b in sorted(set(loops)):
        ns = sum(a <= i <= b for i in scr)
        nm = sum("v_mfma" in x for x in body[a:b])
        if ns:
            print(f"    loop lines {a}-{b} ({'innermost' if (a, b) in inner else 'outer'}): {ns} scratch ops, {nm} MFMAs, {b - a} instructions")
