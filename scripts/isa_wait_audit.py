#!/usr/bin/env python3
"""Where a kernel's assembly waits for memory right behind a load (cross-compiled, no GPU needed).

  scripts/isa_wait_audit.py dp_device.hip [name filter]

Per kernel: vector-memory loads, `s_waitcnt vmcnt(...)` instructions, full waits (`vmcnt(0)`), and full waits at most three instructions
behind a load ("immediate": the pattern a `cond ? p[i] : 0` load, a copied in-flight register or a not-unrolled copy loop compiles into --
each one is a memory round trip nothing else of the wave overlaps with).  The same before the first s_barrier (the prologue).  A count is a
place to LOOK, not a verdict: DESIGN.md section 6 has three cases where it was the bottleneck and two where it was not."""
import os, re, subprocess, sys
src = sys.argv[1] if len(sys.argv) > 1 else "dp_device.hip"
filt = sys.argv[2] if len(sys.argv) > 2 else "."
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "whatshap_amd", "csrc")
out = "/tmp/whamd_isa"; os.makedirs(out, exist_ok=True)
asm = os.path.join(out, src.replace(".hip", ".s"))
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S", src, "-o", asm], cwd=root, check=True, stderr=subprocess.DEVNULL)
name, body, kernels = None, [], {}
for line in open(asm):
    m = re.match(r"^(_Z\w+):", line)
    if m and name is None:
        name, body = m.group(1), []
        continue
    if name is not None:
        body.append(line)
        if re.match(r"^\.Lfunc_end\d+:", line):   # (not the first s_endpgm: a kernel may return early)
            kernels[name] = body
            name = None
def stats(lines):
    loads = waits = full = imm = 0
    last = -10
    for i, l in enumerate(lines):
        if re.search(r"\b(global|buffer|flat)_load", l): loads += 1; last = i
        m = re.search(r"s_waitcnt .*vmcnt\((\d+)\)", l)
        if m:
            waits += 1
            if m.group(1) == "0":
                full += 1
                if i - last <= 3: imm += 1
    return loads, waits, full, imm
print(f"{'kernel':72s} {'loads':>6s} {'waits':>6s} {'full':>5s} {'imm.':>5s} | prologue: loads full imm.")
for k, lines in kernels.items():
    short = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
    short = re.sub(r"whamd::\(anonymous namespace\)::", "", short).split("(")[0].replace("void ", "")
    if not re.search(filt, short): continue
    b = next((i for i, l in enumerate(lines) if "s_barrier" in l), len(lines))
    a, p = stats(lines), stats(lines[:b])
    print(f"{short[:72]:72s} {a[0]:6d} {a[1]:6d} {a[2]:5d} {a[3]:5d} | {p[0]:6d} {p[2]:5d} {p[3]:5d}")
