"""vmcnt waits inside the loops of a kernel's gfx950 assembly -- a wait for (nearly) everything in flight inside a hot loop is a
stall of a full memory round trip per trip (a value computed on at load time, a load inside a branch).
Usage: python tools/loop_waits.py <file.hip> [kernel-name-substring]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else ""
with tempfile.TemporaryDirectory() as tmp:
    out = os.path.join(tmp, "k.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-w", "-I" + os.path.join(ROOT, "include"),
                           "-I" + os.path.join(ROOT, "hpmn_amd", "csrc"), "-S", "--cuda-device-only", "-o", out, src])
    t = open(out).read()
names = re.findall(r"^(_Z\w+):", t, flags=re.M)
for name in names:
    if pat not in name or "kernel" not in name:
        continue
    i = t.index("\n" + name + ":")
    j = t.index("s_endpgm", i)
    body = t[i:j].split("\n")
    labels = {}
    for k, l in enumerate(body):
        mm = re.match(r"^(\.LBB\d+_\d+):", l)
        if mm:
            labels[mm.group(1)] = k
    loops = []
    for k, l in enumerate(body):
        mm = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", l) or re.search(r"s_branch\s+(\.LBB\d+_\d+)", l)
        if mm and mm.group(1) in labels and labels[mm.group(1)] < k:
            loops.append((labels[mm.group(1)], k))
    print(name[:100], len(body), "lines")
    for a, b in loops:
        seg = [l for l in body[a:b + 1] if re.match(r"^\s+[a-z]", l)]
        if len(seg) < 60:
            continue
        vm = [int(re.search(r"vmcnt\((\d+)\)", l).group(1)) for l in seg if "vmcnt" in l]
        nv = sum(1 for l in seg if l.split()[0].startswith(("global_load", "buffer_load")))
        ns = sum(1 for l in seg if l.split()[0].startswith(("global_store", "buffer_store")))
        print("   loop %5d-%5d  %4d instr  %3d loads %3d stores  vmcnt waits: min %s  %s" % (a, b, len(seg), nv, ns, min(vm) if vm else "-", vm[:80]))
