"""Register / scratch / occupancy report of every product kernel (cross-compiles each .hip to gfx950 assembly and reads the
kernel metadata and the compiler's "Kernel info" comments).  Anything with scratch or spills deserves a look: the scan and
read kernels went 15-40 % slower more than once from an innocent-looking change that pushed them over a register cliff.

Two things the plain register count hides (round 4, both found in the ISA, both cost double-digit percentages):
  * OCCUPANCY CLIFFS: arch VGPRs + AGPRs share 512 registers per SIMD lane, so 257-264 of them mean ONE wave per SIMD where 256
    mean two.  The four-wave H = 128 scans sat at 259-263 ("256 VGPRs + a few AGPRs"): a batch of 500 sequences ran as two
    rounds of 256 workgroups.  Kernels within 8 registers ABOVE a boundary (128 / 168 / 256) are marked "<-- cliff".
  * SCRATCH INSIDE LOOPS: a few spilled dwords are harmless in a prologue and poison in a serial chain's loop (a scratch_load
    and its s_waitcnt vmcnt(0) per step).  Loops (backward branches) that touch scratch are listed per kernel.
Usage: python tools/check_resources.py [--fail-on-scratch]"""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "hpmn_amd", "csrc")
PAT = re.compile(r"\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+).*?\.sgpr_count:\s+(\d+).*?"
                 r"\.sgpr_spill_count:\s+(\d+).*?\.vgpr_count:\s+(\d+).*?\.vgpr_spill_count:\s+(\d+)", re.S)


def demangle(names):
    try:
        out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt"] + names, stdout=subprocess.PIPE, text=True).stdout
        return [l.split("(")[0].replace("void ", "") for l in out.splitlines()]
    except OSError:
        return names


def kernel_info(text):
    """mangled name -> (arch VGPRs + AGPRs, waves per SIMD, [(first, last) line of every loop that touches scratch])."""
    out = {}
    for m in re.finditer(r"^(_Z\w+):\s*; @.*?^; TotalNumVgprs: (\d+).*?^; Occupancy: (\d+)", text, re.S | re.M):
        body = text[m.start():m.end()].split("\n")
        labels, loops = {}, []
        for i, line in enumerate(body):
            lm = re.match(r"^(\.LBB\d+_\d+):", line)
            if lm:
                labels[lm.group(1)] = i
        for i, line in enumerate(body):
            bm = re.search(r"s_c?branch\w* (\.LBB\d+_\d+)", line)
            if bm and labels.get(bm.group(1), i) < i:
                a = labels[bm.group(1)]
                # (role-dispatch kernels are one big pseudo-loop over thousands of lines: only real, inner loops count)
                if i - a < 3000 and any("scratch_" in x for x in body[a:i + 1]):
                    loops.append((a, i))
        out[m.group(1)] = (int(m.group(2)), int(m.group(3)), loops)
    return out


def main():
    bad = 0
    with tempfile.TemporaryDirectory() as tmp:
        for src in sorted(glob.glob(os.path.join(CSRC, "*.hip"))):
            out = os.path.join(tmp, os.path.basename(src) + ".s")
            subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-w",
                                   "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-S", "--cuda-device-only",
                                   "-o", out, src])
            text = open(out).read()
            nflat = len(re.findall(r"^\s+flat_(?:load|store|atomic)", text, flags=re.M))
            if nflat:
                # an LDS or global access through a GENERIC pointer: flat_* counts in vmcnt AND lgkmcnt, so waiting for
                # it drains the wave's global stores (DESIGN_HISTORY.md 3.8: progress counters of the fused forward kernel)
                print("%-58s %d flat_load/store/atomic instruction(s)  <-- generic-pointer access" % (os.path.basename(src), nflat))
            rows = PAT.findall(text)
            names = demangle([r[0] for r in rows])
            info = kernel_info(text)
            for name, r in zip(names, rows):
                scratch, sgpr, sspill, vgpr, vspill = map(int, r[1:])
                tot, occ, loops = info.get(r[0], (vgpr, 0, []))
                flag = "  <-- scratch/spill" if (scratch or vspill) else ""
                if any(b < tot <= b + 8 for b in (128, 168, 256)):
                    flag += "  <-- cliff (%d registers over an occupancy boundary)" % min(tot - b for b in (128, 168, 256) if tot > b)
                if loops:
                    flag += "  <-- scratch inside %d loop(s), lines %s" % (len(loops), ",".join("%d-%d" % ab for ab in loops[:3]))
                bad += 1 if (scratch or vspill) else 0
                print("%-58s vgpr+agpr %3d  occ %d  sgpr %3d  scratch %4d  vspill %3d  sspill %3d%s"
                      % (name[:58], tot, occ, sgpr, scratch, vspill, sspill, flag))
    print("%d kernel(s) with scratch or VGPR spills" % bad)
    return 1 if (bad and "--fail-on-scratch" in sys.argv) else 0


if __name__ == "__main__":
    sys.exit(main())
