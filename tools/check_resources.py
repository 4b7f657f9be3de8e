"""Register / scratch report of every product kernel (cross-compiles each .hip to gfx950 assembly and reads the
kernel metadata).  Anything with scratch or spills deserves a look: the scan and read kernels went 15-40 %
slower more than once from an innocent-looking change that pushed them over a register cliff.
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
                # it drains the wave's global stores (progress counters of the fused forward kernel)
                print("%-58s %d flat_load/store/atomic instruction(s)  <-- generic-pointer access" % (os.path.basename(src), nflat))
            rows = PAT.findall(text)
            names = demangle([r[0] for r in rows])
            for name, r in zip(names, rows):
                scratch, sgpr, sspill, vgpr, vspill = map(int, r[1:])
                flag = "  <-- scratch/spill" if (scratch or vspill) else ""
                bad += 1 if (scratch or vspill) else 0
                print("%-58s vgpr %3d  sgpr %3d  scratch %4d  vspill %3d  sspill %3d%s"
                      % (name[:58], vgpr, sgpr, scratch, vspill, sspill, flag))
    print("%d kernel(s) with scratch or VGPR spills" % bad)
    return 1 if (bad and "--fail-on-scratch" in sys.argv) else 0


if __name__ == "__main__":
    sys.exit(main())
