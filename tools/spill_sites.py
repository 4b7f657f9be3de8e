"""List the scratch (spill) instructions of a gfx950 assembly file with the basic block they sit in and whether that
block belongs to a loop (a reload inside the time loop of a chain wave is a latency on the serial chain).
Usage: python tools/spill_sites.py file.s"""
import re
import sys

kernel, block, in_loop = None, None, False
for n, line in enumerate(open(sys.argv[1]), 1):
    m = re.match(r"^(_Z\w+):", line)
    if m:
        kernel, block, in_loop = m.group(1), None, False
        continue
    m = re.match(r"^(\.LBB\d+_\d+):(.*)", line)
    if m:
        block, in_loop = m.group(1), "Loop" in m.group(2)
        continue
    if "scratch_" in line and kernel:
        print("%s  %-10s %-8s line %6d  %s" % (kernel[:48], block, "LOOP" if in_loop else "-", n, line.strip()[:70]))
