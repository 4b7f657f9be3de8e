"""Build-time guards on what the compiler made of the latency-critical kernels (CPU: hipcc cross-compiles gfx950 to assembly;
no GPU).  Two properties that neither the source nor a parity test shows and that cost double-digit percentages when they
slipped (DESIGN_HISTORY.md 3.21, tools/check_resources.py):

* the four-wave H = 128 scans must fit 256 registers (arch VGPRs + AGPRs): 257-264 is ONE wave per SIMD, and a batch of 500
  sequences then runs every scan launch in two rounds;
* no loop of a serial-chain kernel may touch scratch (a scratch_load + s_waitcnt vmcnt(0) per step).
"""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _assembly(tmp_path, name):
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc")
    out = str(tmp_path / (name + ".s"))
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-w", "-I" + os.path.join(ROOT, "include"),
                           "-I" + os.path.join(ROOT, "hpmn_amd", "csrc"), "-S", "--cuda-device-only", "-o", out,
                           os.path.join(ROOT, "hpmn_amd", "csrc", name + ".hip")])
    with open(out) as f:
        return f.read()


def test_h128_scans_fit_two_workgroups_per_cu(tmp_path):
    import check_resources
    info = check_resources.kernel_info(_assembly(tmp_path, "gru_scan128"))
    four_wave = {k: v for k, v in info.items() if "Li4E" in k}
    assert len(four_wave) == 3, sorted(info)                       # forward (training, inference), reverse
    for name, (total, occupancy, _) in four_wave.items():
        assert total <= 256 and occupancy >= 2, "%s: %d registers, %d wave(s) per SIMD" % (name, total, occupancy)
    # the time loops themselves stay free of scratch (the remainder code behind them may reload a spilled value)
    text = _assembly(tmp_path, "gru_scan128")
    for name in four_wave:
        body = text[text.index(name + ":"):]
        body = body[:body.index("s_endpgm")]
        lines = body.split("\n")
        head = [i for i, l in enumerate(lines) if "Inner Loop Header" in l]
        assert head, name
        # the longest inner loop is the time loop (the others are the weight-loading prologue's)
        spans = []
        for first in head:
            label = lines[first].split(":")[0].strip()
            back = [i for i in range(first, len(lines)) if "branch" in lines[i] and lines[i].rstrip().endswith(label)]
            if back:
                spans.append((back[-1] - first, first, back[-1]))
        assert spans, name
        _, first, back = max(spans)
        assert back - first > 500, name
        assert not any("scratch_" in l for l in lines[first:back + 1]), name + ": scratch in the time loop"


def test_two_layer_forward_keeps_nothing_in_scratch(tmp_path):
    import check_resources
    info = check_resources.kernel_info(_assembly(tmp_path, "gru_pair_fwd"))
    pair = {k: v for k, v in info.items() if "gru_pair_fwd_kernel" in k}
    assert len(pair) == 6, sorted(info)
    for name, (total, occupancy, loops) in pair.items():
        assert total <= 256 and occupancy >= 2 and not loops, "%s: %d registers, scratch in loops %s" % (name, total, loops)


def test_tile_evaluation_kernels_keep_their_time_loops_out_of_scratch(tmp_path):
    """r5: the four-wave tile kernels (hpmn_tile_fwd).  H = 64 must fit TWO workgroups per CU (its whole advantage over the
    twelve-wave kernel is two tiles interleaving on a CU: cut for three or four the loop spilled and a pass took 1.7x as long);
    H = 128 may use the whole 512-register budget -- mode 2 even spills while it prepares its weight fragments -- but no
    time loop of any of them may touch scratch."""
    import check_resources
    info = check_resources.kernel_info(_assembly(tmp_path, "gru_tile64"))
    k64 = {k: v for k, v in info.items() if "gru_tile64_fwd_kernel" in k}
    assert len(k64) == 2, sorted(info)
    for name, (total, occupancy, loops) in k64.items():
        assert total <= 256 and occupancy >= 2 and not loops, "%s: %d registers, scratch in loops %s" % (name, total, loops)
    text = _assembly(tmp_path, "gru_tile128")
    info = check_resources.kernel_info(text)
    k128 = {k: v for k, v in info.items() if "gru_tile128_fwd_kernel" in k}
    assert len(k128) == 3, sorted(info)
    for name, (total, occupancy, loops) in k128.items():
        assert total <= 512 and occupancy >= 1, name
        # scratch only in front of the time loop: behind the step loop's first barrier pair nothing touches it
        body = text[text.index(name + ":"):]
        body = body[:body.index("s_endpgm")].split("\n")
        bars = [i for i, l in enumerate(body) if "s_barrier" in l]
        assert len(bars) >= 10, name
        assert not any("scratch_" in l for l in body[bars[2]:]), name + ": scratch behind the prologue"


def test_read_training_launch_on_eight_waves_fits_the_register_file(tmp_path):
    """r5: the read path's training launch on bf16 fragments runs EIGHT waves per workgroup (two per SIMD): it must fit 256
    registers per wave -- 257 would not launch at all -- and what the compiler spills to get there must stay small (the first
    eight-wave build spilled 165 registers; with four column tiles per wave instead of eight: 4).  The fp32 launch and the
    inference launch keep four waves and two workgroups per CU."""
    import re
    text = _assembly(tmp_path, "read_path")

    def facts(prefix):
        m = re.search(r"^(%s\w*):\s*; @" % prefix, text, re.M)
        assert m, prefix
        body = text[m.start():]
        body = body[:body.index("; Occupancy:") + 40]
        regs = int(re.search(r"; TotalNumVgprs: (\d+)", body).group(1))
        occ = int(re.search(r"; Occupancy: (\d+)", body).group(1))
        scratch = sum(1 for l in body[:body.index("s_endpgm")].split("\n") if "scratch_" in l)
        return regs, occ, scratch

    regs, occ, scratch = facts("_ZN4hpmn19read_fwd_bwd_kernelILb1")
    assert regs <= 256 and occ >= 2 and scratch <= 16, (regs, occ, scratch)
    for prefix in ("_ZN4hpmn19read_fwd_bwd_kernelILb0", "_ZN4hpmn15read_fwd_kernel"):
        regs, occ, scratch = facts(prefix)
        assert regs <= 256 and occ >= 2 and scratch == 0, (prefix, regs, occ, scratch)
