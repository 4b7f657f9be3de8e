"""Stand-alone timing of the reverse-scan variants at the C3 layer-0 shape (B=500, T=1024, H=64): one executable per
(HPMN_BWD_HELPER mode, -D knob set), nothing here ships.  Usage (GPU box):
    python tools/micro/feed_bench.py && sh tools/micro/run_feed.sh"""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "hpmn_amd", "csrc")

MAIN = r'''
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <hip/hip_runtime.h>
#include "hpmn_hip.h"
namespace hpmn { void set_last_hip_error(int) {} int gru_scan_bwd128_dispatch(const HpmnGruBwd &, hipStream_t) { return -2; }
                 int gru_scan_bwd_dispatch(const HpmnGruBwd &a, hipStream_t st); }
int main(int argc, char **argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 500, T = 1024, H = 64, D = argc > 2 ? atoi(argv[2]) : 32, dx = argc > 3 ? atoi(argv[3]) : 0;
    float *wg, *wc, *dhl, *dy, *hs, *gates, *dact, *carry;
    hipMalloc(&wg, (D + H) * 2 * H * 4); hipMalloc(&wc, (D + H) * H * 4);
    std::vector<float> w((D + H) * 2 * H, 0.01f);
    hipMemcpy(wg, w.data(), (D + H) * 2 * H * 4, hipMemcpyHostToDevice);
    hipMemcpy(wc, w.data(), (D + H) * H * 4, hipMemcpyHostToDevice);
    hipMalloc(&dhl, B * H * 4); hipMemset(dhl, 0, B * H * 4);
    hipMalloc(&carry, B * H * 4);
    hipMalloc(&dy, (size_t)B * T / 2 * H * 4); hipMemset(dy, 0, (size_t)B * T / 2 * H * 4);
    hipMalloc(&hs, (size_t)B * (T + 1) * H * 4); hipMemset(hs, 0, (size_t)B * (T + 1) * H * 4);
    hipMalloc(&gates, (size_t)B * T * 3 * H * 4); hipMemset(gates, 0, (size_t)B * T * 3 * H * 4);
    hipMalloc(&dact, (size_t)B * T * 3 * H * 4);
    float *dxb; hipMalloc(&dxb, (size_t)B * T * D * 4);
    HpmnGruBwd a = {};
    a.B = B; a.T = T; a.D = D; a.H = H; a.wg = wg; a.wc = wc; a.hs = hs; a.gates = gates;
    a.d_h_last = dhl; a.d_h_last_stride = H; a.d_y = dy; a.period = 2; a.d_act = dact; a.dh_carry = carry; if (dx) a.d_x = dxb;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) hpmn::gru_scan_bwd_dispatch(a, 0);
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) hpmn::gru_scan_bwd_dispatch(a, 0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-52s B=%d D=%d dx=%d %.4f ms/launch  %.0f ns/step\n", VARIANT, B, D, dx, ms / 5, ms / 5 * 1e6 / T);
    return 0;
}
'''

# name -> (HPMN_BWD_HELPER value, extra -D flags)
VARIANTS = {
    "h1 e_u helper wave (r2 default before the feeder)": ("1", []),
    "h2 chain + feeder": ("2", []),
}


def main():
    import sys
    extra = {}
    for arg in sys.argv[1:]:              # name=-DX=1,-DY=2
        name, flags = arg.split("=", 1)
        extra[name] = ("2", flags.split(","))
    variants = dict(VARIANTS)
    variants.update(extra)
    main_cc = os.path.join(HERE, "feed_main.hip")
    lines = ["#!/bin/sh"]
    for i, (name, (mode, flags)) in enumerate(variants.items()):
        exe = os.path.join(HERE, "feedb_%d" % i)
        open(main_cc, "w").write(MAIN)
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-w", "-I" + os.path.join(ROOT, "include"),
               "-I" + CSRC, '-DVARIANT="%s"' % name] + flags + [main_cc, os.path.join(CSRC, "gru_scan_bwd.hip"),
               os.path.join(CSRC, "gru_scan_bwd_feed.hip"), "-o", exe]
        subprocess.check_call(cmd)
        for args in ("500 32 0", "500 32 1", "500 64 0", "500 64 1", "250 64 1"):
            if mode == "1" and args.endswith("1"):
                continue
            lines.append("HPMN_BWD_HELPER=%s ./tools/micro/feedb_%d %s" % (mode, i, args))
    os.remove(main_cc)
    open(os.path.join(HERE, "run_feed.sh"), "w").write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
