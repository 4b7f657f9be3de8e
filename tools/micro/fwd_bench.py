"""Stand-alone timing of the fused forward layer kernels at the C3 shapes: layer 0 (gather, D=32, T=1024) and layer 1
(D=64, T=512), generation 1 (gru_fused_fwd.hip) against generation 3 (gru_fused_fwd3.hip) and its -D knob sets
(e.g. "u=-DMF_UPROD32=1,-DMF_UPROD64=1").
Nothing here ships.  Usage (GPU box):  python tools/micro/fwd_bench.py [name=-DX=1,...] && sh tools/micro/run_fwd.sh"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "hpmn_amd", "csrc")

MAIN = r'''
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#include <hip/hip_runtime.h>
#include "hpmn_hip.h"
namespace hpmn { void set_last_hip_error(int) {} int gru_fused_fwd_dispatch(const HpmnGruFusedFwd &a, hipStream_t st); }
int main(int argc, char **argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 500, layer = argc > 2 ? atoi(argv[2]) : 0;
    const int H = 64, F = 2, E = 16, Tids = 1001, T = layer == 0 ? 1024 : 512, D = layer == 0 ? 32 : 64;
    const long V = 3308019;
    float *emb, *wg, *wc, *bg, *bc, *x, *xo, *hl, *y, *hs, *gates; int *ids;
    hipMalloc(&emb, V * E * 4); hipMemset(emb, 0, V * E * 4);
    hipMalloc(&wg, (D + H) * 2 * H * 4); hipMalloc(&wc, (D + H) * H * 4); hipMalloc(&bg, 2 * H * 4); hipMalloc(&bc, H * 4);
    std::vector<float> w((D + H) * 2 * H, 0.01f);
    hipMemcpy(wg, w.data(), (D + H) * 2 * H * 4, hipMemcpyHostToDevice);
    hipMemcpy(wc, w.data(), (D + H) * H * 4, hipMemcpyHostToDevice);
    hipMemset(bg, 0, 2 * H * 4); hipMemset(bc, 0, H * 4);
    hipMalloc(&x, (size_t)B * T * D * 4); hipMemset(x, 0, (size_t)B * T * D * 4);
    hipMalloc(&xo, (size_t)B * T * D * 4);
    std::vector<int> h((size_t)B * Tids * F); std::mt19937 g(1);
    for (auto &v : h) v = g() % V;
    hipMalloc(&ids, h.size() * 4); hipMemcpy(ids, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMalloc(&hl, B * H * 4); hipMalloc(&y, (size_t)B * T / 2 * H * 4);
    hipMalloc(&hs, (size_t)B * (T + 1) * H * 4); hipMalloc(&gates, (size_t)B * T * 3 * H * 4);
    HpmnGruFusedFwd a = {};
    a.B = B; a.T = T; a.D = D; a.H = H; a.wg = wg; a.bg = bg; a.wc = wc; a.bc = bc; a.h_last = hl; a.h_last_stride = H;
    a.y = y; a.period = 2; a.hs = hs; a.gates = gates;
    if (layer == 0) { a.ids = ids; a.emb = emb; a.Tids = Tids; a.F = F; a.E = E; a.front_zero = 23; a.V = V; a.x_out = xo; }
    else a.x = x;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) hpmn::gru_fused_fwd_dispatch(a, 0);
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) hpmn::gru_fused_fwd_dispatch(a, 0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s B=%d layer %d  %.4f ms/launch  %.0f ns/step  (%s)\n", VARIANT, B, layer, ms / 5, ms / 5 * 1e6 / T,
           hipGetErrorString(hipGetLastError()));
    return 0;
}
'''


def main():
    variants = {"gen1": ("1", []), "gen3": ("3", [])}
    for arg in sys.argv[1:]:
        name, flags = arg.split("=", 1)
        variants[name] = ("3", flags.split(","))
    main_cc = os.path.join(HERE, "fwd_main.hip")
    open(main_cc, "w").write(MAIN)
    lines = ["#!/bin/sh"]
    for i, (name, (gen, flags)) in enumerate(variants.items()):
        exe = os.path.join(HERE, "feedb_fwd_%d" % i)
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-w", "-I" + os.path.join(ROOT, "include"),
               "-I" + CSRC, '-DVARIANT="%s"' % name] + flags + [main_cc, os.path.join(CSRC, "gru_fused_fwd.hip"),
               os.path.join(CSRC, "gru_fused_fwd3.hip"), "-o", exe]
        subprocess.check_call(cmd)
        for B in (500, 250):
            for layer in (0, 1):
                lines.append("HPMN_FUSED_FWD_GEN=%s ./tools/micro/feedb_fwd_%d %d %d" % (gen, i, B, layer))
    os.remove(main_cc)
    open(os.path.join(HERE, "run_fwd.sh"), "w").write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
