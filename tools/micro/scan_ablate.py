"""Where do the cycles of a scan step go?  Builds timing-only variants of the product forward/backward
scan kernels by text substitution on the product sources (nothing here ships), one executable per
variant, and prints a runner script.  Usage (GPU box): python tools/micro/scan_ablate.py && sh tools/micro/run_ablate.sh"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "hpmn_amd", "csrc")

MAIN_FWD = r'''
#include <cstdio>
#include <vector>
namespace hpmn { void set_last_hip_error(int) {} int gru_scan_fwd128_dispatch(const HpmnGruFwd &, hipStream_t) { return -2; } }
int main() {
    const int B = 500, T = 1024, H = 64, D = 32;
    float *xp, *wg, *wc, *hl, *y, *hs, *gates;
    hipMalloc(&xp, (size_t)B * T * 3 * H * 4); hipMemset(xp, 0, (size_t)B * T * 3 * H * 4);
    hipMalloc(&wg, (D + H) * 2 * H * 4); hipMalloc(&wc, (D + H) * H * 4);
    std::vector<float> w((D + H) * 2 * H, 0.01f);
    hipMemcpy(wg, w.data(), (D + H) * 2 * H * 4, hipMemcpyHostToDevice);
    hipMemcpy(wc, w.data(), (D + H) * H * 4, hipMemcpyHostToDevice);
    hipMalloc(&hl, B * H * 4); hipMalloc(&y, (size_t)B * T / 2 * H * 4);
    hipMalloc(&hs, (size_t)B * (T + 1) * H * 4); hipMalloc(&gates, (size_t)B * T * 3 * H * 4);
    HpmnGruFwd a = {};
    a.B = B; a.T = T; a.D = D; a.H = H; a.xp = xp; a.wg = wg; a.wc = wc; a.h_last = hl; a.h_last_stride = H;
    a.y = y; a.period = 2; a.hs = hs; a.gates = gates;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) hpmn::gru_scan_fwd_dispatch(a, 0);
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) hpmn::gru_scan_fwd_dispatch(a, 0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %.4f ms/launch  %.0f cycles/step @2.4GHz\n", VARIANT, ms / 5, ms / 5 * 2.4e6 / T);
    return 0;
}
'''


MAIN_BWD = r'''
#include <cstdio>
#include <vector>
namespace hpmn { void set_last_hip_error(int) {} int gru_scan_bwd128_dispatch(const HpmnGruBwd &, hipStream_t) { return -2; } }
int main() {
    const int B = 500, T = 1024, H = 64, D = 32;
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
    HpmnGruBwd a = {};
    a.B = B; a.T = T; a.D = D; a.H = H; a.wg = wg; a.wc = wc; a.hs = hs; a.gates = gates;
    a.d_h_last = dhl; a.d_h_last_stride = H; a.d_y = dy; a.period = 2; a.d_act = dact; a.dh_carry = carry;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) hpmn::gru_scan_bwd_dispatch(a, 0);
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) hpmn::gru_scan_bwd_dispatch(a, 0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %.4f ms/launch  %.0f cycles/step @2.4GHz\n", VARIANT, ms / 5, ms / 5 * 2.4e6 / T);
    return 0;
}
'''


def bwd_variants(src):
    out = {"b0 baseline": src}
    v1 = src
    for old in ("da[0] = dar;", "da[H] = dau;", "da[2 * H] = dcp;"):
        v1 = sub(v1, old, "")
    out["b1 no global stores"] = v1
    v2 = sub(v1, "p.g[i] = *reinterpret_cast<const f2 *>(gb + (long)t * 3 * H + g_col[i]);", "p.g[i] = f2{(float)t * 1e-6f, 0.5f};")
    v2 = sub(v2, "p.hp = *reinterpret_cast<const f2 *>(hsb + (long)t * H + h_col);", "p.hp = f2{(float)t * 1e-6f, 0.5f};")
    v2 = sub(v2, "p.dy[tt] = dyb[(long)(pf_row > 0 ? pf_row : 0) * dy_stride];", "p.dy[tt] = (float)pf_row * 1e-6f;")
    out["b2 b1 + no prefetch loads"] = v2
    v3 = sub(src, "p.dy[tt] = dyb[(long)(pf_row > 0 ? pf_row : 0) * dy_stride];", "p.dy[tt] = (float)pf_row * 1e-6f;")
    out["b3 no d_y loads only"] = v3
    return out


MAIN_PROJ = r'''
#include <cstdio>
#include <vector>
#include <random>
namespace hpmn { void set_last_hip_error(int) {} }
int main() {
    const int B = 500, T = 1024, Tids = 1001, H = 64, D = 32, F = 2, E = 16;
    const long V = 3308019;
    float *emb, *wg, *wc, *bg, *bc, *xp, *xo; int *ids;
    hipMalloc(&emb, V * E * 4); hipMemset(emb, 0, V * E * 4);
    hipMalloc(&wg, (D + H) * 2 * H * 4); hipMalloc(&wc, (D + H) * H * 4); hipMalloc(&bg, 2 * H * 4); hipMalloc(&bc, H * 4);
    hipMemset(wg, 0, (D + H) * 2 * H * 4); hipMemset(wc, 0, (D + H) * H * 4); hipMemset(bg, 0, 2 * H * 4); hipMemset(bc, 0, H * 4);
    hipMalloc(&xp, (size_t)B * T * 3 * H * 4); hipMalloc(&xo, (size_t)B * T * D * 4);
    std::vector<int> h((size_t)B * Tids * F); std::mt19937 g(1);
    for (auto &v : h) v = g() % V;
    hipMalloc(&ids, h.size() * 4); hipMemcpy(ids, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    HpmnInputProj a = {};
    a.B = B; a.T = T; a.D = D; a.H = H; a.ids = ids; a.emb = emb; a.Tids = Tids; a.F = F; a.E = E; a.front_zero = 23;
    a.V = V; a.wg = wg; a.bg = bg; a.wc = wc; a.bc = bc; a.xp = xp; a.x_out = xo;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) hpmn::input_proj_dispatch(a, 0);
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) hpmn::input_proj_dispatch(a, 0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %.4f ms/launch\n", VARIANT, ms / 5);
    return 0;
}
'''


def proj_variants(src):
    out = {"p0 baseline (gather, L0)": src}
    st = """            float *dst = a.xp + (long)flat_row(tile_row(tile)) * N + n_base + 4 * p;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<float4 *>(dst + 32 * nt + 8 * g) ="""
    v1 = sub(src, st, st.replace("*reinterpret_cast<float4 *>(dst + 32 * nt + 8 * g) =",
                                 "if (acc[nt][4 * g] == 123.f) *reinterpret_cast<float4 *>(dst + 32 * nt + 8 * g) ="))
    out["p1 no xp stores"] = v1
    v2 = sub(src, "*reinterpret_cast<float4 *>(xo + 4 * i) = v;", "if (v.x == 123.f) *reinterpret_cast<float4 *>(xo + 4 * i) = v;")
    out["p2 no x_out stores"] = v2
    v3 = sub(src, "v[q] = *reinterpret_cast<const float4 *>(a.emb + (long)id[q] * a.E + (j - f * a.E));",
             "v[q] = make_float4((float)id[q], 0.f, 0.f, 0.f);")
    out["p3 no embedding row loads (ids only)"] = v3
    v4 = sub(src, "acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(wb[nt][4 * q + e], av[e], acc[nt], 0, 0, 0);\n        }\n        // C/D layout: lane (c, p), reg r -> D row",
             "acc[nt][e] += wb[nt][4 * q + e] * av[e];\n        }\n        // C/D layout: lane (c, p), reg r -> D row")
    out["p4 no MFMA (4 fmas instead)"] = v4
    v6 = sub(src, "float *dst = a.xp + (long)flat_row(tile_row(tile)) * N + n_base + 4 * p;",
             "float *dst = a.xp + ((long)tile * NS + ns) * 32 * NT * 32 + lane * 4;")
    v6 = sub(v6, "*reinterpret_cast<float4 *>(dst + 32 * nt + 8 * g) =", "*reinterpret_cast<float4 *>(dst + (nt * 4 + g) * 256) =")
    out["p6 xp stores fully contiguous (wrong layout)"] = v6
    v7 = sub(src, "float *dst = a.xp + (long)flat_row(tile_row(tile)) * N + n_base + 4 * p;",
             "float *dst = a.xp + ((long)tile * NS + ns) * 32 * NT * 32 + c * 96 + 4 * p;")
    out["p7 xp stores: 384-B row pieces packed (wrong layout)"] = v7
    v8 = sub(src, """                    *reinterpret_cast<float4 *>(dst + 32 * nt + 8 * g) =
                        make_float4(acc[nt][4 * g], acc[nt][4 * g + 1], acc[nt][4 * g + 2], acc[nt][4 * g + 3]);
        }
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            cur[q] = nxt[q];""", """                    { typedef float v4 __attribute__((ext_vector_type(4))); v4 vv = {acc[nt][4 * g], acc[nt][4 * g + 1], acc[nt][4 * g + 2], acc[nt][4 * g + 3]};
                      __builtin_nontemporal_store(vv, reinterpret_cast<v4 *>(dst + 32 * nt + 8 * g)); }
        }
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            cur[q] = nxt[q];""")
    out["p8 xp stores non-temporal"] = v8
    v5 = sub(v1, "*reinterpret_cast<float4 *>(xo + 4 * i) = v;", "if (v.x == 123.f) *reinterpret_cast<float4 *>(xo + 4 * i) = v;")
    v5 = sub(v5, "v[q] = *reinterpret_cast<const float4 *>(a.emb + (long)id[q] * a.E + (j - f * a.E));",
             "v[q] = make_float4((float)id[q], 0.f, 0.f, 0.f);")
    out["p5 only ids loads + MFMA"] = v5
    return out


def sub(s, old, new, count=1):
    assert old in s, old
    return s.replace(old, new, count)


def fwd_variants(src):
    out = {"f0 baseline": src}
    v1 = src
    for old in ("*hsp = h;", "gp[0] = r;", "gp[H] = u;", "gp[2 * H] = cc;"):
        v1 = sub(v1, old, "")
    v1 = sub(v1, "        *yp = h;\n", "")
    out["f1 no global stores"] = v1
    v2 = sub(v1, "v[i] = *reinterpret_cast<const f2 *>(xpb + (long)t * 3 * H + c_col[i]);",
             "v[i] = f2{(float)t * 1e-6f, 0.f};")
    out["f2 f1 + no prefetch loads"] = v2
    v3 = sub(v2, "const float r = sigmoid_scaled(", "const float r = 0.5f + 0.25f * (")
    v3 = sub(v3, "const float u = sigmoid_scaled(", "const float u = 0.5f + 0.25f * (")
    v3 = sub(v3, "const float cc = tanh_scaled(", "const float cc = 0.1f * (")
    out["f3 f2 + linear activations"] = v3
    v4 = sub(v3, "        rhb[lane] = r * h;\n        wave_sync();", "        rhb[lane] = r * h;")
    v4 = sub(v4, "        hb[lane] = h;\n        wave_sync();", "        hb[lane] = h;")
    out["f4 f3 + no fence between write and reads"] = v4
    v6 = src
    for old_, new_ in (("*hsp = h;", "__builtin_nontemporal_store(h, hsp);"), ("gp[0] = r;", "__builtin_nontemporal_store(r, gp);"),
                       ("gp[H] = u;", "__builtin_nontemporal_store(u, gp + H);"), ("gp[2 * H] = cc;", "__builtin_nontemporal_store(cc, gp + 2 * H);")):
        v6 = sub(v6, old_, new_)
    out["f6 non-temporal hs/gates stores"] = v6
    v5 = sub(src, "const float r = sigmoid_scaled(", "const float r = 0.5f + 0.25f * (")
    v5 = sub(v5, "const float u = sigmoid_scaled(", "const float u = 0.5f + 0.25f * (")
    v5 = sub(v5, "const float cc = tanh_scaled(", "const float cc = 0.1f * (")
    out["f5 linear activations only"] = v5
    return out


def main():
    src = open(os.path.join(CSRC, "gru_scan_fwd.hip")).read()
    lines = ["#!/bin/sh"]
    for name, text in fwd_variants(src).items():
        tag = name.split()[0]
        path = os.path.join(HERE, "ablate_%s.hip" % tag)
        with open(path, "w") as f:
            f.write("#include <hip/hip_runtime.h>\n#include \"hpmn_hip.h\"\n" + text +
                    "\n#define VARIANT \"%s\"\n" % name + MAIN_FWD)
        exe = os.path.join(HERE, "ablate_%s" % tag)
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-w",
                               "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-o", exe, path])
        os.remove(path)
        lines.append("./tools/micro/ablate_%s" % tag)
    srcb = open(os.path.join(CSRC, "gru_scan_bwd.hip")).read()
    for name, text in bwd_variants(srcb).items():
        tag = name.split()[0]
        path = os.path.join(HERE, "ablate_%s.hip" % tag)
        with open(path, "w") as f:
            f.write("#include <hip/hip_runtime.h>\n#include \"hpmn_hip.h\"\n" + text +
                    "\n#define VARIANT \"%s\"\n" % name + MAIN_BWD)
        exe = os.path.join(HERE, "ablate_%s" % tag)
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-w",
                               "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-o", exe, path])
        os.remove(path)
        lines.append("./tools/micro/ablate_%s" % tag)
    srcp = open(os.path.join(CSRC, "input_proj.hip")).read()
    for name, text in proj_variants(srcp).items():
        tag = name.split()[0]
        path = os.path.join(HERE, "ablate_%s.hip" % tag)
        with open(path, "w") as f:
            f.write("#include <hip/hip_runtime.h>\n#include \"hpmn_hip.h\"\n" + text +
                    "\n#define VARIANT \"%s\"\n" % name + MAIN_PROJ)
        exe = os.path.join(HERE, "ablate_%s" % tag)
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-w",
                               "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-o", exe, path])
        os.remove(path)
        lines.append("./tools/micro/ablate_%s" % tag)
    with open(os.path.join(HERE, "run_ablate.sh"), "w") as f:
        f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    sys.exit(main())
