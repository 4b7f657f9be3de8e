// Do fp32 MFMAs and independent VALU instructions of the SAME wave overlap?  And of two waves on one SIMD?
// (the reverse scan's input gradient as a concurrent MFMA role slowed the scan by the MFMA pipe time.)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
template <int MODE, bool F16 = false>   // 0: MFMA only, 1: VALU only, 2: both interleaved in one wave, 3: wave 0 MFMA + wave 4 VALU (same SIMD)
__global__ __launch_bounds__(512, 1) void k(float *out, int iters, long long *cyc) {
    const int wave = threadIdx.x >> 6;
    f4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    float v[8];
    h8 ha, hb;
    for (int i = 0; i < 8; ++i) { ha[i] = (_Float16)(a + i); hb[i] = (_Float16)(b + i); }
    for (int i = 0; i < 8; ++i) v[i] = a + i;
    const bool do_m = MODE == 0 || MODE == 2 || (MODE == 3 && wave == 0);
    const bool do_v = MODE == 1 || MODE == 2 || (MODE == 3 && wave == 4);
    if (MODE == 3 && wave != 0 && wave != 4) return;
    if (MODE != 3 && wave != 0) return;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (do_m) {
                if (F16) acc[r & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, acc[r & 3], 0, 0, 0);
                else     acc[r & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[r & 3], 0, 0, 0);
            }
            if (do_v) {
#pragma unroll
                for (int q = 0; q < 5; ++q) v[(r + q) & 7] = __builtin_fmaf(v[(r + q) & 7], b, a);
            }
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += v[i];
    for (int i = 0; i < 4; ++i) s += acc[i].x + acc[i].y;
    out[threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[wave] = t1 - t0;
}
template <int MODE, bool F16 = false> void run(const char *name) {
    float *o; long long *c; hipMalloc(&o, 4096); hipMalloc(&c, 64); hipMemset(c, 0, 64);
    const int iters = 2000;
    hipLaunchKernelGGL((k<MODE, F16>), dim3(1), dim3(512), 0, 0, o, iters, c);
    hipDeviceSynchronize();
    long long h[8]; hipMemcpy(h, c, 64, hipMemcpyDeviceToHost);
    printf("%-46s wave0 %.1f  wave4 %.1f  (s_memtime ticks of 10 ns per iteration of 8 MFMA / 40 FMA)\n", name,
           (double)h[0] / iters, (double)h[4] / iters);
}
int main() {
    run<0>("8 x mfma_f32_16x16x4 only");
    run<1>("40 x v_fma only");
    run<2>("both interleaved in one wave");
    run<3>("wave 0: MFMA, wave 4 (same SIMD): FMA");
    run<0, true>("8 x mfma_f32_16x16x32_f16 only");
    run<2, true>("f16 MFMA + FMA interleaved in one wave");
    run<3, true>("wave 0: f16 MFMA, wave 4 (same SIMD): FMA");
    return 0;
}
