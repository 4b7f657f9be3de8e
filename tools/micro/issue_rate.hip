// What a clock64() tick is worth, and what one wave alone on a SIMD gets: dependent v_fma chain, dependent
// v_mfma_f32_32x32x2_f32 chain, LDS read round trips, L2-hit load round trips.   hipcc --offload-arch=gfx950 -O3 issue_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void k(float *out, const float *in, unsigned long long *clk, int n) {
    __shared__ float lds[4096];
    const int tid = threadIdx.x;
    for (int i = tid; i < 4096; i += 256) lds[i] = in[i];
    __syncthreads();
    unsigned long long t[6];
    float a = in[tid], b = 1.0001f;
    t[0] = clock64();
#pragma unroll 16
    for (int i = 0; i < n; ++i) a = fmaf(a, b, 0.5f);          // n dependent VALU
    asm volatile("" : "+v"(a));
    t[1] = clock64();
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = a;
#pragma unroll 8
    for (int i = 0; i < n / 4; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);    // n/4 dependent MFMA
    a += acc[0] + acc[7];
    asm volatile("" : "+v"(a));
    t[2] = clock64();
    int idx = tid;
#pragma unroll 4
    for (int i = 0; i < n / 4; ++i) idx = (int)lds[(idx & 4095)] & 4095;       // n/4 dependent LDS round trips
    asm volatile("" : "+v"(idx));
    t[3] = clock64();
#pragma unroll 4
    for (int i = 0; i < n / 16; ++i) idx = (int)in[(idx & 4095)] & 4095;       // n/16 dependent global (L2/L1 hit) round trips
    asm volatile("" : "+v"(idx));
    t[4] = clock64();
    out[blockIdx.x * 256 + tid] = a + idx;
    if (tid == 0 && blockIdx.x == 0) for (int i = 0; i < 5; ++i) clk[i] = t[i];
}

int main() {
    float *in, *out; unsigned long long *clk;
    hipMalloc(&in, 4096 * 4); hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&clk, 64);
    float h[4096]; for (int i = 0; i < 4096; ++i) h[i] = (float)((i * 37 + 11) & 4095);
    hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    const int n = 4096;
    for (int grid : {1, 256, 1024}) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        k<<<grid, 256>>>(out, in, clk, n);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        k<<<grid, 256>>>(out, in, clk, n);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long c[5]; hipMemcpy(c, clk, sizeof(c), hipMemcpyDeviceToHost);
        printf("grid %4d: launch %.1f us, ticks total %llu (%.2f ticks/ns) | per op: valu %.1f  mfma32x32x2 %.1f  lds round trip %.1f  global round trip %.1f ticks\n",
               grid, ms * 1e3, c[4] - c[0], (double)(c[4] - c[0]) / (ms * 1e6), (double)(c[1] - c[0]) / n,
               (double)(c[2] - c[1]) / (n / 4), (double)(c[3] - c[2]) / (n / 4), (double)(c[4] - c[3]) / (n / 16));
    }
    return 0;
}
