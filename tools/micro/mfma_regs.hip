// Does a dependent v_mfma_f32_32x32x2_f32 chain care which registers hold its operands?   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHAIN(NAME, A, B, ACC)                                                                              \
    __global__ __launch_bounds__(256) void NAME(float *out, unsigned long long *clk) {                      \
        unsigned long long t0, t1;                                                                          \
        float r;                                                                                            \
        asm volatile("v_mov_b32 " A ", 1.0\n v_mov_b32 " B ", 0x3f800347\n"                                 \
                     "s_memtime %0\n s_waitcnt lgkmcnt(0)\n"                                                \
                     : "=s"(t0) : : A, B);                                                                  \
        for (int i = 0; i < 32; ++i)                                                                        \
            asm volatile("v_mfma_f32_32x32x2_f32 " ACC ", " A ", " B ", " ACC "\n"                          \
                         "v_mfma_f32_32x32x2_f32 " ACC ", " A ", " B ", " ACC "\n"                          \
                         "v_mfma_f32_32x32x2_f32 " ACC ", " A ", " B ", " ACC "\n"                          \
                         "v_mfma_f32_32x32x2_f32 " ACC ", " A ", " B ", " ACC "\n"                          \
                         "v_mfma_f32_32x32x2_f32 " ACC ", " A ", " B ", " ACC "\n"                          \
                         "v_mfma_f32_32x32x2_f32 " ACC ", " A ", " B ", " ACC "\n"                          \
                         "v_mfma_f32_32x32x2_f32 " ACC ", " A ", " B ", " ACC "\n"                          \
                         "v_mfma_f32_32x32x2_f32 " ACC ", " A ", " B ", " ACC "\n" ::: "memory");           \
        asm volatile("s_nop 15\n s_nop 15\n s_memtime %0\n s_waitcnt lgkmcnt(0)\n" : "=s"(t1));            \
        if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;                                          \
        out[threadIdx.x] = 0.f;                                                                             \
    }
CHAIN(k_a0_v1_v0, "v1", "v0", "a[0:15]")
CHAIN(k_a16_v33_v0, "v33", "v0", "a[16:31]")
CHAIN(k_a16_v33_v151, "v33", "v151", "a[16:31]")
CHAIN(k_a16_v33_v150, "v33", "v150", "a[16:31]")
CHAIN(k_a32_v33_v151, "v33", "v151", "a[32:47]")
CHAIN(k_v_v1_v0, "v1", "v0", "v[16:31]")
CHAIN(k_v_v33_v151, "v33", "v151", "v[64:79]")
int main() {
    float *out; unsigned long long *clk;
    (void)hipMalloc(&out, 4096); (void)hipMalloc(&clk, 64);
#define RUN(K) { K<<<1, 256>>>(out, clk); (void)hipDeviceSynchronize(); unsigned long long c; (void)hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost); printf("%-18s %.1f cycles per instruction\n", #K, (double)c / 256); }
    RUN(k_a0_v1_v0) RUN(k_a16_v33_v0) RUN(k_a16_v33_v151) RUN(k_a16_v33_v150) RUN(k_a32_v33_v151) RUN(k_v_v1_v0) RUN(k_v_v33_v151)
    return 0;
}
