// What does a workgroup barrier phase cost on gfx950?  NW waves, each iteration = [R x ds_read_b128] [V dependent
// VALU ops] [T exp+rcp pairs] [ds_write_b64] s_waitcnt lgkmcnt(0) s_barrier.  Prints cycles per iteration.
// Build/run on the GPU box: hipcc --offload-arch=gfx950 -O3 -o barrier_cost barrier_cost.hip && ./barrier_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int R, int V, int T, bool SPLIT>
__global__ void k(float *out, long long *cyc, int iters) {
    __shared__ __attribute__((aligned(16))) char lds[32768];
    const int tid = threadIdx.x, lane = tid & 63;
    float x = out[tid];
    for (int i = tid; i < 32768 / 4; i += blockDim.x) reinterpret_cast<float *>(lds)[i] = 0.001f * i;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        float4 acc = make_float4(0, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const float4 v = *reinterpret_cast<const float4 *>(lds + ((lane * 160 + r * 2560 + (it & 1) * 64) & 32767 & ~15));
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        x += acc.x + acc.y + acc.z + acc.w;
#pragma unroll
        for (int v = 0; v < V; ++v) x = fmaf(x, 1.0001f, 0.5f);
#pragma unroll
        for (int t = 0; t < T; ++t) x = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x));
        if (SPLIT) {
            _Float16 h = (_Float16)x;
            float rem = x - (float)h;
            _Float16 l = (_Float16)rem;
            x += (float)l;
        }
        *reinterpret_cast<float2 *>(lds + ((tid * 8) & 32767)) = make_float2(x, x);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    out[tid] = x;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int R, int V, int T, bool SPLIT>
int run(const char *name, float *out, long long *cyc) {
    for (int nw : {1, 4, 8, 12, 16}) {
        const int iters = 2000;
        hipLaunchKernelGGL((k<R, V, T, SPLIT>), dim3(32), dim3(nw * 64), 0, 0, out, cyc, iters);
        CK(hipDeviceSynchronize());
        long long h[32];
        CK(hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost));
        printf("%-44s waves %2d: %6.1f cycles / iteration\n", name, nw, (double)h[0] / iters);
    }
    return 0;
}

int main() {
    float *out; long long *cyc;
    CK(hipMalloc(&out, 4096 * 4)); CK(hipMalloc(&cyc, 32 * 8));
    CK(hipMemset(out, 0, 4096 * 4));
    run<0, 0, 0, false>("barrier + ds_write only", out, cyc);
    run<0, 8, 0, false>("8 dependent fma", out, cyc);
    run<0, 0, 4, false>("4 dependent exp+rcp pairs", out, cyc);
    run<4, 0, 0, false>("4 ds_read_b128", out, cyc);
    run<4, 8, 4, true>("4 reads + 8 fma + 4 exp/rcp + split", out, cyc);
    return 0;
}
