// The read path's dense products alone (forward, transposed), one workgroup or many: cycles per layer.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I hpmn_amd/csrc tools/micro/dense_micro.hip -o tools/micro/dense_micro
#include "../../hpmn_amd/csrc/read_path.hip"
#include <cstdio>
#include <vector>
namespace hpmn { void set_last_hip_error(int) {} }
using namespace hpmn;

__global__ __launch_bounds__(RT) void k(const float *W, const float *bias, const float *Xg, float *out, unsigned long long *clk,
                                        int R, int I, int N) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int ldx = I + PADF, ldy = N + PADF;
    float *X = sm, *Y = X + 32 * ldx;
    for (int o = threadIdx.x; o < R * I; o += RT) X[(o / I) * ldx + o % I] = Xg[o];
    __syncthreads();
    unsigned long long t3 = clock64();
    dense_fwd_g<1>(X, ldx, R, I, W, bias, N, Y, ldy);
    __syncthreads();
    unsigned long long t4 = clock64();
    dense_bwd_x_g<false>(Y, ldy, R, N, W, I, X, ldx);
    __syncthreads();
    unsigned long long t5 = clock64();
    for (int o = threadIdx.x; o < R * N; o += RT) out[blockIdx.x * R * N + o] = Y[(o / N) * ldy + o % N] + X[o % I];
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t4 - t3; clk[1] = t5 - t4; }
}

int main() {
    const int shapes[][3] = {{14, 256, 80}, {14, 80, 40}, {2, 96, 200}, {2, 200, 80}, {2, 64, 64}, {8, 128, 80}};
    for (auto &sh : shapes) {
        const int R = sh[0], I = sh[1], N = sh[2];
        std::vector<float> hw(I * N), hx(R * I), hb(N, 0.1f);
        for (size_t i = 0; i < hw.size(); ++i) hw[i] = ((int)(i * 37 % 101) - 50) * 0.002f;
        for (size_t i = 0; i < hx.size(); ++i) hx[i] = ((int)(i * 17 % 23) - 11) * 0.05f;
        float *W, *b, *X, *out; unsigned long long *clk;
        hipMalloc(&W, hw.size() * 4); hipMalloc(&b, N * 4); hipMalloc(&X, hx.size() * 4); hipMalloc(&out, 1024 * R * N * 4); hipMalloc(&clk, 64);
        hipMemcpy(W, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(b, hb.data(), N * 4, hipMemcpyHostToDevice);
        hipMemcpy(X, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
        const size_t lds = (32 * (I + PADF) + 32 * (N + PADF)) * 4;
        hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        for (int grid : {1, 250}) {
            for (int rep = 0; rep < 2; ++rep) k<<<grid, RT, lds>>>(W, b, X, out, clk, R, I, N);
            hipDeviceSynchronize();
            unsigned long long c[2]; hipMemcpy(c, clk, sizeof(c), hipMemcpyDeviceToHost);
            printf("R=%2d I=%3d N=%3d grid %3d: forward %6llu | transposed %6llu cycles (matrix instructions alone: %d / %d)\n", R, I, N, grid,
                   c[0], c[1], ((N + 31) / 32 + 3) / 4 * (I / 2) * 64, ((I + 31) / 32 + 3) / 4 * (N / 2) * 64);
        }
        hipFree(W); hipFree(b); hipFree(X); hipFree(out); hipFree(clk);
    }
    return 0;
}
