// Where do the waves of a 2-wave-per-workgroup launch land?  (SIMD sharing between the chain waves of two
// sequences on one CU would explain the B=500 vs B=250 gap of the scan kernels.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
#include <string>
template <int NW>
__global__ __launch_bounds__(NW * 64, 1) void where_kernel(unsigned *out, int spin) {
    __shared__ float pad[3500];
    const int wave = threadIdx.x >> 6;
    unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);
    unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    long long t0 = __builtin_amdgcn_s_memtime();
    float acc = threadIdx.x;
    while (__builtin_amdgcn_s_memtime() - t0 < spin) { acc = acc * 1.0001f + 1.f; pad[threadIdx.x] = acc; }
    if ((threadIdx.x & 63) == 0) { out[(blockIdx.x * NW + wave) * 2] = hw; out[(blockIdx.x * NW + wave) * 2 + 1] = xcc; }
    if (acc == 123.f) out[0] = 0;
}
template <int NW>
void run(int B) {
    unsigned *d; hipMalloc(&d, B * NW * 8);
    hipLaunchKernelGGL(where_kernel<NW>, dim3(B), dim3(NW * 64), 0, 0, d, 5000);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(where_kernel<NW>, dim3(B), dim3(NW * 64), 0, 0, d, 20000);
    std::vector<unsigned> h(B * NW * 2); hipMemcpy(h.data(), d, B * NW * 8, hipMemcpyDeviceToHost);
    printf("---- %d workgroups of %d waves\n", B, NW);
    std::map<unsigned, std::string> cu; std::map<unsigned, std::vector<int>> simds;
    for (int b = 0; b < B; ++b) for (int w = 0; w < NW; ++w) {
        unsigned hw = h[(b * NW + w) * 2], xcc = h[(b * NW + w) * 2 + 1] & 0xf;
        unsigned simd = (hw >> 4) & 3, cuid = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        unsigned key = (xcc << 12) | (se << 8) | (sh << 4) | cuid;
        char buf[64]; snprintf(buf, sizeof buf, " b%d.w%d@simd%u", b, w, simd);
        cu[key] += buf; simds[key].push_back(simd);
    }
    int n = 0, share_chain = 0, share_any = 0, two = 0;
    for (auto &kv : cu) {
        if (n++ < 8) printf("cu %05x:%s\n", kv.first, kv.second.c_str());
        auto &v = simds[kv.first];
        int cnt[4] = {0, 0, 0, 0};
        for (int x : v) cnt[x]++;
        int mx = 0; for (int s = 0; s < 4; ++s) mx = cnt[s] > mx ? cnt[s] : mx;
        if (v.size() >= 4) ++two;
        share_any += (int)v.size() <= 4 && mx > 1;
    }
    printf("%d CUs used, %d with >= 4 waves; %d CUs with <= 4 waves have two of them on one SIMD\n", (int)cu.size(), two, share_any);
    hipFree(d);
}
int main(int argc, char **argv) {
    run<2>(500); run<4>(250); run<6>(250); run<8>(250); run<6>(320);
    return 0;
}
