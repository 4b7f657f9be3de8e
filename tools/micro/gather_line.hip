// Does a random 64-byte table row HAVE to cost a 128-byte fetch?  (north_star: gather at >= 40 % of HBM)
// The in-place gather probe with its row load issued five ways; run plain for timings, and under
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d D -- ./gather_line
// for the bytes each variant really pulls per row (x2 gfx950 correction applies, see profiles/r04_gather_pmc.json: the
// sequential-id calibration reads 34 B raw per 68 B true).
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/gather_line tools/micro/gather_line.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float v4f __attribute__((ext_vector_type(4)));

template <int MODE> __device__ __forceinline__ v4f load_row(const v4f *p) {
    v4f v;
    if constexpr (MODE == 0) v = *p;
    else if constexpr (MODE == 1) v = __builtin_nontemporal_load(p);
    else if constexpr (MODE == 2) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else if constexpr (MODE == 3) asm volatile("global_load_dwordx4 %0, %1, off nt\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else if constexpr (MODE == 4) asm volatile("global_load_dwordx4 %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1 nt\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

// one thread per (lookup, float4 of the row): 4 adjacent lanes move one 64-byte row; U lookups in flight per thread (modes 0/1;
// the asm variants wait per load, so they run with more waves instead)
template <int MODE, int U>
__global__ __launch_bounds__(256) void gather_sum_variant(const int *__restrict__ ids, const float *__restrict__ emb,
                                                          float *__restrict__ out, long n) {
    const long stride = (long)gridDim.x * blockDim.x / 4;
    const int e4 = threadIdx.x & 3;
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    for (long r = ((long)blockIdx.x * blockDim.x + threadIdx.x) / 4; r < n; r += U * stride) {
        long id[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { const long rr = r + u * stride; id[u] = rr < n ? ids[rr] : -1; }
        v4f v[U];
#pragma unroll
        for (int u = 0; u < U; ++u)
            v[u] = id[u] >= 0 ? load_row<MODE>(reinterpret_cast<const v4f *>(emb) + id[u] * 4 + e4) : v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u];
    }
    float s = acc[0] + acc[1] + acc[2] + acc[3];
    if (s == 12345.678f) out[0] = s;          // (keeps the loads alive; never true)
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("hip error %d at %d\n", (int)e_, __LINE__); exit(1); } } while (0)

template <int MODE, int U> void run(const char *name, const std::vector<int *> &ids, const float *emb, float *out, long n) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int blocks = 256 * 16;
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((gather_sum_variant<MODE, U>), dim3(blocks), dim3(256), 0, 0, ids[i % ids.size()], emb, out, n);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    const int it = 8;
    for (int i = 0; i < it; ++i) hipLaunchKernelGGL((gather_sum_variant<MODE, U>), dim3(blocks), dim3(256), 0, 0, ids[i % ids.size()], emb, out, n);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= it;
    printf("%-28s %7.1f us  %6.0f GB/s algorithmic (68 B/lookup) = %.3f of 8 TB/s\n", name, ms * 1e3, n * 68.0 / ms / 1e6, n * 68.0 / ms / 1e6 / 8000);
}

int main() {
    const long V = 64L * 1024 * 1024, n = 500L * 1001 * 2;
    float *emb, *out;
    CK(hipMalloc(&emb, V * 64)); CK(hipMalloc(&out, 64));
    CK(hipMemset(emb, 0, V * 64));
    std::vector<int *> ids(4);
    std::vector<int> h(n);
    unsigned long long s = 88172645463325252ULL;
    for (auto &p : ids) {
        for (long i = 0; i < n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = (int)(s % (unsigned long long)V); }
        CK(hipMalloc(&p, n * 4)); CK(hipMemcpy(p, h.data(), n * 4, hipMemcpyHostToDevice));
    }
    run<0, 8>("plain x8", ids, emb, out, n);
    run<1, 8>("nontemporal x8", ids, emb, out, n);
    run<0, 1>("plain x1", ids, emb, out, n);
    run<2, 1>("asm sc0 sc1 x1", ids, emb, out, n);
    run<3, 1>("asm nt x1", ids, emb, out, n);
    run<4, 1>("asm sc1 x1", ids, emb, out, n);
    run<5, 1>("asm sc0 sc1 nt x1", ids, emb, out, n);
    return 0;
}
