// Instruction-cost micro-benchmarks for the scan kernels on gfx950 (one wave per workgroup).
// Build: hipcc --offload-arch=gfx950 -O3 -o micro micro.hip ; run on the GPU box.
// Prints: permlane swap semantics, and cycles per instruction (s_memtime) for the candidate
// inner-loop instruction mixes of the GRU recurrence.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int N>
__device__ __forceinline__ float bcast_builtin(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x150 + N, 0xf, 0xf, true));
}

__global__ void k_semantics(unsigned* out) {
    unsigned x = threadIdx.x;
    auto r32 = __builtin_amdgcn_permlane32_swap(x, x, false, false);
    auto r16a = __builtin_amdgcn_permlane16_swap(r32[0], r32[0], false, false);
    auto r16b = __builtin_amdgcn_permlane16_swap(r32[1], r32[1], false, false);
    out[threadIdx.x] = r32[0];
    out[64 + threadIdx.x] = r32[1];
    out[128 + threadIdx.x] = r16a[0];
    out[192 + threadIdx.x] = r16a[1];
    out[256 + threadIdx.x] = r16b[0];
    out[320 + threadIdx.x] = r16b[1];
    float f = (float)threadIdx.x;
    out[384 + threadIdx.x] = (unsigned)bcast_builtin<5>(f);
}

#define REP16(X) X X X X X X X X X X X X X X X X
#define REP64(X) REP16(X) REP16(X) REP16(X) REP16(X)

// variant 0: plain v_fmac (4 independent accumulators)
// variant 1: v_fmac_dpp row_newbcast (asm)
// variant 2: v_pk_fma_f32 (2 accumulators pairs)
// variant 3: v_mov_dpp + v_fmac (what the builtin gives without folding)
// variant 4: v_readlane + v_fmac with SGPR
template <int V>
__global__ void k_tput(float* out, long long* cyc, const float* in, int iters) {
    float a0 = in[threadIdx.x], a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    float w0 = in[64 + threadIdx.x], w1 = w0 * 2, w2 = w0 * 3, w3 = w0 * 4;
    float h = in[128 + threadIdx.x];
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {a0, a1}, p1 = {a2, a3}, pw = {w0, w1}, ph = {h, h};
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if constexpr (V == 0) {
            REP16(asm volatile("v_fmac_f32 %0, %4, %5\n v_fmac_f32 %1, %4, %6\n v_fmac_f32 %2, %4, %7\n v_fmac_f32 %3, %4, %8"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(h), "v"(w0), "v"(w1), "v"(w2), "v"(w3));)
        } else if constexpr (V == 1) {
            REP16(asm volatile("v_fmac_f32_dpp %0, %4, %5 row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
                               "v_fmac_f32_dpp %1, %4, %6 row_newbcast:2 row_mask:0xf bank_mask:0xf\n"
                               "v_fmac_f32_dpp %2, %4, %7 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
                               "v_fmac_f32_dpp %3, %4, %8 row_newbcast:4 row_mask:0xf bank_mask:0xf"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(h), "v"(w0), "v"(w1), "v"(w2), "v"(w3));)
        } else if constexpr (V == 2) {
            REP16(asm volatile("v_pk_fma_f32 %0, %2, %3, %0\n v_pk_fma_f32 %1, %2, %3, %1\n"
                               "v_pk_fma_f32 %0, %2, %3, %0\n v_pk_fma_f32 %1, %2, %3, %1"
                               : "+v"(p0), "+v"(p1) : "v"(ph), "v"(pw));)
        } else if constexpr (V == 3) {
            float t1, t2, t3, t4;
            REP16(asm volatile("v_mov_b32_dpp %4, %8 row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
                               "v_mov_b32_dpp %5, %8 row_newbcast:2 row_mask:0xf bank_mask:0xf\n"
                               "v_mov_b32_dpp %6, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
                               "v_mov_b32_dpp %7, %8 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
                               "v_fmac_f32 %0, %4, %9\n v_fmac_f32 %1, %5, %10\n v_fmac_f32 %2, %6, %11\n v_fmac_f32 %3, %7, %12"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4)
                               : "v"(h), "v"(w0), "v"(w1), "v"(w2), "v"(w3));)
        } else if constexpr (V == 4) {
            REP16(asm volatile("v_readlane_b32 s20, %4, 1\n v_readlane_b32 s21, %4, 2\n"
                               "v_readlane_b32 s22, %4, 3\n v_readlane_b32 s23, %4, 4\n"
                               "v_fmac_f32 %0, s20, %5\n v_fmac_f32 %1, s21, %6\n v_fmac_f32 %2, s22, %7\n v_fmac_f32 %3, s23, %8"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(h), "v"(w0), "v"(w1), "v"(w2), "v"(w3)
                               : "s20", "s21", "s22", "s23");)
        } else if constexpr (V == 5) {   // dependent chain: single accumulator plain fmac
            REP64(asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a0) : "v"(h), "v"(w0));)
        } else if constexpr (V == 6) {   // dependent chain: single accumulator dpp fmac
            REP64(asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:1 row_mask:0xf bank_mask:0xf" : "+v"(a0) : "v"(h), "v"(w0));)
        } else if constexpr (V == 7) {   // 2 chains dpp
            REP16(asm volatile("v_fmac_f32_dpp %0, %2, %3 row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
                               "v_fmac_f32_dpp %1, %2, %4 row_newbcast:2 row_mask:0xf bank_mask:0xf\n"
                               "v_fmac_f32_dpp %0, %2, %3 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
                               "v_fmac_f32_dpp %1, %2, %4 row_newbcast:4 row_mask:0xf bank_mask:0xf"
                               : "+v"(a0), "+v"(a1) : "v"(h), "v"(w0), "v"(w1));)
        } else if constexpr (V == 8) {   // pk_fma pairs separated by a (satisfied) s_waitcnt
            REP16(asm volatile("v_pk_fma_f32 %0, %2, %3, %0\n v_pk_fma_f32 %1, %2, %3, %1\n s_waitcnt lgkmcnt(0)\n"
                               "v_pk_fma_f32 %0, %2, %3, %0\n v_pk_fma_f32 %1, %2, %3, %1\n s_waitcnt lgkmcnt(0)"
                               : "+v"(p0), "+v"(p1) : "v"(ph), "v"(pw));)
        } else if constexpr (V == 9) {   // pk_fma pairs separated by s_nop 0
            REP16(asm volatile("v_pk_fma_f32 %0, %2, %3, %0\n v_pk_fma_f32 %1, %2, %3, %1\n s_nop 0\n"
                               "v_pk_fma_f32 %0, %2, %3, %0\n v_pk_fma_f32 %1, %2, %3, %1\n s_nop 0"
                               : "+v"(p0), "+v"(p1) : "v"(ph), "v"(pw));)
        } else if constexpr (V == 10) {  // 6 pk_fma per wave-uniform ds_read_b128 (the scan's mix), waits every read
            typedef float f4 __attribute__((ext_vector_type(4)));
            f4 r0;
            REP16(asm volatile("ds_read_b128 %2, %5\n"
                               "v_pk_fma_f32 %0, %3, %4, %0\n v_pk_fma_f32 %1, %3, %4, %1\n"
                               "v_pk_fma_f32 %0, %3, %4, %0\n v_pk_fma_f32 %1, %3, %4, %1\n"
                               "v_pk_fma_f32 %0, %3, %4, %0\n v_pk_fma_f32 %1, %3, %4, %1\n s_waitcnt lgkmcnt(0)"
                               : "+v"(p0), "+v"(p1), "=&v"(r0) : "v"(ph), "v"(pw), "v"(0));)
        } else if constexpr (V == 11) {  // pk_fma with an SGPR pair as the broadcast operand
            REP16(asm volatile("v_pk_fma_f32 %0, s[20:21], %2, %0\n v_pk_fma_f32 %1, s[22:23], %2, %1\n"
                               "v_pk_fma_f32 %0, s[20:21], %2, %0\n v_pk_fma_f32 %1, s[22:23], %2, %1"
                               : "+v"(p0), "+v"(p1) : "v"(pw) : "s20", "s21", "s22", "s23");)
        } else if constexpr (V == 12) {  // 2 waves' worth of independent pk_fma chains (4 accumulators)
            f2 p2 = p0, p3 = p1;
            REP16(asm volatile("v_pk_fma_f32 %0, %4, %5, %0\n v_pk_fma_f32 %1, %4, %5, %1\n"
                               "v_pk_fma_f32 %2, %4, %5, %2\n v_pk_fma_f32 %3, %4, %5, %3"
                               : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(ph), "v"(pw));)
            p0 += p2; p1 += p3;
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    if constexpr (V == 2 || V >= 8) { a0 = p0.x + p0.y; a1 = p1.x + p1.y; }
    out[threadIdx.x + 64 * blockIdx.x] = a0 + a1 + a2 + a3;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// latency probes: a dependent chain of exchange primitives
template <int V>
__global__ void k_lat(float* out, long long* cyc, const float* in, int iters) {
    __shared__ __attribute__((aligned(16))) float buf[64];
    float v = in[threadIdx.x];
    const int lane = threadIdx.x;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if constexpr (V == 0) {          // LDS write -> b128 broadcast read (4 values) -> combine
            buf[lane] = v;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            float4 q = reinterpret_cast<const float4*>(buf)[(lane & 15)];
            v = q.x + q.y * 0.5f + q.z * 0.25f + q.w * 0.125f;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        } else if constexpr (V == 1) {   // 4 x ds_bpermute
            int x = __builtin_bit_cast(int, v);
            int p = lane & 15;
            float q0 = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(p * 4, x));
            float q1 = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((16 + p) * 4, x));
            float q2 = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((32 + p) * 4, x));
            float q3 = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((48 + p) * 4, x));
            v = q0 + q1 * 0.5f + q2 * 0.25f + q3 * 0.125f;
        } else if constexpr (V == 2) {   // permlane32_swap + 2 x permlane16_swap
            unsigned x = __builtin_bit_cast(unsigned, v);
            auto r32 = __builtin_amdgcn_permlane32_swap(x, x, false, false);
            auto ra = __builtin_amdgcn_permlane16_swap(r32[0], r32[0], false, false);
            auto rb = __builtin_amdgcn_permlane16_swap(r32[1], r32[1], false, false);
            v = __builtin_bit_cast(float, ra[0]) + __builtin_bit_cast(float, ra[1]) * 0.5f +
                __builtin_bit_cast(float, rb[0]) * 0.25f + __builtin_bit_cast(float, rb[1]) * 0.125f;
        } else if constexpr (V == 3) {   // sigmoid chain: exp + rcp
            v = __builtin_amdgcn_rcpf(1.0f + __expf(-v));
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x + 64 * blockIdx.x] = v;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// 4-wave workgroup probes (the 4-waves-per-sequence scan): V=0 partial-sum exchange through LDS with a
// barrier; V=1 the same + rep_row + 32 dpp fmacs (the gates half of a step); V=2 a full synthetic step
template <int V>
__global__ __launch_bounds__(256) void k_wg4(float* out, long long* cyc, const float* in, int iters) {
    __shared__ __attribute__((aligned(16))) float pg[2][64][4];
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    float h = in[l];
    float wr[16], wu[16];
    for (int j = 0; j < 16; ++j) { wr[j] = in[64 + j] * (1 + w); wu[j] = in[128 + j]; }
    for (int j = 0; j < 16; ++j) { asm volatile("" : "+v"(wr[j])); asm volatile("" : "+v"(wu[j])); }
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        float a = h, b2 = h;
        if constexpr (V >= 1) {
            unsigned x = __builtin_bit_cast(unsigned, h);
            auto r32 = __builtin_amdgcn_permlane32_swap(x, x, false, false);
            unsigned src = (w & 2) ? r32[1] : r32[0];
            auto r16 = __builtin_amdgcn_permlane16_swap(src, src, false, false);
            float q = __builtin_bit_cast(float, (w & 1) ? r16[1] : r16[0]);
            float a0 = 0, a1 = 0, c0 = 0, c1 = 0;
#define D16(A0, A1, W) asm("s_nop 1\n" \
            "v_fmac_f32_dpp %0, %2, %3 row_newbcast:0 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %1, %2, %4 row_newbcast:1 row_mask:0xf bank_mask:0xf\n" \
            "v_fmac_f32_dpp %0, %2, %5 row_newbcast:2 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %1, %2, %6 row_newbcast:3 row_mask:0xf bank_mask:0xf\n" \
            "v_fmac_f32_dpp %0, %2, %7 row_newbcast:4 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %1, %2, %8 row_newbcast:5 row_mask:0xf bank_mask:0xf\n" \
            "v_fmac_f32_dpp %0, %2, %9 row_newbcast:6 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %1, %2, %10 row_newbcast:7 row_mask:0xf bank_mask:0xf\n" \
            "v_fmac_f32_dpp %0, %2, %11 row_newbcast:8 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %1, %2, %12 row_newbcast:9 row_mask:0xf bank_mask:0xf\n" \
            "v_fmac_f32_dpp %0, %2, %13 row_newbcast:10 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %1, %2, %14 row_newbcast:11 row_mask:0xf bank_mask:0xf\n" \
            "v_fmac_f32_dpp %0, %2, %15 row_newbcast:12 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %1, %2, %16 row_newbcast:13 row_mask:0xf bank_mask:0xf\n" \
            "v_fmac_f32_dpp %0, %2, %17 row_newbcast:14 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %1, %2, %18 row_newbcast:15 row_mask:0xf bank_mask:0xf\n" \
            : "+v"(A0), "+v"(A1) : "v"(q), "v"(W[0]), "v"(W[1]), "v"(W[2]), "v"(W[3]), "v"(W[4]), "v"(W[5]), "v"(W[6]), "v"(W[7]), \
              "v"(W[8]), "v"(W[9]), "v"(W[10]), "v"(W[11]), "v"(W[12]), "v"(W[13]), "v"(W[14]), "v"(W[15]))
            D16(a0, a1, wr);
            D16(c0, c1, wu);
            a = a0 + a1; b2 = c0 + c1;
        }
        pg[0][l][w] = a;
        pg[1][l][w] = b2;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const float4 sr = *reinterpret_cast<const float4*>(pg[0][l]);
        const float4 su = *reinterpret_cast<const float4*>(pg[1][l]);
        float r = (sr.x + sr.y) + (sr.z + sr.w), u = (su.x + su.y) + (su.z + su.w);
        if constexpr (V >= 2) {
            r = __builtin_amdgcn_rcpf(1.0f + __expf(-r));
            u = __builtin_amdgcn_rcpf(1.0f + __expf(-u));
        }
        h = r * 0.25f + u * 0.125f;
        // second barrier of the step (candidate partials)
        if constexpr (V >= 2) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x + 256 * blockIdx.x] = h;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <typename F>
static double run(F launch, long long* d_cyc, int blocks) {
    launch();
    hipDeviceSynchronize();
    launch();
    hipDeviceSynchronize();
    std::vector<long long> c(blocks);
    hipMemcpy(c.data(), d_cyc, blocks * sizeof(long long), hipMemcpyDeviceToHost);
    double s = 0;
    for (auto v : c) s += (double)v;
    return s / blocks;
}

int main() {
    unsigned* d_sem;
    CK(hipMalloc(&d_sem, 448 * 4));
    hipLaunchKernelGGL(k_semantics, dim3(1), dim3(64), 0, 0, d_sem);
    CK(hipDeviceSynchronize());
    std::vector<unsigned> s(448);
    CK(hipMemcpy(s.data(), d_sem, 448 * 4, hipMemcpyDeviceToHost));
    const char* names[7] = {"perm32[0]", "perm32[1]", "p16(a)[0]", "p16(a)[1]", "p16(b)[0]", "p16(b)[1]", "newbcast5"};
    for (int r = 0; r < 7; ++r) {
        printf("%s:", names[r]);
        for (int i = 0; i < 64; i += 4) printf(" %u", s[r * 64 + i]);
        printf("\n");
    }
    float *d_in, *d_out;
    long long* d_cyc;
    const int blocks = 256;   // 1 wave per CU
    CK(hipMalloc(&d_in, 4096));
    CK(hipMalloc(&d_out, 256 * 512 * 4 * 2));
    CK(hipMalloc(&d_cyc, 512 * 8 * 8));
    std::vector<float> hin(1024, 0.001f);
    CK(hipMemcpy(d_in, hin.data(), 4096, hipMemcpyHostToDevice));
    const int iters = 2000;
    // s_memtime ticks at a fixed 100 MHz reference on gfx9: report both raw ticks and instr/tick
    double c;
#define TP(V, n_instr, label) c = run([&] { hipLaunchKernelGGL((k_tput<V>), dim3(blocks), dim3(64), 0, 0, d_out, d_cyc, d_in, iters); }, d_cyc, blocks); \
    printf("tput %-28s: %.1f ticks total, %.4f ticks/instr\n", label, c, c / ((double)iters * n_instr));
    TP(0, 64, "v_fmac x4 indep");
    TP(1, 64, "v_fmac_dpp x4 indep");
    TP(2, 64, "v_pk_fma_f32 x2 indep");
    TP(3, 128, "mov_dpp+fmac (per instr)");
    TP(4, 128, "readlane+fmac(sgpr) (per instr)");
    TP(5, 64, "v_fmac dependent chain");
    TP(6, 64, "v_fmac_dpp dependent chain");
    TP(7, 64, "v_fmac_dpp 2 chains");
    TP(8, 64, "pk_fma x2 + s_waitcnt (per pk)");
    TP(9, 64, "pk_fma x2 + s_nop (per pk)");
    TP(10, 96, "6 pk_fma + ds_read_b128 + wait (per pk)");
    TP(11, 64, "pk_fma sgpr-pair operand");
    TP(12, 64, "pk_fma x4 indep");
    // blocks = 1024: 1 wave per SIMD on every SIMD
#define LT(V, label) c = run([&] { hipLaunchKernelGGL((k_lat<V>), dim3(blocks), dim3(64), 0, 0, d_out, d_cyc, d_in, iters); }, d_cyc, blocks); \
    printf("lat  %-28s: %.3f ticks/iter\n", label, c / iters);
    LT(0, "lds write + b128 read");
    LT(1, "4 x ds_bpermute");
    LT(2, "permlane32 + 2x permlane16");
    LT(3, "sigmoid (exp+rcp)");
#define W4(V, nb, label) c = run([&] { hipLaunchKernelGGL((k_wg4<V>), dim3(nb), dim3(256), 0, 0, d_out, d_cyc, d_in, iters); }, d_cyc, nb); \
    printf("wg4  %-34s: %.1f ticks/iter\n", label, c / iters);
    W4(0, 250, "lds exchange + barrier (250 WG)");
    W4(0, 500, "lds exchange + barrier (500 WG)");
    W4(1, 500, "+rep_row + 32 dpp fmac (500 WG)");
    W4(2, 500, "+2 sigmoid + 2nd barrier (500 WG)");
    // calibrate tick vs wall: time a known-length kernel with events
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_tput<0>), dim3(blocks), dim3(64), 0, 0, d_out, d_cyc, d_in, iters * 10);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> cc(blocks);
    hipMemcpy(cc.data(), d_cyc, blocks * 8, hipMemcpyDeviceToHost);
    printf("calib: kernel %.3f ms wall, %lld ticks -> %.1f MHz tick rate; v_fmac rate = %.3f ns/instr\n", ms, cc[0],
           cc[0] / (ms * 1e3), ms * 1e6 / ((double)iters * 10 * 64));
    return 0;
}
