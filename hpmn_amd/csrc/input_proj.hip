// Input projection of one GRU layer for gfx950: the time-parallel half of the cell.
//
//   xp[m, 0:2H]  = x[m] Wg[0:D, :] + bg          (r and u pre-activations, input part)
//   xp[m, 2H:3H] = x[m] Wc[0:D, :] + bc          (candidate pre-activation, input part)
//
// for every row m = (b, t) at once -- no serial dependency, so it is hoisted out of the
// scan (gru_scan_fwd.hip).  Layer 0 fuses the embedding gather: the rows of x are built
// from (ids, emb) with the id-0 mask and the zero prefix and never round-trip through HBM
// unless the caller asks for x_out (training needs it for the weight-gradient GEMM).
//
// One workgroup = PR rows x all 3H columns.  The x tile is staged in LDS (coalesced
// float4 loads; 4 lanes per 64-byte embedding row in gather mode), thread n keeps column n
// of the [D, 3H] input weights in registers and walks the tile 4 rows at a time reading x
// as wave-uniform 16-byte LDS broadcasts; output rows are written fully coalesced (3H
// consecutive floats).  HBM-bound by the xp write (12H bytes per row).
#include "common.h"

namespace hpmn {

constexpr int PR = 64;  // rows per workgroup

template <int D, bool GATHER>
__global__ __launch_bounds__(256) void input_proj_kernel(const HpmnInputProj a) {
    constexpr int D4 = D / 4;
    __shared__ __attribute__((aligned(16))) float xs[PR * D];

    const int tid = threadIdx.x;
    const int H = a.H;
    const int N = 3 * H;
    const long M = (long)a.B * a.T;
    const long m0 = (long)blockIdx.x * PR;

    // ---- stage the x tile ------------------------------------------------------------------
    for (int i = tid; i < PR * D4; i += blockDim.x) {
        const int r = i / D4;
        const int d = (i % D4) * 4;
        const long m = m0 + r;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (m < M) {
            if constexpr (GATHER) {
                const long b = m / a.T;
                const int t = (int)(m - b * a.T) - a.front_zero;
                if (t >= 0) {
                    const int f = d / a.E;
                    const int id = a.ids[(b * a.Tids + t) * a.F + f];
                    if (!(a.mask_id0 && id == 0))
                        v = *reinterpret_cast<const float4 *>(a.emb + (long)id * a.E + (d - f * a.E));
                }
                if (a.x_out != nullptr) *reinterpret_cast<float4 *>(a.x_out + m * D + d) = v;
            } else {
                v = *reinterpret_cast<const float4 *>(a.x + m * D + d);
            }
        }
        *reinterpret_cast<float4 *>(&xs[r * D + d]) = v;
    }

    // ---- column weights ----------------------------------------------------------------------
    const int n = tid;
    float w[D];
    float bias = 0.f;
    if (n < N) {
        if (n < 2 * H) {
#pragma unroll
            for (int j = 0; j < D; ++j) w[j] = a.wg[(long)j * 2 * H + n];
            bias = a.bg[n];
        } else {
#pragma unroll
            for (int j = 0; j < D; ++j) w[j] = a.wc[(long)j * H + (n - 2 * H)];
            bias = a.bc[n - 2 * H];
        }
    }
    __syncthreads();
    if (n >= N) return;

    const int rows = (M - m0) < PR ? (int)(M - m0) : PR;
    for (int r = 0; r < rows; r += 4) {
        float acc0 = bias, acc1 = bias, acc2 = bias, acc3 = bias;
        const float4 *x0 = reinterpret_cast<const float4 *>(&xs[(r + 0) * D]);
        const float4 *x1 = reinterpret_cast<const float4 *>(&xs[(r + 1) * D]);
        const float4 *x2 = reinterpret_cast<const float4 *>(&xs[(r + 2) * D]);
        const float4 *x3 = reinterpret_cast<const float4 *>(&xs[(r + 3) * D]);
#pragma unroll
        for (int j = 0; j < D4; ++j) {
            const float4 v0 = x0[j], v1 = x1[j], v2 = x2[j], v3 = x3[j];
            acc0 = fmaf(v0.x, w[4 * j], acc0); acc0 = fmaf(v0.y, w[4 * j + 1], acc0);
            acc0 = fmaf(v0.z, w[4 * j + 2], acc0); acc0 = fmaf(v0.w, w[4 * j + 3], acc0);
            acc1 = fmaf(v1.x, w[4 * j], acc1); acc1 = fmaf(v1.y, w[4 * j + 1], acc1);
            acc1 = fmaf(v1.z, w[4 * j + 2], acc1); acc1 = fmaf(v1.w, w[4 * j + 3], acc1);
            acc2 = fmaf(v2.x, w[4 * j], acc2); acc2 = fmaf(v2.y, w[4 * j + 1], acc2);
            acc2 = fmaf(v2.z, w[4 * j + 2], acc2); acc2 = fmaf(v2.w, w[4 * j + 3], acc2);
            acc3 = fmaf(v3.x, w[4 * j], acc3); acc3 = fmaf(v3.y, w[4 * j + 1], acc3);
            acc3 = fmaf(v3.z, w[4 * j + 2], acc3); acc3 = fmaf(v3.w, w[4 * j + 3], acc3);
        }
        float *o = a.xp + (m0 + r) * N + n;
        o[0] = acc0;
        if (r + 1 < rows) o[N] = acc1;
        if (r + 2 < rows) o[2 * (long)N] = acc2;
        if (r + 3 < rows) o[3 * (long)N] = acc3;
    }
}

template <int D>
static int launch_proj(const HpmnInputProj &a, hipStream_t st) {
    const long M = (long)a.B * a.T;
    const unsigned grid = (unsigned)((M + PR - 1) / PR);
    const int threads = (3 * a.H + 63) / 64 * 64;
    if (a.x == nullptr)
        hipLaunchKernelGGL((input_proj_kernel<D, true>), dim3(grid), dim3(threads), 0, st, a);
    else
        hipLaunchKernelGGL((input_proj_kernel<D, false>), dim3(grid), dim3(threads), 0, st, a);
    return check_launch();
}

bool input_proj_supported(int H, int D) {
    return (H == 32 || H == 64) && (D == 16 || D == 32 || D == 48 || D == 64);
}

int input_proj_dispatch(const HpmnInputProj &a, hipStream_t st) {
    switch (a.D) {
        case 16: return launch_proj<16>(a, st);
        case 32: return launch_proj<32>(a, st);
        case 48: return launch_proj<48>(a, st);
        case 64: return launch_proj<64>(a, st);
        default: return HPMN_EUNSUPPORTED;
    }
}

}  // namespace hpmn
