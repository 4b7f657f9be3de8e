// Online incremental memory update for gfx950 (include/hpmn_hip.h: hpmn_memory_update): one new event
// per user against the persisted [U, K, H] state store -- the cascade of code/srnn.py:727-748 with the
// periods of code/hpmn.py:113-129.
//
// One event touches layer 0 always and layer i with probability 1/(p_0..p_{i-1}): a single GRU cell
// step per fired layer, weights used once.  So this is a weight-STREAMING kernel, the opposite regime
// of the training scans: one wave per event, lane = hidden unit (H/64 units per lane at H = 128), the
// cell's inputs [x | h] broadcast from LDS, the weight rows read coalesced over the lanes (they stay
// L2-resident across the events of a call: 74 KB per H=64 layer).  Same arithmetic as the scan kernels
// (exp2-based sigmoid / tanh), so a sequence fed event by event lands within rounding of hpmn_scan_fwd.
#include "common.h"

namespace hpmn {

template <int H>
__global__ __launch_bounds__(64) void memory_update_kernel(const HpmnOnlineUpdate a) {
    constexpr int UPL = (H + 63) / 64;            // units per lane
    constexpr int NL = H < 64 ? H : 64;           // active lanes
    __shared__ float xin[128];                    // the fired layer's input (x or the state below)
    __shared__ float hb[H], rhb[H];
    const int lane = threadIdx.x;
    const int b = blockIdx.x;
    const int u = a.user[b];
    const int K = a.K;
    float *st = a.state + (long)u * K * H;

    int n = 0;
    if (lane == 0) {
        n = a.count[u] + 1;
        a.count[u] = n;
    }
    n = __shfl(n, 0);

    int Din = a.D;
    for (int k = lane; k < Din; k += 64) xin[k] = a.x[(long)b * a.D + k];
    int prod = 1;                                  // p_0 .. p_{i-1}
    for (int i = 0; i < K; ++i) {
        if (i > 0) {
            prod *= a.periods[i - 1];
            if (n % prod != 0) break;              // wave-uniform: the cascade stops here
        }
        const float *wg = a.wg[i], *bg = a.bg[i], *wc = a.wc[i], *bc = a.bc[i];
        float h[UPL];
#pragma unroll
        for (int q = 0; q < UPL; ++q) {
            const int j = lane + 64 * q;
            h[q] = (lane < NL) ? st[i * H + j] : 0.f;
            if (lane < NL) hb[j] = h[q];
        }
        wave_sync();
        float ar[UPL], au[UPL], ac[UPL];
#pragma unroll
        for (int q = 0; q < UPL; ++q) {
            const int j = (lane < NL) ? lane + 64 * q : 0;
            ar[q] = bg[j];
            au[q] = bg[H + j];
            ac[q] = bc[j];
        }
        // input rows [0, Din) and state rows [Din, Din + H) of the gate kernel; input rows of the candidate
        for (int k = 0; k < Din; ++k) {
            const float xv = xin[k];
#pragma unroll
            for (int q = 0; q < UPL; ++q) {
                const int j = (lane < NL) ? lane + 64 * q : 0;
                ar[q] = fmaf(xv, wg[(long)k * 2 * H + j], ar[q]);
                au[q] = fmaf(xv, wg[(long)k * 2 * H + H + j], au[q]);
                ac[q] = fmaf(xv, wc[(long)k * H + j], ac[q]);
            }
        }
        for (int k = 0; k < H; ++k) {
            const float hv = hb[k];
#pragma unroll
            for (int q = 0; q < UPL; ++q) {
                const int j = (lane < NL) ? lane + 64 * q : 0;
                ar[q] = fmaf(hv, wg[(long)(Din + k) * 2 * H + j], ar[q]);
                au[q] = fmaf(hv, wg[(long)(Din + k) * 2 * H + H + j], au[q]);
            }
        }
        float r[UPL], ug[UPL];
#pragma unroll
        for (int q = 0; q < UPL; ++q) {
            r[q] = sigmoid_scaled(NEG_LOG2E * ar[q]);
            ug[q] = sigmoid_scaled(NEG_LOG2E * au[q]);
            if (lane < NL) rhb[lane + 64 * q] = r[q] * h[q];
        }
        wave_sync();
        for (int k = 0; k < H; ++k) {
            const float rv = rhb[k];
#pragma unroll
            for (int q = 0; q < UPL; ++q) {
                const int j = (lane < NL) ? lane + 64 * q : 0;
                ac[q] = fmaf(rv, wc[(long)(Din + k) * H + j], ac[q]);
            }
        }
        wave_sync();                               // everyone is done with xin / hb before they are rewritten
#pragma unroll
        for (int q = 0; q < UPL; ++q) {
            const float cc = tanh_scaled(2.0f * NEG_LOG2E * ac[q]);
            const float hn = fmaf(ug[q], h[q] - cc, cc);
            if (lane < NL) {
                st[i * H + lane + 64 * q] = hn;
                xin[lane + 64 * q] = hn;           // the next layer's input
            }
        }
        Din = H;
        wave_sync();
    }
}

int memory_update_launch(const HpmnOnlineUpdate &a, hipStream_t st) {
    if (a.B == 0) return HPMN_OK;
    if (a.H == 32) hipLaunchKernelGGL(memory_update_kernel<32>, dim3(a.B), dim3(64), 0, st, a);
    else if (a.H == 64) hipLaunchKernelGGL(memory_update_kernel<64>, dim3(a.B), dim3(64), 0, st, a);
    else if (a.H == 128) hipLaunchKernelGGL(memory_update_kernel<128>, dim3(a.B), dim3(64), 0, st, a);
    else return HPMN_EUNSUPPORTED;
    return check_launch();
}

}  // namespace hpmn
