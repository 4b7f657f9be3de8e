// Forward periodic-GRU layer scan for gfx950 (CDNA4) -- the serial part of build_memory.
//
// Work decomposition (MI355X-first, see DESIGN.md "scan kernel"):
//   * the recurrence is serial in t and every sample is independent, so the unit of
//     parallelism is the SEQUENCE: one 64-lane wave owns 64/H sequences, lane = hidden
//     unit.  A launch is B*H/64 single-wave workgroups; nothing is shared between waves,
//     so there is no s_barrier anywhere in the time loop.
//   * the input half of both GRU kernels (x_t Wg[:D] + bg, x_t Wc[:D] + bc) has no serial
//     dependency and is hoisted into input_proj.hip; this kernel streams the projected
//     rows xp[b,t,0:3H] (prefetched CHF steps ahead into registers).
//   * the recurrent half is register-stationary: lane l keeps column l of the r, u and c
//     blocks of the state rows (3H arch VGPRs -- VALU cannot source AGPRs, which is what
//     bounds the per-lane weight budget at 256 and is why the input half is hoisted).
//   * the per-step broadcast operands (h_{t-1}, r*h_{t-1}) go through LDS as wave-uniform
//     16-byte reads (a broadcast, conflict-free); each value read feeds 2 (r,u) or 1 (c)
//     FMAs per lane.
#include "common.h"

namespace hpmn {

constexpr int CHF = 8;  // steps of projected input prefetched per chunk

template <int H, bool TRAIN>
__global__ __launch_bounds__(64, 1) void gru_scan_fwd_kernel(const HpmnGruFwd a) {
    constexpr int SPW = 64 / H;  // sequences per wave
    static_assert(64 % H == 0, "shape");

    __shared__ __attribute__((aligned(16))) float hb[SPW * H];
    __shared__ __attribute__((aligned(16))) float rhb[SPW * H];

    const int lane = threadIdx.x;
    const int s = lane / H;  // which of this wave's sequences
    const int l = lane % H;  // hidden unit
    const int B = a.B, T = a.T, D = a.D;
    const long b_raw = (long)blockIdx.x * SPW + s;
    const bool live = b_raw < B;
    const long b = live ? b_raw : (long)B - 1;

    // register-stationary recurrent weights (TF layout: rows [D, D+H) are the state rows)
    float whr[H], whu[H], whc[H];
#pragma unroll
    for (int k = 0; k < H; ++k) {
        whr[k] = a.wg[(long)(D + k) * 2 * H + l];
        whu[k] = a.wg[(long)(D + k) * 2 * H + H + l];
        whc[k] = a.wc[(long)(D + k) * H + l];
    }

    struct XP { float r, u, c; };
    const float *xpb = a.xp + b * (long)T * 3 * H + l;
    auto fetch = [&](int t, XP &o) {
        o.r = o.u = o.c = 0.f;
        if (t < T) {
            const float *p = xpb + (long)t * 3 * H;
            o.r = p[0];
            o.u = p[H];
            o.c = p[2 * H];
        }
    };

    XP cur[CHF], nxt[CHF];
#pragma unroll
    for (int i = 0; i < CHF; ++i) fetch(i, cur[i]);

    float h = 0.f;
    hb[lane] = 0.f;
    if constexpr (TRAIN) {
        if (live) a.hs[(b * (T + 1)) * H + l] = 0.f;
    }
    wave_sync();

    const int period = a.period;
    const int nchunk = (T + CHF - 1) / CHF;
    for (int c = 0; c < nchunk; ++c) {
        const int t0 = c * CHF;
#pragma unroll
        for (int i = 0; i < CHF; ++i) fetch(t0 + CHF + i, nxt[i]);
#pragma unroll
        for (int tt = 0; tt < CHF; ++tt) {
            const int t = t0 + tt;
            if (t < T) {
                const float4 *hrow = reinterpret_cast<const float4 *>(&hb[s * H]);
                float ar = cur[tt].r, au = cur[tt].u;
                float ar2 = 0.f, au2 = 0.f;
#pragma unroll
                for (int k = 0; k < H / 4; ++k) {
                    const float4 v = hrow[k];
                    ar = fmaf(v.x, whr[4 * k + 0], ar);   au = fmaf(v.x, whu[4 * k + 0], au);
                    ar2 = fmaf(v.y, whr[4 * k + 1], ar2); au2 = fmaf(v.y, whu[4 * k + 1], au2);
                    ar = fmaf(v.z, whr[4 * k + 2], ar);   au = fmaf(v.z, whu[4 * k + 2], au);
                    ar2 = fmaf(v.w, whr[4 * k + 3], ar2); au2 = fmaf(v.w, whu[4 * k + 3], au2);
                }
                const float r = fast_sigmoid(ar + ar2);
                const float u = fast_sigmoid(au + au2);
                const float rh = r * h;
                rhb[lane] = rh;
                wave_sync();
                const float4 *rrow = reinterpret_cast<const float4 *>(&rhb[s * H]);
                float ac = cur[tt].c, ac2 = 0.f, ac3 = 0.f, ac4 = 0.f;
#pragma unroll
                for (int k = 0; k < H / 4; ++k) {
                    const float4 v = rrow[k];
                    ac = fmaf(v.x, whc[4 * k + 0], ac);
                    ac2 = fmaf(v.y, whc[4 * k + 1], ac2);
                    ac3 = fmaf(v.z, whc[4 * k + 2], ac3);
                    ac4 = fmaf(v.w, whc[4 * k + 3], ac4);
                }
                const float cc = fast_tanh((ac + ac2) + (ac3 + ac4));
                h = fmaf(u, h - cc, cc);  // u*h + (1-u)*c
                hb[lane] = h;
                wave_sync();
                if (live) {
                    if constexpr (TRAIN) {
                        a.hs[(b * (T + 1) + t + 1) * H + l] = h;
                        float *g = a.gates + (b * T + t) * 4 * H;
                        g[l] = r;
                        g[H + l] = u;
                        g[2 * H + l] = cc;
                        g[3 * H + l] = rh;
                    }
                    if (a.y != nullptr && (t + 1) % period == 0)
                        a.y[(b * (T / period) + (t + 1) / period - 1) * H + l] = h;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < CHF; ++i) cur[i] = nxt[i];
    }
    if (live) a.h_last[b * a.h_last_stride + l] = h;
}

template <int H>
static int launch_fwd(const HpmnGruFwd &a, hipStream_t st) {
    constexpr int SPW = 64 / H;
    const int grid = (a.B + SPW - 1) / SPW;
    if (a.hs != nullptr) hipLaunchKernelGGL((gru_scan_fwd_kernel<H, true>), dim3(grid), dim3(64), 0, st, a);
    else                 hipLaunchKernelGGL((gru_scan_fwd_kernel<H, false>), dim3(grid), dim3(64), 0, st, a);
    return check_launch();
}

bool gru_shape_supported(int H, int D) {
    return (H == 32 || H == 64) && D >= 4 && D <= 128 && D % 4 == 0;
}

int gru_scan_fwd_dispatch(const HpmnGruFwd &a, hipStream_t st) {
    if (a.H == 32) return launch_fwd<32>(a, st);
    if (a.H == 64) return launch_fwd<64>(a, st);
    return HPMN_EUNSUPPORTED;
}

}  // namespace hpmn
