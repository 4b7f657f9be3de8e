// Reverse (BPTT) periodic-GRU layer scan for gfx950 -- the serial part of the gradient.
//
// Same decomposition as the forward: one wave owns 64/H sequences, lane = hidden unit k.
// The two transposed mat-vecs of the reverse step
//     d(r*h)[k]  = sum_n dc_pre[n] * Wc[D+k][n]
//     dh_prev[k] = dh[k]*u[k] + d(r*h)[k]*r[k] + sum_n (da_r[n]*Wg[D+k][n] + da_u[n]*Wg[D+k][H+n])
// use ROW k of the recurrent blocks, kept register-stationary in lane k (3H VGPRs); the
// broadcast operands (dc_pre, da_r, da_u) go through LDS as wave-uniform 16-byte reads.
// Saved activations (r,u,c,h_prev) and the incoming output gradients are prefetched one
// CHB-step chunk ahead into registers, so the only HBM access on the serial chain is the
// fire-and-forget store of d_act.  Weight/input gradients are GEMMs over d_act (host).
#include "common.h"

namespace hpmn {

constexpr int CHB = 8;  // reverse steps per prefetched chunk

template <int H>
__global__ __launch_bounds__(64, 1) void gru_scan_bwd_kernel(const HpmnGruBwd a) {
    constexpr int SPW = 64 / H;
    __shared__ __attribute__((aligned(16))) float bufA[SPW * H];
    __shared__ __attribute__((aligned(16))) float bufB[SPW * 2 * H];

    const int lane = threadIdx.x;
    const int s = lane / H;
    const int l = lane % H;
    const int B = a.B, T = a.T, D = a.D;
    const long b_raw = (long)blockIdx.x * SPW + s;
    const bool live = b_raw < B;
    const long b = live ? b_raw : (long)B - 1;

    float wcT[H], wgT[2 * H];
#pragma unroll
    for (int n = 0; n < H; n += 4) {
        const float4 v = *reinterpret_cast<const float4 *>(a.wc + (long)(D + l) * H + n);
        wcT[n] = v.x; wcT[n + 1] = v.y; wcT[n + 2] = v.z; wcT[n + 3] = v.w;
    }
#pragma unroll
    for (int n = 0; n < 2 * H; n += 4) {
        const float4 v = *reinterpret_cast<const float4 *>(a.wg + (long)(D + l) * 2 * H + n);
        wgT[n] = v.x; wgT[n + 1] = v.y; wgT[n + 2] = v.z; wgT[n + 3] = v.w;
    }

    const int period = a.period;
    const bool has_dy = a.d_y != nullptr;
    const int Ty = has_dy ? T / period : 0;

    struct Saved { float r, u, c, hp, dy; };
    auto fetch = [&](int j, Saved &o) {
        const int t = T - 1 - j;
        o.r = o.u = o.c = o.hp = o.dy = 0.f;
        if (t >= 0) {
            const float *g = a.gates + (b * T + t) * 4 * H;
            o.r = g[l];
            o.u = g[H + l];
            o.c = g[2 * H + l];
            o.hp = a.hs[(b * (T + 1) + t) * H + l];
            if (has_dy && (t + 1) % period == 0)
                o.dy = a.d_y[(b * Ty + (t + 1) / period - 1) * H + l];
        }
    };

    Saved cur[CHB], nxt[CHB];
#pragma unroll
    for (int jj = 0; jj < CHB; ++jj) fetch(jj, cur[jj]);

    float dh = a.d_h_last[b * a.d_h_last_stride + l];
    const int nchunk = (T + CHB - 1) / CHB;
    for (int q = 0; q < nchunk; ++q) {
#pragma unroll
        for (int jj = 0; jj < CHB; ++jj) fetch((q + 1) * CHB + jj, nxt[jj]);
#pragma unroll
        for (int jj = 0; jj < CHB; ++jj) {
            const int j = q * CHB + jj;
            if (j < T) {
                const int t = T - 1 - j;
                const Saved sv = cur[jj];
                dh += sv.dy;
                const float omu = 1.f - sv.u;
                const float dcp = dh * omu * (1.f - sv.c * sv.c);
                const float dau = dh * (sv.hp - sv.c) * sv.u * omu;
                bufA[lane] = dcp;
                wave_sync();
                const float4 *ra = reinterpret_cast<const float4 *>(&bufA[s * H]);
                float d0 = 0.f, d1 = 0.f;
#pragma unroll
                for (int n = 0; n < H / 4; ++n) {
                    const float4 v = ra[n];
                    d0 = fmaf(v.x, wcT[4 * n + 0], d0);
                    d1 = fmaf(v.y, wcT[4 * n + 1], d1);
                    d0 = fmaf(v.z, wcT[4 * n + 2], d0);
                    d1 = fmaf(v.w, wcT[4 * n + 3], d1);
                }
                const float drh = d0 + d1;
                const float dar = drh * sv.hp * sv.r * (1.f - sv.r);
                bufB[s * 2 * H + l] = dar;
                bufB[s * 2 * H + H + l] = dau;
                wave_sync();
                const float4 *rb = reinterpret_cast<const float4 *>(&bufB[s * 2 * H]);
                float e0 = 0.f, e1 = 0.f;
#pragma unroll
                for (int n = 0; n < 2 * H / 4; ++n) {
                    const float4 v = rb[n];
                    e0 = fmaf(v.x, wgT[4 * n + 0], e0);
                    e1 = fmaf(v.y, wgT[4 * n + 1], e1);
                    e0 = fmaf(v.z, wgT[4 * n + 2], e0);
                    e1 = fmaf(v.w, wgT[4 * n + 3], e1);
                }
                if (live) {
                    float *da = a.d_act + (b * T + t) * 3 * H;
                    da[l] = dar;
                    da[H + l] = dau;
                    da[2 * H + l] = dcp;
                }
                dh = fmaf(dh, sv.u, fmaf(drh, sv.r, e0 + e1));
                wave_sync();
            }
        }
#pragma unroll
        for (int jj = 0; jj < CHB; ++jj) cur[jj] = nxt[jj];
    }
}

int gru_scan_bwd_dispatch(const HpmnGruBwd &a, hipStream_t st) {
    if (a.H == 32) {
        hipLaunchKernelGGL((gru_scan_bwd_kernel<32>), dim3((a.B + 1) / 2), dim3(64), 0, st, a);
    } else if (a.H == 64) {
        hipLaunchKernelGGL((gru_scan_bwd_kernel<64>), dim3(a.B), dim3(64), 0, st, a);
    } else {
        return HPMN_EUNSUPPORTED;
    }
    return check_launch();
}

}  // namespace hpmn
