// Reverse (BPTT) periodic-GRU scan, H = 64, two specialised waves per sequence: a CHAIN wave and a FEEDER wave.
//
// Why: a single wave issues one instruction per ~5.5 cycles whatever the instruction is, so the
// length of a reverse step is the NUMBER of instructions the wave on the serial chain has to issue.  In
// gru_scan_bwd_helper_kernel that wave still issued ~230 per step, of which only 96 (two 64x64 products) and a
// dozen more are the recurrence: the rest was the prefetch of the saved activations (address arithmetic, loads,
// parking them in LDS), the coefficient arithmetic on them, the d_act stores and their addresses.  None of that
// depends on the gradient that is being propagated.  Here the second wave of the workgroup (on another SIMD of the
// CU) does all of it:
//
//   feeder (wave 1), per step:  loads r,u,c,h_prev,d_y of a step 4..7 steps ahead straight in unit layout (lane = unit),
//       turns them into the step's coefficients
//           k1 = (1-u)(1-c^2)     dc_pre = dh k1          k3 = h_prev r (1-r)     da_r = d(rh) k3
//           k2 = (h_prev-c)u(1-u) da_u   = dh k2
//       and parks {dy,k1,k2,k3 | r,u} in an LDS ring (one 16-byte + one 8-byte write, read back the same way);
//       computes e_u = da_u Wg[D+k][H:2H] for the chain wave as before (16 broadcast reads + 32 packed FMAs);
//       stores the step's d_act = [da_r | da_u | dc_pre] to HBM out of the LDS operand buffers the chain wave fills
//       anyway for its own broadcasts (one step late, so that it never waits for da_r).
//   chain (wave 0), per step:   dh += dy; dc_pre, da_u (2 multiplies) -> LDS; d(rh) = dc_pre Wc^T; da_r -> LDS;
//       e_r = da_r Wg_r^T; dh = dh u + d(rh) r + e_r + e_u.   No global memory access, no address arithmetic, no vmcnt.
//
// Hand-offs are LDS progress counters (common.h: data, lgkmcnt(0), counter; cached copies, re-read only when the
// cached value says "wait"); no barrier in the loop.  Buffers are double-buffered by step parity: the feeder reads
// step k-1's operands at the start of its step k, before it publishes e_u(k), and the chain wave cannot reach step
// k+1 (which overwrites them) without e_u(k).
#include <cstdlib>

#include "common.h"

namespace hpmn {

constexpr int FR_STEPS = 8;     // ring depth in steps (4 chunks of 2)
constexpr int FR_AHEAD = 3;     // chunks the feeder parks ahead of the chunk the chain wave is on

// tuning knobs (tools/micro/feed_bench.py builds variants with -D; the defaults are the measured best)
#ifndef FEED_DAU_SLEEP
#define FEED_DAU_SLEEP 0        // s_sleep argument of the feeder's poll for da_u (0: spin)
#endif
#ifndef FEED_EU_MID
#define FEED_EU_MID 0           // chain wave picks e_u up in the middle of its second product (0: at its end)
#endif
#ifndef FEED_KS
#define FEED_KS 2               // k-split of the three 64x64 products (common.h split_matvec; 1: broadcast reads)
#endif
#ifndef FEED_CHAIN_EU
#define FEED_CHAIN_EU 0         // float4s (of 16) of the e_u product the chain wave computes itself
#endif
constexpr int CEU = FEED_CHAIN_EU;

// TWO sequences per workgroup: the hardware places the waves of a workgroup on consecutive SIMDs of its rotation,
// but starts the next workgroup of the CU on the SIMD the previous one ended on (tools/micro/where.hip: with 2-wave
// workgroups every CU had the chain wave of one sequence and the feeder of the other on ONE SIMD and a SIMD idle;
// 4-wave workgroups land on four distinct SIMDs).  Waves 0,1 are the chain waves of sequences 2 blockIdx.x + 0,1,
// waves 2,3 their feeders; the two halves share nothing but the launch.
template <int KS>
__global__ __launch_bounds__(256, 1) void gru_scan_bwd_feed_kernel(const HpmnGruBwd a) {
    constexpr int H = 64;
    constexpr int KSS = KS == 1 ? 2 : KS;     // array extents of the unused form stay legal
    __shared__ __attribute__((aligned(16))) v4f ringA_[2][FR_STEPS][H];    // dy, k1, k2, k3
    __shared__ __attribute__((aligned(16))) f2 ringB_[2][FR_STEPS][H];     // r, u
    __shared__ __attribute__((aligned(16))) float bufA_[2][2][H];          // dc_pre
    __shared__ __attribute__((aligned(16))) float bufB_[2][2][2 * H];      // da_r | da_u
    __shared__ float eU_[2][2][H];
    __shared__ int ctr_[2][4];

    const int lane = threadIdx.x & 63;
    const int seq = (threadIdx.x >> 6) & 1, wave = threadIdx.x >> 7;     // wave: 0 chain, 1 feeder
    const int l = lane;
    const int T = a.T, D = a.D;
    const long b = 2 * (long)blockIdx.x + seq;
    if (b >= a.B) return;                        // odd batch: the last workgroup runs one sequence (before the barrier:
                                                 // ended waves do not take part in it)
    v4f (&ringA)[FR_STEPS][H] = ringA_[seq];
    f2 (&ringB)[FR_STEPS][H] = ringB_[seq];
    float (&bufA)[2][H] = bufA_[seq];
    float (&bufB)[2][2 * H] = bufB_[seq];
    float (&eU)[2][H] = eU_[seq];
    int &dau_pub = ctr_[seq][0], &eu_pub = ctr_[seq][1], &fed = ctr_[seq][2];
    const int t_lo0 = a.t_begin;
    const int t_hi = a.t_end > 0 ? a.t_end : T;
    const int nsteps = t_hi - t_lo0;
    const int nfull = nsteps >> 1;               // 2-step chunks; an odd last step is peeled
    if (lane == 0 && wave == 0) { dau_pub = 0; eu_pub = 0; fed = 0; }
    __syncthreads();

    if (wave == 1) {
        // ================================================================== feeder
        __builtin_amdgcn_s_setprio(2);
        f2 wuT[KS == 1 ? H / 2 : 1];       // row D+l of the update-gate block, packed over consecutive n
        f2 wuS[KSS][32 / KSS];
        if constexpr (KS == 1) {
#pragma unroll
            for (int n = 0; n < H / 2; ++n) wuT[n] = *reinterpret_cast<const f2 *>(a.wg + (long)(D + l) * 2 * H + H + 2 * n);
#pragma unroll
            for (int n = 0; n < H / 2; ++n) settle(wuT[n]);
        } else {
            split_matvec_weights<KSS>(a.wg + (long)D * 2 * H + H, 2 * H, lane, wuS);
        }

        const int period = a.period;
        const bool has_dy = a.d_y != nullptr;
        const float *gb = a.gates + b * (long)T * 3 * H + l;
        const float *hsb = a.hs + b * (long)(T + 1) * H + l;
        const float *dyb = has_dy ? a.d_y + b * (long)(T / period) * H + l : a.d_h_last + b * a.d_h_last_stride + l;
        const long dy_stride = has_dy ? H : 0;
        // d_y row j belongs to step (j+1)*period - 1 (t_hi is a multiple of period, so step t_hi-1 has one)
        int pf_fire = t_hi - 1, pf_row = t_hi / period - 1;

        struct Raw { float r[2], u[2], c[2], hp[2], dy[2]; bool m[2]; };
        // chunk q = iterations 2q, 2q+1 = steps t_hi-1-2q, t_hi-2-2q; rows before the sequence start are clamped
        // (loaded, parked, never consumed)
        auto load_chunk = [&](int q, Raw &w) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int t_raw = t_hi - 1 - 2 * q - j;
                const int t = t_raw > 0 ? t_raw : 0;
                const float *g = gb + (long)t * 3 * H;
                w.r[j] = g[0];
                w.u[j] = g[H];
                w.c[j] = g[2 * H];
                w.hp[j] = hsb[(long)t * H];
                const bool fire = has_dy && t_raw == pf_fire && pf_row >= 0;
                w.dy[j] = dyb[(long)(pf_row > 0 ? pf_row : 0) * dy_stride];
                w.m[j] = fire;
                pf_row -= fire ? 1 : 0;
                pf_fire -= fire ? period : 0;
            }
        };
        auto park_chunk = [&](int q, const Raw &w) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int slot = (2 * q + j) & (FR_STEPS - 1);
                const float r = w.r[j], u = w.u[j], c = w.c[j], hp = w.hp[j];
                const float omu = 1.f - u;
                const float k1 = omu * (1.f - c * c);
                const float k2 = (hp - c) * u * omu;
                const float k3 = hp * r * (1.f - r);
                ringA[slot][l] = v4f{w.m[j] ? w.dy[j] : 0.f, k1, k2, k3};
                ringB[slot][l] = f2{r, u};
            }
            lds_counter_set(&fed, 2 * q + 2);
        };

        {   // chunks 0 .. FR_AHEAD-1 before the loop
            Raw w;
#pragma unroll
            for (int q = 0; q < FR_AHEAD; ++q) {
                load_chunk(q, w);
                park_chunk(q, w);
            }
        }
        Raw w0, w1;                          // two chunks of loads in flight
        load_chunk(FR_AHEAD, w1);

        float *dap = a.d_act + (b * (long)T + (t_hi - 1)) * 3 * H + l;     // row of iteration 0
        int seen = 0;
        // one iteration: e_u(k); the d_act row of iteration k-1 goes out behind it
        auto iter = [&](int k, int p, bool store_prev) {
            while (seen <= k) {
                seen = lds_counter_peek(&dau_pub);
                if (FEED_DAU_SLEEP && seen <= k) __builtin_amdgcn_s_sleep(FEED_DAU_SLEEP);
            }
            asm volatile("" ::: "memory");
            float euv;
            if constexpr (KS == 1) {
                f2 e0, e1;
                bcast_matvec_first<H / 4 - CEU>(reinterpret_cast<const float4 *>(&bufB[p][H + 4 * CEU]), wuT + 2 * CEU, e0, e1);
                euv = (e0.x + e0.y) + (e1.x + e1.y);
            } else {
                euv = split_matvec<KSS>(&bufB[p][H], wuS, lane);
            }
            const float o_dar = bufB[p ^ 1][l], o_dau = bufB[p ^ 1][H + l], o_dcp = bufA[p ^ 1][l];
            eU[p][l] = euv;
            lds_counter_set(&eu_pub, k + 1);
            if (store_prev) {
                dap[0] = o_dar;
                dap[H] = o_dau;
                dap[2 * H] = o_dcp;
                dap -= 3 * H;
            }
        };

        // iteration 0 has no previous row to store: peeled together with iteration 1
        int q = 0;
        if (nfull > 0) {
            load_chunk(FR_AHEAD + 1, w0);
            iter(0, 0, false);
            iter(1, 1, true);
            park_chunk(FR_AHEAD, w1);
            q = 1;
        }
        for (; q + 1 < nfull; q += 2) {
            load_chunk(q + FR_AHEAD + 1, w1);
            iter(2 * q, 0, true);
            iter(2 * q + 1, 1, true);
            park_chunk(q + FR_AHEAD, w0);
            load_chunk(q + FR_AHEAD + 2, w0);
            iter(2 * q + 2, 0, true);
            iter(2 * q + 3, 1, true);
            park_chunk(q + FR_AHEAD + 1, w1);
        }
        if (q < nfull) {
            iter(2 * q, 0, true);
            iter(2 * q + 1, 1, true);
            park_chunk(q + FR_AHEAD, w0);
            q += 1;
        }
        if (nsteps & 1) iter(nsteps - 1, 0, nsteps > 1);
        // the last iteration's row: the chain wave reports "da_r of the last step is written" as dau_pub = nsteps+1
        {
            while (seen <= nsteps) {
                seen = lds_counter_peek(&dau_pub);
                if (seen <= nsteps) __builtin_amdgcn_s_sleep(1);
            }
            asm volatile("" ::: "memory");
            const int p = (nsteps - 1) & 1;
            dap[0] = bufB[p][l];
            dap[H] = bufB[p][H + l];
            dap[2 * H] = bufA[p][l];
        }
        return;
    }

    // ====================================================================== chain wave
    __builtin_amdgcn_s_setprio(3);
    f2 wcT[KS == 1 ? H / 2 : 1], wrT[KS == 1 ? H / 2 : 1];
    f2 wcS[KSS][32 / KSS], wrS[KSS][32 / KSS];
    if constexpr (KS == 1) {
#pragma unroll
        for (int n = 0; n < H / 2; ++n) wcT[n] = *reinterpret_cast<const f2 *>(a.wc + (long)(D + l) * H + 2 * n);
#pragma unroll
        for (int n = 0; n < H / 2; ++n) wrT[n] = *reinterpret_cast<const f2 *>(a.wg + (long)(D + l) * 2 * H + 2 * n);
#pragma unroll
        for (int n = 0; n < H / 2; ++n) { settle(wcT[n]); settle(wrT[n]); }
    } else {
        split_matvec_weights<KSS>(a.wc + (long)D * H, H, lane, wcS);
        split_matvec_weights<KSS>(a.wg + (long)D * 2 * H, 2 * H, lane, wrS);
    }
    f2 wuC[CEU > 0 ? 2 * CEU : 1];       // the chain wave's share of the update-gate rows
    if constexpr (CEU > 0) {
#pragma unroll
        for (int n = 0; n < 2 * CEU; ++n) wuC[n] = *reinterpret_cast<const f2 *>(a.wg + (long)(D + l) * 2 * H + H + 2 * n);
#pragma unroll
        for (int n = 0; n < 2 * CEU; ++n) settle(wuC[n]);
    }

    float dh = (t_hi == T) ? a.d_h_last[b * a.d_h_last_stride + l] : a.dh_carry[b * H + l];
    settle(dh);

    int fed_seen = 0, eu_seen = 0;
    auto wait_fed = [&](int need) {
        while (fed_seen < need) {
            fed_seen = lds_counter_peek(&fed);
            if (fed_seen < need) __builtin_amdgcn_s_sleep(1);
        }
        asm volatile("" ::: "memory");
    };
    wait_fed(1);
    v4f ca = ringA[0][l];
    f2 cb = ringB[0][l];

    auto step = [&](int k, int p) {
        const float dhin = dh + ca.x;
        const float dcp = dhin * ca.y;
        const float dau = dhin * ca.z;
        const float k3 = ca.w, r = cb.x, u = cb.y;
        bufB[p][H + l] = dau;
        bufA[p][l] = dcp;
        lds_counter_set(&dau_pub, k + 1);                            // the feeder may start on e_u(k)
        wave_sync();
        float drh;
        if constexpr (KS == 1) {
            f2 d0, d1;
            bcast_matvec_first<H / 4>(reinterpret_cast<const float4 *>(&bufA[p][0]), wcT, d0, d1);
            drh = (d0.x + d0.y) + (d1.x + d1.y);
        } else {
            drh = split_matvec<KSS>(&bufA[p][0], wcS, lane);
        }
        bufB[p][l] = drh * k3;
        wave_sync();
        // e_r in two halves; between them -- i.e. underneath the second half -- the reads whose latency would
        // otherwise sit on the chain: e_u (the feeder has normally published it by now) and the next step's
        // coefficients (parked several steps ago; the feeder parks past the end as well)
        f2 e0, e1;
        float er = 0.f;
        if constexpr (KS == 1) {
            bcast_matvec_first<H / 8>(reinterpret_cast<const float4 *>(&bufB[p][0]), wrT, e0, e1);
            if constexpr (CEU > 0) bcast_matvec<CEU>(reinterpret_cast<const float4 *>(&bufB[p][H]), wuC, e0, e1);
        }
        auto wait_eu = [&]() {
            while (eu_seen <= k) {
                eu_seen = lds_counter_peek(&eu_pub);
                if (eu_seen <= k) __builtin_amdgcn_s_sleep(1);
            }
            asm volatile("" ::: "memory");
        };
        float eu = 0.f;
        if (FEED_EU_MID) { wait_eu(); eu = eU[p][l]; }
        wait_fed(k + 2);
        const int slot = (k + 1) & (FR_STEPS - 1);
        ca = ringA[slot][l];
        cb = ringB[slot][l];
        if constexpr (KS == 1) {
            bcast_matvec<H / 8>(reinterpret_cast<const float4 *>(&bufB[p][H / 2]), wrT + H / 4, e0, e1);
            er = (e0.x + e0.y) + (e1.x + e1.y);
        } else {
            er = split_matvec<KSS>(&bufB[p][0], wrS, lane);
        }
        const float part = fmaf(dhin, u, fmaf(drh, r, er));
        if (!FEED_EU_MID) { wait_eu(); eu = eU[p][l]; }
        dh = part + eu;
        wave_sync();
    };

    for (int q = 0; q < nfull; ++q) {
        step(2 * q, 0);
        step(2 * q + 1, 1);
    }
    if (nsteps & 1) step(nsteps - 1, 0);
    lds_counter_set(&dau_pub, nsteps + 1);                           // da_r of the last step is in LDS
    if (t_lo0 > 0) a.dh_carry[b * H + l] = dh;
}

int gru_scan_bwd_feed_launch(const HpmnGruBwd &a, hipStream_t st) {
    hipLaunchKernelGGL(gru_scan_bwd_feed_kernel<FEED_KS>, dim3((a.B + 1) / 2), dim3(256), 0, st, a);
    return check_launch();
}

}  // namespace hpmn
