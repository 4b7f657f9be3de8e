// Fused forward of one periodic-GRU layer for gfx950 (H = 64), second generation: a CHAIN wave and a PRODUCER wave
// per sequence, two sequences per workgroup.
//
// What changed against gru_fused_fwd.hip, each point measured on the reverse scan first:
//   * two sequences per 4-wave workgroup.  The hardware starts a CU's next workgroup on the SIMD the previous one
//     ended on (tools/micro/where.hip): with 2-wave workgroups the scan wave of one sequence shared a SIMD with the
//     projection wave of the other on every CU while a SIMD sat idle;
//   * the 64x64 recurrent products are k-split over lane pairs (common.h split_matvec): half the LDS return traffic;
//   * the wave on the serial chain only does what depends on h: the producer wave also stores the saved states
//     (h from the LDS state buffer, r,u,c from a small LDS hand-off), and -- where its registers allow, D = 32 -- forms
//     the UPDATE gate u_t = sigmoid(xu_t + h_{t-1} Wu), which the chain wave needs only at the very end of step t;
//   * the two waves run in lockstep on the h_pub counter (the producer projects step t+8 while the chain wave works on
//     step t), so ring space and ring contents need no counters of their own.
#include <cstdlib>

#include "common.h"

namespace hpmn {

constexpr int GH = 64;        // hidden size of this kernel
constexpr int GRING = 16;     // ring slots
constexpr int GAHEAD = 8;     // steps the projection runs ahead of the recurrence (= the producer's prefetch block)

template <int NQ, int G>
__device__ __forceinline__ void bcast_matvec3b(const float4 *row4, const f2 *wa, const f2 *wb, const f2 *wc, f2 &a,
                                               f2 &b, f2 &c) {
    static_assert(NQ % G == 0, "groups");
    const v4f *row = reinterpret_cast<const v4f *>(row4);
    v4f cur[G], nxt[G];
#pragma unroll
    for (int i = 0; i < G; ++i) cur[i] = row[i];
#pragma unroll
    for (int g = 0; g < NQ / G; ++g) {
        if (g + 1 < NQ / G) {
#pragma unroll
            for (int i = 0; i < G; ++i) nxt[i] = row[(g + 1) * G + i];
        }
        asm volatile("" ::: "memory");
        land_group<G>(cur, a, b);
#pragma unroll
        for (int i = 0; i < G; ++i) {
            const int q = g * G + i;
            const f2 lo = {cur[i].x, cur[i].y}, hi = {cur[i].z, cur[i].w};
            a = __builtin_elementwise_fma(lo, wa[2 * q], a);
            b = __builtin_elementwise_fma(lo, wb[2 * q], b);
            c = __builtin_elementwise_fma(lo, wc[2 * q], c);
            a = __builtin_elementwise_fma(hi, wa[2 * q + 1], a);
            b = __builtin_elementwise_fma(hi, wb[2 * q + 1], b);
            c = __builtin_elementwise_fma(hi, wc[2 * q + 1], c);
        }
#pragma unroll
        for (int i = 0; i < G; ++i) cur[i] = nxt[i];
    }
}

// UPROD: the producer wave forms the update gate
template <int D, bool GATHER, bool TRAIN, bool UPROD>
__global__ __launch_bounds__(256, 1) void gru_fwd_duo_kernel(const HpmnGruFusedFwd a) {
    constexpr int H = GH;
    __shared__ __attribute__((aligned(16))) float ring_[2][GRING][3 * H];   // xp of steps t .. t+GAHEAD (r | u | c)
    __shared__ __attribute__((aligned(16))) float xb_[2][D];
    __shared__ __attribute__((aligned(16))) float hb_[2][2][H];            // h_{t-1} lives in hb[t & 1]
    __shared__ __attribute__((aligned(16))) float rhb_[2][H];
    __shared__ float ubuf_[2][2][H];
    __shared__ __attribute__((aligned(16))) v4f rcb_[2][2][H];             // r, u, c of step t in rcb[t & 1]
    __shared__ int ctr_[2][4];

    const int lane = threadIdx.x & 63;
    const int seq = (threadIdx.x >> 6) & 1, role = threadIdx.x >> 7;       // role 0: chain, 1: producer
    const int l = lane;
    const int T = a.T;
    const long b = 2 * (long)blockIdx.x + seq;
    if (b >= a.B) return;        // odd batch (before the barrier: ended waves do not take part in it)
    float (&ring)[GRING][3 * H] = ring_[seq];
    float (&xb)[D] = xb_[seq];
    float (&hb)[2][H] = hb_[seq];
    float (&rhb)[H] = rhb_[seq];
    float (&ubuf)[2][H] = ubuf_[seq];
    v4f (&rcb)[2][H] = rcb_[seq];
    int &produced = ctr_[seq][0], &h_pub = ctr_[seq][1], &u_pub = ctr_[seq][2];
    if (role == 0) {
        hb[0][l] = 0.f;
        if (lane == 0) { produced = 0; h_pub = 0; u_pub = 0; }
    }
    __syncthreads();             // the only barrier: counters and h_{-1} = 0 are in place

    if (role == 1) {
        // ================================================================== producer
        __builtin_amdgcn_s_setprio(2);
        // columns lane (r), 64 + lane (u), 128 + lane (c) of [wg[0:D] | wc[0:D]], exponent scale folded in
        f2 wr[D / 2], wu[D / 2], wcd[D / 2];
#pragma unroll
        for (int k = 0; k < D / 2; ++k) {
            wr[k] = f2{a.wg[(long)(2 * k) * 2 * H + lane], a.wg[(long)(2 * k + 1) * 2 * H + lane]} * NEG_LOG2E;
            wu[k] = f2{a.wg[(long)(2 * k) * 2 * H + H + lane], a.wg[(long)(2 * k + 1) * 2 * H + H + lane]} * NEG_LOG2E;
            wcd[k] = f2{a.wc[(long)(2 * k) * H + lane], a.wc[(long)(2 * k + 1) * H + lane]} * (2.0f * NEG_LOG2E);
        }
        float br = a.bg[lane] * NEG_LOG2E, bu = a.bg[H + lane] * NEG_LOG2E, bcn = a.bc[lane] * (2.0f * NEG_LOG2E);
#pragma unroll
        for (int k = 0; k < D / 2; ++k) { settle(wr[k]); settle(wu[k]); settle(wcd[k]); }
        settle(br); settle(bu); settle(bcn);
        f2 whu[2][16];
        if constexpr (UPROD) split_matvec_weights_t<2>(a.wg + (long)D * 2 * H + H, 2 * H, NEG_LOG2E, lane, whu);

        const int lx = lane < D ? lane : D - 1;          // lanes past D repeat the last column (never used)
        float xq[GAHEAD], xn[GAHEAD];
        int idn[GAHEAD];
        bool keep_q[GAHEAD], keep_n[GAHEAD];
        auto fetch_ids = [&](int t0, int (&id)[GAHEAD]) {
            if constexpr (GATHER) {
                const int f = lx / a.E;
#pragma unroll
                for (int i = 0; i < GAHEAD; ++i) {
                    int t = t0 + i;
                    t = t < T ? t : T - 1;
                    const int ti = t - a.front_zero;
                    id[i] = a.ids[(b * a.Tids + (ti > 0 ? ti : 0)) * a.F + f];      // clamped address, never examined here
                }
            }
        };
        auto fetch_rows = [&](int t0, const int (&id)[GAHEAD], float (&x)[GAHEAD], bool (&keep)[GAHEAD]) {
#pragma unroll
            for (int i = 0; i < GAHEAD; ++i) {
                int t = t0 + i;
                t = t < T ? t : T - 1;
                if constexpr (GATHER) {
                    const int e = lx % a.E;
                    x[i] = a.emb[(long)id[i] * a.E + e];
                    keep[i] = (t >= a.front_zero) && !(a.mask_id0 && id[i] == 0);
                } else {
                    x[i] = a.x[(b * (long)T + t) * D + lx];
                    keep[i] = true;
                }
            }
        };
        const float4 *xrow = reinterpret_cast<const float4 *>(xb);
        // project step t (its input row in xv) into ring slot t % GRING
        auto project = [&](int t, float xv) {
            if (lane < D) xb[lane] = xv;
            if constexpr (GATHER) {
                if (a.x_out != nullptr && lane < D) a.x_out[(b * (long)T + t) * D + lane] = xv;
            }
            wave_sync();
            f2 pr = {br, 0.f}, pu = {bu, 0.f}, pc = {bcn, 0.f};
            bcast_matvec3b<D / 4, 4>(xrow, wr, wu, wcd, pr, pu, pc);
            float *slot = ring[t & (GRING - 1)];
            slot[lane] = pr.x + pr.y;
            slot[H + lane] = pu.x + pu.y;
            slot[2 * H + lane] = pc.x + pc.y;
            wave_sync();                             // xb may be rewritten
        };

        // steps 0 .. GAHEAD-1 before the loop
        {
            int id0[GAHEAD];
            fetch_ids(0, id0);
            fetch_ids(GAHEAD, idn);
            fetch_rows(0, id0, xq, keep_q);
#pragma unroll
            for (int i = 0; i < GAHEAD; ++i)
                if (i < T) project(i, keep_q[i] ? xq[i] : 0.f);
            lds_counter_set(&produced, GAHEAD);
            fetch_rows(GAHEAD, idn, xq, keep_q);
            fetch_ids(2 * GAHEAD, idn);
#pragma unroll
            for (int i = 0; i < GAHEAD; ++i) { settle(xq[i]); if constexpr (GATHER) asm volatile("" : "+v"(idn[i])); }
        }

        const int period = a.period;
        const bool has_y = a.y != nullptr;
        int next_fire = period;                                            // h_{t-1} is an output row when t == next_fire
        float *yp = has_y ? a.y + (b * (long)(T / period)) * H + l : a.h_last + b * a.h_last_stride + l;
        const int y_adv = has_y ? H : 0;
        float *hsp = TRAIN ? a.hs + (b * (long)(T + 1)) * H + l : nullptr;  // row t <- h_{t-1}
        float *gp = TRAIN ? a.gates + (b * (long)T) * 3 * H + l : nullptr;  // row t-1 (row 0 twice, see below)
        int h_seen = 0;
        float u_prev = 0.f;

        // iteration t: u_t for the chain wave; then the saved rows of step t-1; then the projection of step t+GAHEAD
        auto iteration = [&](int t, float xv, bool proj) {
            while (h_seen < t) {                         // h_{t-1} (and r,c of step t-1) are in LDS
                h_seen = lds_counter_peek(&h_pub);
            }
            asm volatile("" ::: "memory");
            const int p = t & 1;
            float u_now = 0.f;
            // everything this iteration reads of the chain wave's buffers is read BEFORE the counter that lets the
            // chain wave move on (LDS operations of a wave complete in order): hb[p] and rcb[p^1] are rewritten at
            // the end of step t+1
            float hprev;
            v4f rc = {0.f, 0.f, 0.f, 0.f};
            if constexpr (UPROD) {
                const float xu = ring[t & (GRING - 1)][H + l];
                const float su = split_matvec<2>(&hb[p][0], whu, lane);
                hprev = hb[p][l];
                if constexpr (TRAIN) rc = rcb[p ^ 1][l];
                u_now = sigmoid_scaled(xu + su);
                ubuf[p][l] = u_now;
                lds_counter_set(&u_pub, t + 1);
            } else {
                hprev = hb[p][l];
                if constexpr (TRAIN) rc = rcb[p ^ 1][l];
                asm volatile("" : "+v"(hprev), "+v"(rc));          // (the reads have landed)
                lds_counter_set(&u_pub, t + 1);                    // here: "iteration t has read its inputs"
            }
            if constexpr (TRAIN) {
                *hsp = hprev;
                hsp += H;
                // (t == 0 writes an undefined row 0, which iteration 1 overwrites: same wave, same address, in order)
                gp[0] = rc.x;
                gp[H] = UPROD ? u_prev : rc.y;
                gp[2 * H] = rc.z;
                gp += t > 0 ? 3 * H : 0;
                u_prev = u_now;
            }
            *yp = hprev;
            const bool fire = t == next_fire;
            next_fire += fire ? period : 0;
            yp += fire ? y_adv : 0;
            if (proj) project(t + GAHEAD, xv);
        };

        for (int t0 = 0; t0 < T; t0 += GAHEAD) {
            // xq: rows of steps t0+GAHEAD .. ; fetch the block after that, ids one block further
            fetch_rows(t0 + 2 * GAHEAD, idn, xn, keep_n);
            fetch_ids(t0 + 3 * GAHEAD, idn);
#pragma unroll
            for (int i = 0; i < GAHEAD; ++i) {
                const int t = t0 + i;
                if (t < T) iteration(t, keep_q[i] ? xq[i] : 0.f, t + GAHEAD < T);     // wave-uniform
            }
#pragma unroll
            for (int i = 0; i < GAHEAD; ++i) { xq[i] = xn[i]; keep_q[i] = keep_n[i]; }
        }
        // the rows of the last step
        while (h_seen < T) h_seen = lds_counter_peek(&h_pub);
        asm volatile("" ::: "memory");
        {
            const int p = T & 1;
            const float hlast = hb[p][l];
            if constexpr (TRAIN) {
                const v4f rc = rcb[p ^ 1][l];
                *hsp = hlast;
                gp[0] = rc.x;
                gp[H] = UPROD ? u_prev : rc.y;
                gp[2 * H] = rc.z;
            }
            *yp = hlast;                                  // T is a multiple of period: the last output row (or h_last)
            a.h_last[b * a.h_last_stride + l] = hlast;
        }
        return;
    }

    // ====================================================================== chain wave
    __builtin_amdgcn_s_setprio(3);
    f2 whr[2][16], whu[2][16], whc[2][16];
    split_matvec_weights_t<2>(a.wg + (long)D * 2 * H, 2 * H, NEG_LOG2E, lane, whr);
    if constexpr (!UPROD) split_matvec_weights_t<2>(a.wg + (long)D * 2 * H + H, 2 * H, NEG_LOG2E, lane, whu);
    split_matvec_weights_t<2>(a.wc + (long)D * H, H, 2.0f * NEG_LOG2E, lane, whc);

    {
        int seen = 0;
        const int need = GAHEAD < T ? GAHEAD : T;
        while (seen < need) seen = lds_counter_peek(&produced);
        asm volatile("" ::: "memory");
    }
    float h = 0.f;
    int u_seen = 0;
    float xr = ring[0][l], xu = UPROD ? 0.f : ring[0][H + l], xcand = ring[0][2 * H + l];

    auto step = [&](int t, int p) {
        float r, u = 0.f;
        if constexpr (UPROD) {
            r = sigmoid_scaled(xr + split_matvec<2>(&hb[p][0], whr, lane));
        } else {
            float sr, su;
            split_matvec2x(&hb[p][0], whr, whu, lane, sr, su);
            r = sigmoid_scaled(xr + sr);
            u = sigmoid_scaled(xu + su);
        }
        rhb[lane] = r * h;
        wave_sync();
        const float cc = tanh_scaled(xcand + split_matvec<2>(&rhb[0], whc, lane));
        // next step's projected input (the producer is GAHEAD steps ahead; past the end: a stale slot, unused)
        const float *nx = ring[(t + 1) & (GRING - 1)];
        xr = nx[l];
        if constexpr (!UPROD) xu = nx[H + l];
        xcand = nx[2 * H + l];
        // UPROD: u_t from the producer.  Otherwise the same counter says "the producer has read h_{t-2}, r,u,c of
        // step t-2" (= its iteration t-1), the buffers this step is about to overwrite
        if constexpr (UPROD) {
            while (u_seen <= t) u_seen = lds_counter_peek(&u_pub);
            asm volatile("" ::: "memory");
            u = ubuf[p][l];
        } else {
            while (u_seen < t) u_seen = lds_counter_peek(&u_pub);
            asm volatile("" ::: "memory");
        }
        h = fmaf(u, h - cc, cc);
        hb[p ^ 1][lane] = h;
        if constexpr (TRAIN) rcb[p][l] = v4f{r, u, cc, 0.f};
        lds_counter_set(&h_pub, t + 1);
        wave_sync();
    };
    const int nfull = T >> 1;
    for (int q = 0; q < nfull; ++q) {
        step(2 * q, 0);
        step(2 * q + 1, 1);
    }
    if (T & 1) step(T - 1, 0);
}

template <int D, bool UPROD>
static int launch_duo(const HpmnGruFusedFwd &a, hipStream_t st) {
    const bool train = a.hs != nullptr;
    const dim3 grid((a.B + 1) / 2), blk(256);
    if (a.x == nullptr) {
        if (train) hipLaunchKernelGGL((gru_fwd_duo_kernel<D, true, true, UPROD>), grid, blk, 0, st, a);
        else       hipLaunchKernelGGL((gru_fwd_duo_kernel<D, true, false, UPROD>), grid, blk, 0, st, a);
    } else {
        if (train) hipLaunchKernelGGL((gru_fwd_duo_kernel<D, false, true, UPROD>), grid, blk, 0, st, a);
        else       hipLaunchKernelGGL((gru_fwd_duo_kernel<D, false, false, UPROD>), grid, blk, 0, st, a);
    }
    return check_launch();
}

#ifndef FWD_UPROD32
#define FWD_UPROD32 0
#endif
#ifndef FWD_UPROD64
#define FWD_UPROD64 0
#endif

int gru_fwd_duo_dispatch(const HpmnGruFusedFwd &a, hipStream_t st) {
    if (a.B == 0) return HPMN_OK;
    if (a.H != GH) return HPMN_EUNSUPPORTED;
    if (a.D == 32) return launch_duo<32, FWD_UPROD32>(a, st);
    if (a.D == 64) return launch_duo<64, FWD_UPROD64>(a, st);
    return HPMN_EUNSUPPORTED;
}

}  // namespace hpmn
