// Reverse (BPTT) periodic-GRU layer scan for gfx950 -- the serial part of the gradient.
//
// Same decomposition as the forward: one wave owns 64/H sequences, lane = hidden unit k.
// The two transposed mat-vecs of the reverse step
//     d(r*h)[k]  = sum_n dc_pre[n] * Wc[D+k][n]
//     dh_prev[k] = dh[k]*u[k] + d(r*h)[k]*r[k] + sum_n (da_r[n]*Wg[D+k][n] + da_u[n]*Wg[D+k][H+n])
// use ROW k of the recurrent blocks, kept register-stationary in lane k (3H VGPRs, packed in
// pairs for v_pk_fma_f32); the broadcast operands (dc_pre, da_r, da_u) go through LDS as
// wave-uniform 16-byte reads.  Saved activations (r,u,c | h_prev) are contiguous per 4-step
// chunk in HBM: they are fetched three 2-step chunks ahead (4 x 8 B per lane), parked in an LDS ring
// and read back per step, so the only HBM access on the serial chain is the fire-and-forget
// store of d_act.  Weight/input gradients are MFMA reductions over d_act (gru_wgrad.hip).
#include <cstdlib>

#include "common.h"

namespace hpmn {

constexpr int BCS = 2;    // steps per staged chunk
constexpr int BPD = 3;    // prefetch distance in chunks
constexpr int BRING = 4;  // chunks in the LDS ring (> BPD)

template <int H>
__global__ __launch_bounds__(64, 1) void gru_scan_bwd_kernel(const HpmnGruBwd a) {
    constexpr int SPW = 64 / H;
    constexpr int GF = BCS * 3 * H;   // gates floats per sequence per chunk (r,u,c)
    constexpr int HF = BCS * H;       // h_prev floats per sequence per chunk
    static_assert(GF / 2 == 3 * H && HF / 2 == H, "3 + 1 float2 per lane per chunk");
    __shared__ __attribute__((aligned(16))) float gring[BRING][SPW * GF];
    __shared__ __attribute__((aligned(16))) float hring[BRING][SPW * HF];
    __shared__ float dring[BRING][BCS * 64];
    __shared__ __attribute__((aligned(16))) float bufA[SPW * H];
    __shared__ __attribute__((aligned(16))) float bufB[SPW * 2 * H];

    __builtin_amdgcn_s_setprio(3);   // see gru_scan_fwd.hip
    const int lane = threadIdx.x;
    const int s = lane / H;
    const int l = lane % H;
    const int B = a.B, T = a.T, D = a.D;
    const long b_raw = SPW == 1 ? (long)blockIdx.x : (long)blockIdx.x * SPW + s;
    if (SPW != 1 && b_raw >= B) return;   // partial last wave at H=32: the dead half is not needed below
    const long b = b_raw;

    f2 wcT[H / 2], wgT[H];
#pragma unroll
    for (int n = 0; n < H / 2; ++n) wcT[n] = *reinterpret_cast<const f2 *>(a.wc + (long)(D + l) * H + 2 * n);
#pragma unroll
    for (int n = 0; n < H; ++n) wgT[n] = *reinterpret_cast<const f2 *>(a.wg + (long)(D + l) * 2 * H + 2 * n);
#pragma unroll
    for (int n = 0; n < H / 2; ++n) settle(wcT[n]);
#pragma unroll
    for (int n = 0; n < H; ++n) settle(wgT[n]);

    const int period = a.period;
    const bool has_dy = a.d_y != nullptr;
    const float *gb = a.gates + b * (long)T * 3 * H;
    const float *hsb = a.hs + b * (long)(T + 1) * H;
    // (without d_y the stream below still loads -- from a valid dummy row -- and masks the value out: the
    // time loop must stay free of branches, see the note at the loop)
    const float *dyb = has_dy ? a.d_y + b * (long)(T / period) * H + l : a.d_h_last + b * a.d_h_last_stride + l;
    const long dy_stride = has_dy ? H : 0;

    // steps t_hi-1 .. t_lo0 of this launch, walking backwards (whole sequence unless time-chunked)
    const int t_lo0 = a.t_begin;
    const int t_hi = a.t_end > 0 ? a.t_end : T;

    // chunk q covers steps t in [t_hi - BCS*(q+1), t_hi - BCS*q); float2 i of lane (s,l) covers elements
    // e = 2*(i*H + l), e+1 of the [BCS x 3H] gates image; the [BCS x H] h_prev image takes one float2.
    // Rows with t < 0 (tail chunk / run-ahead past the start) are clamped to 0: loaded, never consumed.
    int g_row[3], g_col[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int e = 2 * (i * H + l);
        g_row[i] = e / (3 * H);
        g_col[i] = e - g_row[i] * 3 * H;
    }
    const int h_row = (2 * l) / H, h_col = 2 * l - h_row * H;
    // incoming output gradients: d_y row j belongs to step (j+1)*period - 1; walking backwards the
    // prefetch stream keeps "the next step that has one" instead of dividing every step
    // (t_hi is a multiple of period, so step t_hi-1 has one)
    int pf_fire = t_hi - 1, pf_row = t_hi / period - 1;

    struct Pre { f2 g[3]; f2 hp; float dy[BCS]; bool m[BCS]; };
    auto load_chunk = [&](int q, Pre &p) {
        const int t_lo = t_hi - BCS * (q + 1);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            int t = t_lo + g_row[i];
            t = t > 0 ? t : 0;
            p.g[i] = *reinterpret_cast<const f2 *>(gb + (long)t * 3 * H + g_col[i]);
        }
        {
            int t = t_lo + h_row;
            t = t > 0 ? t : 0;
            p.hp = *reinterpret_cast<const f2 *>(hsb + (long)t * H + h_col);
        }
#pragma unroll
        for (int tt = BCS - 1; tt >= 0; --tt) {
            // unconditional load from a clamped row + a wave-uniform "this step has one" flag applied when
            // the value is parked
            const bool fire = has_dy && t_lo + tt == pf_fire && pf_row >= 0;
            p.dy[tt] = dyb[(long)(pf_row > 0 ? pf_row : 0) * dy_stride];
            p.m[tt] = fire;
            pf_row -= fire ? 1 : 0;
            pf_fire -= fire ? period : 0;
        }
    };
    auto park_chunk = [&](int q, const Pre &p) {
        float *gd = &gring[q % BRING][s * GF];
#pragma unroll
        for (int i = 0; i < 3; ++i) *reinterpret_cast<f2 *>(gd + 2 * (i * H + l)) = p.g[i];
        *reinterpret_cast<f2 *>(&hring[q % BRING][s * HF + 2 * l]) = p.hp;
#pragma unroll
        for (int tt = 0; tt < BCS; ++tt) dring[q % BRING][tt * 64 + lane] = p.m[tt] ? p.dy[tt] : 0.f;
    };

    const int nchunk = (t_hi - t_lo0 + BCS - 1) / BCS;
    {
        Pre p;
#pragma unroll
        for (int q = 0; q < BPD; ++q) {
            load_chunk(q, p);
            park_chunk(q, p);
        }
    }
    float dh = (t_hi == T) ? a.d_h_last[b * a.d_h_last_stride + l] : a.dh_carry[b * H + l];
    settle(dh);   // land the load here, not (per the waitcnt pass's loop merge) in every step
    wave_sync();

    // One step.  The loop around it has NO branches and every store is unconditional: with control flow
    // in the body the compiler's s_waitcnt pass loses count of the stores in flight and the wait for the
    // prefetched chunk becomes vmcnt(0), i.e. a wait for every store of the chunk to reach memory (and a
    // conditional d_y load made each chunk wait out a full HBM round trip at the join); see
    // tools/micro/scan_ablate.py and the note in gru_scan_fwd.hip.
    auto step = [&](int t, int tt, const float *gc, const float *hc, const float *dc) {
        const float r = gc[tt * 3 * H], u = gc[tt * 3 * H + H], c = gc[tt * 3 * H + 2 * H];
        const float hp = hc[tt * H];
        dh += dc[tt * 64];
        const float omu = 1.f - u;
        const float dcp = dh * omu * (1.f - c * c);
        const float dau = dh * (hp - c) * u * omu;
        bufA[lane] = dcp;
        wave_sync();
        f2 d0 = {0.f, 0.f}, d1 = {0.f, 0.f};
        bcast_matvec<H / 4>(reinterpret_cast<const float4 *>(&bufA[s * H]), wcT, d0, d1);
        const float drh = (d0.x + d0.y) + (d1.x + d1.y);
        const float dar = drh * hp * r * (1.f - r);
        bufB[s * 2 * H + l] = dar;
        bufB[s * 2 * H + H + l] = dau;
        wave_sync();
        f2 e0 = {0.f, 0.f}, e1 = {0.f, 0.f};
        bcast_matvec<2 * H / 4>(reinterpret_cast<const float4 *>(&bufB[s * 2 * H]), wgT, e0, e1);
        float *da = a.d_act + (b * T + t) * 3 * H + l;
        da[0] = dar;
        da[H] = dau;
        da[2 * H] = dcp;
        dh = fmaf(dh, u, fmaf(drh, r, (e0.x + e0.y) + (e1.x + e1.y)));
        wave_sync();
    };

    const int nfull = (t_hi - t_lo0) / BCS;    // chunks whose BCS steps all lie inside [t_lo0, t_hi)
    for (int q = 0; q < nfull; ++q) {
        Pre pre;
        load_chunk(q + BPD, pre);
        const int t_lo = t_hi - BCS * (q + 1);
        const float *gc = &gring[q % BRING][s * GF + l];
        const float *hc = &hring[q % BRING][s * HF + l];
        const float *dc = &dring[q % BRING][lane];
#pragma unroll
        for (int jj = 0; jj < BCS; ++jj) step(t_lo + BCS - 1 - jj, BCS - 1 - jj, gc, hc, dc);   // backwards in time
        park_chunk(q + BPD, pre);
        wave_sync();
    }
    if (nfull < nchunk) {   // odd number of steps: the oldest step sits alone in the last chunk's top row
        const int q = nfull;
        step(t_lo0, BCS - 1, &gring[q % BRING][s * GF + l], &hring[q % BRING][s * HF + l], &dring[q % BRING][lane]);
    }
    if (t_lo0 > 0) a.dh_carry[b * H + l] = dh;
}

// ---------------------------------------------------------------------------------------------------------------
// H = 64 with a HELPER wave.  A single wave issues one instruction per ~5.3 cycles whatever it is, and the reverse
// step is ~300 of them, so its length is issue time.  The update-gate half of the second product,
//     e_u[k] = sum_n da_u[n] Wg[D+k][H+n]                       (16 broadcast reads + 32 packed FMAs per step),
// needs nothing but da_u, which the main wave knows at the very START of the step: a second wave of the sequence's
// workgroup (on another SIMD of the CU -- at the reference batch half of them idle) computes it while the main
// wave is busy with d(rh) and da_r, and hands it back through LDS.  No barrier: the main wave publishes "da_u of
// iteration k is in LDS" (data, then a counter: LDS operations of a wave execute in order), the helper publishes
// "e_u of iteration k is in LDS"; the main wave cannot overwrite da_u before it has consumed e_u, which the helper
// only produces after reading all of da_u.  Everything else is gru_scan_bwd_kernel<64>.
// r4: measured slower than the chain + feeder kernel (gru_scan_bwd_feed.hip); compiled only with -DHPMN_LEGACY_KERNELS.
#ifdef HPMN_LEGACY_KERNELS
__device__ __forceinline__ int bw_peek(int *p) { return lds_counter_peek(p); }

// DX = true adds a THIRD wave: the gradient wrt the layer's input rows, d_x[t] = [da_r | da_u | dc_pre] [Wg[:D] | Wc[:D]]^T
// (row d of the input blocks register-stationary in lane d, the three operand rows are already in LDS for the main
// wave's own products).  That is the whole gru_dx launch of this layer -- for layers >= 1 the d_y the next reverse
// scan waits for, a kernel on the serial chain -- done underneath the scan.  The operand buffers are double-buffered
// by step parity so that wave has a full step to read them; it reports what it has read (dx_read) and the main wave
// checks that (a cached counter) before it reuses a buffer.
template <bool DX>
__global__ __launch_bounds__(DX ? 192 : 128, DX ? 2 : 1) void gru_scan_bwd_helper_kernel(const HpmnGruBwd a) {
    constexpr int H = 64;
    constexpr int GF = BCS * 3 * H, HF = BCS * H;
    __shared__ __attribute__((aligned(16))) float gring[BRING][GF];
    __shared__ __attribute__((aligned(16))) float hring[BRING][HF];
    __shared__ float dring[BRING][BCS * 64];
    __shared__ __attribute__((aligned(16))) float bufA[2][H];
    __shared__ __attribute__((aligned(16))) float bufB[2][2 * H];
    __shared__ float eU[2][H];
    __shared__ int dau_pub, eu_pub, dar_pub, dx_read;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l = lane;
    const int T = a.T, D = a.D;
    const long b = blockIdx.x;
    const int t_lo0 = a.t_begin;
    const int t_hi = a.t_end > 0 ? a.t_end : T;
    const int nsteps = t_hi - t_lo0;
    if (threadIdx.x == 0) { dau_pub = 0; eu_pub = 0; dar_pub = 0; dx_read = 0; }
    __syncthreads();

    if (wave == 1) {
        // ------------------------------------------------------------------ helper: e_u of every iteration
        __builtin_amdgcn_s_setprio(2);
        f2 wuT[H / 2];       // row D+l of the update-gate block, packed over consecutive n
#pragma unroll
        for (int n = 0; n < H / 2; ++n) wuT[n] = *reinterpret_cast<const f2 *>(a.wg + (long)(D + l) * 2 * H + H + 2 * n);
#pragma unroll
        for (int n = 0; n < H / 2; ++n) settle(wuT[n]);
        int seen = 0;
        for (int k = 0; k < nsteps; ++k) {
            while (seen <= k) {
                seen = bw_peek(&dau_pub);
                if (seen <= k) __builtin_amdgcn_s_sleep(1);
            }
            asm volatile("" ::: "memory");
            f2 e0 = {0.f, 0.f}, e1 = {0.f, 0.f};
            bcast_matvec<H / 4>(reinterpret_cast<const float4 *>(&bufB[k & 1][H]), wuT, e0, e1);
            eU[k & 1][l] = (e0.x + e0.y) + (e1.x + e1.y);
            lds_counter_set(&eu_pub, k + 1);
        }
        return;
    }
    if constexpr (DX) {
        if (wave == 2) {
            // -------------------------------------------------------------- input gradient of every iteration
            const int d = lane < D ? lane : D - 1;     // lanes past D repeat the last row (never stored)
            f2 wxg[H], wxc[H / 2];                     // row d of Wg[:D] (r | u columns) and of Wc[:D]
#pragma unroll
            for (int n = 0; n < H; ++n) wxg[n] = *reinterpret_cast<const f2 *>(a.wg + (long)d * 2 * H + 2 * n);
#pragma unroll
            for (int n = 0; n < H / 2; ++n) wxc[n] = *reinterpret_cast<const f2 *>(a.wc + (long)d * H + 2 * n);
#pragma unroll
            for (int n = 0; n < H; ++n) settle(wxg[n]);
#pragma unroll
            for (int n = 0; n < H / 2; ++n) settle(wxc[n]);
            float *dxp = a.d_x + (b * (long)T + (t_hi - 1)) * D + d;
            int seen = 0;
            for (int k = 0; k < nsteps; ++k) {
                while (seen <= k) {
                    seen = bw_peek(&dar_pub);
                    if (seen <= k) __builtin_amdgcn_s_sleep(2);
                }
                asm volatile("" ::: "memory");
                f2 x0 = {0.f, 0.f}, x1 = {0.f, 0.f}, x2 = {0.f, 0.f}, x3 = {0.f, 0.f};
                // (two reads in flight instead of four: this wave holds 192 stationary weights and must stay under
                //  256 registers so that two workgroups fit a CU)
                bcast_matvec<2 * H / 4, 2>(reinterpret_cast<const float4 *>(&bufB[k & 1][0]), wxg, x0, x1);
                bcast_matvec<H / 4, 2>(reinterpret_cast<const float4 *>(&bufA[k & 1][0]), wxc, x2, x3);
                lds_counter_set(&dx_read, k + 1);            // (every read above has been consumed)
                const float v = ((x0.x + x0.y) + (x1.x + x1.y)) + ((x2.x + x2.y) + (x3.x + x3.y));
                if (lane < D) *dxp = v;
                dxp -= D;
            }
            return;
        }
    }

    __builtin_amdgcn_s_setprio(3);
    f2 wcT[H / 2], wrT[H / 2];
#pragma unroll
    for (int n = 0; n < H / 2; ++n) wcT[n] = *reinterpret_cast<const f2 *>(a.wc + (long)(D + l) * H + 2 * n);
#pragma unroll
    for (int n = 0; n < H / 2; ++n) wrT[n] = *reinterpret_cast<const f2 *>(a.wg + (long)(D + l) * 2 * H + 2 * n);
#pragma unroll
    for (int n = 0; n < H / 2; ++n) { settle(wcT[n]); settle(wrT[n]); }

    const int period = a.period;
    const bool has_dy = a.d_y != nullptr;
    const float *gb = a.gates + b * (long)T * 3 * H;
    const float *hsb = a.hs + b * (long)(T + 1) * H;
    const float *dyb = has_dy ? a.d_y + b * (long)(T / period) * H + l : a.d_h_last + b * a.d_h_last_stride + l;
    const long dy_stride = has_dy ? H : 0;

    int g_row[3], g_col[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int e = 2 * (i * H + l);
        g_row[i] = e / (3 * H);
        g_col[i] = e - g_row[i] * 3 * H;
    }
    const int h_row = (2 * l) / H, h_col = 2 * l - h_row * H;
    int pf_fire = t_hi - 1, pf_row = t_hi / period - 1;

    struct Pre { f2 g[3]; f2 hp; float dy[BCS]; bool m[BCS]; };
    auto load_chunk = [&](int q, Pre &p) {
        const int t_lo = t_hi - BCS * (q + 1);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            int t = t_lo + g_row[i];
            t = t > 0 ? t : 0;
            p.g[i] = *reinterpret_cast<const f2 *>(gb + (long)t * 3 * H + g_col[i]);
        }
        {
            int t = t_lo + h_row;
            t = t > 0 ? t : 0;
            p.hp = *reinterpret_cast<const f2 *>(hsb + (long)t * H + h_col);
        }
#pragma unroll
        for (int tt = BCS - 1; tt >= 0; --tt) {
            const bool fire = has_dy && t_lo + tt == pf_fire && pf_row >= 0;
            p.dy[tt] = dyb[(long)(pf_row > 0 ? pf_row : 0) * dy_stride];
            p.m[tt] = fire;
            pf_row -= fire ? 1 : 0;
            pf_fire -= fire ? period : 0;
        }
    };
    auto park_chunk = [&](int q, const Pre &p) {
        float *gd = &gring[q % BRING][0];
#pragma unroll
        for (int i = 0; i < 3; ++i) *reinterpret_cast<f2 *>(gd + 2 * (i * H + l)) = p.g[i];
        *reinterpret_cast<f2 *>(&hring[q % BRING][2 * l]) = p.hp;
#pragma unroll
        for (int tt = 0; tt < BCS; ++tt) dring[q % BRING][tt * 64 + lane] = p.m[tt] ? p.dy[tt] : 0.f;
    };

    const int nchunk = (nsteps + BCS - 1) / BCS;
    {
        Pre p;
#pragma unroll
        for (int q = 0; q < BPD; ++q) {
            load_chunk(q, p);
            park_chunk(q, p);
        }
    }
    float dh = (t_hi == T) ? a.d_h_last[b * a.d_h_last_stride + l] : a.dh_carry[b * H + l];
    settle(dh);
    wave_sync();

    int kk = 0, eu_seen = 0, dxr_seen = 0;       // iteration count of this launch
    auto step = [&](int t, int tt, const float *gc, const float *hc, const float *dc) {
        const int p = kk & 1;
        const float r = gc[tt * 3 * H], u = gc[tt * 3 * H + H], c = gc[tt * 3 * H + 2 * H];
        const float hp = hc[tt * H];
        dh += dc[tt * 64];
        const float omu = 1.f - u;
        const float dcp = dh * omu * (1.f - c * c);
        const float dau = dh * (hp - c) * u * omu;
        if constexpr (DX) {
            // buffer p was last used in iteration kk-2: the dx wave must have read it
            while (dxr_seen < kk - 1) {
                dxr_seen = bw_peek(&dx_read);
                if (dxr_seen < kk - 1) __builtin_amdgcn_s_sleep(1);
            }
            asm volatile("" ::: "memory");
        }
        bufB[p][H + l] = dau;
        bufA[p][lane] = dcp;
        lds_counter_set(&dau_pub, kk + 1);                           // the helper may start on da_u
        wave_sync();
        f2 d0 = {0.f, 0.f}, d1 = {0.f, 0.f};
        bcast_matvec<H / 4>(reinterpret_cast<const float4 *>(&bufA[p][0]), wcT, d0, d1);
        const float drh = (d0.x + d0.y) + (d1.x + d1.y);
        const float dar = drh * hp * r * (1.f - r);
        bufB[p][l] = dar;
        if constexpr (DX) lds_counter_set(&dar_pub, kk + 1);         // all three operand rows of this step are in LDS
        wave_sync();
        f2 e0 = {0.f, 0.f}, e1 = {0.f, 0.f};
        bcast_matvec<H / 4>(reinterpret_cast<const float4 *>(&bufB[p][0]), wrT, e0, e1);
        float *da = a.d_act + (b * T + t) * 3 * H + l;
        da[0] = dar;
        da[H] = dau;
        da[2 * H] = dcp;
        while (eu_seen <= kk) {
            eu_seen = bw_peek(&eu_pub);
            if (eu_seen <= kk) __builtin_amdgcn_s_sleep(1);
        }
        asm volatile("" ::: "memory");
        const float eu = eU[p][l];
        dh = fmaf(dh, u, fmaf(drh, r, ((e0.x + e0.y) + (e1.x + e1.y)) + eu));
        kk += 1;
        wave_sync();
    };

    const int nfull = nsteps / BCS;
    for (int q = 0; q < nfull; ++q) {
        Pre pre;
        load_chunk(q + BPD, pre);
        const int t_lo = t_hi - BCS * (q + 1);
        const float *gc = &gring[q % BRING][l];
        const float *hc = &hring[q % BRING][l];
        const float *dc = &dring[q % BRING][lane];
#pragma unroll
        for (int jj = 0; jj < BCS; ++jj) step(t_lo + BCS - 1 - jj, BCS - 1 - jj, gc, hc, dc);
        park_chunk(q + BPD, pre);
        wave_sync();
    }
    if (nfull < nchunk) {
        const int q = nfull;
        step(t_lo0, BCS - 1, &gring[q % BRING][l], &hring[q % BRING][l], &dring[q % BRING][lane]);
    }
    if (t_lo0 > 0) a.dh_carry[b * H + l] = dh;
}

#endif  // HPMN_LEGACY_KERNELS

// HPMN_BWD_HELPER: 2 (default) chain + feeder wave (gru_scan_bwd_feed.hip), 1 the e_u helper wave above, 0 one wave
static int bwd_helper_enabled() {
    static const int helper = [] { const char *e = getenv("HPMN_BWD_HELPER"); return e ? atoi(e) : 2; }();
#ifdef HPMN_LEGACY_KERNELS
    return helper;
#else
    return helper == 1 ? 2 : helper;         // (the helper-wave kernel is not in this build: 1 means the feeder kernel too)
#endif
}
int gru_scan_bwd_feed_launch(const HpmnGruBwd &a, hipStream_t st);   // gru_scan_bwd_feed.hip
bool gru_scan_bwd_feed_dx_width(int D);
// does hpmn_gru_scan_bwd produce d_x itself (HpmnGruBwd.d_x) for this shape?  (and input widths 16, 32, 64)
bool gru_scan_bwd_fuses_dx(int H, int B) {
    // HPMN_BWD_DX_WAVE: does the scan launch produce d_x itself?  Default 1 with the chain + feeder kernel, whose waves
    // compute it on the matrix cores as an EPILOGUE, once their scan is done (gru_scan_bwd_feed.hip).  The same product
    // CONCURRENT with the scan was built twice and lost twice: with the e_u helper kernel above (HPMN_BWD_HELPER=1; a
    // third wave, packed FMAs) 4.12 vs 3.78 ms/step at C3; with the chain + feeder kernel as a third MFMA role out of an
    // LDS operand ring 3.62 vs 3.36 (a SIMD does not issue its other wave's VALU instructions while an fp32 MFMA is
    // passing, and every SIMD of the CU hosts a latency-critical wave).
    static const int dxw = [] { const char *e = getenv("HPMN_BWD_DX_WAVE"); return e ? atoi(e) : -1; }();
    if (dxw >= 0) return H == 64 && B <= 640 && bwd_helper_enabled() && dxw;
    return H == 64 && B <= 640 && bwd_helper_enabled() >= 2;
}
// HPMN_FWD_NO_CANDIDATE / HPMN_BWD_CANDIDATE_FROM_HS (include/hpmn_hip.h): the chain + feeder reverse kernels (single layer
// and pair) recover the candidate's coefficients from the saved states; the forward kernels that honour the flag are the
// current fused ones (gru_fused_fwd3.hip, gru_pair_fwd.hip).  HPMN_CANDIDATE_ELISION=0: off.
bool gru_candidate_elision(int H, int B) {
    static const int on = [] { const char *e = getenv("HPMN_CANDIDATE_ELISION"); return e ? atoi(e) : 1; }();
    return on && H == 64 && B <= 640 && bwd_helper_enabled() >= 2;
}
bool gru_scan_bwd_dx_width_ok(int D) { return bwd_helper_enabled() >= 2 ? gru_scan_bwd_feed_dx_width(D) : D <= 64; }

bool gru_scan_bwd_feed_scatter_ok(int D, int F, int E);      // gru_scan_bwd_feed.hip
// does hpmn_gru_scan_bwd add the input gradient into the table gradient itself (HpmnGruBwd.d_emb)?
// HPMN_FUSED_SCATTER=1 turns it on in hpmn_scan_bwd.  Built, parity-green, measured SLOWER at C3 (2.884 vs 2.799 ms/step):
// the atomics lengthen layer 0's launch by 67 us (732 vs 665), and the 140 us scatter launch it removes was hidden anyway --
// the step's tail is bounded by layer 0's weight gradient (300 us), which now shares the memory system with the late
// table-Adam pass alone and both get slower (396 + 290 us instead of 303 + 186).
// HPMN_FUSED_SCATTER=2 (r5): the scatter inside the LOOP of the chain + feeder kernel (D <= 32: the in-loop input gradient's
// tiles go straight into the table gradient; hpmn_scan_bwd then keeps d_x as the scratch the kernel wants)
bool gru_scan_bwd_scatter_inloop(int D) {
    static const int on = [] { const char *e = getenv("HPMN_FUSED_SCATTER"); return e ? atoi(e) : 0; }();
    return on == 2 && D <= 32;
}
bool gru_scan_bwd_fuses_scatter(int H, int B, int D, int F, int E) {
    static const int on = [] { const char *e = getenv("HPMN_FUSED_SCATTER"); return e ? atoi(e) : 0; }();
    return on && H == 64 && bwd_helper_enabled() >= 2 && B <= 640 && gru_scan_bwd_fuses_dx(H, B) &&
           gru_scan_bwd_feed_scatter_ok(D, F, E);
}

int gru_scan_bwd128_dispatch(const HpmnGruBwd &a, hipStream_t st);   // gru_scan128.hip

int gru_scan_bwd_dispatch(const HpmnGruBwd &a, hipStream_t st) {
    if ((a.flags & HPMN_BWD_CANDIDATE_FROM_HS) && !gru_candidate_elision(a.H, a.B)) return HPMN_EUNSUPPORTED;
    if (a.H == 128) return gru_scan_bwd128_dispatch(a, st);
    if (a.H == 32) {
        hipLaunchKernelGGL((gru_scan_bwd_kernel<32>), dim3((a.B + 1) / 2), dim3(64), 0, st, a);
    } else if (a.H == 64) {
        // helper-wave variant (default; HPMN_BWD_HELPER=0 selects the one-wave kernel): 0.651 vs 0.678 ms alone at C3
        // layer 0, and 0.683 vs 0.744 ms inside the step now that the weight-gradient launches leave room on every
        // CU (gru_wgrad.hip: one workgroup per CU; before that change the variant LOST in-step, 0.873 vs 0.815)
        if (bwd_helper_enabled() && a.B <= 640) {
            if (bwd_helper_enabled() >= 2)
                return gru_scan_bwd_feed_launch(a, st);
#ifdef HPMN_LEGACY_KERNELS
            else if (a.d_x != nullptr && gru_scan_bwd_fuses_dx(a.H, a.B))
                hipLaunchKernelGGL(gru_scan_bwd_helper_kernel<true>, dim3(a.B), dim3(192), 0, st, a);
            else
                hipLaunchKernelGGL(gru_scan_bwd_helper_kernel<false>, dim3(a.B), dim3(128), 0, st, a);
#endif
        } else {
            hipLaunchKernelGGL((gru_scan_bwd_kernel<64>), dim3(a.B), dim3(64), 0, st, a);
        }
    } else {
        return HPMN_EUNSUPPORTED;
    }
    return check_launch();
}

}  // namespace hpmn
