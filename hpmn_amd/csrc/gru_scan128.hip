// Periodic-GRU layer scans for H = 128 on gfx950 (BASELINE.json configs[4]): forward and reverse.
//
// At H = 128 the recurrent kernels hold 3*128*128 floats -- 768 per lane of one wave, three times
// the 256 architectural VGPRs -- so the one-wave-per-sequence design of gru_scan_fwd.hip /
// gru_scan_bwd.hip cannot keep them stationary.  Here ONE WORKGROUP OF FOUR WAVES owns a sequence:
//
//   wave w, lane (c = lane & 31, p = lane >> 5)  <->  hidden unit u = 32 w + c, k-half p
//
// i.e. the output units are split over the waves (N-split) and the reduction index over the two
// half-waves (K-split), which again gives 3 * 64 = 192 stationary weights per lane, packed in pairs
// over consecutive k for v_pk_fma_f32.  The two halves of a dot product are joined with ONE
// v_permlane32_swap + add (no LDS); the broadcast operand is read from LDS as half-wave-uniform
// 16-byte loads through the same grouped pipeline as the H <= 64 kernels (bcast_matvec).  The price
// of several waves per sequence is two workgroup barriers per step (raw s_barrier behind
// lgkmcnt(0) only -- __syncthreads() would also drain the prefetch loads and the output stores).
//
// Streams that are known in advance (projected input; saved r,u,c,h_prev; incoming d_y) are
// prefetched PF steps ahead straight into registers with unconditional, clamped loads; every store
// is unconditional too (the halves of a wave split the output columns by select, not by branch), so
// the time loop has no control flow for the s_waitcnt pass to lose count in (see gru_scan_fwd.hip).
// LDS reads go two (not four) 16-byte groups deep: with four the kernel needs ~275 registers and the
// compiler parks ~50 stationary weights per lane in AGPRs (one v_accvgpr_read per use, every step).
#include <cstdlib>

#include "common.h"

namespace hpmn {

constexpr int H128 = 128;
constexpr int HPMN_SCAN128_SOLO_DEFAULT = 0;   // (see scan128_solo_mask)
constexpr int PF4 = 4;         // prefetch distance (steps) == unroll factor of the time loop (four-wave form; eight-wave: 2)

// A wave-uniform global pointer pinned to an SGPR pair and made opaque, so that an access through it with a 32-bit lane
// offset is emitted as `global_load/store v, v_off, s[base:base+1]` -- one VGPR per lane offset instead of a 64-bit lane
// pointer per stream (the optimiser otherwise folds the lane offset into the base and keeps base + lane in two VGPRs).
typedef __attribute__((address_space(1))) float gfloat;
__device__ __forceinline__ gfloat *sgpr_base(const float *p) {
    const unsigned long v = reinterpret_cast<unsigned long>(p);
    unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    asm("" : "+s"(lo), "+s"(hi));
    return reinterpret_cast<gfloat *>(((unsigned long)hi << 32) | lo);
}

__device__ __forceinline__ void wg_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// x[lane] + x[lane ^ 32] in every lane
__device__ __forceinline__ float join_halves(float x) {
    const unsigned v = __float_as_uint(x);
    auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// x[lane] + x[lane ^ 16] + x[lane ^ 32] + x[lane ^ 48] in every lane (v_permlane16_swap: odd rows of 16 <-> even rows)
__device__ __forceinline__ float join_quarters(float x) {
    const unsigned v = __float_as_uint(x);
    auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    return join_halves(__uint_as_float(r[0]) + __uint_as_float(r[1]));
}
template <int KQ> __device__ __forceinline__ float join_parts(float x) {
    if constexpr (KQ == 2) return join_halves(x);
    else return join_quarters(x);
}

// r4: the same design with EIGHT waves per sequence (NW = 8) -- built, parity-green, MEASURED SLOWER, kept
// behind HPMN_SCAN128_WAVES=8.  unit u = 16 w + (lane & 15), k-quarter p = lane >> 4: 3 * 32 = 96 stationary weights per lane
// instead of 192, half the packed FMAs and half the LDS reads on every wave's step; the four partial dot products are joined
// with v_permlane16_swap + v_permlane32_swap.  Two things decide against it:
//  * registers: 96 weights + prefetch slots + two LDS read groups come to 146-160 (forcing 128 spills 46-117 of them into the
//    time loop), so ONE eight-wave workgroup fits a CU, not two: it can only serve batches of at most one sequence per CU;
//  * and there -- C4 training steps at the data-parallel shard sizes, 1x MI355X -- it loses to the four-wave form all the same:
//    B = 250: 5.85 vs 5.48 ms/step, B = 63: 4.16 vs 3.85 (profiles/r04_experiments.txt).  The step is two barriers, two LDS
//    round trips and the activations; twice the waves make each barrier and each join longer (eight waves to collect, two
//    swaps instead of one) by more than the halved FMA/LDS issue saves.
template <bool TRAIN, int NW>
__global__ __launch_bounds__(64 * NW, 2) void gru_scan_fwd128_kernel(const HpmnGruFwd a) {
    constexpr int H = H128;
    constexpr int KQ = NW / 2;            // k-parts per dot product (lanes of a wave that share a unit)
    constexpr int UW = 64 / KQ;           // units per wave
    constexpr int KP = H / KQ;            // k per part
    constexpr int PF = NW == 8 ? 2 : PF4;
    __shared__ __attribute__((aligned(16))) float hb[H];
    __shared__ __attribute__((aligned(16))) float rhb[H];
    __builtin_amdgcn_s_setprio(3);

    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = lane % UW, p = lane / UW;
    const int u = UW * w + c;
    const int T = a.T, D = a.D;
    const long b = blockIdx.x;

    // stationary recurrent weights of unit u, k in [KP p, KP p + KP), exponent scale folded in
    f2 whr[KP / 2], whu[KP / 2], whc[KP / 2];
#pragma unroll
    for (int i = 0; i < KP / 2; ++i) {
        const long k = D + KP * p + 2 * i;
        whr[i] = f2{a.wg[k * 2 * H + u], a.wg[(k + 1) * 2 * H + u]} * NEG_LOG2E;
        whu[i] = f2{a.wg[k * 2 * H + H + u], a.wg[(k + 1) * 2 * H + H + u]} * NEG_LOG2E;
        whc[i] = f2{a.wc[k * H + u], a.wc[(k + 1) * H + u]} * (2.0f * NEG_LOG2E);
    }
#pragma unroll
    for (int i = 0; i < KP / 2; ++i) { settle(whr[i]); settle(whu[i]); settle(whc[i]); }

    const int t0 = a.t_begin;
    const int t1 = a.t_end > 0 ? a.t_end : T;
    // Addresses are (wave-uniform base: the sequence's row, in SGPRs) + (the lane's 32-bit offset): a 64-bit pointer per lane
    // and stream cost the four-wave form 260 registers -- four over the 256 that let TWO workgroups share a CU (occupancy 1:
    // a batch of 500 ran as two rounds of 256 workgroups).
    const unsigned uo = (unsigned)u;
    const float *xp_seq = a.xp + b * (long)T * 3 * H;            // (uniform)
    float xr[PF], xu[PF], xc[PF];
    auto fetch = [&](int t, int slot) {
        const int tc = t < T ? t : T - 1;
        const gfloat *row = sgpr_base(xp_seq + (long)tc * 3 * H);   // (uniform)
        xr[slot] = row[uo];
        xu[slot] = row[H + uo];
        xc[slot] = row[2 * H + uo];
    };
#pragma unroll
    for (int i = 0; i < PF; ++i) fetch(t0 + i, i);

    float h = a.h_init != nullptr ? a.h_init[b * a.h_init_stride + u] : 0.f;
    settle(h);
#pragma unroll
    for (int i = 0; i < PF; ++i) { settle(xr[i]); settle(xu[i]); settle(xc[i]); }
    hb[u] = h;
    if constexpr (TRAIN) {
        if (t0 == 0 && p == 0) a.hs[(b * (T + 1)) * H + u] = 0.f;
    }
    wg_barrier();

    const int period = a.period;
    const bool has_y = a.y != nullptr;
    int next_fire = t0 + period - 1;
    float *yp = has_y ? a.y + (b * (long)(T / period) + t0 / period) * H : a.h_last + b * a.h_last_stride;   // (uniform)
    const int y_adv = has_y ? H : 0;
    // TRAIN stores, split over the lane parts by select: (KQ = 2) p == 0 writes (hs, r), p == 1 writes (u, c);
    // (KQ = 4) one value per part: hs, r, u, c
    // (KQ = 2: the second store goes to the gates row alone -- uniform row + the lane's column, r or c)
    float *s0 = nullptr, *grow = nullptr;
    int adv0 = 0;
    unsigned s1o = 0;
    if constexpr (TRAIN) {
        float *hsp = a.hs + (b * (long)(T + 1) + t0 + 1) * H + u;
        grow = a.gates + (b * (long)T + t0) * 3 * H;             // (uniform)
        float *gp = grow + u;
        if constexpr (KQ == 2) {
            s0 = p == 0 ? hsp : gp + H;
            s1o = p == 0 ? uo : 2 * H + uo;
            adv0 = p == 0 ? H : 3 * H;
        } else {
            s0 = p == 0 ? hsp : gp + (p - 1) * H;
            adv0 = p == 0 ? H : 3 * H;
        }
    }
    const float4 *hrow = reinterpret_cast<const float4 *>(&hb[KP * p]);
    const float4 *rrow = reinterpret_cast<const float4 *>(&rhb[KP * p]);

    auto step = [&](int t, int slot) {
        f2 ar = {0.f, 0.f}, au = {0.f, 0.f};
        bcast_matvec2<KP / 4, 2>(hrow, whr, whu, ar, au);
        const float r = sigmoid_scaled(xr[slot] + join_parts<KQ>(ar.x + ar.y));
        const float ug = sigmoid_scaled(xu[slot] + join_parts<KQ>(au.x + au.y));
        rhb[u] = r * h;                     // both halves write the same value: no branch
        wg_barrier();                       // every wave's r*h is in place; every wave is done reading hb
        f2 ac = {0.f, 0.f}, ac2 = {0.f, 0.f};
        bcast_matvec<KP / 4, 2>(rrow, whc, ac, ac2);
        ac += ac2;
        const float cc = tanh_scaled(xc[slot] + join_parts<KQ>(ac.x + ac.y));
        h = fmaf(ug, h - cc, cc);
        hb[u] = h;
        fetch(t + PF, slot);
        if constexpr (TRAIN) {
            if constexpr (KQ == 2) {
                *s0 = p == 0 ? h : ug;
                sgpr_base(grow)[s1o] = p == 0 ? r : cc;
                s0 += adv0;
                grow += 3 * H;
            } else {
                *s0 = p == 0 ? h : (p == 1 ? r : (p == 2 ? ug : cc));
                s0 += adv0;
            }
        }
        yp[uo] = h;
        const bool fire = t == next_fire;
        next_fire += fire ? period : 0;
        yp += fire ? y_adv : 0;
        wg_barrier();                       // new h visible; every wave is done reading rhb
    };

    int t = t0;
    for (; t + PF <= t1; t += PF) {
#pragma unroll
        for (int i = 0; i < PF; ++i) step(t + i, i);
    }
    // remainder (fewer than PF steps): slots were refilled in order, slot i holds step t + i
#pragma unroll
    for (int i = 0; i < PF - 1; ++i)
        if (t + i < t1) step(t + i, i);
    if (p == 0) a.h_last[b * a.h_last_stride + u] = h;
}

template <int NW>
__global__ __launch_bounds__(64 * NW, 2) void gru_scan_bwd128_kernel(const HpmnGruBwd a) {
    constexpr int H = H128;
    constexpr int KQ = NW / 2, UW = 64 / KQ;
    constexpr int KC = H / KQ, KG = 2 * H / KQ;      // columns of wc / wg per lane part
    constexpr int PF = NW == 8 ? 2 : PF4;
    __shared__ __attribute__((aligned(16))) float bufA[H];
    __shared__ __attribute__((aligned(16))) float bufB[2 * H];
    __builtin_amdgcn_s_setprio(3);

    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = lane % UW, p = lane / UW;
    const int j = UW * w + c;
    const int T = a.T, D = a.D;
    const long b = blockIdx.x;

    // row D + j of the recurrent kernels (transposed products): wc columns [KC p, KC p + KC), wg columns [KG p, KG p + KG)
    f2 wcT[KC / 2], wgT[KG / 2];
#pragma unroll
    for (int n = 0; n < KC / 2; ++n) wcT[n] = *reinterpret_cast<const f2 *>(a.wc + (long)(D + j) * H + KC * p + 2 * n);
#pragma unroll
    for (int n = 0; n < KG / 2; ++n) wgT[n] = *reinterpret_cast<const f2 *>(a.wg + (long)(D + j) * 2 * H + KG * p + 2 * n);
#pragma unroll
    for (int n = 0; n < KC / 2; ++n) settle(wcT[n]);
#pragma unroll
    for (int n = 0; n < KG / 2; ++n) settle(wgT[n]);

    const int period = a.period;
    const bool has_dy = a.d_y != nullptr;
    // (uniform bases + the lane's 32-bit offset, as in the forward kernel: 263 -> <= 256 registers, two workgroups per CU)
    const unsigned jo = (unsigned)j;
    const float *gb = a.gates + b * (long)T * 3 * H;
    const float *hsb = a.hs + b * (long)(T + 1) * H;
    const float *dyb = has_dy ? a.d_y + b * (long)(T / period) * H : a.d_h_last + b * a.d_h_last_stride;
    const long dy_stride = has_dy ? H : 0;
    const int t_lo0 = a.t_begin;
    const int t_hi = a.t_end > 0 ? a.t_end : T;

    // prefetch stream, walking backwards: slot i of the ring holds step (current - i)
    int pf_fire = t_hi - 1, pf_row = t_hi / period - 1;       // t_hi is a multiple of period
    float gr[PF], gu[PF], gc[PF], ghp[PF], gdy[PF];
    bool gm[PF];
    auto fetch = [&](int t, int slot) {
        const int tc = t > 0 ? t : 0;
        const gfloat *row = sgpr_base(gb + (long)tc * 3 * H);
        gr[slot] = row[jo];
        gu[slot] = row[H + jo];
        gc[slot] = row[2 * H + jo];
        ghp[slot] = sgpr_base(hsb + (long)tc * H)[jo];
        const bool fire = has_dy && t == pf_fire && pf_row >= 0;
        gdy[slot] = sgpr_base(dyb + (long)(pf_row > 0 ? pf_row : 0) * dy_stride)[jo];
        gm[slot] = fire;
        pf_row -= fire ? 1 : 0;
        pf_fire -= fire ? period : 0;
    };
#pragma unroll
    for (int i = 0; i < PF; ++i) fetch(t_hi - 1 - i, i);
    float dh = (t_hi == T) ? a.d_h_last[b * a.d_h_last_stride + j] : a.dh_carry[b * H + j];
    settle(dh);
#pragma unroll
    for (int i = 0; i < PF; ++i) { settle(gr[i]); settle(gu[i]); settle(gc[i]); settle(ghp[i]); settle(gdy[i]); }

    const float4 *arow = reinterpret_cast<const float4 *>(&bufA[KC * p]);
    const float4 *brow = reinterpret_cast<const float4 *>(&bufB[KG * p]);
    // d_act stores split over the lane parts by select: (KQ = 2) p == 0 writes (da_r, da_u), p == 1 writes (dc_pre, dc_pre);
    // (KQ = 4) one value per part: da_r, da_u, dc_pre, dc_pre
    float *darow = a.d_act + (b * (long)T + (t_hi - 1)) * 3 * H;          // (uniform: the step's d_act row)
    const unsigned da0o = jo + (KQ == 2 ? (p == 0 ? 0 : 2 * H) : (p < 2 ? p : 2) * H);
    const unsigned da1o = jo + (p == 0 ? H : 2 * H);

    auto step = [&](int t, int slot) {
        const float r = gr[slot], ug = gu[slot], cc = gc[slot], hp = ghp[slot];
        dh += gm[slot] ? gdy[slot] : 0.f;
        const float omu = 1.f - ug;
        const float dcp = dh * omu * (1.f - cc * cc);
        const float dau = dh * (hp - cc) * ug * omu;
        bufA[j] = dcp;                      // both halves write the same value: no branch
        fetch(t - PF, slot);
        wg_barrier();                       // dc_pre of all units in place; everyone is done reading bufB
        f2 d0 = {0.f, 0.f}, d1 = {0.f, 0.f};
        bcast_matvec<KC / 4, 2>(arow, wcT, d0, d1);
        d0 += d1;
        const float drh = join_parts<KQ>(d0.x + d0.y);
        const float dar = drh * hp * r * (1.f - r);
        bufB[j] = dar;
        bufB[H + j] = dau;
        wg_barrier();                       // [da_r | da_u] in place; everyone is done reading bufA
        f2 e0 = {0.f, 0.f}, e1 = {0.f, 0.f};
        bcast_matvec<KG / 4, 2>(brow, wgT, e0, e1);
        e0 += e1;
        const float e = join_parts<KQ>(e0.x + e0.y);
        {
            gfloat *drow = sgpr_base(darow);
            if constexpr (KQ == 2) {
                drow[da0o] = p == 0 ? dar : dcp;
                drow[da1o] = p == 0 ? dau : dcp;
            } else {
                drow[da0o] = p == 0 ? dar : (p == 1 ? dau : dcp);
            }
            darow -= 3 * H;
        }
        dh = fmaf(dh, ug, fmaf(drh, r, e));
    };

    int t = t_hi - 1;
    for (; t - PF + 1 >= t_lo0; t -= PF) {
#pragma unroll
        for (int i = 0; i < PF; ++i) step(t - i, i);
    }
#pragma unroll
    for (int i = 0; i < PF - 1; ++i)
        if (t - i >= t_lo0) step(t - i, i);
    if (t_lo0 > 0 && p == 0) a.dh_carry[b * H + j] = dh;
}

// four waves per sequence; HPMN_SCAN128_WAVES=8: the eight-wave form (measured slower, see the top of the file)
static int scan128_waves(int) {
    static const int env = [] { const char *e = getenv("HPMN_SCAN128_WAVES"); return e ? atoi(e) : 0; }();
    return env == 8 ? 8 : 4;
}

// The four-wave kernels fit 256 registers (r4: lane offsets against uniform bases), so TWO workgroups share a CU and a batch
// of up to 2 x CUs sequences runs in one round.  `solo` launches pad the workgroup's LDS (unused dynamic LDS) to more than
// half of the CU's, which caps the occupancy at one workgroup per CU again and leaves half of every CU's registers to a
// kernel on another stream (the weight gradients beside the reverse scans).  HPMN_SCAN128_SOLO: bit 0 forward, bit 1 reverse.
static int scan128_solo_mask() {
    static const int env = [] { const char *e = getenv("HPMN_SCAN128_SOLO"); return e ? atoi(e) : HPMN_SCAN128_SOLO_DEFAULT; }();
    return env;
}
// (a batch that fits one workgroup per CU is spread that way: left to the dispatcher, 250 workgroups doubled up on some CUs
//  while others stayed empty -- C4 at B = 250: 4.83 vs 4.77 ms/step)
static bool scan128_fits_one_per_cu(int B) {
    static const int cus = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess) return 256;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) return 256;
        return n;
    }();
    return B <= cus;
}
template <typename K>
static size_t solo_pad(K kernel, bool solo) {
    if (!solo) return 0;
    hipFuncAttributes fa = {};
    const void *fn = reinterpret_cast<const void *>(kernel);
    if (hipFuncGetAttributes(&fa, fn) != hipSuccess) return 0;
    const size_t want = 82 * 1024;
    const size_t q = fa.sharedSizeBytes < want ? want - fa.sharedSizeBytes : 0;
    (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)q);
    return q;
}

int gru_scan_fwd128_dispatch(const HpmnGruFwd &a, hipStream_t st) {
    if (scan128_waves(a.B) == 8) {
        if (a.hs != nullptr) hipLaunchKernelGGL((gru_scan_fwd128_kernel<true, 8>), dim3(a.B), dim3(512), 0, st, a);
        else                 hipLaunchKernelGGL((gru_scan_fwd128_kernel<false, 8>), dim3(a.B), dim3(512), 0, st, a);
    } else {
        const bool solo = (scan128_solo_mask() & 1) != 0 || scan128_fits_one_per_cu(a.B);
        if (a.hs != nullptr) {
            static const size_t pad = solo_pad(gru_scan_fwd128_kernel<true, 4>, true);
            hipLaunchKernelGGL((gru_scan_fwd128_kernel<true, 4>), dim3(a.B), dim3(256), solo ? pad : 0, st, a);
        } else {
            static const size_t pad = solo_pad(gru_scan_fwd128_kernel<false, 4>, true);
            hipLaunchKernelGGL((gru_scan_fwd128_kernel<false, 4>), dim3(a.B), dim3(256), solo ? pad : 0, st, a);
        }
    }
    return check_launch();
}

int gru_scan_bwd128_dispatch(const HpmnGruBwd &a, hipStream_t st) {
    if (scan128_waves(a.B) == 8) hipLaunchKernelGGL((gru_scan_bwd128_kernel<8>), dim3(a.B), dim3(512), 0, st, a);
    else {
        const bool solo = (scan128_solo_mask() & 2) != 0 || scan128_fits_one_per_cu(a.B);
        static const size_t pad = solo_pad(gru_scan_bwd128_kernel<4>, true);
        hipLaunchKernelGGL((gru_scan_bwd128_kernel<4>), dim3(a.B), dim3(256), solo ? pad : 0, st, a);
    }
    return check_launch();
}

}  // namespace hpmn
