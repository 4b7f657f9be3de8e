// Forward periodic-GRU layer scan for gfx950 (CDNA4) -- the serial part of build_memory.
//
// Work decomposition (MI355X-first, see DESIGN_HISTORY.md 3.1):
//   * the recurrence is serial in t and every sample is independent, so the unit of
//     parallelism is the SEQUENCE: one 64-lane wave owns 64/H sequences, lane = hidden
//     unit.  A launch is B*H/64 single-wave workgroups; nothing is shared between waves,
//     so there is no s_barrier anywhere in the time loop.
//   * the input half of both GRU kernels (x_t Wg[:D] + bg, x_t Wc[:D] + bc) has no serial
//     dependency and is hoisted into input_proj.hip; this kernel streams the projected
//     rows xp[b,t,0:3H].  They are fetched one 2-step chunk (3 x 8 B per lane, contiguous
//     in HBM) three chunks ahead, parked in an LDS ring, and read back 4 bytes per lane per
//     gate when the step needs them, so HBM latency never sits on the serial chain and the
//     prefetch costs 6 VGPRs.
//   * the recurrent half is register-stationary: lane l keeps column l of the r, u and c
//     blocks of the state rows (3H arch VGPRs -- VALU cannot source AGPRs, which bounds the
//     per-lane weight budget at 256 and is why the input half is hoisted).
//   * a single wave issues one VALU instruction per ~5.3 cycles whatever it is (measured,
//     tools/micro), so the step is instruction-count bound: all mat-vec FMAs are packed
//     (v_pk_fma_f32 over two consecutive k: 3H/2 instructions per step instead of 3H) with
//     the broadcast operands (h_{t-1}, r*h_{t-1}) read from LDS as wave-uniform 16-byte
//     loads, each feeding two packed FMAs.
//   * NOT splitting a sequence over several waves is deliberate: a 4-wave, K-split variant with DPP
//     row broadcasts (no LDS broadcast reads, 48 weights per lane) was built and measured -- each
//     cross-wave partial-sum exchange (ds_write, s_barrier, ds_read) costs ~380 cycles against ~110
//     for the in-wave LDS round trip (tools/micro: "wg4" rows), and the step needs two of them:
//     0.71 ms vs 0.67 ms for this kernel at layer 0 of C3.
#include "common.h"

namespace hpmn {

constexpr int CS = 2;    // steps per staged chunk of projected input
constexpr int PD = 3;    // prefetch distance in chunks
constexpr int RING = 4;  // chunks in the LDS ring (> PD)

template <int H, bool TRAIN>
__global__ __launch_bounds__(64, 1) void gru_scan_fwd_kernel(const HpmnGruFwd a) {
    constexpr int SPW = 64 / H;          // sequences per wave
    constexpr int CF = CS * 3 * H;       // floats per sequence per chunk
    static_assert(64 % H == 0 && CF / 2 == 3 * H, "3 float2 per lane per chunk");

    __shared__ __attribute__((aligned(16))) float ring[RING][SPW * CF];
    __shared__ __attribute__((aligned(16))) float hb[SPW * H];
    __shared__ __attribute__((aligned(16))) float rhb[SPW * H];
    // the serial chain is latency-bound: win issue arbitration against co-resident waves of other kernels
    __builtin_amdgcn_s_setprio(3);

    const int lane = threadIdx.x;
    const int s = lane / H;  // which of this wave's sequences
    const int l = lane % H;  // hidden unit
    const int B = a.B, T = a.T, D = a.D;
    // SPW == 1: the sequence index is wave-uniform (addresses stay in SGPRs)
    const long b_raw = SPW == 1 ? (long)blockIdx.x : (long)blockIdx.x * SPW + s;
    const bool live = SPW == 1 ? true : (b_raw < B);
    if (!live) return;   // partial last wave at H=32: nothing below needs the dead half's lanes
    const long b = b_raw;

    // register-stationary recurrent weights, packed over consecutive k (TF layout: rows [D, D+H))
    f2 whr[H / 2], whu[H / 2], whc[H / 2];
#pragma unroll
    for (int k = 0; k < H / 2; ++k) {
        whr[k] = f2{a.wg[(long)(D + 2 * k) * 2 * H + l], a.wg[(long)(D + 2 * k + 1) * 2 * H + l]};
        whu[k] = f2{a.wg[(long)(D + 2 * k) * 2 * H + H + l], a.wg[(long)(D + 2 * k + 1) * 2 * H + H + l]};
        whc[k] = f2{a.wc[(long)(D + 2 * k) * H + l], a.wc[(long)(D + 2 * k + 1) * H + l]};
    }
    // The exponent scale of sigmoid / tanh (exp(-x) = exp2(-log2e * x)) is folded into the stationary
    // weights here and into the projected input by input_proj_kernel, so a step's serial chain is
    // matvec -> v_exp -> +1 -> v_rcp with no multiply in front of the v_exp.
#pragma unroll
    for (int k = 0; k < H / 2; ++k) {
        whr[k] *= NEG_LOG2E;
        whu[k] *= NEG_LOG2E;
        whc[k] *= 2.0f * NEG_LOG2E;
        settle(whr[k]); settle(whu[k]); settle(whc[k]);
    }

    // chunk staging: float2 i of lane (s,l) covers elements e = 2*(i*H + l), e+1 of the sequence's
    // [CS x 3H] chunk image, i.e. row e / 3H, column e % 3H.  Rows past the end are clamped to T-1
    // (they are loaded but never consumed), so the prefetch has no branches.
    const float *xpb = a.xp + b * (long)T * 3 * H;
    int c_row[3], c_col[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int e = 2 * (i * H + l);
        c_row[i] = e / (3 * H);
        c_col[i] = e - c_row[i] * 3 * H;
    }
    auto load_chunk = [&](int c, f2 (&v)[3]) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            int t = c * CS + c_row[i];
            t = t < T ? t : T - 1;
            v[i] = *reinterpret_cast<const f2 *>(xpb + (long)t * 3 * H + c_col[i]);
        }
    };
    auto park_chunk = [&](int c, const f2 (&v)[3]) {
        float *dst = &ring[c % RING][s * CF];
#pragma unroll
        for (int i = 0; i < 3; ++i) *reinterpret_cast<f2 *>(dst + 2 * (i * H + l)) = v[i];
    };

    // steps [t0, t1) of this launch (whole sequence unless the caller pipelines layers in time chunks)
    const int t0 = a.t_begin;
    const int t1 = a.t_end > 0 ? a.t_end : T;
    const int c_begin = t0 / CS, c_end = (t1 + CS - 1) / CS;
    {
        f2 v[3];
#pragma unroll
        for (int c = 0; c < PD; ++c) {
            load_chunk(c_begin + c, v);
            park_chunk(c_begin + c, v);
        }
    }

    float h = a.h_init != nullptr ? a.h_init[b * a.h_init_stride + l] : 0.f;
    hb[lane] = h;
    if constexpr (TRAIN) {
        if (t0 == 0) a.hs[(b * (T + 1)) * H + l] = 0.f;
    }
    wave_sync();

    // subsampled outputs y[:, j] = outputs[:, (j+1)*period - 1].  The store is UNCONDITIONAL: every step
    // writes h to the current slot and the slot pointer advances after a firing step, so the slot ends up
    // holding the firing step's state (same lane, same address: stores stay ordered).  Layers without y
    // aim the pointer at their h_last slot.  Why: with a conditional store in the loop the compiler's
    // s_waitcnt pass loses count of the stores in flight and waits vmcnt(0) for the prefetch loads at the
    // end of each chunk, i.e. for every store of the chunk to reach memory -- measured 300 of the 1520
    // cycles of a step (tools/micro/scan_ablate.py).  For the same reason the loop body has no branches:
    // an odd last step is peeled, and the dead half of a partial H=32 wave has left the kernel above.
    const int period = a.period;
    const bool has_y = a.y != nullptr;
    int next_fire = t0 + period - 1;                                    // t0 is a multiple of period
    float *yp = has_y ? a.y + (b * (long)(T / period) + t0 / period) * H + l : a.h_last + b * a.h_last_stride + l;
    const int y_adv = has_y ? H : 0;
    float *hsp = TRAIN ? a.hs + (b * (long)(T + 1) + t0 + 1) * H + l : nullptr;
    float *gp = TRAIN ? a.gates + (b * (long)T + t0) * 3 * H + l : nullptr;

    auto step = [&](int t, const float *xc) {
        // (xr, xu, xcand and the weights carry the exponent scale)
        const float xr = xc[0], xu = xc[H], xcand = xc[2 * H];
        f2 ar = {0.f, 0.f}, au = {0.f, 0.f};
        bcast_matvec2<H / 4>(reinterpret_cast<const float4 *>(&hb[s * H]), whr, whu, ar, au);
        const float r = sigmoid_scaled(xr + (ar.x + ar.y));
        const float u = sigmoid_scaled(xu + (au.x + au.y));
        rhb[lane] = r * h;
        wave_sync();
        f2 ac = {0.f, 0.f}, ac2 = {0.f, 0.f};
        bcast_matvec<H / 4>(reinterpret_cast<const float4 *>(&rhb[s * H]), whc, ac, ac2);
        ac += ac2;
        const float cc = tanh_scaled(xcand + (ac.x + ac.y));
        h = fmaf(u, h - cc, cc);  // u*h + (1-u)*c
        hb[lane] = h;
        wave_sync();
        if constexpr (TRAIN) {
            *hsp = h;
            gp[0] = r;
            gp[H] = u;
            gp[2 * H] = cc;
            hsp += H;
            gp += 3 * H;
        }
        *yp = h;
        const bool fire = t == next_fire;
        next_fire += fire ? period : 0;
        yp += fire ? y_adv : 0;
    };

    const int c_full = t1 / CS;            // chunks [c_begin, c_full) run all CS steps
    for (int c = c_begin; c < c_full; ++c) {
        f2 pre[3];
        load_chunk(c + PD, pre);          // in flight for the steps below
        const float *xc = &ring[c % RING][s * CF + l];
#pragma unroll
        for (int tt = 0; tt < CS; ++tt) step(c * CS + tt, xc + tt * 3 * H);
        park_chunk(c + PD, pre);
        wave_sync();
    }
    if (c_full < c_end) step(c_full * CS, &ring[c_full % RING][s * CF + l]);   // odd T: one step left
    a.h_last[b * a.h_last_stride + l] = h;
}

template <int H>
static int launch_fwd(const HpmnGruFwd &a, hipStream_t st) {
    constexpr int SPW = 64 / H;
    const int grid = (a.B + SPW - 1) / SPW;
    if (a.hs != nullptr) hipLaunchKernelGGL((gru_scan_fwd_kernel<H, true>), dim3(grid), dim3(64), 0, st, a);
    else                 hipLaunchKernelGGL((gru_scan_fwd_kernel<H, false>), dim3(grid), dim3(64), 0, st, a);
    return check_launch();
}

int gru_scan_fwd128_dispatch(const HpmnGruFwd &a, hipStream_t st);   // gru_scan128.hip

bool gru_shape_supported(int H, int D) {
    return (H == 32 || H == 64 || H == 128) && D >= 4 && D <= 128 && D % 4 == 0;
}

int gru_scan_fwd_dispatch(const HpmnGruFwd &a, hipStream_t st) {
    if (a.H == 32) return launch_fwd<32>(a, st);
    if (a.H == 64) return launch_fwd<64>(a, st);
    if (a.H == 128) return gru_scan_fwd128_dispatch(a, st);
    return HPMN_EUNSUPPORTED;
}

}  // namespace hpmn
