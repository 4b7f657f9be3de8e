// Fused forward of one periodic-GRU layer for gfx950 (H = 64): input projection + recurrence in one
// launch, two SPECIALISED WAVES per sequence.
//
// Why: at the reference's batch (500 sequences) the one-wave-per-sequence scan keeps one SIMD per
// sequence busy and leaves the other half of the chip idle, while the time-parallel input projection
// runs as a separate HBM-bound kernel before it (0.37 ms of a 4.4 ms step over the 7 layers) and
// hands 3H floats per step through HBM (393 MB written + 393 MB re-read per layer-0 launch).  Here
// the projection moves onto a second wave of the sequence's workgroup -- i.e. onto one of the idle
// SIMDs of the same CU:
//
//   wave 1 (producer): per step, x_t (64-byte embedding rows gathered from (ids, emb) for layer 0,
//       or the 256-byte row of the layer below) -> LDS broadcast -> xp_t = x_t Wx + b with Wx
//       register-stationary (3 columns per lane, packed FMAs) -> LDS ring slot t % RING -> counter.
//       Its inputs are prefetched one 8-step block (rows) / two blocks (ids) ahead.
//   wave 0 (consumer): the recurrence of gru_scan_fwd.hip, reading xp_t from the ring instead of
//       from an HBM prefetch (no global loads, no park writes, no vmcnt waits left in its loop).
//
// The waves never meet at a barrier: each publishes a progress counter in LDS (data first, then the
// counter -- LDS operations of a wave execute in order) and keeps a cached copy of the other's, which it
// re-reads only when the cached value says "wait"; the producer runs up to RING steps ahead, so the
// consumer re-reads about once per RING steps.  Everything else (exp2-domain weights, branch-free
// stores, unconditional y slot) is as in gru_scan_fwd.hip; results are bit-identical to the two-kernel
// path up to the summation order of the projection.
#include <cstdlib>

#include "common.h"

namespace hpmn {

// r4: the first generation below is a measured-slower fallback (0.571 vs 0.508 ms at C3 layer 0); it is compiled only with
// -DHPMN_LEGACY_KERNELS (HPMN_HIPCC_FLAGS) -- the default library dispatches every call to gru_fused_fwd3.hip.
#ifdef HPMN_LEGACY_KERNELS
constexpr int FH = 64;        // hidden size of this kernel
constexpr int FRING = 8;      // ring slots (steps the producer may run ahead)
constexpr int FBLK = 8;       // producer prefetch block (steps)

__device__ __forceinline__ int lds_peek(int *p) { return lds_counter_peek(p); }
__device__ __forceinline__ void lds_publish(int *p, int v) { lds_counter_set(p, v); }

// acc_c += sum over NQ float4 of a wave-uniform LDS row times three packed weight sets (bcast_matvec, x3)
template <int NQ, int G>
__device__ __forceinline__ void bcast_matvec3(const float4 *row4, const f2 *wa, const f2 *wb, const f2 *wc, f2 &a,
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

// HELP = true adds a third wave: the UPDATE gate u_t = sigmoid(xu_t + h_{t-1} Wg[D:, H:2H]) needs nothing the scan
// wave computes within the step, so a helper wave forms it (16 broadcast reads + 32 packed FMAs + one sigmoid, and
// the u store of the saved gates) while the scan wave does r, r*h and the candidate; the scan wave picks u_t up from
// LDS just before the state update.  No barrier: h_pub ("h_{t-1} is in hb") and u_pub ("u_t is in ubuf") are LDS
// counters; the scan wave cannot overwrite hb before it has consumed u_t, which exists only once the helper has
// read all of hb.
template <int D, bool GATHER, bool TRAIN, bool HELP>
__global__ __launch_bounds__(HELP ? 192 : 128, 1) void gru_fused_fwd_kernel(const HpmnGruFusedFwd a) {
    constexpr int H = FH;
    __shared__ __attribute__((aligned(16))) float ring[FRING][3 * H];
    __shared__ __attribute__((aligned(16))) float xb[D];
    __shared__ __attribute__((aligned(16))) float hb[H];
    __shared__ __attribute__((aligned(16))) float rhb[H];
    __shared__ float ubuf[2][H];
    __shared__ int produced, consumed, h_pub, u_pub;

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int T = a.T;
    const long b = blockIdx.x;
    if (threadIdx.x == 0) { produced = 0; consumed = 0; h_pub = 0; u_pub = 0; }
    if (threadIdx.x < H) hb[threadIdx.x] = 0.f;
    __syncthreads();                       // the only barrier: the counters start at zero for every wave

    if constexpr (HELP) {
        if (wave == 2) {
            // -------------------------------------------------------------- helper: the update gate of every step
            __builtin_amdgcn_s_setprio(2);
            const int l = lane;
            f2 whu[H / 2];
#pragma unroll
            for (int k = 0; k < H / 2; ++k)
                whu[k] = f2{a.wg[(long)(D + 2 * k) * 2 * H + H + l], a.wg[(long)(D + 2 * k + 1) * 2 * H + H + l]} * NEG_LOG2E;
#pragma unroll
            for (int k = 0; k < H / 2; ++k) settle(whu[k]);
            float *gpu = TRAIN ? a.gates + (b * (long)T) * 3 * H + H + l : nullptr;
            int p_seen = 0, h_seen = 0;
            for (int t = 0; t < T; ++t) {
                while (p_seen <= t) {
                    p_seen = lds_peek(&produced);
                    if (p_seen <= t) __builtin_amdgcn_s_sleep(1);
                }
                while (h_seen < t) {                     // h_{t-1} is in hb once the scan wave has finished step t-1
                    h_seen = lds_peek(&h_pub);
                    if (h_seen < t) __builtin_amdgcn_s_sleep(1);
                }
                asm volatile("" ::: "memory");
                const float xu = ring[t % FRING][H + l];
                f2 au = {0.f, 0.f}, au2 = {0.f, 0.f};
                bcast_matvec<H / 4>(reinterpret_cast<const float4 *>(hb), whu, au, au2);
                const float u = sigmoid_scaled(xu + ((au.x + au.y) + (au2.x + au2.y)));
                ubuf[t & 1][l] = u;
                lds_publish(&u_pub, t + 1);
                if constexpr (TRAIN) {
                    *gpu = u;
                    gpu += 3 * H;
                }
            }
            return;
        }
    }

    if (wave == 1) {
        // ------------------------------------------------------------------ producer: x_t -> xp_t
        __builtin_amdgcn_s_setprio(1);
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

        const int l = lane < D ? lane : D - 1;          // lanes past D repeat the last column (never used)
        // input stream: block of FBLK steps, rows one block ahead, ids two blocks ahead
        float xq[FBLK], xn[FBLK];
        long idn[FBLK];
        bool keep_q[FBLK], keep_n[FBLK];
        auto fetch_ids = [&](int t0, long (&id)[FBLK]) {
            if constexpr (GATHER) {
                const int f = l / a.E;
#pragma unroll
                for (int i = 0; i < FBLK; ++i) {
                    int t = t0 + i;
                    t = t < T ? t : T - 1;
                    const int ti = t - a.front_zero;
                    id[i] = load_id(a.ids, (b * (long)a.Tids + (ti > 0 ? ti : 0)) * a.F + f, a.mask_id0);      // clamped address, never examined here
                }
            }
        };
        auto fetch_rows = [&](int t0, const long (&id)[FBLK], float (&x)[FBLK], bool (&keep)[FBLK]) {
#pragma unroll
            for (int i = 0; i < FBLK; ++i) {
                int t = t0 + i;
                t = t < T ? t : T - 1;
                if constexpr (GATHER) {
                    const int e = l % a.E;
                    x[i] = a.emb[id[i] * a.E + e];
                    keep[i] = (t >= a.front_zero) && !id_masked(id[i], a.mask_id0);
                } else {
                    x[i] = a.x[(b * (long)T + t) * D + l];
                    keep[i] = true;
                }
            }
        };
        long id0[FBLK];
        fetch_ids(0, id0);
        fetch_ids(FBLK, idn);
        fetch_rows(0, id0, xq, keep_q);
#pragma unroll
        for (int i = 0; i < FBLK; ++i) { settle(xq[i]); if constexpr (GATHER) asm volatile("" : "+v"(idn[i])); }

        int cons_seen = 0;
        const float4 *xrow = reinterpret_cast<const float4 *>(xb);
        for (int t0 = 0; t0 < T; t0 += FBLK) {
            fetch_rows(t0 + FBLK, idn, xn, keep_n);
            fetch_ids(t0 + 2 * FBLK, idn);
#pragma unroll
            for (int i = 0; i < FBLK; ++i) {
                const int t = t0 + i;
                if (t < T) {                                 // wave-uniform; only the last block is partial
                    const float xv = keep_q[i] ? xq[i] : 0.f;
                    if (lane < D) xb[lane] = xv;
                    if constexpr (GATHER) {
                        if (a.x_out != nullptr && lane < D) a.x_out[(b * (long)T + t) * D + lane] = xv;
                    }
                    wave_sync();
                    f2 pr = {br, 0.f}, pu = {bu, 0.f}, pc = {bcn, 0.f};
                    bcast_matvec3<D / 4, 4>(xrow, wr, wu, wcd, pr, pu, pc);
                    // ring space: the consumer must have finished step t - FRING
                    while (cons_seen + FRING <= t) {
                        cons_seen = lds_peek(&consumed);
                        if (cons_seen + FRING <= t) __builtin_amdgcn_s_sleep(4);
                    }
                    float *slot = ring[t % FRING];
                    slot[lane] = pr.x + pr.y;
                    slot[H + lane] = pu.x + pu.y;
                    slot[2 * H + lane] = pc.x + pc.y;
                    lds_publish(&produced, t + 1);
                    wave_sync();                             // xb may be rewritten
                }
            }
#pragma unroll
            for (int i = 0; i < FBLK; ++i) { xq[i] = xn[i]; keep_q[i] = keep_n[i]; }
        }
        return;
    }

    // ---------------------------------------------------------------------- consumer: the recurrence
    __builtin_amdgcn_s_setprio(3);
    const int l = lane;
    f2 whr[H / 2], whu[HELP ? 1 : H / 2], whc[H / 2];
#pragma unroll
    for (int k = 0; k < H / 2; ++k) {
        whr[k] = f2{a.wg[(long)(D + 2 * k) * 2 * H + l], a.wg[(long)(D + 2 * k + 1) * 2 * H + l]} * NEG_LOG2E;
        if constexpr (!HELP)
            whu[k] = f2{a.wg[(long)(D + 2 * k) * 2 * H + H + l], a.wg[(long)(D + 2 * k + 1) * 2 * H + H + l]} * NEG_LOG2E;
        whc[k] = f2{a.wc[(long)(D + 2 * k) * H + l], a.wc[(long)(D + 2 * k + 1) * H + l]} * (2.0f * NEG_LOG2E);
    }
#pragma unroll
    for (int k = 0; k < H / 2; ++k) {
        settle(whr[k]); settle(whc[k]);
        if constexpr (!HELP) settle(whu[k]);
    }

    float h = 0.f;
    if constexpr (TRAIN) a.hs[(b * (T + 1)) * H + l] = 0.f;
    wave_sync();
    int u_seen = 0;

    const int period = a.period;
    const bool has_y = a.y != nullptr;
    int next_fire = period - 1;
    float *yp = has_y ? a.y + (b * (long)(T / period)) * H + l : a.h_last + b * a.h_last_stride + l;
    const int y_adv = has_y ? H : 0;
    float *hsp = TRAIN ? a.hs + (b * (long)(T + 1) + 1) * H + l : nullptr;
    float *gp = TRAIN ? a.gates + (b * (long)T) * 3 * H + l : nullptr;

    int avail = 0;
    for (int t = 0; t < T; ++t) {
        while (avail <= t) {
            avail = lds_peek(&produced);
            if (avail <= t) __builtin_amdgcn_s_sleep(1);
        }
        asm volatile("" ::: "memory");                       // ring reads stay behind the counter check
        const float *xc = &ring[t % FRING][l];
        const float xr = xc[0], xcand = xc[2 * H];
        float r, u = 0.f;
        if constexpr (HELP) {
            f2 ar = {0.f, 0.f}, ar2 = {0.f, 0.f};
            bcast_matvec<H / 4>(reinterpret_cast<const float4 *>(hb), whr, ar, ar2);
            r = sigmoid_scaled(xr + ((ar.x + ar.y) + (ar2.x + ar2.y)));
        } else {
            const float xu = xc[H];
            f2 ar = {0.f, 0.f}, au = {0.f, 0.f};
            bcast_matvec2<H / 4>(reinterpret_cast<const float4 *>(hb), whr, whu, ar, au);
            r = sigmoid_scaled(xr + (ar.x + ar.y));
            u = sigmoid_scaled(xu + (au.x + au.y));
        }
        rhb[lane] = r * h;
        wave_sync();
        f2 ac = {0.f, 0.f}, ac2 = {0.f, 0.f};
        bcast_matvec<H / 4>(reinterpret_cast<const float4 *>(rhb), whc, ac, ac2);
        ac += ac2;
        const float cc = tanh_scaled(xcand + (ac.x + ac.y));
        if constexpr (HELP) {
            while (u_seen <= t) {
                u_seen = lds_peek(&u_pub);
                if (u_seen <= t) __builtin_amdgcn_s_sleep(1);
            }
            asm volatile("" ::: "memory");
            u = ubuf[t & 1][l];
        }
        h = fmaf(u, h - cc, cc);
        hb[lane] = h;
        if constexpr (HELP) lds_publish(&h_pub, t + 1);       // the helper may start on u_{t+1}
        if ((t & 1) == 1) lds_publish(&consumed, t + 1);     // slot t (and t-1) may be refilled
        wave_sync();
        if constexpr (TRAIN) {
            *hsp = h;
            gp[0] = r;
            if constexpr (!HELP) gp[H] = u;
            gp[2 * H] = cc;
            hsp += H;
            gp += 3 * H;
        }
        *yp = h;
        const bool fire = t == next_fire;
        next_fire += fire ? period : 0;
        yp += fire ? y_adv : 0;
    }
    a.h_last[b * a.h_last_stride + l] = h;
}

template <int D, bool HELP>
static int launch_fused_h(const HpmnGruFusedFwd &a, hipStream_t st) {
    const bool train = a.hs != nullptr;
    const dim3 grid(a.B), blk(HELP ? 192 : 128);
    if (a.x == nullptr) {
        if (train) hipLaunchKernelGGL((gru_fused_fwd_kernel<D, true, true, HELP>), grid, blk, 0, st, a);
        else       hipLaunchKernelGGL((gru_fused_fwd_kernel<D, true, false, HELP>), grid, blk, 0, st, a);
    } else {
        if (train) hipLaunchKernelGGL((gru_fused_fwd_kernel<D, false, true, HELP>), grid, blk, 0, st, a);
        else       hipLaunchKernelGGL((gru_fused_fwd_kernel<D, false, false, HELP>), grid, blk, 0, st, a);
    }
    return check_launch();
}

template <int D>
static int launch_fused(const HpmnGruFusedFwd &a, hipStream_t st) {
    static const int help = [] { const char *e = getenv("HPMN_FWD_HELPER"); return e ? atoi(e) : 0; }();
    // (D = 64: the projection wave alone needs ~290 registers, so a three-wave workgroup would take three SIMDs to
    //  itself and only half the sequences would be resident: D = 32 only)
    if constexpr (D == 32) return help ? launch_fused_h<D, true>(a, st) : launch_fused_h<D, false>(a, st);
    else                   return launch_fused_h<D, false>(a, st);
}

#endif  // HPMN_LEGACY_KERNELS

bool gru_fused_fwd_supported(int H, int D, int gather) {
    (void)gather;
    return H == 64 && (D == 32 || D == 64);
}

int gru_fwd_mfma_dispatch(const HpmnGruFusedFwd &a, hipStream_t st);  // gru_fused_fwd3.hip

static int fused_gen() {
#ifdef HPMN_LEGACY_KERNELS
    static const int gen = [] { const char *e = getenv("HPMN_FUSED_FWD_GEN"); return e ? atoi(e) : 3; }();
    return gen;
#else
    return 3;
#endif
}
// does the launch write HpmnGruFusedFwd.last itself?
bool gru_fused_fwd_writes_last() { return fused_gen() >= 3; }

int gru_fused_fwd_dispatch(const HpmnGruFusedFwd &a, hipStream_t st) {
    // HPMN_FUSED_FWD_GEN: 3 (default) chain + MFMA-producer waves, two sequences per workgroup (gru_fused_fwd3.hip); 1 the
    // first generation below
    const int gen = fused_gen();
    if (gen >= 3) return gru_fwd_mfma_dispatch(a, st);
#ifdef HPMN_LEGACY_KERNELS
    if (a.last != nullptr || (a.flags & HPMN_FWD_NO_CANDIDATE)) return HPMN_EUNSUPPORTED;
    if (a.B == 0) return HPMN_OK;
    if (a.H != FH) return HPMN_EUNSUPPORTED;
    if (a.D == 32) return launch_fused<32>(a, st);
    if (a.D == 64) return launch_fused<64>(a, st);
#endif
    return HPMN_EUNSUPPORTED;
}

}  // namespace hpmn
