// Forward of TWO consecutive periodic-GRU layers in ONE launch, H = 64: layer i+1 runs WHILE layer i runs
// (code/hpmn.py:113-131: layer i+1 consumes every period-th output of layer i -- a pipeline, which one launch per layer
// serialises: at C3 the forward of layers 1-6 ran 526 us strictly AFTER the 465 us of layer 0).
//
// A workgroup owns two sequences and BOTH layers of them: eight waves, two per SIMD.
//
//   wave   role                              SIMD (a workgroup's waves go to SIMDs cyclically: wave w and w+4 share one)
//   0, 1   LOWER chain    of sequence 0, 1   a, b
//   2, 3   LOWER producer of sequence 0, 1   c, d
//   4, 5   UPPER chain    of sequence 0, 1   a, b      (flags & 1: the two upper roles swapped)
//   6, 7   UPPER producer of sequence 0, 1   c, d
//
// Chain and producer are the two roles of gru_fused_fwd3.hip (the recurrence, k-split products, LDS progress counters;
// the input projection on the matrix cores, the saved-state stores).  What is new is the hand-over between the layers:
// it never leaves the CU.  The lower producer, which reads every h_{t-1} out of the chain wave's state buffer anyway
// (it stores the saved states), also drops the rows that FIRE into a 32-row LDS ring and publishes their count; the upper
// producer fetches its 16-step blocks of input rows straight from that ring in MFMA operand layout (one ds_read_b128 per
// 16 features), and reports what it has taken so that the lower producer never overwrites a row that is still needed.
// No global flag, no agent-scope store, no dependence on dispatch order or placement -- the price list of
// MI355X_MICROARCH.md puts an in-launch hand-off between workgroups at 1-5 us per hop, an LDS counter costs a ds_read.
//
// Both waves of a SIMD share its VALU issue (priority, then age) and its matrix pipe: the lower layer, which sets the
// pace, runs at s_setprio 3 / 2, the upper layer -- half the steps, so half the time idle -- at 1 / 0 and sleeps in its
// polls.  Registers: two waves per SIMD leave each 256, which the chain wave (192 stationary recurrent weights) fits
// and a producer with 192 stationary projection weights (D = 64) does not; a D = 64 producer therefore re-reads its
// MFMA A operands for every tile of every 16-step block from a pre-arranged image in L2 (lane-contiguous 16-byte
// loads, written by pair_wimg_kernel in front of the launch: 48 KB per layer).
#include <cstdlib>
#include <type_traits>

#include "common.h"

namespace hpmn {

constexpr int QH = 64;        // hidden size
constexpr int QB = 16;        // steps per projection block (= MFMA N)
constexpr int QRING = 32;     // projected-input ring: the block the chain wave is on + the block being projected
constexpr int QNT = 12;       // 16-column tiles of [r | u | c]
constexpr int QSL = 4;        // slots of the state / gate hand-off buffers
constexpr int QYR = 32;       // rows of the inter-layer ring
constexpr int QYROW = 68;     // floats per row of it (64 + 4: rows 16 apart in one ds_read_b128 lane group stay off one bank)

typedef float q4 __attribute__((ext_vector_type(4)));

enum { SRC_GATHER = 0, SRC_GLOBAL = 1, SRC_LDS = 2 };

template <bool TRAIN>
struct SeqLds {                                   // one sequence of one layer
    float ring[QRING][3 * QH + 4];                // xp (r | u | c), exponent domain; rows padded off the bank period (gru_fused_fwd3.hip)
    float hb[QSL][QH];                            // h_{t-1} in hb[t % QSL]
    float rhb[QH];
    v4f rcb[TRAIN ? QSL : 1][QH];                 // r, u, c of step t in rcb[t % QSL]
    int produced, h_pub, u_pub, pad;
};
struct YLds {                                     // lower -> upper rows of one sequence
    float row[QYR][QYROW];
    int pub, taken, pad0, pad1;
};

// ---------------------------------------------------------------------------------------------------- chain wave
template <bool TRAIN, bool SLEEPY>
__device__ __forceinline__ void pair_chain_wave(const HpmnGruFusedFwd &a, SeqLds<TRAIN> &S, const int lane) {
    constexpr int H = QH;
    const int T = a.T, D = a.D, l = lane;
    f2 whr[2][16], whu[2][16], whc[2][16];
    split_matvec_weights_t<2>(a.wg + (long)D * 2 * H, 2 * H, NEG_LOG2E, lane, whr);
    split_matvec_weights_t<2>(a.wg + (long)D * 2 * H + H, 2 * H, NEG_LOG2E, lane, whu);
    split_matvec_weights_t<2>(a.wc + (long)D * H, H, 2.0f * NEG_LOG2E, lane, whc);
    {
        int seen = 0;
        const int need = QB < T ? QB : T;
        while (seen < need) {
            seen = lds_counter_peek(&S.produced);
            if (SLEEPY && seen < need) __builtin_amdgcn_s_sleep(4);
        }
        asm volatile("" ::: "memory");
    }
    float h = 0.f;
    int u_seen = 0, p_seen = 0;
    float xr = S.ring[0][l], xu = S.ring[0][H + l], xcand = S.ring[0][2 * H + l];

    auto step = [&](int t, int p) {
        // The lane index is RECOMPUTED in every step (two v_mbcnt, opaque to the optimiser): with 192 stationary weights and the
        // LDS read groups the loop has no register left for loop-invariant address terms, and the training instantiations that
        // start at layer 0 kept `lane` and its k-group offset in SCRATCH -- a scratch_load + s_waitcnt vmcnt(0) in every step
        // of the serial chain (tools/check_resources.py reports the kernel's scratch, the ISA shows where it is touched).
        int lane, l;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
        __builtin_assume(lane >= 0 && lane < 64);
        l = lane;
        float sr, su;
        split_matvec2x(&S.hb[p][0], whr, whu, lane, sr, su);
        const float r = sigmoid_scaled(xr + sr);
        const float u = sigmoid_scaled(xu + su);
        S.rhb[lane] = r * h;
        wave_sync();
        const float cc = tanh_scaled(xcand + split_matvec<2>(&S.rhb[0], whc, lane));
        // crossing into the next 16-step block: the producer has published it as projected
        if (((t + 1) & (QB - 1)) == 0 && t + 1 < T) {
            while (p_seen <= t + 1) {
                p_seen = lds_counter_peek(&S.produced);
                if (SLEEPY && p_seen <= t + 1) __builtin_amdgcn_s_sleep(2);
            }
            asm volatile("" ::: "memory");
        }
        const float *nx = S.ring[(t + 1) & (QRING - 1)];
        xr = nx[l];
        xu = nx[H + l];
        xcand = nx[2 * H + l];
        // the slots this step is about to overwrite (h_{t-4}, gates of step t-4) were read by the producer's iteration t-3
        while (u_seen < t - (QSL - 2)) {
            u_seen = lds_counter_peek(&S.u_pub);
            if (SLEEPY && u_seen < t - (QSL - 2)) __builtin_amdgcn_s_sleep(1);
        }
        asm volatile("" ::: "memory");
        h = fmaf(u, h - cc, cc);
        S.hb[(p + 1) & (QSL - 1)][lane] = h;
        if constexpr (TRAIN) S.rcb[p][l] = v4f{r, u, cc, 0.f};
        lds_counter_set(&S.h_pub, t + 1);
        wave_sync();
    };
    static_assert(QSL == 4, "the loop below is unrolled by the slot count");
    const int nfull = T >> 2;
    for (int q = 0; q < nfull; ++q) {
        step(4 * q, 0);
        step(4 * q + 1, 1);
        step(4 * q + 2, 2);
        step(4 * q + 3, 3);
    }
    if ((T & 3) > 0) step(4 * nfull, 0);
    if ((T & 3) > 1) step(4 * nfull + 1, 1);
    if ((T & 3) > 2) step(4 * nfull + 2, 2);
}

// ---------------------------------------------------------------------------------------------------- producer wave
// D: input width; SRC: where the input rows come from; YOUT: the rows that fire also go to the inter-layer ring `yo`;
// WIMG: A operands re-read per tile from the image `wimg` ([QNT][D/16][64 lanes][4]; bias image behind it) instead of
// held in registers.
template <int D, int SRC, bool TRAIN, bool YOUT, bool WIMG, bool SLEEPY>
__device__ __forceinline__ void pair_producer_wave(const HpmnGruFusedFwd &a, SeqLds<TRAIN> &S, YLds *yi, YLds *yo,
                                                   const float *wimg, const long b, const int lane, const bool alone = false) {
    constexpr int H = QH;
    constexpr int NJ = D / 16;
    const int T = a.T, l = lane;
    const int n16 = lane & 15, g = lane >> 4;
    constexpr int NST = WIMG ? 1 : QNT;
    float wA[NST][NJ][4], wBias[NST];
    if constexpr (!WIMG) {
#pragma unroll
        for (int ct = 0; ct < QNT; ++ct) {
            const int col = 16 * ct + n16;
#pragma unroll
            for (int kq = 0; kq < NJ; ++kq)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const long f = 16 * kq + 4 * g + c;
                    wA[ct][kq][c] = ct < 8 ? a.wg[f * 2 * H + col] * NEG_LOG2E : a.wc[f * H + (col - 2 * H)] * (2.0f * NEG_LOG2E);
                }
            const float bias = ct < 8 ? a.bg[col] * NEG_LOG2E : a.bc[col - 2 * H] * (2.0f * NEG_LOG2E);
            wBias[ct] = g == 0 ? bias : 0.f;
        }
#pragma unroll
        for (int ct = 0; ct < QNT; ++ct) {
            settle(wBias[ct]);
#pragma unroll
            for (int kq = 0; kq < NJ; ++kq)
#pragma unroll
                for (int c = 0; c < 4; ++c) settle(wA[ct][kq][c]);
        }
    }
    float one = g == 0 ? 1.f : 0.f;
    settle(one);
    const q4 *wimg4 = reinterpret_cast<const q4 *>(wimg) + lane;              // tile (ct, kq) at [(ct * NJ + kq) * 64]
    const float *bimg = wimg + (long)QNT * NJ * 256 + lane;                    // bias of tile ct at [ct * 64]

    struct Rows { v4f v[NJ]; bool keep[NJ]; };
    auto fetch_ids = [&](int s0, long (&id)[NJ]) {
        if constexpr (SRC == SRC_GATHER) {
            int t = s0 + n16;
            t = t < T ? t : T - 1;
            const int ti = t - a.front_zero;
#pragma unroll
            for (int kq = 0; kq < NJ; ++kq)
                id[kq] = load_id(a.ids, (b * (long)a.Tids + (ti > 0 ? ti : 0)) * a.F + (16 * kq + 4 * g) / a.E, a.mask_id0);
        }
    };
    auto fetch_rows = [&](int s0, const long (&id)[NJ], Rows &r) {
        int t = s0 + n16;
        t = t < T ? t : T - 1;
#pragma unroll
        for (int kq = 0; kq < NJ; ++kq) {
            const int e0 = 16 * kq + 4 * g;
            if constexpr (SRC == SRC_GATHER) {
                r.v[kq] = *reinterpret_cast<const v4f *>(a.emb + id[kq] * a.E + e0 % a.E);
                r.keep[kq] = (t >= a.front_zero) && !id_masked(id[kq], a.mask_id0);
            } else if constexpr (SRC == SRC_GLOBAL) {
                r.v[kq] = *reinterpret_cast<const v4f *>(a.x + (b * (long)T + t) * D + e0);
                r.keep[kq] = true;
            } else {
                r.v[kq] = *reinterpret_cast<const v4f *>(&yi->row[t & (QYR - 1)][e0]);
                r.keep[kq] = true;
            }
        }
    };
    // SRC_LDS: the rows of block s0 exist once the lower layer has published min(s0 + 16, T) of them; taken at once
    int y_seen = 0;
    auto take_rows = [&](int s0, Rows &r) {
        const int need = s0 + QB < T ? s0 + QB : T;
        while (y_seen < need) {
            y_seen = lds_counter_peek(&yi->pub);
            if (y_seen < need) __builtin_amdgcn_s_sleep(8);
        }
        asm volatile("" ::: "memory");
        long none[NJ];
        fetch_rows(s0, none, r);
        lds_counter_set(&yi->taken, need);            // (LDS executes a wave's operations in order: the reads are done)
    };
    auto finish_rows = [&](int s0, Rows &r) {
#pragma unroll
        for (int kq = 0; kq < NJ; ++kq) {
            if constexpr (SRC == SRC_GATHER) {
                if (!r.keep[kq]) r.v[kq] = v4f{0.f, 0.f, 0.f, 0.f};
                if (a.x_out != nullptr && s0 + n16 < T)
                    *reinterpret_cast<v4f *>(a.x_out + (b * (long)T + s0 + n16) * D + 16 * kq + 4 * g) = r.v[kq];
                if (a.last != nullptr && s0 + n16 == a.last_t)
                    *reinterpret_cast<v4f *>(a.last + b * (long)D + 16 * kq + 4 * g) = r.v[kq];
            }
        }
    };
    auto load_tile_w = [&](int ct, q4 (&w)[NJ], float &bias) {
#pragma unroll
        for (int kq = 0; kq < NJ; ++kq) w[kq] = wimg4[(ct * NJ + kq) * 64];
        bias = bimg[ct * 64];
    };
    auto tile_mfma = [&](const q4 (&w)[NJ], float bias, const Rows &r) -> q4 {
        q4 acc = {0.f, 0.f, 0.f, 0.f};
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(bias, one, acc, 0, 0, 0);
#pragma unroll
        for (int kq = 0; kq < NJ; ++kq)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w[kq][c], r.v[kq][c], acc, 0, 0, 0);
        return acc;
    };
    auto tile_stationary = [&](int ct, const Rows &r) -> q4 {
        q4 acc = {0.f, 0.f, 0.f, 0.f};
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wBias[WIMG ? 0 : ct], one, acc, 0, 0, 0);
#pragma unroll
        for (int kq = 0; kq < NJ; ++kq)
#pragma unroll
            for (int c = 0; c < 4; ++c)
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wA[WIMG ? 0 : ct][kq][c], r.v[kq][c], acc, 0, 0, 0);
        return acc;
    };

    long idA[NJ], idB[NJ];
    Rows rA, rB;
    q4 wT[NJ];          // WIMG: the A operands of the tile the next iteration issues
    float bT = 0.f;

    const int period = a.period;
    const bool has_y = a.y != nullptr;
    int next_fire = period;
    float *yp = has_y ? a.y + (b * (long)(T / period)) * H + l : a.h_last + b * a.h_last_stride + l;
    const int y_adv = has_y ? H : 0;
    float *hsp = TRAIN ? a.hs + (b * (long)(T + 1)) * H + l : nullptr;
    float *gp = TRAIN ? a.gates + (b * (long)T) * 3 * H + l : nullptr;
    int h_seen = 0;
    int yrow = 0, taken_seen = 0;
    const bool keep_c = !(a.flags & HPMN_FWD_NO_CANDIDATE);                // (wave-uniform)

    // iteration t: the saved rows of step t-1 (h_{t-1} from the chain wave's state buffer, r,u,c from the hand-off), the
    // row that fires; TILE >= 0: one 16-column tile of the NEXT block's projection is issued right behind the wait
    auto iteration = [&](int t, auto tile_c, int s_next) {
        constexpr int TILE = decltype(tile_c)::value;
        while (h_seen < t) {
            h_seen = lds_counter_peek(&S.h_pub);
            if (SLEEPY && h_seen < t) __builtin_amdgcn_s_sleep(1);
        }
        asm volatile("" ::: "memory");
        const int p = t & (QSL - 1), pm = (t - 1) & (QSL - 1);
        q4 acc = {0.f, 0.f, 0.f, 0.f};
        if constexpr (TILE >= 0) {
            if constexpr (WIMG) {
                acc = tile_mfma(wT, bT, rA);
                load_tile_w(TILE + 1 < QNT ? TILE + 1 : 0, wT, bT);      // (tile 0 again: the next block's first)
            } else {
                acc = tile_stationary(TILE, rA);
            }
        }
        float hprev = S.hb[p][l];
        v4f rc = {0.f, 0.f, 0.f, 0.f};
        if constexpr (TRAIN) rc = S.rcb[pm][l];
        asm volatile("" : "+v"(hprev), "+v"(rc));          // (the reads have landed)
        lds_counter_set(&S.u_pub, t + 1);                   // "iteration t has read its inputs"
        if constexpr (TRAIN) {
            *hsp = hprev;
            hsp += H;
            gp[0] = rc.x;
            gp[H] = rc.y;
            if (keep_c) gp[2 * H] = rc.z;               // (HPMN_FWD_NO_CANDIDATE, as in gru_fused_fwd3.hip)
            gp += t > 0 ? 3 * H : 0;
        }
        *yp = hprev;
        const bool fire = t == next_fire;
        next_fire += fire ? period : 0;
        yp += fire ? y_adv : 0;
        if constexpr (YOUT) {
            yo->row[yrow & (QYR - 1)][l] = hprev;           // (the slot of the next row to fire: final when it does)
            yrow += fire ? 1 : 0;
            lds_counter_set(&yo->pub, yrow);
        }
        if constexpr (TILE >= 0) {
            *reinterpret_cast<q4 *>(&S.ring[(s_next + n16) & (QRING - 1)][16 * TILE + 4 * g]) = acc;
            if constexpr (TILE == QNT - 1) lds_counter_set(&S.produced, s_next + QB);
        }
    };
    // all twelve tiles of the block at s0 in one go (rows in r): block 0, before the loop.  (Doing this for EVERY block of
    // the upper layer -- run the block it has, then project the next the moment its rows arrive, which would keep the upper
    // chain one block closer to the lower -- was measured: 599 vs 519 us for layers 0+1 at C3; the twelve tiles back to back
    // hold the matrix pipe of the SIMD the lower producer lives on for 2.7 us per block, and the lower chain waits it out.)
    auto project_block = [&](int s0, const Rows &r) {
        if constexpr (WIMG) {
            // (the A operands of tile ct + 3 leave L2 while tile ct is on the matrix pipe: twelve dependent round trips
            //  in a row made this 8 us instead of 3)
            q4 wq[3][NJ];
            float bq[3];
            load_tile_w(0, wq[0], bq[0]);
            load_tile_w(1, wq[1], bq[1]);
            load_tile_w(2, wq[2], bq[2]);
#pragma unroll
            for (int ct = 0; ct < QNT; ++ct) {
                const q4 acc = tile_mfma(wq[ct % 3], bq[ct % 3], r);
                if (ct + 3 < QNT) load_tile_w(ct + 3, wq[ct % 3], bq[ct % 3]);
                *reinterpret_cast<q4 *>(&S.ring[(s0 + n16) & (QRING - 1)][16 * ct + 4 * g]) = acc;
            }
        } else {
#pragma unroll
            for (int ct = 0; ct < QNT; ++ct) {
                const q4 acc = tile_stationary(ct, r);
                *reinterpret_cast<q4 *>(&S.ring[(s0 + n16) & (QRING - 1)][16 * ct + 4 * g]) = acc;
            }
        }
        lds_counter_set(&S.produced, s0 + QB);
    };
    auto full_block = [&](int s0, auto... is) {
        (iteration(s0 + decltype(is)::value, std::integral_constant<int, (decltype(is)::value < QNT ? decltype(is)::value : -1)>{},
                   s0 + QB), ...);
    };
    // block 0 before the loop
    {
        if constexpr (SRC == SRC_LDS) {
            take_rows(0, rA);
        } else {
            fetch_ids(0, idA);
            fetch_ids(QB, idB);
            fetch_rows(0, idA, rA);
            fetch_ids(2 * QB, idA);
            fetch_rows(QB, idB, rB);
        }
        finish_rows(0, rA);
        project_block(0, rA);
        if constexpr (SRC != SRC_LDS) rA = rB;        // rows of block 1 (in flight); idA: ids of block 2
        if constexpr (WIMG) load_tile_w(0, wT, bT);
    }

    for (int s0 = 0; s0 < T; s0 += QB) {
        if (YOUT && !alone) {
            // the ring slots this block writes (rows up to (s0 + 16) / period) must have been taken by the upper layer
            const int idx_max = (s0 + QB) / period;
            while (taken_seen <= idx_max - QYR) {
                taken_seen = lds_counter_peek(&yo->taken);
                if (taken_seen <= idx_max - QYR) __builtin_amdgcn_s_sleep(2);
            }
            asm volatile("" ::: "memory");
        }
        using std::integral_constant;
        if constexpr (SRC == SRC_LDS) {
            if (s0 + QB < T) take_rows(s0 + QB, rA);
        } else {
            fetch_rows(s0 + 2 * QB, idA, rB);
            fetch_ids(s0 + 3 * QB, idA);
        }
        if (s0 + QB < T) {                            // wave-uniform
            finish_rows(s0 + QB, rA);
            {
                full_block(s0, integral_constant<int, 0>{}, integral_constant<int, 1>{}, integral_constant<int, 2>{},
                           integral_constant<int, 3>{}, integral_constant<int, 4>{}, integral_constant<int, 5>{},
                           integral_constant<int, 6>{}, integral_constant<int, 7>{}, integral_constant<int, 8>{},
                           integral_constant<int, 9>{}, integral_constant<int, 10>{}, integral_constant<int, 11>{},
                           integral_constant<int, 12>{}, integral_constant<int, 13>{}, integral_constant<int, 14>{},
                           integral_constant<int, 15>{});
            }
        } else {                                      // the last block: nothing left to project, maybe partial
#pragma unroll
            for (int i = 0; i < QB; ++i)
                if (s0 + i < T) iteration(s0 + i, std::integral_constant<int, -1>{}, 0);
        }
        if constexpr (SRC != SRC_LDS) rA = rB;
    }
    // the rows of the last step
    while (h_seen < T) {
        h_seen = lds_counter_peek(&S.h_pub);
        if (SLEEPY && h_seen < T) __builtin_amdgcn_s_sleep(1);
    }
    asm volatile("" ::: "memory");
    {
        const float hlast = S.hb[T & (QSL - 1)][l];
        if constexpr (TRAIN) {
            const v4f rc = S.rcb[(T - 1) & (QSL - 1)][l];
            *hsp = hlast;
            gp[0] = rc.x;
            gp[H] = rc.y;
            if (keep_c) gp[2 * H] = rc.z;
        }
        *yp = hlast;                                  // T is a multiple of period: the last output row (or h_last)
        a.h_last[b * a.h_last_stride + l] = hlast;
        if constexpr (YOUT) {
            yo->row[yrow & (QYR - 1)][l] = hlast;
            lds_counter_set(&yo->pub, yrow + 1);
        }
    }
}

struct PairFwdArgs {
    HpmnGruFusedFwd lo, up;
    const float *wimg_lo, *wimg_up;      // A-operand images (pair_wimg_kernel) of the producers that stream them
    int32_t flags, pad;
};

template <int D0, int SRC0, bool TRAIN>
__global__ __launch_bounds__(512, 2) void gru_pair_fwd_kernel(const PairFwdArgs p) {
    __shared__ __attribute__((aligned(16))) SeqLds<TRAIN> lds_[2][2];      // [layer][sequence]
    __shared__ __attribute__((aligned(16))) YLds y_[2];

    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // wave-uniform, and known to be
    // (r6) flags bit 2: ONE sequence per workgroup, four waves (launched with 256 threads): the roles land on four SIMDs instead of
    // sharing two -- for batches that leave the chip's CUs to spare (B <= CUs: Taobao's 128), where two chain waves per SIMD
    // cost the pace-setting layer 8-17 % (DESIGN_HISTORY 3.12) and buy nothing
    const bool single = (p.flags & 4) != 0;
    const int seq = single ? 0 : (w & 1);
    int role = single ? w : (w >> 1);                               // 0 lower chain, 1 lower producer, 2 upper chain, 3 upper producer
    if ((p.flags & 1) && role >= 2) role ^= 1;
    const long b = single ? (long)blockIdx.x : 2 * (long)blockIdx.x + seq;
    if (b >= p.lo.B) return;                         // odd batch (before the barrier: ended waves do not take part in it)
    const bool alone = (p.flags & 2) != 0;           // (measurement: the lower layer only)
    if (alone && role >= 2) return;
    SeqLds<TRAIN> &SL = lds_[0][seq], &SU = lds_[1][seq];
    YLds &Y = y_[seq];
    if (role == 0) {
        SL.hb[0][lane] = 0.f;
        if (lane == 0) { SL.produced = 0; SL.h_pub = 0; SL.u_pub = 0; Y.pub = 0; Y.taken = 0; }
    } else if (role == 2) {
        SU.hb[0][lane] = 0.f;
        if (lane == 0) { SU.produced = 0; SU.h_pub = 0; SU.u_pub = 0; }
    }
    __syncthreads();                                 // the only barrier

    if (role == 0) {
        __builtin_amdgcn_s_setprio(3);
        pair_chain_wave<TRAIN, false>(p.lo, SL, lane);
    } else if (role == 1) {
        __builtin_amdgcn_s_setprio(2);
        pair_producer_wave<D0, SRC0, TRAIN, true, (D0 > 32), false>(p.lo, SL, nullptr, &Y, p.wimg_lo, b, lane, alone);
    } else if (role == 2) {
        __builtin_amdgcn_s_setprio(1);
        pair_chain_wave<TRAIN, true>(p.up, SU, lane);
    } else {
        __builtin_amdgcn_s_setprio(0);
        pair_producer_wave<QH, SRC_LDS, TRAIN, false, true, true>(p.up, SU, &Y, nullptr, p.wimg_up, b, lane);
    }
}

// A-operand image of one layer's input projection for the producers that stream it: tile (ct, kq), lane (n16, g),
// component c  <-  W[feature 16 kq + 4 g + c][column 16 ct + n16] in the exponent domain; behind the tiles the bias of
// every tile as the A operand of its extra k-step (lane group g == 0 only).  blockIdx.y = layer of the batch.
struct WimgBatch {
    const float *wg[HPMN_MAX_LAYERS], *bg[HPMN_MAX_LAYERS], *wc[HPMN_MAX_LAYERS], *bc[HPMN_MAX_LAYERS];
    float *img[HPMN_MAX_LAYERS];
    int D[HPMN_MAX_LAYERS];
};

__global__ void pair_wimg_kernel(const WimgBatch w) {
    constexpr int H = QH;
    const int L = blockIdx.y;
    const int D = w.D[L], NJ = D / 16;
    const float *wg = w.wg[L], *bg = w.bg[L], *wc = w.wc[L], *bc = w.bc[L];
    float *img = w.img[L];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int ntile = QNT * NJ * 64;
    if (i < ntile) {
        const int lane = i & 63, kq = (i >> 6) % NJ, ct = (i >> 6) / NJ;
        const int n16 = lane & 15, g = lane >> 4, col = 16 * ct + n16;
        q4 v;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const long f = 16 * kq + 4 * g + c;
            v[c] = ct < 8 ? wg[f * 2 * H + col] * NEG_LOG2E : wc[f * H + (col - 2 * H)] * (2.0f * NEG_LOG2E);
        }
        reinterpret_cast<q4 *>(img)[i] = v;
    } else if (i < ntile + QNT * 64) {
        const int j = i - ntile, lane = j & 63, ct = j >> 6;
        const int n16 = lane & 15, g = lane >> 4, col = 16 * ct + n16;
        const float bias = ct < 8 ? bg[col] * NEG_LOG2E : bc[col - 2 * H] * (2.0f * NEG_LOG2E);
        img[(long)QNT * NJ * 256 + j] = g == 0 ? bias : 0.f;
    }
}

size_t gru_proj_image_floats(int D) { return (size_t)QNT * (D / 16) * 256 + QNT * 64; }

bool gru_pair_fwd_supported(int H, int D_lo, int gather) {
    return H == QH && ((D_lo == 32 && gather) || D_lo == 64);
}

// images of n layers in ONE launch (D[i] in {32, 64}, H = 64); img[i]: gru_proj_image_floats(D[i]) floats, 16-byte aligned
int gru_proj_images_launch(int n, const float *const *wg, const float *const *bg, const float *const *wc,
                           const float *const *bc, const int *D, float *const *img, hipStream_t st) {
    if (n < 1 || n > HPMN_MAX_LAYERS) return HPMN_EINVAL;
    WimgBatch w = {};
    for (int i = 0; i < n; ++i) {
        if (D[i] != 32 && D[i] != 64) return HPMN_EUNSUPPORTED;
        w.wg[i] = wg[i]; w.bg[i] = bg[i]; w.wc[i] = wc[i]; w.bc[i] = bc[i]; w.img[i] = img[i]; w.D[i] = D[i];
    }
    const int nmax = QNT * 4 * 64 + QNT * 64;
    hipLaunchKernelGGL(pair_wimg_kernel, dim3((nmax + 255) / 256, n), dim3(256), 0, st, w);
    return check_launch();
}

// scratch of a launch that builds its own images
size_t gru_pair_fwd_scratch_bytes() { return 2 * gru_proj_image_floats(64) * sizeof(float); }

// img_lo / img_up: ready-made images (gru_proj_images_launch) or NULL: built here, into scratch
// (r6) one sequence per workgroup where the batch leaves at least half of the CUs empty even so: HPMN_PAIR_SINGLE=0 / 1, default
// B <= CUs / 2 (measured: C3 at 128 sequences 2.03 -> 1.84 ms/step, C2 at 128 0.917 -> 0.911; at 256 sequences -- every CU taken by
// a four-wave workgroup, none left for the weight gradients -- 0.97 -> 1.02)
static bool pair_single_seq(int B) {
    static const int env = [] { const char *e = getenv("HPMN_PAIR_SINGLE"); return e ? atoi(e) : -1; }();
    if (env >= 0) return env != 0;
    static const int cus = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess) return 256;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) return 256;
        return n;
    }();
    return 2 * B <= cus;
}

int gru_pair_fwd_launch(const HpmnGruFusedFwd &lo, const HpmnGruFusedFwd &up, int flags, float *scratch,
                        const float *img_lo, const float *img_up, hipStream_t st) {
    PairFwdArgs p = {};
    p.lo = lo; p.up = up; p.flags = flags;
    const bool need_lo = lo.D > 32;
    if (img_up == nullptr || (need_lo && img_lo == nullptr)) {
        if (scratch == nullptr) return HPMN_EINVAL;
        float *imgs[2] = {scratch, scratch + gru_proj_image_floats(64)};
        const float *wg[2] = {up.wg, lo.wg}, *bg[2] = {up.bg, lo.bg}, *wc[2] = {up.wc, lo.wc}, *bc[2] = {up.bc, lo.bc};
        const int D[2] = {up.D, lo.D};
        const int rc = gru_proj_images_launch(need_lo ? 2 : 1, wg, bg, wc, bc, D, imgs, st);
        if (rc != HPMN_OK) return rc;
        img_up = imgs[0];
        img_lo = imgs[1];
    }
    p.wimg_up = img_up;
    p.wimg_lo = need_lo ? img_lo : nullptr;
    const bool train = lo.hs != nullptr;
    const bool single = pair_single_seq(lo.B);
    if (single) p.flags |= 4;
    const dim3 grid(single ? lo.B : (lo.B + 1) / 2), blk(single ? 256 : 512);
    const bool gather = lo.x == nullptr;
#define PAIR_LAUNCH(D0, SRC0)                                                                              \
    do {                                                                                                     \
        if (train) hipLaunchKernelGGL((gru_pair_fwd_kernel<D0, SRC0, true>), grid, blk, 0, st, p);           \
        else       hipLaunchKernelGGL((gru_pair_fwd_kernel<D0, SRC0, false>), grid, blk, 0, st, p);          \
    } while (0)
    if (lo.D == 32 && gather) PAIR_LAUNCH(32, SRC_GATHER);
    else if (lo.D == 64 && gather) PAIR_LAUNCH(64, SRC_GATHER);
    else if (lo.D == 64) PAIR_LAUNCH(64, SRC_GLOBAL);
    else return HPMN_EUNSUPPORTED;
#undef PAIR_LAUNCH
    return check_launch();
}

}  // namespace hpmn
