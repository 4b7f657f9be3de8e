// Reverse (BPTT) scan of TWO consecutive periodic-GRU layers in ONE launch, H = 64: the lower layer's reverse scan runs
// WHILE the upper layer's does (the mirror of gru_pair_fwd.hip; reference semantics code/hpmn.py:113-131 under TF's
// autodiff: the gradient wrt every period-th output of layer i is the input gradient of layer i+1).  One launch per layer
// serialises that pipeline: at C3 the reverse scans of layers 6..1 ran 726 us strictly BEFORE the 589 us of layer 0.
//
// A workgroup owns two sequences and both layers of them, eight waves, two per SIMD (wave w and w + 4 share one):
//
//   wave   role
//   0, 1   LOWER chain  of sequence 0, 1     the serial recurrence of gru_scan_bwd_feed.hip, unchanged
//   2, 3   LOWER feeder of sequence 0, 1     coefficients, e_u, d_act stores; its d_y rows come from the LDS ring below
//   4, 5   UPPER chain
//   6, 7   UPPER feeder                      ... and the upper layer's INPUT GRADIENT, interleaved with its iterations
//
// The hand-over never leaves the CU.  The upper chain wave keeps its operand rows [da_r | da_u | dc_pre] in a 32-row LDS
// ring instead of two parity slots; while it fills one half (16 iterations), the upper feeder multiplies the other half by
// [Wg[:D] | Wc[:D]]^T on the matrix cores -- 12 of the block's 192 v_mfma_f32_16x16x4_f32 per iteration, B operands 16
// bytes per lane straight out of the ring, A operands 16 bytes per lane straight out of the weight matrices in L2 (a row
// of Wg is contiguous in the gate index) -- and drops the 16 finished rows of d_x into a second 32-row ring, which is where
// the lower feeder picks up the d_y of its firing steps.  Two LDS counters (rows published / rows taken) keep either side
// from overrunning the other; nothing depends on dispatch order.  The upper layer's d_x is never written to memory (nobody
// else reads it), its d_act is (the weight gradient does).
//
// The interleaving (not one burst of 192 matrix instructions per block) is what the forward pair measured: back-to-back
// matrix instructions of a co-resident wave cost the latency-critical layer their full duration (gru_pair_fwd.hip).
// (Also measured: holding each chunk back until the LOWER feeder on the same SIMD has just published its e_u, i.e. aiming the
//  matrix instructions at that wave's idle window -- 378.8 / 388.8 us for layers 2+1 at C3 with a wait of up to 4 / 12 polls
//  against 378.2 without: no gain, removed.)
//
// Epilogue: the LOWER layer's input gradient, as in gru_scan_bwd_feed.hip -- but the upper layer's two waves have long
// finished by then and take half of the blocks.
#include <cstdlib>

#include "common.h"

namespace hpmn {

constexpr int RH = 64;
constexpr int RFS = 8;          // coefficient ring depth in steps
constexpr int RAHEAD = 3;       // chunks the feeder parks ahead of the chunk the chain wave is on
constexpr int RROW = 192 + 4;   // floats per operand row [da_r | da_u | dc_pre], padded off the bank period (gru_scan_bwd_feed.hip: DROW)
constexpr int RXB = 16;         // steps per input-gradient block (= MFMA N)
constexpr int RDR = 32;         // rows of the upper layer's operand ring and of the d_y ring
constexpr int RDYROW = 68;      // floats per d_y ring row (64 + 4, see gru_pair_fwd.hip)

typedef float r4 __attribute__((ext_vector_type(4)));

template <int NS>
struct BwdLds {                                   // one sequence of one layer; NS operand rows
    v4f ringA[RFS][RH];                           // dy, k1, k2, k3
    f2 ringB[RFS][RH];                            // r, u
    float dact[NS][RROW];                         // da_r | da_u | dc_pre of iteration k in row k % NS
    float eU[2][RH];
    int dau_pub, eu_pub, fed, pad;
};
struct DyLds {                                    // upper -> lower: d_x rows of the upper layer, by upper ITERATION index
    float row[RDR][RDYROW];
    int pub, taken, pad0, pad1;
};

// ------------------------------------------------------------------------------------------------ chain wave (either layer)
template <int NS, bool SLEEPY>
__device__ __forceinline__ void pair_bwd_chain(const HpmnGruBwd &a, BwdLds<NS> &S, const long b, const int lane) {
    constexpr int H = RH;
    const int T = a.T, D = a.D, l = lane;
    f2 wcS[2][16], wrS[2][16];
    split_matvec_weights<2>(a.wc + (long)D * H, H, lane, wcS);
    split_matvec_weights<2>(a.wg + (long)D * 2 * H, 2 * H, lane, wrS);
    float dh = a.d_h_last[b * a.d_h_last_stride + l];
    settle(dh);
    int fed_seen = 0, eu_seen = 0;
    auto wait_fed = [&](int need) {
        while (fed_seen < need) {
            fed_seen = lds_counter_peek(&S.fed);
            if (fed_seen < need) __builtin_amdgcn_s_sleep(SLEEPY ? 2 : 1);
        }
        asm volatile("" ::: "memory");
    };
    wait_fed(1);
    v4f ca = S.ringA[0][l];
    f2 cb = S.ringB[0][l];

    auto step = [&](int k, int p) {
        float *row = S.dact[NS == 2 ? p : (k & (NS - 1))];
        const float dhin = dh + ca.x;
        const float dcp = dhin * ca.y;
        const float dau = dhin * ca.z;
        const float k3 = ca.w, r = cb.x, u = cb.y;
        row[H + l] = dau;
        row[2 * H + l] = dcp;
        lds_counter_set(&S.dau_pub, k + 1);                          // the feeder may start on e_u(k)
        wave_sync();
        const float drh = split_matvec<2>(row + 2 * H, wcS, lane);
        row[l] = drh * k3;
        wave_sync();
        wait_fed(k + 2);
        const int slot = (k + 1) & (RFS - 1);
        ca = S.ringA[slot][l];
        cb = S.ringB[slot][l];
        const float er = split_matvec<2>(row, wrS, lane);
        const float part = fmaf(dhin, u, fmaf(drh, r, er));
        while (eu_seen <= k) {
            eu_seen = lds_counter_peek(&S.eu_pub);
            if (eu_seen <= k) __builtin_amdgcn_s_sleep(1);
        }
        asm volatile("" ::: "memory");
        dh = part + S.eU[p][l];
        wave_sync();
    };
    const int nfull = T >> 1;
    for (int q = 0; q < nfull; ++q) {
        step(2 * q, 0);
        step(2 * q + 1, 1);
    }
    if (T & 1) step(T - 1, 0);
    lds_counter_set(&S.dau_pub, T + 1);                              // da_r of the last step is in LDS
}

// ------------------------------------------------------------------------------------------------ feeder wave (either layer)
// DY_LDS: the d_y rows of the firing steps come from the upper layer's ring `yi` (the lower layer of the pair), else from
// a.d_y / a.d_h_last in memory.  DXOUT: this is the upper layer -- it also produces its input gradient into `yo`.
template <int NS, bool DY_LDS, bool DXOUT, bool SLEEPY, bool CFH>
__device__ __forceinline__ void pair_bwd_feeder(const HpmnGruBwd &a, BwdLds<NS> &S, DyLds *yi, DyLds *yo, const long b,
                                                const int lane) {
    constexpr int H = RH;
    const int T = a.T, D = a.D, l = lane;
    f2 wuS[2][16];
    split_matvec_weights<2>(a.wg + (long)D * 2 * H + H, 2 * H, lane, wuS);

    const int period = a.period;
    const bool has_dy = DY_LDS || a.d_y != nullptr;
    const float *gb = a.gates + b * (long)T * 3 * H + l;
    const float *hsb = a.hs + b * (long)(T + 1) * H + l;
    const float *dyb = nullptr;
    long dy_stride = 0;
    if constexpr (!DY_LDS) {
        dyb = a.d_y != nullptr ? a.d_y + b * (long)(T / period) * H + l : a.d_h_last + b * a.d_h_last_stride + l;
        dy_stride = a.d_y != nullptr ? H : 0;
    }
    // iteration m = step T-1-m fires iff m % period == 0 (T is a multiple of period); it is the (m / period)-th firing
    // step from the end: d_y row T/period - 1 - m/period in memory, row m/period of the upper layer's ring
    int pf_next = 0, pf_cnt = 0;
    int dy_seen = 0;

    // (HPMN_BWD_CANDIDATE_FROM_HS: c[] holds the state after the step instead of the candidate, see gru_scan_bwd_feed.hip)
    constexpr bool c_from_hs = CFH;                    // (a template switch: see gru_scan_bwd_feed.hip)
    float h_after = c_from_hs ? hsb[(long)T * H] : 0.f;
    struct Raw { float r[2], u[2], c[2], hp[2], dy[2]; bool m[2]; };
    auto load_chunk = [&](int q, Raw &w) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int m = 2 * q + j;
            const int t_raw = T - 1 - m;
            const int t = t_raw > 0 ? t_raw : 0;
            const float *g = gb + (long)t * 3 * H;
            w.r[j] = g[0];
            w.u[j] = g[H];
            w.hp[j] = hsb[(long)t * H];
            w.c[j] = c_from_hs ? 0.f : g[2 * H];
            const bool fire = has_dy && m == pf_next && m < T;
            if constexpr (DY_LDS) {
                // (no branch around the wait: need = -1 never waits; the cached count answers 15 times out of 16)
                const int need = fire ? pf_cnt : -1;
                while (dy_seen <= need) {
                    dy_seen = lds_counter_peek(&yi->pub);
                    if (dy_seen <= need) __builtin_amdgcn_s_sleep(2);
                }
                asm volatile("" ::: "memory");
                w.dy[j] = yi->row[pf_cnt & (RDR - 1)][l];             // (not firing: some row, masked below)
            } else {
                const int row = T / period - 1 - pf_cnt;
                w.dy[j] = dyb[(long)(row > 0 ? row : 0) * dy_stride];
            }
            w.m[j] = fire;
            pf_cnt += fire ? 1 : 0;
            pf_next += fire ? period : 0;
            if constexpr (DY_LDS) lds_counter_set(&yi->taken, pf_cnt);   // rows taken (LDS runs a wave's operations in order)
        }
    };
    auto park_chunk = [&](int q, const Raw &w) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int slot = (2 * q + j) & (RFS - 1);
            const float r = w.r[j], u = w.u[j], hp = w.hp[j];
            const float omu = 1.f - u;
            float k1, k2;
            if (c_from_hs) gru_coeff_from_states(h_after, hp, u, omu, k1, k2);
            else { const float c = w.c[j]; k1 = omu * (1.f - c * c); k2 = (hp - c) * u * omu; }
            h_after = hp;
            const float k3 = hp * r * (1.f - r);
            S.ringA[slot][l] = v4f{w.m[j] ? w.dy[j] : 0.f, k1, k2, k3};
            S.ringB[slot][l] = f2{r, u};
        }
        lds_counter_set(&S.fed, 2 * q + 2);
    };

    // ---- DXOUT: the input gradient of block kb (iterations 16 kb .. 16 kb + 15), one chunk of 12 matrix instructions per
    //      iteration of block kb + 1: chunk c = (column tile ct = c / 4, k-quads 3 (c % 4) .. + 2)
    const int xj = lane & 15, xg = lane >> 4;
    r4 xacc = {0.f, 0.f, 0.f, 0.f};
    r4 xw[3];                                                        // A operands of the chunk about to be issued
    int taken_seen = 0;
    auto dx_load_w = [&](int c) {
        const int ct = c >> 2;
        const long col = 16 * ct + xj;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int kq = 3 * (c & 3) + i;
            xw[i] = kq < 8 ? *reinterpret_cast<const r4 *>(a.wg + col * 2 * H + 16 * kq + 4 * xg)
                           : *reinterpret_cast<const r4 *>(a.wc + col * H + 16 * (kq - 8) + 4 * xg);
        }
    };
    // B operands of chunk c of block kb: lane (j, g) <- d_act[iteration 16 kb + j][16 kq + 4 g ..].  Issued BEFORE the
    // iteration publishes e_u: the chain wave, once it has e_u of the last iteration of block kb + 1, starts overwriting
    // the ring half these rows live in, and the LDS unit executes one wave's operations in order.
    auto dx_rows = [&](int kb, int c, r4 (&bq)[3]) {
        int it = RXB * kb + xj;
        it = it < T ? it : T - 1;
        const float *src = &S.dact[it & (NS - 1)][4 * xg];
#pragma unroll
        for (int i = 0; i < 3; ++i) bq[i] = *reinterpret_cast<const r4 *>(src + 16 * (3 * (c & 3) + i));
    };
    int pub_val = 0;
    // (branch-free: the partial sums of a column tile are written to the ring after every chunk, the last write wins and the
    //  rows only count once `pub` says so; the wait for ring space has need = -1 except in front of a block's first write)
    auto dx_chunk = [&](int kb, int c, const r4 (&bq)[3]) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) xacc = __builtin_amdgcn_mfma_f32_16x16x4f32(xw[i][e], bq[i][e], xacc, 0, 0, 0);
        // the ring slots of block kb held block kb - 2: the lower layer must be through with those rows
        const int need = c == 0 ? RXB * (kb - 1) : -1;
        while (taken_seen < need) {
            taken_seen = lds_counter_peek(&yo->taken);
            if (taken_seen < need) __builtin_amdgcn_s_sleep(2);
        }
        asm volatile("" ::: "memory");
        *reinterpret_cast<r4 *>(&yo->row[(RXB * kb + xj) & (RDR - 1)][16 * (c >> 2) + 4 * xg]) = xacc;
        const bool last = (c & 3) == 3;
        xacc = last ? r4{0.f, 0.f, 0.f, 0.f} : xacc;
        const int done = RXB * (kb + 1) < T ? RXB * (kb + 1) : T;
        pub_val = c == 15 ? done : pub_val;
        lds_counter_set(&yo->pub, pub_val);
        dx_load_w((c + 1) & 15);
    };

    {   // chunks 0 .. RAHEAD-1 before the loop
        Raw w;
#pragma unroll
        for (int q = 0; q < RAHEAD; ++q) {
            load_chunk(q, w);
            park_chunk(q, w);
        }
    }
    Raw w0, w1;
    load_chunk(RAHEAD, w1);
    if constexpr (DXOUT) dx_load_w(0);

    float *dap = a.d_act + (b * (long)T + (T - 1)) * 3 * H + l;      // row of iteration 0
    int seen = 0;

    auto iter = [&](int k, int p, bool store_prev) {
        while (seen <= k) {
            seen = lds_counter_peek(&S.dau_pub);
            if (SLEEPY && seen <= k) __builtin_amdgcn_s_sleep(1);
        }
        asm volatile("" ::: "memory");
        const float *row = S.dact[NS == 2 ? p : (k & (NS - 1))];
        const float *old = S.dact[NS == 2 ? (p ^ 1) : ((k - 1) & (NS - 1))];
        r4 bq[3];
        if constexpr (DXOUT) dx_rows(k >= RXB ? (k >> 4) - 1 : 0, k & 15, bq);
        const float euv = split_matvec<2>(row + H, wuS, lane);
        const float o_dar = old[l], o_dau = old[H + l], o_dcp = old[2 * H + l];
        S.eU[p][l] = euv;
        lds_counter_set(&S.eu_pub, k + 1);
        if (store_prev) {
            dap[0] = o_dar;
            dap[H] = o_dau;
            dap[2 * H] = o_dcp;
            dap -= 3 * H;
        }
        if constexpr (DXOUT) {
            if (k >= RXB) dx_chunk((k >> 4) - 1, k & 15, bq);        // (wave-uniform; false only in the first block)
        }
    };

    const int nfull = T >> 1;
    int q = 0;
    if (nfull > 0) {
        load_chunk(RAHEAD + 1, w0);
        iter(0, 0, false);
        iter(1, 1, true);
        park_chunk(RAHEAD, w1);
        q = 1;
    }
    auto group = [&](int qq) {                       // four iterations, two chunks
        load_chunk(qq + RAHEAD + 1, w1);
        iter(2 * qq, 0, true);
        iter(2 * qq + 1, 1, true);
        park_chunk(qq + RAHEAD, w0);
        load_chunk(qq + RAHEAD + 2, w0);
        iter(2 * qq + 2, 0, true);
        iter(2 * qq + 3, 1, true);
        park_chunk(qq + RAHEAD + 1, w1);
    };
    // (Sixteen iterations per trip instead of four -- at a loop's back edge the compiler settles the wave's outstanding loads
    //  down to the last few, s_waitcnt vmcnt(6) in front of the s_branch, i.e. it waits for the chunk prefetched one and a half
    //  iterations ago four iterations before it is needed -- measured: C3 step 2.540 vs 2.538 ms, no gain, not kept.)
    for (; q + 1 < nfull; q += 2) group(q);
    if (q < nfull) {
        iter(2 * q, 0, true);
        iter(2 * q + 1, 1, true);
        park_chunk(q + RAHEAD, w0);
        q += 1;
    }
    if (T & 1) iter(T - 1, 0, T > 1);
    // the last iteration's row: the chain wave reports "da_r of the last step is written" as dau_pub = T + 1
    {
        while (seen <= T) {
            seen = lds_counter_peek(&S.dau_pub);
            if (seen <= T) __builtin_amdgcn_s_sleep(1);
        }
        asm volatile("" ::: "memory");
        const float *row = S.dact[(T - 1) & (NS - 1)];
        dap[0] = row[l];
        dap[H] = row[H + l];
        dap[2 * H] = row[2 * H + l];
    }
    if constexpr (DXOUT) {
        // what the loop has not covered: the rest of the block that was in progress when the iterations ran out (T not a
        // multiple of 16), and the last block, which had no iterations of a following block to hide under
        const int nblk = (T + RXB - 1) / RXB;
        int kb = T >= RXB ? (T >> 4) - 1 : 0, c = T >= RXB ? (T & 15) : 0;
        // (T >= 16: block (T >> 4) - 1 has had chunks 0 .. (T & 15) - 1)
        for (; kb < nblk; ++kb, c = 0)
            for (; c < 16; ++c) {
                r4 bq[3];
                dx_rows(kb, c, bq);
                dx_chunk(kb, c, bq);
            }
    }
}

// ------------------------------------------------------------------------------------------------ lower layer's input gradient
// d_x[t] = d_act[t] [Wg[:D] | Wc[:D]]^T for this sequence, blocks of 16 steps dealt round-robin to `nw` waves (me = 0..nw-1),
// column tiles two at a time (their A operands stationary: 96 registers).
template <int DXD>
__device__ __forceinline__ void pair_bwd_dx_epilogue(const HpmnGruBwd &a, const long b, const int lane, const int me, const int nw) {
    constexpr int H = RH;
    constexpr int NCT = DXD / 16, CTG = NCT < 2 ? NCT : 2;
    const int T = a.T;
    const int j = lane & 15, g = lane >> 4;
    const int nblk = (T + RXB - 1) / RXB;
    const float *src = a.d_act + (b * (long)T) * 3 * H + 4 * g;
    float *dst = a.d_x + (b * (long)T) * DXD + 4 * g;
    auto fetch = [&](int q, v4f (&v)[12]) {
        int tr = RXB * q + j;
        tr = tr < T ? tr : T - 1;
#pragma unroll
        for (int kq = 0; kq < 12; ++kq) v[kq] = *reinterpret_cast<const v4f *>(src + (long)tr * 3 * H + 16 * kq);
    };
    for (int ct0 = 0; ct0 < NCT; ct0 += CTG) {
        float wx[CTG][12][4];
#pragma unroll
        for (int ci = 0; ci < CTG; ++ci)
#pragma unroll
            for (int kq = 0; kq < 12; ++kq) {
                const long col = 16 * (ct0 + ci) + j;
                const v4f v = kq < 8 ? *reinterpret_cast<const v4f *>(a.wg + col * 2 * H + 16 * kq + 4 * g)
                                     : *reinterpret_cast<const v4f *>(a.wc + col * H + 16 * (kq - 8) + 4 * g);
                wx[ci][kq][0] = v.x; wx[ci][kq][1] = v.y; wx[ci][kq][2] = v.z; wx[ci][kq][3] = v.w;
            }
        v4f cur[12], nxt[12];
        if (me < nblk) fetch(me, cur);
        for (int q = me; q < nblk; q += nw) {
            if (q + nw < nblk) fetch(q + nw, nxt);
            const int tr = RXB * q + j;
#pragma unroll
            for (int ci = 0; ci < CTG; ++ci) {
                r4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kq = 0; kq < 12; ++kq)
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wx[ci][kq][c], cur[kq][c], acc, 0, 0, 0);
                if (tr < T) *reinterpret_cast<r4 *>(dst + (long)tr * DXD + 16 * (ct0 + ci)) = acc;
            }
#pragma unroll
            for (int kq = 0; kq < 12; ++kq) cur[kq] = nxt[kq];
        }
    }
}

struct PairBwdArgs {
    HpmnGruBwd lo, up;
    int32_t flags, pad;
};

template <int DXD, bool CFH>
__global__ __launch_bounds__(512, 2) void gru_pair_bwd_kernel(const PairBwdArgs p) {
    __shared__ __attribute__((aligned(16))) BwdLds<2> lo_[2];
    __shared__ __attribute__((aligned(16))) BwdLds<RDR> up_[2];
    __shared__ __attribute__((aligned(16))) DyLds dy_[2];

    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // (r6) flags bit 2: ONE sequence per workgroup, four waves (launched with 256 threads): the roles land on four SIMDs instead of
    // sharing two -- for batches that leave the chip's CUs to spare (B <= CUs: Taobao's 128), where two chain waves per SIMD
    // cost the pace-setting layer 8-17 % (DESIGN_HISTORY 3.12) and buy nothing
    const bool single = (p.flags & 4) != 0;
    const int seq = single ? 0 : (w & 1);
    int role = single ? w : (w >> 1);                               // 0 lower chain, 1 lower feeder, 2 upper chain, 3 upper feeder
    if ((p.flags & 1) && role >= 2) role ^= 1;
    const long b = single ? (long)blockIdx.x : 2 * (long)blockIdx.x + seq;
    if (b >= p.lo.B) return;                         // odd batch (before the barrier: ended waves do not take part in it)
    BwdLds<2> &SL = lo_[seq];
    BwdLds<RDR> &SU = up_[seq];
    DyLds &Y = dy_[seq];
    if (lane == 0) {
        if (role == 0) { SL.dau_pub = 0; SL.eu_pub = 0; SL.fed = 0; Y.pub = 0; Y.taken = 0; }
        if (role == 2) { SU.dau_pub = 0; SU.eu_pub = 0; SU.fed = 0; }
    }
    __syncthreads();

    if (role == 0) {
        __builtin_amdgcn_s_setprio(3);
        pair_bwd_chain<2, false>(p.lo, SL, b, lane);
    } else if (role == 1) {
        __builtin_amdgcn_s_setprio(2);
        pair_bwd_feeder<2, true, false, false, CFH>(p.lo, SL, &Y, nullptr, b, lane);
    } else if (role == 2) {
        __builtin_amdgcn_s_setprio(1);
        pair_bwd_chain<RDR, true>(p.up, SU, b, lane);
    } else {
        __builtin_amdgcn_s_setprio(0);
        pair_bwd_feeder<RDR, false, true, true, CFH>(p.up, SU, nullptr, &Y, b, lane);
    }

    if constexpr (DXD > 0) {
        // the lower layer's input gradient: every d_act row of this workgroup's sequences has to be in memory
        __builtin_amdgcn_s_setprio(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        // waves of the sequence in the order (lower chain, lower feeder, upper chain, upper feeder) = role
        pair_bwd_dx_epilogue<DXD>(p.lo, b, lane, role, 4);
    }
}

bool gru_pair_bwd_supported(int H, int D_lo) { return H == RH && (D_lo == 16 || D_lo == 32 || D_lo == 64); }

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

int gru_pair_bwd_launch(const HpmnGruBwd &lo, const HpmnGruBwd &up, int flags, hipStream_t st) {
    PairBwdArgs p = {};
    p.lo = lo; p.up = up; p.flags = flags;
    const bool single = pair_single_seq(lo.B);
    if (single) p.flags |= 4;
    const dim3 grid(single ? lo.B : (lo.B + 1) / 2), blk(single ? 256 : 512);
    // (the candidate switch is a template argument of the whole launch: both layers the same way)
    const bool cfh = (lo.flags & HPMN_BWD_CANDIDATE_FROM_HS) != 0;
    if (cfh != ((up.flags & HPMN_BWD_CANDIDATE_FROM_HS) != 0)) return HPMN_EUNSUPPORTED;
    if (cfh) {
        if (lo.d_x == nullptr) hipLaunchKernelGGL((gru_pair_bwd_kernel<0, true>), grid, blk, 0, st, p);
        else if (lo.D == 16) hipLaunchKernelGGL((gru_pair_bwd_kernel<16, true>), grid, blk, 0, st, p);
        else if (lo.D == 32) hipLaunchKernelGGL((gru_pair_bwd_kernel<32, true>), grid, blk, 0, st, p);
        else if (lo.D == 64) hipLaunchKernelGGL((gru_pair_bwd_kernel<64, true>), grid, blk, 0, st, p);
        else return HPMN_EUNSUPPORTED;
        return check_launch();
    }
    if (lo.d_x == nullptr) hipLaunchKernelGGL((gru_pair_bwd_kernel<0, false>), grid, blk, 0, st, p);
    else if (lo.D == 16) hipLaunchKernelGGL((gru_pair_bwd_kernel<16, false>), grid, blk, 0, st, p);
    else if (lo.D == 32) hipLaunchKernelGGL((gru_pair_bwd_kernel<32, false>), grid, blk, 0, st, p);
    else if (lo.D == 64) hipLaunchKernelGGL((gru_pair_bwd_kernel<64, false>), grid, blk, 0, st, p);
    else return HPMN_EUNSUPPORTED;
    return check_launch();
}

}  // namespace hpmn
