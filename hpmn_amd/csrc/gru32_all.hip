// build_memory for H = 32 -- the reference's own hidden size (code/hpmn.py:586, :614, :653) -- with ALL K layers and the
// embedding gather in ONE launch per direction: the kernel north_star describes ("a fused embedding-gather + K-layer
// periodic-GRU scan kernel that stages hidden states ... in LDS, uses wavefront shuffles for the per-layer reductions").
//
// At the reference shapes (Amazon: B = 128, T = 100, K = 3-4) a layer's scan is 5-50 us of pure latency and the step was
// ~35 launches of ~10 us each; the layers form a pipeline (layer i+1 consumes every period-th output of layer i,
// code/hpmn.py:124-128) that one launch per layer serialises.  Here a workgroup owns ONE sequence and gives every layer a
// wave of its own (K <= 7 -> at most eight waves with the loader, on up to four SIMDs):
//
//   wave K (loader)  gathers the layer-0 input rows (ids -> embedding rows, id-0 mask, zero prefix) a few steps ahead into
//                    an LDS ring; in training it also writes the materialised gather (the weight gradient's input) and the
//                    read path's `last` row
//   wave i (layer i) the whole layer: input product AND recurrence.  H = 32 makes both small enough for one wave if the
//                    64 lanes split k: lane (unit j = lane / 2, half h = lane % 2) holds the weights of unit j's three
//                    gates over half of the input features and half of the state -- 3 (D/2 + 16) <= 144 registers --,
//                    reads its half of x_t and of h_{t-1} as broadcast 16-byte LDS loads, and the two halves meet in ONE
//                    quad_perm DPP add per gate.  Rows that fire go to the next layer's wave through an 8-row LDS ring
//                    behind a pair of counters (rows published / rows taken); nothing but the saved states of a training
//                    pass ever leaves the CU.
//
// All layers advance concurrently, so the launch takes about as long as layer 0 alone.
//
// The reverse launch mirrors it: wave i runs layer i's reverse scan, forms the gradient wrt its own input rows
// (d_x = d_act [Wg[:D] | Wc[:D]]^T, 96 x D per step on the lane pairs) right behind every step and hands it down a ring as
// the d_y of the layer below; layer 0's d_x goes to memory for the embedding scatter.
#include <cstdlib>

#include "gru32_all.h"

namespace hpmn {

constexpr int AH = 32;
constexpr int AXR = 16;          // rows of the layer-0 input ring
constexpr int AYR = 8;           // rows of an inter-layer ring
struct RingCtr { int pub, taken; };

// the two lanes of a unit add their halves
__device__ __forceinline__ float pair_sum(float v) { return v + dpp_quad<0xB1>(v); }

// ---------------------------------------------------------------------------------------------------- forward, one layer
// XP: the input ring holds PROJECTED rows (x_t [Wg[:D] | Wc[:D]] + b, exponent domain, [r | u | c] x 32) -- layer 0, whose
// input half the loader wave forms (it has nothing else to do and layer 0 sets the pace of the launch)
template <int D, bool TRAIN, bool XP = false>
__device__ __forceinline__ void all32_layer_fwd(const All32Args &a, const int i, const long b, const int lane,
                                                const float *in_ring, const int in_depth, RingCtr *in_ctr,
                                                float *out_ring, RingCtr *out_ctr, float (*hb)[AH], float *rhb) {
    constexpr int H = AH, DX = D / 2;
    static_assert(D % 8 == 0, "half rows are read 16 bytes at a time");
    const int j = lane >> 1, h = lane & 1;
    const int T = a.len[i], period = a.period[i];
    const float *wg = a.wg[i], *wc = a.wc[i];
    constexpr int NX = XP ? 1 : DX / 2;
    f2 wxr[NX], wxu[NX], wxc[NX], whr[8], whu[8], whc[8];
    if constexpr (!XP) {
#pragma unroll
        for (int q = 0; q < DX / 2; ++q) {
            const long k = h * DX + 2 * q;
            wxr[q] = f2{wg[k * 2 * H + j], wg[(k + 1) * 2 * H + j]} * NEG_LOG2E;
            wxu[q] = f2{wg[k * 2 * H + H + j], wg[(k + 1) * 2 * H + H + j]} * NEG_LOG2E;
            wxc[q] = f2{wc[k * H + j], wc[(k + 1) * H + j]} * (2.0f * NEG_LOG2E);
        }
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const long k = D + 16 * h + 2 * q;
        whr[q] = f2{wg[k * 2 * H + j], wg[(k + 1) * 2 * H + j]} * NEG_LOG2E;
        whu[q] = f2{wg[k * 2 * H + H + j], wg[(k + 1) * 2 * H + H + j]} * NEG_LOG2E;
        whc[q] = f2{wc[k * H + j], wc[(k + 1) * H + j]} * (2.0f * NEG_LOG2E);
    }
    if constexpr (!XP) {
#pragma unroll
        for (int q = 0; q < DX / 2; ++q) { settle(wxr[q]); settle(wxu[q]); settle(wxc[q]); }
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) { settle(whr[q]); settle(whu[q]); settle(whc[q]); }
    // (the bias once per unit: on the h == 0 half; XP: already in the projected row)
    float br = (h == 0 && !XP) ? a.bg[i][j] * NEG_LOG2E : 0.f, bu = (h == 0 && !XP) ? a.bg[i][H + j] * NEG_LOG2E : 0.f;
    float bcc = (h == 0 && !XP) ? a.bc[i][j] * (2.0f * NEG_LOG2E) : 0.f;
    settle(br); settle(bu); settle(bcc);

    float hj = 0.f;
    hb[0][j] = 0.f;
    wave_sync();
    int in_seen = 0, out_taken = 0, nout = 0, next_fire = period - 1;
    const bool has_out = out_ring != nullptr;
    // saved states: lane h == 0 stores r and h_t, lane h == 1 stores u and c (one 128-byte line each)
    float *gpa = nullptr, *gpb = nullptr, *yp = nullptr;
    long gb_stride = 0;
    if constexpr (TRAIN) {
        float *g = a.gates[i] + b * (long)T * 3 * H, *hs = a.hs[i] + b * (long)(T + 1) * H;
        gpa = h == 0 ? g + j : g + H + j;
        gpb = h == 0 ? hs + H + j : g + 2 * H + j;                       // hs row t+1 / c
        gb_stride = h == 0 ? H : 3 * H;
        if (h == 0) hs[j] = 0.f;
        if (a.y[i] != nullptr) yp = a.y[i] + b * (long)(T / period) * H + j;
    }
    const long y_adv = yp != nullptr ? H : 0;
    if (yp == nullptr) yp = a.memory + (b * a.K + i) * H + j;      // (no subsampled outputs wanted: the final-state slot)

    for (int t = 0; t < T; ++t) {
        while (in_seen <= t) {
            in_seen = lds_counter_peek(&in_ctr->pub);
            if (in_seen <= t) __builtin_amdgcn_s_sleep(1);
        }
        asm volatile("" ::: "memory");
        f2 ar = {0.f, 0.f}, au = {0.f, 0.f}, ac = {0.f, 0.f};
        if constexpr (XP) {
            // (one lane of the pair carries the projected value, the pair sum below adds the other's zero)
            const float *xp = in_ring + (long)(t & (in_depth - 1)) * 3 * H + j;
            const float pr = xp[0], pu = xp[H], pc = xp[2 * H];
            ar.x = h == 0 ? pr : 0.f;
            au.x = h == 0 ? pu : 0.f;
            ac.x = h == 0 ? pc : 0.f;
        } else {
            const v4f *x4 = reinterpret_cast<const v4f *>(in_ring + (long)(t & (in_depth - 1)) * D + h * DX);
#pragma unroll
            for (int q = 0; q < DX / 4; ++q) {
                const v4f v = x4[q];
                const f2 lo = {v.x, v.y}, hi = {v.z, v.w};
                ar = __builtin_elementwise_fma(lo, wxr[2 * q], ar);     ar = __builtin_elementwise_fma(hi, wxr[2 * q + 1], ar);
                au = __builtin_elementwise_fma(lo, wxu[2 * q], au);     au = __builtin_elementwise_fma(hi, wxu[2 * q + 1], au);
                ac = __builtin_elementwise_fma(lo, wxc[2 * q], ac);     ac = __builtin_elementwise_fma(hi, wxc[2 * q + 1], ac);
            }
        }
        lds_counter_set(&in_ctr->taken, t + 1);          // (LDS runs a wave's operations in order: the row has been read)
        const v4f *h4 = reinterpret_cast<const v4f *>(&hb[t & 1][16 * h]);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const v4f v = h4[q];
            const f2 lo = {v.x, v.y}, hi = {v.z, v.w};
            ar = __builtin_elementwise_fma(lo, whr[2 * q], ar);     ar = __builtin_elementwise_fma(hi, whr[2 * q + 1], ar);
            au = __builtin_elementwise_fma(lo, whu[2 * q], au);     au = __builtin_elementwise_fma(hi, whu[2 * q + 1], au);
        }
        const float r = sigmoid_scaled(pair_sum(ar.x + ar.y + br));
        const float u = sigmoid_scaled(pair_sum(au.x + au.y + bu));
        rhb[j] = r * hj;                                 // (both lanes of the pair write the same value)
        wave_sync();
        const v4f *r4 = reinterpret_cast<const v4f *>(&rhb[16 * h]);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const v4f v = r4[q];
            ac = __builtin_elementwise_fma(f2{v.x, v.y}, whc[2 * q], ac);
            ac = __builtin_elementwise_fma(f2{v.z, v.w}, whc[2 * q + 1], ac);
        }
        const float c = tanh_scaled(pair_sum(ac.x + ac.y + bcc));
        hj = fmaf(u, hj - c, c);
        hb[(t + 1) & 1][j] = hj;
        if constexpr (TRAIN) {
            *gpa = h == 0 ? r : u;
            *gpb = h == 0 ? hj : c;
            gpa += 3 * H;
            gpb += gb_stride;
        }
        *yp = hj;                                        // (the slot of the next row to fire: final when it does)
        const bool fire = t == next_fire;
        if (has_out) {
            // room in the ring -- every step, not only when the row is final: the slot is written on the way there
            const int need = nout - (AYR - 1);
            while (out_taken < need) {
                out_taken = lds_counter_peek(&out_ctr->taken);
                if (out_taken < need) __builtin_amdgcn_s_sleep(1);
            }
            asm volatile("" ::: "memory");
            out_ring[(nout & (AYR - 1)) * H + j] = hj;
        }
        nout += fire ? 1 : 0;
        yp += fire ? y_adv : 0;
        next_fire += fire ? period : 0;
        if (has_out) lds_counter_set(&out_ctr->pub, nout);
        wave_sync();
    }
    if (h == 0) a.memory[(b * a.K + i) * H + j] = hj;
}

// ---------------------------------------------------------------------------------------------------- forward, loader wave
template <int D0, bool TRAIN>
__device__ __forceinline__ void all32_loader(const All32Args &a, const long b, const int lane, float *xrow, float *xpring,
                                             RingCtr *ctr) {
    constexpr int H = AH, DX = D0 / 2;
    const int T = a.len[0];
    const bool live = lane < D0;
    // layer 0's input projection, on the lane pairs like the layers' own products
    const int j = lane >> 1, h = lane & 1;
    f2 wxr[DX / 2], wxu[DX / 2], wxc[DX / 2];
    {
        const float *wg = a.wg[0], *wc = a.wc[0];
#pragma unroll
        for (int q = 0; q < DX / 2; ++q) {
            const long k = h * DX + 2 * q;
            wxr[q] = f2{wg[k * 2 * H + j], wg[(k + 1) * 2 * H + j]} * NEG_LOG2E;
            wxu[q] = f2{wg[k * 2 * H + H + j], wg[(k + 1) * 2 * H + H + j]} * NEG_LOG2E;
            wxc[q] = f2{wc[k * H + j], wc[(k + 1) * H + j]} * (2.0f * NEG_LOG2E);
        }
#pragma unroll
        for (int q = 0; q < DX / 2; ++q) { settle(wxr[q]); settle(wxu[q]); settle(wxc[q]); }
    }
    float br = h == 0 ? a.bg[0][j] * NEG_LOG2E : 0.f, bu = h == 0 ? a.bg[0][H + j] * NEG_LOG2E : 0.f;
    float bcc = h == 0 ? a.bc[0][j] * (2.0f * NEG_LOG2E) : 0.f;
    settle(br); settle(bu); settle(bcc);
    const int f = live ? lane / a.E : 0, e = live ? lane % a.E : 0;
    const long idb = b * (long)a.Tids * a.F + f;      // index of ids[b, 0, f]
    constexpr int LB = 4;                               // steps per batch of loads in flight
    int taken = 0;
    auto fetch_ids = [&](int t0, long (&id)[LB]) {
#pragma unroll
        for (int s = 0; s < LB; ++s) {
            int ti = t0 + s - a.front_zero;
            ti = ti < 0 ? 0 : (ti < a.Tids ? ti : a.Tids - 1);
            id[s] = load_id(a.ids, idb + (long)ti * a.F, a.mask_id0);
        }
    };
    auto fetch_rows = [&](const long (&id)[LB], float (&v)[LB]) {
#pragma unroll
        for (int s = 0; s < LB; ++s) v[s] = a.emb[id[s] * a.E + e];
    };
    long idA[LB], idB[LB];
    float vA[LB];
    fetch_ids(0, idA);
    fetch_ids(LB, idB);
    fetch_rows(idA, vA);
    for (int t0 = 0; t0 < T; t0 += LB) {
        // idA / vA: this batch; idB: the next batch's ids (its rows are requested now, used next iteration)
        float vB[LB];
        fetch_rows(idB, vB);
        long idC[LB];
        fetch_ids(t0 + 2 * LB, idC);
#pragma unroll
        for (int s = 0; s < LB; ++s) {
            const int t = t0 + s;
            if (t < T) {                                // (wave-uniform)
                while (t - taken >= AXR) {
                    taken = lds_counter_peek(&ctr->taken);
                    if (t - taken >= AXR) __builtin_amdgcn_s_sleep(2);
                }
                asm volatile("" ::: "memory");
                const bool keep = t >= a.front_zero && !id_masked(idA[s], a.mask_id0);
                const float v = keep ? vA[s] : 0.f;
                if (live) {
                    xrow[lane] = v;
                    if constexpr (TRAIN) a.x0[(b * (long)T + t) * D0 + lane] = v;
                    if (a.last != nullptr && t == a.last_t) a.last[b * D0 + lane] = v;
                }
                wave_sync();
                {
                    const v4f *x4 = reinterpret_cast<const v4f *>(xrow + h * DX);
                    f2 ar = {0.f, 0.f}, au = {0.f, 0.f}, ac = {0.f, 0.f};
#pragma unroll
                    for (int q = 0; q < DX / 4; ++q) {
                        const v4f xv = x4[q];
                        const f2 lo = {xv.x, xv.y}, hi = {xv.z, xv.w};
                        ar = __builtin_elementwise_fma(lo, wxr[2 * q], ar);     ar = __builtin_elementwise_fma(hi, wxr[2 * q + 1], ar);
                        au = __builtin_elementwise_fma(lo, wxu[2 * q], au);     au = __builtin_elementwise_fma(hi, wxu[2 * q + 1], au);
                        ac = __builtin_elementwise_fma(lo, wxc[2 * q], ac);     ac = __builtin_elementwise_fma(hi, wxc[2 * q + 1], ac);
                    }
                    const float pr = pair_sum(ar.x + ar.y + br), pu = pair_sum(au.x + au.y + bu), pc = pair_sum(ac.x + ac.y + bcc);
                    float *xp = xpring + (t & (AXR - 1)) * 3 * H;
                    xp[(h == 0 ? 0 : H) + j] = h == 0 ? pr : pu;          // lane h == 0: r and c parts, lane h == 1: the u part
                    if (h == 0) xp[2 * H + j] = pc;
                }
                wave_sync();
                lds_counter_set(&ctr->pub, t + 1);
            }
        }
#pragma unroll
        for (int s = 0; s < LB; ++s) { idA[s] = idB[s]; idB[s] = idC[s]; vA[s] = vB[s]; }
    }
}

template <int D0, bool TRAIN>
__global__ __launch_bounds__(512) void gru32_fwd_all_kernel(const All32Args a) {
    __shared__ __attribute__((aligned(16))) float xrow[64];              // the loader's current input row
    __shared__ __attribute__((aligned(16))) float xpring[AXR * 3 * AH];   // layer 0's projected input rows
    __shared__ __attribute__((aligned(16))) float yring[AMAXK][AYR * AH];
    __shared__ __attribute__((aligned(16))) float hb[AMAXK][2][AH];
    __shared__ __attribute__((aligned(16))) float rhb[AMAXK][AH];
    __shared__ RingCtr ctr[AMAXK + 1];                  // ctr[0]: the input ring; ctr[i + 1]: the ring behind layer i

    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long b = blockIdx.x;
    const int K = a.K;
    if (threadIdx.x <= AMAXK) { ctr[threadIdx.x].pub = 0; ctr[threadIdx.x].taken = 0; }
    __syncthreads();
    if (w == K) {
        all32_loader<D0, TRAIN>(a, b, lane, xrow, xpring, &ctr[0]);
    } else if (w == 0) {
        __builtin_amdgcn_s_setprio(3);
        all32_layer_fwd<D0, TRAIN, true>(a, 0, b, lane, xpring, AXR, &ctr[0], K > 1 ? yring[0] : nullptr, &ctr[1], hb[0], rhb[0]);
    } else {
        __builtin_amdgcn_s_setprio(2);
        all32_layer_fwd<AH, TRAIN>(a, w, b, lane, yring[w - 1], AYR, &ctr[w], w + 1 < K ? yring[w] : nullptr, &ctr[w + 1],
                                   hb[w], rhb[w]);
    }
}

// ---------------------------------------------------------------------------------------------------- reverse, one layer
// Iteration k = step t = T-1-k.  Per step (lane pair of unit j, halves over the summed index n):
//   dh_in = dh + dy (firing steps);  dc_pre = dh_in (1-u)(1-c^2);  da_u = dh_in (h_prev - c) u (1-u)
//   d(rh)_j = sum_n dc_pre_n Wc[D+j][n];  da_r = d(rh) h_prev r (1-r)
//   dh_j = dh_in u + d(rh) r + sum_n (da_r_n Wg[D+j][n] + da_u_n Wg[D+j][32+n])
//   d_x_f = sum_n (da_r_n Wg[f][n] + da_u_n Wg[f][32+n] + dc_pre_n Wc[f][n]),  f = j, 32 + j, ... < D
template <int D>
__device__ __forceinline__ void all32_layer_bwd(const All32Args &a, const int i, const long b, const int lane,
                                                const float *in_ring, RingCtr *in_ctr, float *out_ring, RingCtr *out_ctr,
                                                float (*row)[3 * AH]) {
    constexpr int H = AH;
    constexpr int NF = (D + 31) / 32;                   // passes of the input gradient (32 features per pass)
    const int j = lane >> 1, h = lane & 1;
    const int T = a.len[i], period = a.period[i];
    const float *wg = a.wg[i], *wc = a.wc[i];
    // transposed operands: unit j's ROW of the state block, my half of its columns
    f2 wcT[8], wrT[8], wuT[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int n = 16 * h + 2 * q;
        wcT[q] = *reinterpret_cast<const f2 *>(wc + (long)(D + j) * H + n);
        wrT[q] = *reinterpret_cast<const f2 *>(wg + (long)(D + j) * 2 * H + n);
        wuT[q] = *reinterpret_cast<const f2 *>(wg + (long)(D + j) * 2 * H + H + n);
    }
    f2 xr[NF][8], xu[NF][8], xc[NF][8];
    bool fok[NF];
#pragma unroll
    for (int p = 0; p < NF; ++p) {
        const int f = 32 * p + j;
        fok[p] = f < D;
        const int fc = fok[p] ? f : 0;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int n = 16 * h + 2 * q;
            xr[p][q] = *reinterpret_cast<const f2 *>(wg + (long)fc * 2 * H + n);
            xu[p][q] = *reinterpret_cast<const f2 *>(wg + (long)fc * 2 * H + H + n);
            xc[p][q] = *reinterpret_cast<const f2 *>(wc + (long)fc * H + n);
        }
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) { settle(wcT[q]); settle(wrT[q]); settle(wuT[q]); }
#pragma unroll
    for (int p = 0; p < NF; ++p)
#pragma unroll
        for (int q = 0; q < 8; ++q) { settle(xr[p][q]); settle(xu[p][q]); settle(xc[p][q]); }

    const float *gb = a.gates[i] + b * (long)T * 3 * H + j;
    const float *hsb = a.hs[i] + b * (long)(T + 1) * H + j;
    float *dab = a.d_act[i] + b * (long)T * 3 * H;
    float dh = a.d_memory[(b * a.K + i) * H + j];
    const bool has_dy = in_ring != nullptr, has_out = out_ring != nullptr;
    int in_seen = 0, out_taken = 0, nin = 0, next_fire = 0;
    // saved activations of the steps ahead (the same four values on both lanes of a pair), four steps deep
    constexpr int PF = 4;
    float pr[PF], pu[PF], pc[PF], ph[PF];
    auto prefetch = [&](int k, int slot) {
        int t = T - 1 - k;
        t = t > 0 ? t : 0;
        const float *g = gb + (long)t * 3 * H;
        pr[slot] = g[0]; pu[slot] = g[H]; pc[slot] = g[2 * H];
        ph[slot] = hsb[(long)t * H];
    };
#pragma unroll
    for (int s = 0; s < PF; ++s) prefetch(s, s);

    auto step = [&](int k, int slot) {
        const int t = T - 1 - k;
        const float r = pr[slot], u = pu[slot], c = pc[slot], hp = ph[slot];
        prefetch(k + PF, slot);
        const bool fire = has_dy && k == next_fire;
        // (no branch around the wait: need = -1 never waits)
        const int need = fire ? nin : -1;
        while (in_seen <= need) {
            in_seen = lds_counter_peek(&in_ctr->pub);
            if (in_seen <= need) __builtin_amdgcn_s_sleep(1);
        }
        asm volatile("" ::: "memory");
        float dy = 0.f;
        if (has_dy) {
            dy = in_ring[(nin & (AYR - 1)) * H + j];
            nin += fire ? 1 : 0;
            next_fire += fire ? period : 0;
            lds_counter_set(&in_ctr->taken, nin);
        }
        const float dhin = dh + (fire ? dy : 0.f);
        const float omu = 1.f - u;
        const float dcp = dhin * omu * (1.f - c * c);
        const float dau = dhin * (hp - c) * u * omu;
        float *rw = row[k & 1];
        rw[H + j] = dau;
        rw[2 * H + j] = dcp;
        wave_sync();
        const v4f *c4 = reinterpret_cast<const v4f *>(rw + 2 * H + 16 * h);
        v4f vc[4];
        f2 acc = {0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            vc[q] = c4[q];
            acc = __builtin_elementwise_fma(f2{vc[q].x, vc[q].y}, wcT[2 * q], acc);
            acc = __builtin_elementwise_fma(f2{vc[q].z, vc[q].w}, wcT[2 * q + 1], acc);
        }
        const float drh = pair_sum(acc.x + acc.y);
        const float dar = drh * hp * r * (1.f - r);
        rw[j] = dar;
        wave_sync();
        const v4f *r4 = reinterpret_cast<const v4f *>(rw + 16 * h), *u4 = reinterpret_cast<const v4f *>(rw + H + 16 * h);
        v4f vr[4], vu[4];
        f2 e = {0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            vr[q] = r4[q];
            vu[q] = u4[q];
            e = __builtin_elementwise_fma(f2{vr[q].x, vr[q].y}, wrT[2 * q], e);
            e = __builtin_elementwise_fma(f2{vr[q].z, vr[q].w}, wrT[2 * q + 1], e);
            e = __builtin_elementwise_fma(f2{vu[q].x, vu[q].y}, wuT[2 * q], e);
            e = __builtin_elementwise_fma(f2{vu[q].z, vu[q].w}, wuT[2 * q + 1], e);
        }
        dh = fmaf(dhin, u, fmaf(drh, r, pair_sum(e.x + e.y)));
        // the step's d_act row (the weight gradient's operand): lane h == 0 stores da_r and dc_pre, lane h == 1 da_u
        {
            float *dst = dab + (long)t * 3 * H;
            dst[h == 0 ? j : H + j] = h == 0 ? dar : dau;
            if (h == 0) dst[2 * H + j] = dcp;
        }
        // the input gradient of this step, from the operand halves already in registers
        float dx[NF];
#pragma unroll
        for (int p = 0; p < NF; ++p) {
            f2 s = {0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                s = __builtin_elementwise_fma(f2{vr[q].x, vr[q].y}, xr[p][2 * q], s);
                s = __builtin_elementwise_fma(f2{vr[q].z, vr[q].w}, xr[p][2 * q + 1], s);
                s = __builtin_elementwise_fma(f2{vu[q].x, vu[q].y}, xu[p][2 * q], s);
                s = __builtin_elementwise_fma(f2{vu[q].z, vu[q].w}, xu[p][2 * q + 1], s);
                s = __builtin_elementwise_fma(f2{vc[q].x, vc[q].y}, xc[p][2 * q], s);
                s = __builtin_elementwise_fma(f2{vc[q].z, vc[q].w}, xc[p][2 * q + 1], s);
            }
            dx[p] = pair_sum(s.x + s.y);
        }
        if (has_out) {
            // D == 32 here (layers >= 1): one row of the ring the layer below reads its d_y from, index = this iteration
            const int needo = k - (AYR - 1);
            while (out_taken < needo) {
                out_taken = lds_counter_peek(&out_ctr->taken);
                if (out_taken < needo) __builtin_amdgcn_s_sleep(1);
            }
            asm volatile("" ::: "memory");
            out_ring[(k & (AYR - 1)) * H + j] = dx[0];
            lds_counter_set(&out_ctr->pub, k + 1);
        } else {
            float *dst = a.d_x0 + (b * (long)T + t) * D;
#pragma unroll
            for (int p = 0; p < NF; ++p)
                if (fok[p] && h == 0) dst[32 * p + j] = dx[p];
        }
        wave_sync();
    };
    const int nfull = T / PF;
    for (int q = 0; q < nfull; ++q) {
        step(PF * q, 0);
        step(PF * q + 1, 1);
        step(PF * q + 2, 2);
        step(PF * q + 3, 3);
    }
    for (int k = PF * nfull; k < T; ++k) {
        const int slot = k & (PF - 1);
        if (slot == 0) step(k, 0);
        else if (slot == 1) step(k, 1);
        else step(k, 2);
    }
}

template <int D0>
__global__ __launch_bounds__(512) void gru32_bwd_all_kernel(const All32Args a) {
    __shared__ __attribute__((aligned(16))) float dyring[AMAXK][AYR * AH];      // ring i: d_y rows of layer i (from layer i + 1)
    __shared__ __attribute__((aligned(16))) float rows[AMAXK][2][3 * AH];
    __shared__ RingCtr ctr[AMAXK];

    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long b = blockIdx.x;
    const int K = a.K;
    if (threadIdx.x < AMAXK) { ctr[threadIdx.x].pub = 0; ctr[threadIdx.x].taken = 0; }
    __syncthreads();
    const bool top = w == K - 1;
    if (w == 0) {
        __builtin_amdgcn_s_setprio(3);
        all32_layer_bwd<D0>(a, 0, b, lane, top ? nullptr : dyring[0], &ctr[0], nullptr, nullptr, rows[0]);
    } else {
        __builtin_amdgcn_s_setprio(2);
        all32_layer_bwd<AH>(a, w, b, lane, top ? nullptr : dyring[w], &ctr[w], dyring[w - 1], &ctr[w - 1], rows[w]);
    }
}

int gru32_bwd_all_launch(const All32Args &a, int D0, hipStream_t st) {
    const dim3 grid(a.B), blk(64 * a.K);
    if (D0 == 16) hipLaunchKernelGGL(gru32_bwd_all_kernel<16>, grid, blk, 0, st, a);
    else if (D0 == 32) hipLaunchKernelGGL(gru32_bwd_all_kernel<32>, grid, blk, 0, st, a);
    else if (D0 == 48) hipLaunchKernelGGL(gru32_bwd_all_kernel<48>, grid, blk, 0, st, a);
    else if (D0 == 64) hipLaunchKernelGGL(gru32_bwd_all_kernel<64>, grid, blk, 0, st, a);
    else return HPMN_EUNSUPPORTED;
    return check_launch();
}

bool gru32_all_supported(int H, int D0, int K, int E) {
    return H == AH && K >= 1 && K <= AMAXK && (D0 == 16 || D0 == 32 || D0 == 48 || D0 == 64) && E >= 1 && D0 % E == 0;
}

int gru32_fwd_all_launch(const All32Args &a, int D0, bool train, hipStream_t st) {
    const dim3 grid(a.B), blk(64 * (a.K + 1));
#define ALL32_FWD(DD)                                                                          \
    do {                                                                                       \
        if (train) hipLaunchKernelGGL((gru32_fwd_all_kernel<DD, true>), grid, blk, 0, st, a);    \
        else       hipLaunchKernelGGL((gru32_fwd_all_kernel<DD, false>), grid, blk, 0, st, a);   \
    } while (0)
    if (D0 == 16) ALL32_FWD(16);
    else if (D0 == 32) ALL32_FWD(32);
    else if (D0 == 48) ALL32_FWD(48);
    else if (D0 == 64) ALL32_FWD(64);
    else return HPMN_EUNSUPPORTED;
#undef ALL32_FWD
    return check_launch();
}

}  // namespace hpmn
