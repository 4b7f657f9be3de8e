// Fused forward of one periodic-GRU layer for gfx950 (H = 64): a CHAIN wave and a PRODUCER wave per sequence, two
// sequences per workgroup, the input projection on the MATRIX cores.
//
// Against the first generation (gru_fused_fwd.hip), each point measured (DESIGN_HISTORY.md 3.10):
//   * two sequences per 4-wave workgroup.  The hardware starts a CU's next workgroup on the SIMD the previous one ended
//     on (tools/micro/where.hip): with 2-wave workgroups the scan wave of one sequence shared a SIMD with the projection
//     wave of the other on every CU while a SIMD sat idle;
//   * the 64x64 recurrent products are k-split over lane pairs (common.h split_matvec): half the LDS return traffic;
//   * the wave on the serial chain only does what depends on h; the saved states are stored by the producer wave.
//
// The projection xp_t = x_t [Wg[:D] | Wc[:D]] + b is the time-parallel half of the layer: for a block of 16 steps it is
// a real [192 x D] x [D x 16] product.  The producer wave of gru_fused_fwd3.hip did it step by step with packed FMAs
// (48 / 96 per step plus an LDS broadcast of x_t), which made it as long as the wave on the serial chain and left it no
// room for anything else.  Here it issues v_mfma_f32_16x16x4_f32 (true fp32), 9 / 17 per 16-column tile and 16-step
// block = 7 / 13 instructions per step, and the matrix pipe works underneath the wave's other duties:
//   D[i][j] = sum_k A[i][k] B[k][j],   i = output column within the tile, j = step within the block
//   A (weights, stationary):  lane (i = lane % 16, g = lane / 16) holds W[feature(kq, g, c)][16 ct + i]
//   B (inputs):               lane (j = lane % 16, g)             holds x[step j][feature(kq, g, c)]
//   feature(kq, g, c) = 16 kq + 4 g + c -- the k order of a product is free as long as both operands agree, and with
//   this one a lane's B operands of k-steps (kq, c = 0..3) are the four components of ONE 16-byte load of its step's input
//   row (for layer 0: of the embedding row; no LDS broadcast, no staging);
//   D: lane (j, g), register r = xp[step j][16 ct + 4 g + r]: one ds_write_b128 into the ring;
//   the bias rides in as one more k-step (B = 1 for g = 0).
// The producer wave also stores the saved states (h from the LDS state buffer, r,u,c from a small LDS hand-off) and runs
// in lockstep with the chain wave on the h_pub counter, so ring space and ring contents need no counters of their own.
//
// UPROD (built, parity-green, measured slower, default off): the producer also forms the UPDATE gate
// u_t = sigmoid(xu_t + h_{t-1} Wu), which the chain wave needs only at the very end of step t, leaving the chain wave 64
// of the 96 packed FMAs.  Stand-alone layer 0 at C3: 0.508 ms without, 0.547-0.627 ms with (0.489 ms with every other
// duty of the producer removed): a step of this recurrence is a LATENCY chain (two LDS round trips, two dependent
// products, two activations ~ 1000 cycles), and the hand-off of u (publish h, poll, product, sigmoid, publish, poll,
// read) is as long as the chain wave's own path -- unlike the reverse scan's e_u, which carries no activation.
#include <cstdlib>
#include <type_traits>

#include "common.h"

namespace hpmn {

constexpr int MH = 64;        // hidden size of this kernel
constexpr int MB = 16;        // steps per projection block (= MFMA N)
constexpr int MRING = 32;     // ring slots: the block the chain wave is on + the block being projected
constexpr int MNT = 12;       // 16-column tiles of [r | u | c]
constexpr int MSL = 4;        // slots of the state / gate hand-off buffers (h_{t-1} in slot t % MSL): the chain wave may run
                              // MSL - 2 steps ahead of the producer's reads, so its check of the producer's counter is
                              // almost always answered by the cached copy

typedef float f4v __attribute__((ext_vector_type(4)));

// UPROD: the producer wave forms the update gate
template <int D, bool GATHER, bool TRAIN, bool UPROD>
__global__ __launch_bounds__(256, 1) void gru_fwd_mfma_kernel(const HpmnGruFusedFwd a) {
    constexpr int H = MH;
    constexpr int NJ = D / 16;                                              // 16-byte pieces of an input row per lane
    // (rows padded by 4 floats: 3 H = 192 floats are three bank periods, and the producer writes a projected tile as 16 rows x
    //  16 bytes -- one row per lane group -- which with a 192-float stride was a 16-way bank conflict per write, on the LDS pipe
    //  the chain wave's round trips queue in: SQ_LDS_BANK_CONFLICT 28 % of this launch's LDS cycles)
    constexpr int RINGW = 3 * H + 4;
    __shared__ __attribute__((aligned(16))) float ring_[2][MRING][RINGW];   // xp (r | u | c)
    __shared__ __attribute__((aligned(16))) float hb_[2][MSL][H];          // h_{t-1} lives in hb[t % MSL]
    __shared__ __attribute__((aligned(16))) float rhb_[2][H];
    __shared__ float ubuf_[2][2][H];
    __shared__ __attribute__((aligned(16))) v4f rcb_[2][MSL][H];           // r, u, c of step t in rcb[t % MSL]
    __shared__ int ctr_[2][4];

    const int lane = threadIdx.x & 63;
    const int seq = (threadIdx.x >> 6) & 1, role = threadIdx.x >> 7;       // role 0: chain, 1: producer
    const int l = lane;
    const int T = a.T;
    const long b = 2 * (long)blockIdx.x + seq;
    if (b >= a.B) return;        // odd batch (before the barrier: ended waves do not take part in it)
    float (&ring)[MRING][RINGW] = ring_[seq];
    float (&hb)[MSL][H] = hb_[seq];
    float (&rhb)[H] = rhb_[seq];
    float (&ubuf)[2][H] = ubuf_[seq];
    v4f (&rcb)[MSL][H] = rcb_[seq];
    int &produced = ctr_[seq][0], &h_pub = ctr_[seq][1], &u_pub = ctr_[seq][2];
    if (role == 0) {
        hb[0][l] = 0.f;
        if (lane == 0) { produced = 0; h_pub = 0; u_pub = 0; }
    }
    __syncthreads();             // the only barrier: counters and h_{-1} = 0 are in place

    if (role == 1) {
        // ================================================================== producer
        __builtin_amdgcn_s_setprio(2);
        const int n16 = lane & 15, g = lane >> 4;
        // stationary A operands, exponent scale folded in (columns < 2H: gates, then the candidate)
        float wA[MNT][NJ][4], wBias[MNT];
#pragma unroll
        for (int ct = 0; ct < MNT; ++ct) {
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
        for (int ct = 0; ct < MNT; ++ct) {
            settle(wBias[ct]);
#pragma unroll
            for (int kq = 0; kq < NJ; ++kq)
#pragma unroll
                for (int c = 0; c < 4; ++c) settle(wA[ct][kq][c]);
        }
        float one = g == 0 ? 1.f : 0.f;
        settle(one);
        f2 whu[2][16];
        if constexpr (UPROD) split_matvec_weights_t<2>(a.wg + (long)D * 2 * H + H, 2 * H, NEG_LOG2E, lane, whu);

        struct Rows { v4f v[NJ]; bool keep[NJ]; };
        // lane (n16, g) of block s0: step s0 + n16, features 16 kq + 4 g .. + 3
        auto fetch_ids = [&](int s0, long (&id)[NJ]) {
            if constexpr (GATHER) {
                int t = s0 + n16;
                t = t < T ? t : T - 1;
                const int ti = t - a.front_zero;
#pragma unroll
                for (int kq = 0; kq < NJ; ++kq)
                    id[kq] = load_id(a.ids, (b * (long)a.Tids + (ti > 0 ? ti : 0)) * a.F + (16 * kq + 4 * g) / a.E, a.mask_id0);   // clamped, never examined here
            }
        };
        auto fetch_rows = [&](int s0, const long (&id)[NJ], Rows &r) {
            int t = s0 + n16;
            t = t < T ? t : T - 1;
#pragma unroll
            for (int kq = 0; kq < NJ; ++kq) {
                const int e0 = 16 * kq + 4 * g;
                if constexpr (GATHER) {
                    r.v[kq] = *reinterpret_cast<const v4f *>(a.emb + id[kq] * a.E + e0 % a.E);
                    r.keep[kq] = (t >= a.front_zero) && !id_masked(id[kq], a.mask_id0);
                } else {
                    r.v[kq] = *reinterpret_cast<const v4f *>(a.x + (b * (long)T + t) * D + e0);
                    r.keep[kq] = true;
                }
            }
        };
        // masked rows of a block, and the materialised gather (training: the weight gradient's input)
        auto finish_rows = [&](int s0, Rows &r) {
#pragma unroll
            for (int kq = 0; kq < NJ; ++kq) {
                if (!r.keep[kq]) r.v[kq] = v4f{0.f, 0.f, 0.f, 0.f};
                if constexpr (GATHER) {
                    if (a.x_out != nullptr && s0 + n16 < T)
                        *reinterpret_cast<v4f *>(a.x_out + (b * (long)T + s0 + n16) * D + 16 * kq + 4 * g) = r.v[kq];
                    if (a.last != nullptr && s0 + n16 == a.last_t)      // the read path's uinp[:, last_index, :]
                        *reinterpret_cast<v4f *>(a.last + b * (long)D + 16 * kq + 4 * g) = r.v[kq];
                }
            }
        };
        // one 16-column tile of a block -> ring
        auto tile = [&](int s0, int ct, const Rows &r) {
            f4v acc = {0.f, 0.f, 0.f, 0.f};
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wBias[ct], one, acc, 0, 0, 0);
#pragma unroll
            for (int kq = 0; kq < NJ; ++kq)
#pragma unroll
                for (int c = 0; c < 4; ++c) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wA[ct][kq][c], r.v[kq][c], acc, 0, 0, 0);
            *reinterpret_cast<f4v *>(&ring[(s0 + n16) & (MRING - 1)][16 * ct + 4 * g]) = acc;
        };

        long idA[NJ], idB[NJ];
        Rows rA, rB;
        // block 0 before the loop
        {
            fetch_ids(0, idA);
            fetch_ids(MB, idB);
            fetch_rows(0, idA, rA);
            fetch_ids(2 * MB, idA);
            fetch_rows(MB, idB, rB);
            finish_rows(0, rA);
#pragma unroll
            for (int ct = 0; ct < MNT; ++ct) tile(0, ct, rA);
            lds_counter_set(&produced, MB);
            rA = rB;                                     // rows of block 1 (in flight); idA: ids of block 2
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
        const bool keep_c = !(a.flags & HPMN_FWD_NO_CANDIDATE);            // (wave-uniform)

        // iteration t: u_t for the chain wave; then the saved rows of step t-1.  TILE >= 0: one 16-column tile of the
        // NEXT block's projection is issued right behind the wait, so that the matrix pipe works underneath the packed
        // FMAs of the update gate (one basic block: the scheduler interleaves the two)
        auto iteration = [&](int t, auto tile_c, int s_next) {
            constexpr int TILE = decltype(tile_c)::value;
            while (h_seen < t) {                         // h_{t-1} (and r,c of step t-1) are in LDS
                h_seen = lds_counter_peek(&h_pub);
            }
            asm volatile("" ::: "memory");
            const int p = t & (MSL - 1), pm = (t - 1) & (MSL - 1);
            f4v acc = {0.f, 0.f, 0.f, 0.f};
            if constexpr (TILE >= 0) {
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wBias[TILE], one, acc, 0, 0, 0);
#pragma unroll
                for (int kq = 0; kq < NJ; ++kq)
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wA[TILE][kq][c], rA.v[kq][c], acc, 0, 0, 0);
            }
            float u_now = 0.f;
            // everything this iteration reads of the chain wave's buffers is read BEFORE the counter that lets the
            // chain wave move on (LDS operations of a wave complete in order): hb[t % MSL] and rcb[(t-1) % MSL] are
            // rewritten MSL - 1 steps later
            float hprev;
            v4f rc = {0.f, 0.f, 0.f, 0.f};
            if constexpr (UPROD) {
                const float xu = ring[t & (MRING - 1)][H + l];
                const float su = split_matvec<2>(&hb[p][0], whu, lane);
                hprev = hb[p][l];
                if constexpr (TRAIN) rc = rcb[pm][l];
                u_now = sigmoid_scaled(xu + su);
                ubuf[t & 1][l] = u_now;
                lds_counter_set(&u_pub, t + 1);
            } else {
                hprev = hb[p][l];
                if constexpr (TRAIN) rc = rcb[pm][l];
                asm volatile("" : "+v"(hprev), "+v"(rc));          // (the reads have landed)
                lds_counter_set(&u_pub, t + 1);                    // here: "iteration t has read its inputs"
            }
            if constexpr (TRAIN) {
                *hsp = hprev;
                hsp += H;
                // (t == 0 writes an undefined row 0, which iteration 1 overwrites: same wave, same address, in order)
                gp[0] = rc.x;
                gp[H] = UPROD ? u_prev : rc.y;
                if (keep_c) gp[2 * H] = rc.z;            // (HPMN_FWD_NO_CANDIDATE: two 128-byte lines per step stay unwritten)
                gp += t > 0 ? 3 * H : 0;
                u_prev = u_now;
            }
            *yp = hprev;
            const bool fire = t == next_fire;
            next_fire += fire ? period : 0;
            yp += fire ? y_adv : 0;
            if constexpr (TILE >= 0)
                *reinterpret_cast<f4v *>(&ring[(s_next + n16) & (MRING - 1)][16 * TILE + 4 * g]) = acc;
        };
        auto full_block = [&](int s0, auto... is) {      // 16 steps, all inside the sequence, and a block to project
            (iteration(s0 + decltype(is)::value, std::integral_constant<int, (decltype(is)::value < MNT ? decltype(is)::value : -1)>{},
                       s0 + MB), ...);
        };

        for (int s0 = 0; s0 < T; s0 += MB) {
            // chain wave on block s0; rA = rows of block s0 + MB (loaded a block ago), idA = ids of block s0 + 2 MB
            fetch_rows(s0 + 2 * MB, idA, rB);
            fetch_ids(s0 + 3 * MB, idA);
            if (s0 + MB < T) {                            // wave-uniform
                finish_rows(s0 + MB, rA);
                {
                    using std::integral_constant;
                    full_block(s0, integral_constant<int, 0>{}, integral_constant<int, 1>{}, integral_constant<int, 2>{},
                               integral_constant<int, 3>{}, integral_constant<int, 4>{}, integral_constant<int, 5>{},
                               integral_constant<int, 6>{}, integral_constant<int, 7>{}, integral_constant<int, 8>{},
                               integral_constant<int, 9>{}, integral_constant<int, 10>{}, integral_constant<int, 11>{},
                               integral_constant<int, 12>{}, integral_constant<int, 13>{}, integral_constant<int, 14>{},
                               integral_constant<int, 15>{});
                }
            } else {                                      // the last block: nothing left to project, maybe partial
#pragma unroll
                for (int i = 0; i < MB; ++i)
                    if (s0 + i < T) iteration(s0 + i, std::integral_constant<int, -1>{}, 0);
            }
            rA = rB;
        }
        // the rows of the last step
        while (h_seen < T) h_seen = lds_counter_peek(&h_pub);
        asm volatile("" ::: "memory");
        {
            const float hlast = hb[T & (MSL - 1)][l];
            if constexpr (TRAIN) {
                const v4f rc = rcb[(T - 1) & (MSL - 1)][l];
                *hsp = hlast;
                gp[0] = rc.x;
                gp[H] = UPROD ? u_prev : rc.y;
                if (keep_c) gp[2 * H] = rc.z;
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
        const int need = MB < T ? MB : T;
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
        // next step's projected input (the producer is MB steps ahead; past the end: a stale slot, unused).  Crossing
        // into the next 16-step block, its last tile must be in the ring: the producer writes tile i at the END of
        // iteration 16 q + i, i < 12, so "iteration 16 q + 12 has started" (u_pub >= 16 q + 13 = t - 2) is the condition
        // -- the end-of-step check below only guarantees one iteration less.  (Found by running the forward beside an
        // unrelated 2.5 GB fill, tools/dbg_fwd_race.py: without this wait 5 of 20 such runs read a stale last tile.)
        if (((t + 1) & (MB - 1)) == 0) {
            while (u_seen < t - 2) u_seen = lds_counter_peek(&u_pub);
            asm volatile("" ::: "memory");
        }
        const float *nx = ring[(t + 1) & (MRING - 1)];
        xr = nx[l];
        if constexpr (!UPROD) xu = nx[H + l];
        xcand = nx[2 * H + l];
        // UPROD: u_t from the producer.  Otherwise the same counter says "the producer's iteration i has read its
        // inputs": the slots this step is about to overwrite (h_{t-4}, gates of step t-4) were read by iteration t-3.
        // With MSL = 4 slots the cached copy of the counter answers that almost every time -- with 2 slots it was by
        // construction one step too old, and the forced re-read (an LDS round trip on the chain) cost 50 ns per step.
        if constexpr (UPROD) {
            while (u_seen <= t) u_seen = lds_counter_peek(&u_pub);
            asm volatile("" ::: "memory");
            u = ubuf[p & 1][l];
        } else {
            while (u_seen < t - (MSL - 2)) u_seen = lds_counter_peek(&u_pub);
            asm volatile("" ::: "memory");
        }
        h = fmaf(u, h - cc, cc);
        hb[(p + 1) & (MSL - 1)][lane] = h;
        if constexpr (TRAIN) rcb[p][l] = v4f{r, u, cc, 0.f};
        lds_counter_set(&h_pub, t + 1);
        wave_sync();
    };
    static_assert(MSL == 4, "the loop below is unrolled by the slot count");
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

template <int D, bool UPROD>
static int launch_mf(const HpmnGruFusedFwd &a, hipStream_t st) {
    const bool train = a.hs != nullptr;
    const dim3 grid((a.B + 1) / 2), blk(256);
    if (a.x == nullptr) {
        if (train) hipLaunchKernelGGL((gru_fwd_mfma_kernel<D, true, true, UPROD>), grid, blk, 0, st, a);
        else       hipLaunchKernelGGL((gru_fwd_mfma_kernel<D, true, false, UPROD>), grid, blk, 0, st, a);
    } else {
        if (train) hipLaunchKernelGGL((gru_fwd_mfma_kernel<D, false, true, UPROD>), grid, blk, 0, st, a);
        else       hipLaunchKernelGGL((gru_fwd_mfma_kernel<D, false, false, UPROD>), grid, blk, 0, st, a);
    }
    return check_launch();
}

#ifndef MF_UPROD32
#define MF_UPROD32 0
#endif
#ifndef MF_UPROD64
#define MF_UPROD64 0
#endif

int gru_fwd_mfma_dispatch(const HpmnGruFusedFwd &a, hipStream_t st) {
    if (a.B == 0) return HPMN_OK;
    if (a.H != MH) return HPMN_EUNSUPPORTED;
    if (a.D == 32) return launch_mf<32, MF_UPROD32>(a, st);
    if (a.D == 64) return launch_mf<64, MF_UPROD64>(a, st);
    return HPMN_EUNSUPPORTED;
}

}  // namespace hpmn
