// build_memory forward, ALL layers in ONE launch: batch-tiled split-f16 MFMA recurrence, layers pipelined across
// workgroups (the north_star kernel: "K-layer periodic-GRU scan ... fires every 2^k steps", code/hpmn.py:113-131).
//
//   workgroup (layer i, tile b) = 4 waves, 16 sequences.  Wave w owns hidden units [16w, 16w+16) of all three
//   gates; per step it issues 12 (gates) + 6 (candidate) recurrent MFMAs and 9 or 18 for the input product of the
//   NEXT step (off the serial chain), see pipe_common.h for the operand layout.  Two LDS hand-overs per step
//   (r*h before the candidate product, h' before the next step), each one 8-byte write + raw s_barrier +
//   16-byte reads.
//   Layer i+1 consumes every period-th h of layer i straight from the y rows layer i writes anyway (they are
//   the next layer's weight-gradient input): the producer writes them through (sc1) and publishes, per wave,
//   "rows complete" in a progress word a few steps late -- s_waitcnt vmcnt(N) with N = the stores issued since,
//   never a drain -- and the consumer wave that needs units [16w,16w+16) polls the producer wave that made them.
//   Workgroups take their role from a ticket counter in (tile, layer) order, so a consumer only ever waits for
//   a workgroup that is already running: placement- and dispatch-order independent, no co-residency assumption.
//
// A step of the chain costs ~? cycles here against ~1250 (scan) / ~1450 (fused) for one sequence per wave, on
// 1/16 of the waves; with K x ceil(B/16) <= 256 workgroups every layer runs concurrently on its own CU, so the
// forward of all K layers takes about as long as layer 0 alone.
#include "pipe_common.h"

namespace hpmn {

constexpr int FWD_IMGS = 8;                       // operand images: h hi/lo, r*h hi/lo, x ring 2 x hi/lo
constexpr int F32ROW = 288;                       // bytes per sequence row of an fp32 image (256 data + 32 pad)
constexpr int F32IMG = TS * F32ROW;
constexpr int ROLE_OFF = FWD_IMGS * IMG + 2 * F32IMG;
constexpr int FWD_LDS = ROLE_OFF + 64;
constexpr int PUB_DELAY = 3;                      // publish rows whose stores were issued >= 3 steps ago
constexpr int FWD_WAVES = 12;                     // (r, u, c) x 4 unit blocks
constexpr int FWD_THREADS = FWD_WAVES * 64;

// A wave issues one instruction per ~5 cycles whatever it is, so the length of a step is set by the number of
// instructions the waves ON THE CHAIN have to issue, not by the arithmetic: the step is therefore cut into
// twelve role-specialised waves, three per SIMD --
//   R(ub), U(ub): reset / update gate of unit block ub (16 units): 6 recurrent MFMAs, 4 sigmoids per lane;
//                 R also forms r*h and writes its operand image, U leaves u in an fp32 image
//   C(ub):        candidate + state update of unit block ub: 6 MFMAs, 4 tanh, h' = u h + (1-u) c, writes the
//                 operand image of h' (and an fp32 copy for R), the saved states' h and c, the output row y
// Phase 1 (R, U on the chain) and phase 2 (C on the chain) are separated by the two workgroup barriers; the
// wave that is off the chain in a phase does the time-parallel work there: the input product of the next step
// (R, U in phase 2; C in phase 1) and the input pipeline (C in phase 1: row t+4 leaves memory, row t+3 is
// split into f16 halves and parked as an operand image).
template <bool TRAIN, bool DEP>
__device__ __forceinline__ void pipe_fwd_body(const PipeArgs &a, const PipeLayer &L, const int layer, const int tile,
                                              char *smem) {
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, g = lane >> 4, n = lane & 15;
    const int role = wv >> 2, ub = wv & 3;                 // 0 = R, 1 = U, 2 = C
    const int B = a.B, T = L.T, D = L.D;
    const bool live = tile * TS + n < B;
    const long b = live ? (long)tile * TS + n : (long)B - 1;
    const int u0 = 16 * ub + 4 * g;         // the lane's four hidden units (outputs) / input features
    const bool ks2 = D > 32;                // input product has a second k-step
    const bool has_x = 16 * ub < D;         // this unit block's feature block exists (D % 16 == 0)

    char *Hhi = smem, *Hlo = smem + IMG, *Rhi = smem + 2 * IMG, *Rlo = smem + 3 * IMG;
    char *Xhi = smem + 4 * IMG, *Xlo = smem + 5 * IMG;     // ring slot q at + 2*q*IMG
    char *H32 = smem + FWD_IMGS * IMG, *U32 = H32 + F32IMG;
    const int wr = img_wr_off(ub, g, n);
    const int rd0 = img_rd_off(0, g, n), rd1 = img_rd_off(1, g, n);
    const int f32off = n * F32ROW + ub * 64 + g * 16;

    // ---- stationary A operands of the wave's tile: recurrent rows and input rows, exponent scale of the
    //      exp2-based sigmoid / tanh folded in, split into f16 halves once
    h8 Ah_hi[2], Ah_lo[2], Ai_hi[2], Ai_lo[2];
    f4 bias;
    {
        const int col = 16 * ub + n;        // A's row index m = lane % 16 -> output column of the tile
        const float sc = role < 2 ? NEG_LOG2E : 2.0f * NEG_LOG2E;
        const float *W = role < 2 ? L.wg + role * PH + col : L.wc + col;
        const int ld = role < 2 ? 2 * PH : PH;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            float vh[8], vi[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int unit = slot_unit(s, g, e);
                vh[e] = W[(long)(D + unit) * ld] * sc;
                const float wi = W[(long)(unit < D ? unit : 0) * ld] * sc;     // (clamped: no divergent load)
                vi[e] = unit < D ? wi : 0.f;
            }
            split8(vh, Ah_hi[s], Ah_lo[s]);
            split8(vi, Ai_hi[s], Ai_lo[s]);
        }
        const float *bp = role < 2 ? L.bg + role * PH : L.bc;
#pragma unroll
        for (int j = 0; j < 4; ++j) bias[j] = bp[u0 + j] * sc;
    }

    // ---- output pointers; a partial tile's dead columns write to a dump row with stride 0
    const int period = L.period;
    const bool has_y = L.y != nullptr;
    float *dump = a.dump + (tid & 255) * 4;
    float *hsp = dump, *gp = dump;           // gp: this role's third of the gates row
    int s_adv = 0, g_adv = 0;
    if constexpr (TRAIN) {
        if (live) {
            hsp = L.hs + (b * (long)(T + 1) + 1) * PH + u0;
            gp = L.gates + (b * (long)T) * 3 * PH + role * PH + u0;
            s_adv = PH;
            g_adv = 3 * PH;
            if (role == 2) *reinterpret_cast<f4 *>(L.hs + b * (long)(T + 1) * PH + u0) = f4{0.f, 0.f, 0.f, 0.f};
        }
    }
    float *yp = live ? (has_y ? L.y + b * (long)(T / period) * PH + u0 : L.h_last + b * L.h_last_stride + u0) : dump;
    const int y_adv = (live && has_y) ? PH : 0;
    const float *xb_ = L.x + b * (long)T * D + u0;

    // ---- hand-off words (C waves: unit block ub of this layer's output <- / -> feature block ub of the next)
    unsigned *my_flag = a.sync + 2 + ((long)layer * a.ntiles + tile) * 4 + ub;
    const unsigned *dep_flag = a.sync + 2 + ((long)(layer - 1) * a.ntiles + tile) * 4 + ub;
    int avail = 0;                          // rows of the layer below known complete (wave-uniform)
    // (hysteresis: a poll is a round trip through memory, ~1.5 us; once the consumer has caught up with the
    //  producer it would pay one per step -- more than the producer needs for a row.  When it has to wait it
    //  waits for WAIT_AHEAD rows beyond the one it needs, and then runs that many steps without polling.)
    const int wait_ahead = layer < 4 ? (8 >> layer) : 0;      // about the same TIME at every layer (rows are 2^i apart)
    auto wait_rows = [&](int need, int limit) {
        if constexpr (DEP) {
            if (need > avail) {
                const int want = need + wait_ahead < limit ? need + wait_ahead : limit;
                unsigned spins = 0;
                do {
                    const unsigned v = __hip_atomic_load(dep_flag, RLX_AGENT);
                    avail = __builtin_amdgcn_readfirstlane((int)v);
                    if (avail >= want) break;
                    __builtin_amdgcn_s_sleep(2);
                    if (++spins > PIPE_SPIN_LIMIT) {          // lost hand-off: flag the error, stop waiting
                        if (lane == 0) __hip_atomic_store(a.sync + 1, 1u + (unsigned)layer, RLX_AGENT);
                        avail = 0x7fffffff;
                    }
                } while (avail < want);
            }
        }
    };
    auto load_row = [&](int rho) -> f4 {
        const int rc = rho < T ? rho : T - 1;
        if constexpr (DEP) return load4_agent(xb_ + (long)rc * D);
        else               return *reinterpret_cast<const f4 *>(xb_ + (long)rc * D);
    };
    auto park = [&](const f4 v, int slot) {
        uint2 hi, lo;
        split4(v, hi, lo);
        *reinterpret_cast<uint2 *>(Xhi + 2 * slot * IMG + wr) = hi;
        *reinterpret_cast<uint2 *>(Xlo + 2 * slot * IMG + wr) = lo;
    };
#define MF(A, Bv, C) __builtin_amdgcn_mfma_f32_16x16x32_f16(A, Bv, C, 0, 0, 0)
    // bias + x W[:D] of the wave's tile for the row parked in `slot` (three-product split)
    auto project = [&](int slot) -> f4 {
        const h8 x0h = *reinterpret_cast<const h8 *>(Xhi + 2 * slot * IMG + rd0);
        const h8 x0l = *reinterpret_cast<const h8 *>(Xlo + 2 * slot * IMG + rd0);
        f4 p = bias;
        p = MF(Ai_hi[0], x0h, p);
        p = MF(Ai_hi[0], x0l, p);
        p = MF(Ai_lo[0], x0h, p);
        if (ks2) {
            const h8 x1h = *reinterpret_cast<const h8 *>(Xhi + 2 * slot * IMG + rd1);
            const h8 x1l = *reinterpret_cast<const h8 *>(Xlo + 2 * slot * IMG + rd1);
            p = MF(Ai_hi[1], x1h, p);
            p = MF(Ai_hi[1], x1l, p);
            p = MF(Ai_lo[1], x1h, p);
        }
        return p;
    };
    // recurrent product of the wave's tile on the operand image at (hi, lo), on top of `init`
    auto recur = [&](const char *hi, const char *lo, const f4 init) -> f4 {
#ifdef HPMN_DBG_NOLDSRD
        h8 b0h = Ah_hi[0], b1h = Ah_hi[1], b0l = Ah_lo[0], b1l = Ah_lo[1];
        asm volatile("" : "+v"(b0h), "+v"(b1h), "+v"(b0l), "+v"(b1l));
#else
        const h8 b0h = *reinterpret_cast<const h8 *>(hi + rd0), b1h = *reinterpret_cast<const h8 *>(hi + rd1);
        const h8 b0l = *reinterpret_cast<const h8 *>(lo + rd0), b1l = *reinterpret_cast<const h8 *>(lo + rd1);
#endif
#ifdef HPMN_DBG_NOMFMA
        {
            f4 z_ = init;
            z_[0] += (float)b0h[0] + (float)b1h[0] + (float)b0l[0] + (float)b1l[0];
            return z_;
        }
#endif
        f4 p = MF(Ah_hi[0], b0h, init);
        const f4 zero = {0.f, 0.f, 0.f, 0.f};
        f4 q = MF(Ah_hi[1], b1h, zero);
        p = MF(Ah_hi[0], b0l, p);
        q = MF(Ah_hi[1], b1l, q);
        p = MF(Ah_lo[0], b0h, p);
        q = MF(Ah_lo[1], b1h, q);
        return p + q;
    };

    // ---- zero every image (h_0 = 0; feature slots beyond D stay zero for good)
    for (int i = tid; i < (FWD_IMGS * IMG + 2 * F32IMG) / 16; i += FWD_THREADS)
        reinterpret_cast<uint4 *>(smem)[i] = uint4{0u, 0u, 0u, 0u};
    __syncthreads();

    // ---- input pipeline (C waves).  Row rho: leaves memory in phase 1 of step rho-4, becomes an operand image
    //      (slot rho & 1) in phase 1 of step rho-2, and every wave runs its input product on it while it is off
    //      the chain in step rho-1.  Prologue: rows 0,1 parked and row 0 projected, rows 2,3 in flight.
    //      (An LDS-DMA variant -- global_load_lds_dwordx4 into 8 staging slots, rows six steps in flight behind
    //      one counted s_waitcnt -- was built and measured SLOWER: 1273 vs 972 us at K=1, DESIGN_HISTORY.md 3.7.)
    f4 ra = {0.f, 0.f, 0.f, 0.f}, rb = ra;
    if (role == 2 && has_x) {
        wait_rows(T < 4 ? T : 4, T);
        const f4 r0 = load_row(0), r1 = load_row(1);
        ra = load_row(2);
        rb = load_row(3);
        park(r0, 0);
        park(r1, 1);
    }
    lds_barrier();
    f4 xp = project(0);                      // input product of step 0 for this wave's tile
    unsigned pend = 0;
    if constexpr (DEP) {
        if (role == 2 && has_x) pend = __hip_atomic_load(dep_flag, RLX_AGENT);
    }
    lds_barrier();

#ifdef HPMN_PIPE_PROF
    long long prof[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pt = 0;
#define PROF0() pt = __builtin_amdgcn_s_memtime()
#define PROF(i) { const long long now_ = __builtin_amdgcn_s_memtime(); prof[i] += now_ - pt; pt = now_; }
#else
#define PROF0()
#define PROF(i)
#endif
    f4 h = {0.f, 0.f, 0.f, 0.f};             // C waves: the state of the wave's units, fp32
    int next_fire = period - 1, fired = 0, q1 = 0, q2 = 0, q3 = 0;

    // One step.  (C waves) RX holds row t+2, loaded two steps ago: it is parked now and the register then receives
    // row t+4.  Called with two registers on alternate steps, so a load's destination is never moved in flight.
    auto step = [&](const int t, f4 &RX) {
        if (role < 2) {
            // ---------------- phase 1, on the chain: gate of this unit block
            __builtin_amdgcn_s_setprio(2);
            PROF0();
            const f4 hown = *reinterpret_cast<const f4 *>(H32 + f32off);
            f4 z = recur(Hhi, Hlo, xp);
#ifdef HPMN_PIPE_PROF
            asm volatile("" : "+v"(z));
#endif
            PROF(0);
            f4 gate;
#pragma unroll
            for (int j = 0; j < 4; ++j) gate[j] = sigmoid_scaled(z[j]);
#ifdef HPMN_PIPE_PROF
            asm volatile("" : "+v"(gate));
#endif
            PROF(1);
            if (role == 0) {
                uint2 hi, lo;
                split4(gate * hown, hi, lo);
                *reinterpret_cast<uint2 *>(Rhi + wr) = hi;
                *reinterpret_cast<uint2 *>(Rlo + wr) = lo;
            } else {
                *reinterpret_cast<f4 *>(U32 + f32off) = gate;
            }
#ifndef HPMN_DBG_NOSTORE
            if constexpr (TRAIN) {
                *reinterpret_cast<f4 *>(gp) = gate;
                gp += g_adv;
            }
#endif
            __builtin_amdgcn_s_setprio(0);
            PROF(2);
            lds_barrier();                                                 // A
            PROF(3);
            // ---------------- phase 2, off the chain: input product of step t+1 (row t+1 is in slot (t+1)&1)
            xp = project((t + 1) & 1);
#ifdef HPMN_PIPE_PROF
            asm volatile("" : "+v"(xp));
#endif
            PROF(4);
            lds_barrier();                                                 // B
            PROF(5);
        } else {
            // ---------------- phase 1, off the chain: input pipeline + input product of step t+1
            if (has_x) {
                if constexpr (DEP) {
                    const int seen = __builtin_amdgcn_readfirstlane((int)pend);
                    avail = seen > avail ? seen : avail;
                    wait_rows(t + 5 < T ? t + 5 : T, T);
                    pend = __hip_atomic_load(dep_flag, RLX_AGENT);
                }
                park(RX, t & 1);                                           // row t+2 -> slot (t+2)&1 (row t's slot)
#ifndef HPMN_DBG_NOLOAD
                RX = load_row(t + 4);
#endif
            }
            PROF0();
            f4 xn = project((t + 1) & 1);
#ifdef HPMN_PIPE_PROF
            asm volatile("" : "+v"(xn));
#endif
            PROF(0);
            lds_barrier();                                                 // A: r*h, u are in LDS
            PROF(1);
            // ---------------- phase 2, on the chain: candidate and state update
            __builtin_amdgcn_s_setprio(2);
            const f4 u = *reinterpret_cast<const f4 *>(U32 + f32off);
            f4 z = recur(Rhi, Rlo, xp);
#ifdef HPMN_PIPE_PROF
            asm volatile("" : "+v"(z));
#endif
            PROF(2);
            f4 c;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                c[j] = tanh_scaled(z[j]);
                h[j] = fmaf(u[j], h[j] - c[j], c[j]);                      // u*h + (1-u)*c
            }
            {
                uint2 hi, lo;
                split4(h, hi, lo);
                *reinterpret_cast<uint2 *>(Hhi + wr) = hi;
                *reinterpret_cast<uint2 *>(Hlo + wr) = lo;
                *reinterpret_cast<f4 *>(H32 + f32off) = h;
            }
            __builtin_amdgcn_s_setprio(0);
            PROF(3);
#ifndef HPMN_DBG_NOSTORE
            if constexpr (TRAIN) {
                *reinterpret_cast<f4 *>(hsp) = h;
                *reinterpret_cast<f4 *>(gp) = c;
                hsp += s_adv;
                gp += g_adv;
            }
#endif
            // subsampled output: unconditional store to the current slot, the slot advances after a firing step
            // (no control flow in the loop, see gru_scan_fwd.hip)
#ifndef HPMN_DBG_NOY
            store4_agent(yp, h);
#endif
            const bool fire = t == next_fire;
            next_fire += fire ? period : 0;
            yp += fire ? y_adv : 0;
            fired += fire ? 1 : 0;
            xp = xn;
            PROF(4);
            lds_barrier();                                                 // B: h' (and row t+2) are in LDS
            PROF(5);
            // publish: every store of step t-3 and older has been acknowledged once at most 3 steps' worth of
            // newer memory operations are outstanding (each step issues >= PMIN of them, in order)
            constexpr int PMIN = (TRAIN ? 2 : 0) + 2;
#ifndef HPMN_DBG_NOWAIT
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PUB_DELAY * PMIN) : "memory");
#endif
            if (has_y && lane == 0) __hip_atomic_store(my_flag, (unsigned)q3, RLX_AGENT);
            q3 = q2; q2 = q1; q1 = fired;
            PROF(6);
        }
    };

#ifdef HPMN_PIPE_CLK
    const long long clk0 = __builtin_amdgcn_s_memtime(), wall0 = (long long)wall_clock64();
#endif
    int t = 0;
    for (; t + 1 < T; t += 2) {
        step(t, ra);
        step(t + 1, rb);
    }
    if (t < T) step(t, ra);
#undef MF
#ifdef HPMN_PIPE_CLK
    if (tile == 0 && lane == 0 && ub == 0) {
        long long *o = reinterpret_cast<long long *>(a.dump) + 1024 + (layer * 3 + role) * 8;
        o[0] = __builtin_amdgcn_s_memtime() - clk0;
        o[1] = (long long)wall_clock64() - wall0;
        o[2] = wall0;
        for (int i = 3; i < 8; ++i) o[i] = 0;
    }
#endif
#ifdef HPMN_PIPE_PROF
    if (tile == 0 && lane == 0 && ub == 0) {
        long long *o = reinterpret_cast<long long *>(a.dump) + 1024 + (layer * 3 + role) * 8;
        for (int i = 0; i < 8; ++i) o[i] = prof[i];
    }
#endif
    if (role == 2) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (has_y && lane == 0) __hip_atomic_store(my_flag, (unsigned)fired, RLX_AGENT);
        if (live) *reinterpret_cast<f4 *>(L.h_last + b * L.h_last_stride + u0) = h;
    }
}

template <bool TRAIN>
__global__ __launch_bounds__(FWD_THREADS, 1) void gru_pipe_fwd_kernel(const PipeArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int *role = reinterpret_cast<int *>(smem + ROLE_OFF);
    if (threadIdx.x == 0) role[0] = (int)atomicAdd(a.sync, 1u);
    __syncthreads();
    const int ticket = __builtin_amdgcn_readfirstlane(role[0]);
    __syncthreads();
    const int layer = ticket % a.K, tile = ticket / a.K;
    // (selected with static indices: a dynamically indexed by-value kernel argument would be copied to scratch)
    PipeLayer L = a.L[0];
#pragma unroll
    for (int i = 1; i < HPMN_MAX_LAYERS; ++i)
        if (i == layer) L = a.L[i];
#ifdef HPMN_DBG_NODEP
    pipe_fwd_body<TRAIN, false>(a, L, layer, tile, smem);
#else
    if (layer == 0) pipe_fwd_body<TRAIN, false>(a, L, layer, tile, smem);
    else            pipe_fwd_body<TRAIN, true>(a, L, layer, tile, smem);
#endif
}

// x0[b, t, :] = (t < front_zero) ? 0 : emb[ids[b, t - front_zero, f]] * (mask ? id != 0 : 1) -- the layer-0 input
// rows (Hpmn.embedding, code/hpmn.py:414-423 / :266-276, with the zero prefix of :288-289), one float4 per thread:
// E/4 adjacent lanes move one 64-byte table row, the id stream is read coalesced.
__global__ __launch_bounds__(256) void embed_gather_seq_kernel(const void *__restrict__ ids,
                                                               const float *__restrict__ emb, float *__restrict__ out,
                                                               long total4, int E4, int F, int Tids, int front_zero,
                                                               int mask_id0) {
    // Four items per thread and pass: the four ids first, then the four (dependent) row loads, then the stores --
    // a cold table (rows are 64-byte random HBM reads) is latency-bound unless many rows are in flight per wave.
    constexpr int U = 4;
    const long stride = (long)gridDim.x * blockDim.x;
    const int T0 = Tids + front_zero;
    for (long i0 = (long)blockIdx.x * blockDim.x + threadIdx.x; i0 < total4; i0 += U * stride) {
        long id[U];
        int e4[U];
        bool real[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long i = i0 + u * stride;
            const long ic = i < total4 ? i : total4 - 1;
            const long row = ic / E4;                 // (b, t, f)
            e4[u] = (int)(ic - row * E4);
            const long bt = row / F;
            const int f = (int)(row - bt * F);
            const long bb = bt / T0;
            const int t = (int)(bt - bb * T0) - front_zero;
            real[u] = t >= 0;
            id[u] = load_id(ids, (bb * Tids + (t >= 0 ? t : 0)) * F + f, mask_id0);
        }
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            // (non-temporal, as in embed_gather_sum_kernel: a gathered row is used once -- r5)
            typedef float v4f_ __attribute__((ext_vector_type(4)));
            const v4f_ w = __builtin_nontemporal_load(reinterpret_cast<const v4f_ *>(emb) + id[u] * E4 + e4[u]);
            v[u] = make_float4(w[0], w[1], w[2], w[3]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long i = i0 + u * stride;
            if (!real[u] || id_masked(id[u], mask_id0)) v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < total4) reinterpret_cast<float4 *>(out)[i] = v[u];
        }
    }
}

int embed_gather_seq_launch(const void *ids, const float *emb, float *out, int B, int Tids, int F, int E,
                            int front_zero, int mask_id0, hipStream_t st) {
    const long total4 = (long)B * (Tids + front_zero) * F * E / 4;
    if (total4 == 0) return HPMN_OK;
    long blocks = (total4 + 4 * 256 - 1) / (4 * 256);
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(embed_gather_seq_kernel, dim3((unsigned)blocks), dim3(256), 0, st, ids, emb, out, total4, E / 4,
                       F, Tids, front_zero, mask_id0);
    return check_launch();
}

size_t pipe_sync_bytes(int K, int ntiles) { return (2 + (size_t)K * ntiles * 4) * sizeof(unsigned); }

bool pipe_shape_supported(int H, int D) { return H == PH && D >= 16 && D <= 64 && D % 16 == 0; }

// a.sync must hold pipe_sync_bytes(); it is zeroed here, on the stream, before every launch
int pipe_fwd_launch(const PipeArgs &a, int num_cus, hipStream_t st) {
    hipError_t e = hipMemsetAsync(a.sync, 0, pipe_sync_bytes(a.K, a.ntiles), st);
    if (e != hipSuccess) { set_last_hip_error((int)e); return HPMN_EHIP; }
    const int grid = a.K * a.ntiles;
    // one workgroup per CU while the whole pipeline fits the chip (a second workgroup on a CU would share its
    // VALU and matrix pipe with a latency-bound chain): ask for more than half of the 160 KiB of LDS
    const size_t lds = grid <= num_cus ? (size_t)96 * 1024 : (size_t)FWD_LDS;
    static bool attr_set = false;
    if (!attr_set) {     // dynamic LDS above 64 KiB has to be allowed per function
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(gru_pipe_fwd_kernel<true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(gru_pipe_fwd_kernel<false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        attr_set = true;
    }
    if (a.train) hipLaunchKernelGGL((gru_pipe_fwd_kernel<true>), dim3(grid), dim3(FWD_THREADS), lds, st, a);
    else         hipLaunchKernelGGL((gru_pipe_fwd_kernel<false>), dim3(grid), dim3(FWD_THREADS), lds, st, a);
    return check_launch();
}

}  // namespace hpmn
