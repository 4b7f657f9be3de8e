// Reverse (BPTT) periodic-GRU scan, H = 64, two specialised waves per sequence: a CHAIN wave and a FEEDER wave.
//
// Why (DESIGN_HISTORY.md 3.10): a single wave issues one instruction per ~5.5 cycles whatever the instruction is, so the
// length of a reverse step is the NUMBER of instructions the wave on the serial chain has to issue.  In
// gru_scan_bwd_helper_kernel that wave still issued ~230 per step, of which only 96 (two 64x64 products) and a
// dozen more are the recurrence: the rest was the prefetch of the saved activations (address arithmetic, loads,
// parking them in LDS), the coefficient arithmetic on them, the d_act stores and their addresses.  None of that
// depends on the gradient that is being propagated.  Here the second wave of the workgroup (on another SIMD of the
// CU) does all of it:
//
//   feeder (wave 1), per step:  loads r,u,c,h_prev,d_y of a step 4..7 steps ahead straight in unit layout (lane = unit),
//       turns them into the step's coefficients
//           k1 = (1-u)(1-c^2)     dc_pre = dh k1          k3 = h_prev r (1-r)     da_r = d(rh) k3
//           k2 = (h_prev-c)u(1-u) da_u   = dh k2
//       and parks {dy,k1,k2,k3 | r,u} in an LDS ring (one 16-byte + one 8-byte write, read back the same way);
//       computes e_u = da_u Wg[D+k][H:2H] for the chain wave as before (16 broadcast reads + 32 packed FMAs);
//       stores the step's d_act = [da_r | da_u | dc_pre] to HBM out of the LDS operand buffers the chain wave fills
//       anyway for its own broadcasts (one step late, so that it never waits for da_r).
//   chain (wave 0), per step:   dh += dy; dc_pre, da_u (2 multiplies) -> LDS; d(rh) = dc_pre Wc^T; da_r -> LDS;
//       e_r = da_r Wg_r^T; dh = dh u + d(rh) r + e_r + e_u.   No global memory access, no address arithmetic, no vmcnt.
//
// DXD > 0: the layer's INPUT GRADIENT d_x[t] = d_act[t] [Wg[:D] | Wc[:D]]^T (what hpmn_gru_input_grad computes as a launch
// of its own -- for layers >= 1 the d_y the next reverse scan waits for, i.e. a kernel on the serial chain, 17-90 us each
// and ~10 us of launch latency even for the 16-step top layer; for layer 0 the first kernel of the step's tail) comes out
// of THIS launch, as an epilogue: once a sequence's chain and feeder waves have finished the scan, their SIMDs are idle
// until the launch ends, so the two waves split the sequence's 16-step blocks between them and issue the product on the
// matrix cores (v_mfma_f32_16x16x4_f32, true fp32, 96 / 192 per block), reading the d_act rows back from memory (they are
// the feeder's own stores, ordered by a barrier) 16 bytes per lane straight into operand layout.  Nobody is
// latency-critical any more at that point -- which is what sank the same product as a CONCURRENT third role (waves on the
// feeders' SIMDs, operands from a 32-step LDS ring): 3.62 vs 3.36 ms/step, because a SIMD does not issue its other wave's
// VALU instructions while an fp32 MFMA is passing (DESIGN_HISTORY.md 3.10).
//
// Hand-offs are LDS progress counters (common.h: data, lgkmcnt(0), counter; cached copies, re-read only when the
// cached value says "wait"); no barrier in the loop.  Buffers are double-buffered by step parity: the feeder reads
// step k-1's operands at the start of its step k, before it publishes e_u(k), and the chain wave cannot reach step
// k+1 (which overwrites them) without e_u(k).
#include <cstdlib>
#include <type_traits>

#include "common.h"

namespace hpmn {

constexpr int FR_STEPS = 8;     // coefficient ring depth in steps (4 chunks of 2)
constexpr int FR_AHEAD = 3;     // chunks the feeder parks ahead of the chunk the chain wave is on
constexpr int DROW = 192 + 4;   // floats per operand row [da_r | da_u | dc_pre] in LDS, PADDED: 192 floats are three bank periods, so
                                // the 16 rows of an input-gradient block (lane j reads row 16 kb + j, 16 bytes) all started in
                                // the same banks -- a 16-way conflict on every operand read of the in-loop product, on the LDS
                                // pipe the chain wave's round trips queue in (SQ_LDS_BANK_CONFLICT: 22 % of the launch's LDS
                                // cycles); 196 = 4 x 49 puts them 4 banks apart
constexpr int DXB = 16;         // steps per input-gradient block (= MFMA N)

typedef float f4m __attribute__((ext_vector_type(4)));
typedef float xf4 __attribute__((ext_vector_type(4)));
typedef __bf16 xbf8 __attribute__((ext_vector_type(8)));

// x = p0 + p1 (+ p2) in bf16 planes, eight values at once: two planes leave 2^-17 |x|, three 2^-25 |x| (gru_wgrad_bf16.hip)
template <int NP>
__device__ __forceinline__ void split_bf16x8(const v4f a, const v4f b, xbf8 (&p)[NP]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        float rest = i < 4 ? a[i] : b[i - 4];
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            p[q][i] = (__bf16)rest;
            if (q + 1 < NP) rest -= (float)p[q][i];
        }
    }
}

// TWO sequences per workgroup: the hardware places the waves of a workgroup on consecutive SIMDs of its rotation,
// but starts the next workgroup of the CU on the SIMD the previous one ended on (tools/micro/where.hip: with 2-wave
// workgroups every CU had the chain wave of one sequence and the feeder of the other on ONE SIMD and a SIMD idle;
// 4-wave workgroups land on four distinct SIMDs, the 5th and 6th wave of a 6-wave workgroup on the SIMDs of the 1st
// and 2nd).  Waves 0,1: chain waves of sequences 2 blockIdx.x + 0,1; waves 2,3: their feeders.
// LOOPDX (D <= 32): the input gradient is formed INSIDE the loop, by the feeder, on the bf16 matrix pipe with split operands
// (x = hi + lo, three products, fp32 accumulate -- gru_wgrad_bf16.hip's arithmetic): the chain wave keeps its operand rows in
// a 32-row LDS ring instead of two parity slots, and while it fills one half (16 iterations) the feeder multiplies the
// other half by [Wg[:D] | Wc[:D]]^T, one (column tile, 32-wide k slice) unit per iteration BEHIND the iteration's e_u hand-off:
// two 16-byte LDS reads, the split (24 VALU), three v_mfma_f32_16x16x32_bf16 -- 48 cycles of matrix pipe per iteration where
// the fp32 form of round 2 held it for 384 (which is what sank the concurrent input gradient then).  The weights' operand
// fragments are stationary in the feeder's registers (96 at D = 32: the feeder had them to spare, the kernel's register
// count is set by the chain wave + epilogue).  No d_act row is read back from memory (0.38 GB per step at C3's layer 0) and
// the launch ends with the scan.
// NP (round 6): planes of the in-loop product's operands -- 3: the six products of order <= 2, fp32-equivalent (default);
// 2: rounds 4/5 (three products, ~5e-6 of max|grad|; HPMN_DX_PLANES=2).
template <int DXD, bool SCAT = false, bool LOOPDX = false, bool CFH = false, int NP = 3>
__global__ __launch_bounds__(256, 1) void gru_scan_bwd_feed_kernel(const HpmnGruBwd a) {
    constexpr int H = 64;
    constexpr bool DX = DXD > 0;
    static_assert(!LOOPDX || (DXD > 0 && DXD <= 32), "in-loop input gradient: D = 16 / 32");
    constexpr int NSLOT = LOOPDX ? 32 : 2;                                 // operand rows kept in LDS (step parity / ring)
    __shared__ __attribute__((aligned(16))) v4f ringA_[2][FR_STEPS][H];    // dy, k1, k2, k3
    __shared__ __attribute__((aligned(16))) f2 ringB_[2][FR_STEPS][H];     // r, u
    __shared__ __attribute__((aligned(16))) float dact_[2][NSLOT][DROW];   // da_r | da_u | dc_pre of iteration k in row k % NSLOT
    __shared__ float eU_[2][2][H];
    __shared__ int ctr_[2][4];
    // (r6) three planes: the LO plane of the in-loop product's stationary weight fragments lives here (12 KB at D = 32, one copy
    // for the workgroup's two feeders; read once per product that uses it, behind the e_u hand-off) -- in registers it took the
    // kernel from 280 to 336 registers per wave, and a weight-gradient workgroup (192) no longer fitted beside the scan on a
    // SIMD's 512: the weight gradients of layers 1-6 then ran BEHIND layer 0's reverse scan instead of underneath it and the C3
    // step went from 2.39 to 2.79 ms (profiles/r06_experiments.txt: the scan launch itself was 30 us FASTER).
    constexpr int NWLO = (LOOPDX && NP == 3) ? 6 * (DXD / 16) : 1;
    __shared__ __attribute__((aligned(16))) xbf8 wlo_[NWLO][64];
    if constexpr (LOOPDX && NP == 3) {
        constexpr int NCT_ = DXD / 16;
        for (int sidx = threadIdx.x; sidx < NWLO * 64; sidx += 256) {
            const int u = sidx >> 6, ln = sidx & 63;
            const int ks = u / NCT_, ct = u % NCT_;
            const long col = 16 * ct + (ln & 15);
            const int gc = 32 * ks + 8 * (ln >> 4);
            const float *src = gc < 2 * H ? a.wg + col * 2 * H + gc : a.wc + col * H + (gc - 2 * H);
            const v4f v0 = *reinterpret_cast<const v4f *>(src), v1 = *reinterpret_cast<const v4f *>(src + 4);
            xbf8 pl[3];
            split_bf16x8<3>(v0, v1, pl);
            wlo_[u][ln] = pl[2];
        }
    }

    const int lane = threadIdx.x & 63;
    const int w = threadIdx.x >> 6;
    const int seq = w & 1;
    const int role = w >> 1;                                              // 0 chain, 1 feeder
    const int l = lane;
    const int T = a.T, D = a.D;
    const long b = 2 * (long)blockIdx.x + seq;
    if (b >= a.B) return;                        // odd batch: the last workgroup runs one sequence (before the barrier:
                                                 // ended waves do not take part in it)
    v4f (&ringA)[FR_STEPS][H] = ringA_[seq];
    f2 (&ringB)[FR_STEPS][H] = ringB_[seq];
    float (&dact)[NSLOT][DROW] = dact_[seq];
    float (&eU)[2][H] = eU_[seq];
    int &dau_pub = ctr_[seq][0], &eu_pub = ctr_[seq][1], &fed = ctr_[seq][2];
    const int t_lo0 = a.t_begin;
    const int t_hi = a.t_end > 0 ? a.t_end : T;
    const int nsteps = t_hi - t_lo0;
    const int nfull = nsteps >> 1;               // 2-step chunks; an odd last step is peeled
    if (lane == 0 && role == 0) { dau_pub = 0; eu_pub = 0; fed = 0; }
    __syncthreads();

    if (role == 1) {
        // ================================================================== feeder
        __builtin_amdgcn_s_setprio(2);
        f2 wuS[2][16];                       // rows of the update-gate block, k-split (common.h)
        split_matvec_weights<2>(a.wg + (long)D * 2 * H + H, 2 * H, lane, wuS);

        const int period = a.period;
        const bool has_dy = a.d_y != nullptr;
        const float *gb = a.gates + b * (long)T * 3 * H + l;
        const float *hsb = a.hs + b * (long)(T + 1) * H + l;
        const float *dyb = has_dy ? a.d_y + b * (long)(T / period) * H + l : a.d_h_last + b * a.d_h_last_stride + l;
        const long dy_stride = has_dy ? H : 0;
        // d_y row j belongs to step (j+1)*period - 1 (t_hi is a multiple of period, so step t_hi-1 has one)
        int pf_fire = t_hi - 1, pf_row = t_hi / period - 1;

        // c[]: the stored candidate, or (HPMN_BWD_CANDIDATE_FROM_HS: the forward did not store it) the state AFTER the step,
        // h_t = hs[t + 1] -- the h_prev of the iteration before, kept in a register; the candidate's two 128-byte lines of
        // every gates row are never fetched
        // (CFH: a template switch -- as a run-time flag it cost the feeder a redundant load, a select and a branch per step,
        //  +1.5 % on the whole C3 step with the flag off)
        constexpr bool c_from_hs = CFH;
        float h_after = c_from_hs ? hsb[(long)t_hi * H] : 0.f;
        struct Raw { float r[2], u[2], c[2], hp[2], dy[2]; bool m[2]; };
        // chunk q = iterations 2q, 2q+1 = steps t_hi-1-2q, t_hi-2-2q; rows before the sequence start are clamped
        // (loaded, parked, never consumed)
        auto load_chunk = [&](int q, Raw &w) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int t_raw = t_hi - 1 - 2 * q - j;
                const int t = t_raw > 0 ? t_raw : 0;
                const float *g = gb + (long)t * 3 * H;
                w.r[j] = g[0];
                w.u[j] = g[H];
                w.hp[j] = hsb[(long)t * H];
                w.c[j] = c_from_hs ? 0.f : g[2 * H];
                // (NOTHING is computed on a loaded value here -- not even the select between the candidate and the state
                //  after the step: it would wait for the loads just issued.  park_chunk does it, chunks are parked in order)
                const bool fire = has_dy && t_raw == pf_fire && pf_row >= 0;
                w.dy[j] = dyb[(long)(pf_row > 0 ? pf_row : 0) * dy_stride];
                w.m[j] = fire;
                pf_row -= fire ? 1 : 0;
                pf_fire -= fire ? period : 0;
            }
        };
        auto park_chunk = [&](int q, const Raw &w) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int slot = (2 * q + j) & (FR_STEPS - 1);
                const float r = w.r[j], u = w.u[j], hp = w.hp[j];
                const float omu = 1.f - u;
                float k1, k2;
                if (c_from_hs) gru_coeff_from_states(h_after, hp, u, omu, k1, k2);
                else { const float c = w.c[j]; k1 = omu * (1.f - c * c); k2 = (hp - c) * u * omu; }
                h_after = hp;
                const float k3 = hp * r * (1.f - r);
                ringA[slot][l] = v4f{w.m[j] ? w.dy[j] : 0.f, k1, k2, k3};
                ringB[slot][l] = f2{r, u};
            }
            lds_counter_set(&fed, 2 * q + 2);
        };

        {   // chunks 0 .. FR_AHEAD-1 before the loop
            Raw w;
#pragma unroll
            for (int q = 0; q < FR_AHEAD; ++q) {
                load_chunk(q, w);
                park_chunk(q, w);
            }
        }
        Raw w0, w1;                          // two chunks of loads in flight
        load_chunk(FR_AHEAD, w1);

        // ---- LOOPDX: unit U = (column tile U / 6, k slice U % 6) of a 16-iteration block.  A operand (stationary): lane
        //      (j, kg) = W[input column 16 ct + j][gate columns 32 ks + 8 kg .. + 7], W = [wg[0:D] | wc[0:D]]; B operand: lane
        //      (n, kg) = d_act[iteration 16 kb + n][the same gate columns], out of the ring; C: lane (n, kg) holds input columns
        //      16 ct + 4 kg .. + 3 of iteration n -- 16 contiguous bytes of the d_x row.
        // (r6) A block's product is 6 k slices (32 of the 192 gate columns each) x NCT column tiles (16 input columns each).  The
        // column tiles of a slice multiply the SAME rows of the ring: the slice's fragment is read and split ONCE (rounds 4/5:
        // per (column tile, slice)), and with three planes per operand a tile takes six products where it took three.  What
        // shapes the schedule is the ISA's view of the feeder (an in-order wave pays >= 32 cycles of issue per matrix
        // instruction; the chain wave waits for e_u and for nothing else):
        //  * D = 32, units U = 0 .. 12, one per iteration:  U = 2 ks: read + split slice ks (44 VALU) WHILE the matrix pipe does
        //    tile 1 of slice ks - 1 (six products on the planes split two units ago: independent of the split, so the compiler
        //    interleaves them);  U = 2 ks + 1: tile 0 of slice ks;  U = 12: tile 1 of slice 5.  Never more than six matrix
        //    instructions per iteration (rounds 4/5: three).
        //  * D = 16, units U = ks: read, split, six products.
        //  * every unit is PINNED between the iteration's e_u hand-off and the next iteration's wait: an empty statement
        //    "produces" the planes the unit's products read (nothing of the unit can be hoisted above the hand-off -- the first
        //    shared-split version had the compiler issue the second tile's products in front of the e_u product, 2.80 ms per C3
        //    step against 2.39) and one "uses" the accumulators at its end (nothing can sink into the next iteration).
        constexpr int NCT = LOOPDX ? DXD / 16 : 1;
        constexpr int NU = LOOPDX ? (NCT == 2 ? 13 : 6) : 1;
        constexpr int NBUF = NCT == 2 ? 2 : 1;
        const int j16 = lane & 15, kg = lane >> 4;
        constexpr int NR = NP == 3 ? 2 : NP;      // planes of the stationary fragments kept in registers (the lo plane: wlo_)
        xbf8 wA[6 * NCT][NR];                     // [ks * NCT + ct]
        if constexpr (LOOPDX) {
#pragma unroll
            for (int u = 0; u < 6 * NCT; ++u) {
                const int ks = u / NCT, ct = u % NCT;
                const long col = 16 * ct + j16;
                const int gc = 32 * ks + 8 * kg;
                const float *src = gc < 2 * H ? a.wg + col * 2 * H + gc : a.wc + col * H + (gc - 2 * H);
                const v4f v0 = *reinterpret_cast<const v4f *>(src), v1 = *reinterpret_cast<const v4f *>(src + 4);
                xbf8 pl[NP];
                split_bf16x8<NP>(v0, v1, pl);
#pragma unroll
                for (int q2 = 0; q2 < NR; ++q2) wA[u][q2] = pl[q2];
            }
        }
        xf4 xacc[NCT];
        xbf8 bp[NBUF][NP];                        // slice fragments, split (slice ks in buffer ks % NBUF)
#pragma unroll
        for (int c = 0; c < NCT; ++c) xacc[c] = xf4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q1 = 0; q1 < NBUF; ++q1)
#pragma unroll
            for (int q2 = 0; q2 < NP; ++q2)
#pragma unroll
                for (int e = 0; e < 8; ++e) bp[q1][q2][e] = (__bf16)0.f;
        float *dxb = LOOPDX ? a.d_x + (b * (long)T) * DXD + 4 * kg : nullptr;
        // LOOPDX + SCAT (r5, HPMN_FUSED_SCATTER=2): a finished 16-iteration x 16-column tile goes straight into the table gradient --
        // column tile ct IS id column ct (E = 16) -- instead of into d_x and through a scatter launch behind the scan.  Branch-free
        // like the rest of the feeder's loop: the id of the lane's row and the read path's d_last piece are requested when the
        // block's first unit starts (iterations before they are needed); rows that must not be added (the zero prefix, the
        // masked id 0, clamped lanes) add into the lane's own place of the d_x buffer, which nobody reads.
        int sc_id[NCT], sc_flag = 0;
        xf4 sc_dl[NCT];
#pragma unroll
        for (int c = 0; c < NCT; ++c) { sc_id[c] = 0; sc_dl[c] = xf4{0.f, 0.f, 0.f, 0.f}; }
        const float *dlb = nullptr;
        float dlm = 0.f;
        if constexpr (SCAT && LOOPDX) {
            dlm = a.d_last != nullptr ? 1.f : 0.f;
            dlb = (a.d_last != nullptr ? a.d_last + b * (long)DXD : a.wc) + 4 * kg;      // (no d_last: any finite floats, times 0)
        }
        auto dx_unit = [&](auto uc, int kb) __attribute__((always_inline)) {
            constexpr int U = decltype(uc)::value;
            if constexpr (LOOPDX && U >= 0 && U < NU) {
                const int it_raw = DXB * kb + j16;
                const int it = it_raw < nsteps ? it_raw : nsteps - 1;          // (clamped: computed, not stored)
                // what this unit does (compile-time): split slice `ss` (-1: none), multiply tile `mc` of slice `ms` (-1: none)
                constexpr int ss = NCT == 2 ? ((U % 2 == 0 && U < 12) ? U / 2 : -1) : U;
                constexpr int mc = NCT == 2 ? (U % 2 == 0 ? 1 : 0) : 0;
                constexpr int ms = NCT == 2 ? (U % 2 == 0 ? U / 2 - 1 : U / 2) : U;
                if constexpr (ms >= 0) {
                    // (pin: the planes the products read are "produced" here, behind the e_u hand-off)
                    xbf8 (&mp)[NP] = bp[ms % NBUF];
#pragma unroll
                    for (int q2 = 0; q2 < NP; ++q2) asm volatile("" : "+v"(mp[q2]));
                }
                if constexpr (ss >= 0) {
                    const float *src = &dact[it & (NSLOT - 1)][32 * ss + 8 * kg];
                    const v4f b0 = *reinterpret_cast<const v4f *>(src), b1 = *reinterpret_cast<const v4f *>(src + 4);
                    split_bf16x8<NP>(b0, b1, bp[ss % NBUF]);
                    if constexpr (ss == 0 && SCAT) {
                        const int t = t_hi - 1 - it, ti = t - a.front_zero;
#pragma unroll
                        for (int c = 0; c < NCT; ++c) {
                            sc_id[c] = reinterpret_cast<const int *>(a.scatter_ids)[(b * (long)a.Tids + (ti > 0 ? ti : 0)) * a.F + c];
                            sc_dl[c] = *reinterpret_cast<const xf4 *>(dlb + 16 * c);
                        }
                        sc_flag = (it_raw < nsteps && ti >= 0) ? (t == a.last_t ? 2 : 1) : 0;
                    }
                }
                if constexpr (ms < 0 && ss >= 0) {
                    xbf8 (&sp)[NP] = bp[ss % NBUF];
#pragma unroll
                    for (int q2 = 0; q2 < NP; ++q2) asm volatile("" : "+v"(sp[q2]));
                }
                if constexpr (ms >= 0) {
                    xbf8 (&mp)[NP] = bp[ms % NBUF];
                    xbf8 (&w)[NR] = wA[ms * NCT + mc];
                    xf4 acc = ms == 0 ? xf4{0.f, 0.f, 0.f, 0.f} : xacc[mc];
                    // products of order <= NP - 1, smallest terms first
                    if constexpr (NP == 3) {
                        const xbf8 w2 = wlo_[ms * NCT + mc][lane];
                        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[0], mp[2], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[1], mp[1], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2, mp[0], acc, 0, 0, 0);
                    }
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[0], mp[1], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[1], mp[0], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[0], mp[0], acc, 0, 0, 0);
                    // (pin: "used" here -- not sunk into the next iteration)
                    asm volatile("" : "+a"(acc[0]), "+a"(acc[1]), "+a"(acc[2]), "+a"(acc[3]));
                    xacc[mc] = acc;
                    if constexpr (ss >= 0) {                 // (... nor is the split: its reads' latency lies under the products)
                        xbf8 (&sp)[NP] = bp[ss % NBUF];
#pragma unroll
                        for (int q2 = 0; q2 < NP; ++q2) asm volatile("" : "+v"(sp[q2]));
                    }
                    // (no branch in here -- the feeder's prefetch loads are in flight and a branch's join would wait for all of
                    //  them: a clamped lane holds the last row's result and stores it to the last row's place once more)
                    if constexpr (ms == 5) {
                        constexpr int c = mc;
                        if constexpr (!SCAT) {
                            *reinterpret_cast<xf4 *>(dxb + (long)(t_hi - 1 - it) * DXD + 16 * c) = acc;
                        } else {
                            const bool ok = sc_flag != 0 && !((a.mask_id0 & HPMN_ID_MASK0) && sc_id[c] == 0);
                            const float m = sc_flag == 2 ? dlm : 0.f;
                            float *dst = ok ? a.d_emb + (long)sc_id[c] * 16 + 4 * kg : dxb + (long)(t_hi - 1 - it) * DXD + 16 * c;
                            atomicAdd(dst, fmaf(m, sc_dl[c][0], acc[0]));
                            atomicAdd(dst + 1, fmaf(m, sc_dl[c][1], acc[1]));
                            atomicAdd(dst + 2, fmaf(m, sc_dl[c][2], acc[2]));
                            atomicAdd(dst + 3, fmaf(m, sc_dl[c][3], acc[3]));
                        }
                    }
                }
            }
        };
        auto dx_block = [&](int kb) __attribute__((always_inline)) {            // a whole block at once (behind the unrolled loop, and the last blocks)
            dx_unit(std::integral_constant<int, 0>{}, kb); dx_unit(std::integral_constant<int, 1>{}, kb);
            dx_unit(std::integral_constant<int, 2>{}, kb); dx_unit(std::integral_constant<int, 3>{}, kb);
            dx_unit(std::integral_constant<int, 4>{}, kb); dx_unit(std::integral_constant<int, 5>{}, kb);
            dx_unit(std::integral_constant<int, 6>{}, kb); dx_unit(std::integral_constant<int, 7>{}, kb);
            dx_unit(std::integral_constant<int, 8>{}, kb); dx_unit(std::integral_constant<int, 9>{}, kb);
            dx_unit(std::integral_constant<int, 10>{}, kb); dx_unit(std::integral_constant<int, 11>{}, kb);
            dx_unit(std::integral_constant<int, 12>{}, kb);
        };

        float *dap = a.d_act + (b * (long)T + (t_hi - 1)) * 3 * H + l;     // row of iteration 0
        int seen = 0;
        // one iteration: e_u(k); the d_act row of iteration k-1 goes out behind it; then (LOOPDX) one unit of block kb
        auto iter = [&](int k, int p, bool store_prev, auto uc, int kb) {
            while (seen <= k) seen = lds_counter_peek(&dau_pub);
            asm volatile("" ::: "memory");
            const float *row = dact[LOOPDX ? (k & (NSLOT - 1)) : p];
            const float *old = dact[LOOPDX ? ((k - 1) & (NSLOT - 1)) : (p ^ 1)];
            const float euv = split_matvec<2>(row + H, wuS, lane);
            const float o_dar = old[l], o_dau = old[H + l], o_dcp = old[2 * H + l];
            eU[p][l] = euv;
            lds_counter_set(&eu_pub, k + 1);
            if (store_prev) {
                dap[0] = o_dar;
                dap[H] = o_dau;
                dap[2 * H] = o_dcp;
                dap -= 3 * H;
            }
            dx_unit(uc, kb);                           // (the reads of block kb end 2 iterations before the chain wave can
                                                       //  overwrite its rows, see the loop below)
        };
        const std::integral_constant<int, -1> no_unit{};

        // iteration 0 has no previous row to store: peeled together with iteration 1
        int q = 0;
        if (nfull > 0) {
            load_chunk(FR_AHEAD + 1, w0);
            iter(0, 0, false, no_unit, -1);
            iter(1, 1, true, no_unit, -1);
            park_chunk(FR_AHEAD, w1);
            q = 1;
        }
        // four iterations (two chunks); U0 >= 0: they carry units U0 .. U0 + 3 of block kb
        auto group = [&](int qq, auto u0c, int kb) {
            constexpr int U0 = decltype(u0c)::value;
            constexpr int S = U0 >= 0 ? 1 : 0;
            load_chunk(qq + FR_AHEAD + 1, w1);
            iter(2 * qq, 0, true, std::integral_constant<int, U0>{}, kb);
            iter(2 * qq + 1, 1, true, std::integral_constant<int, U0 + S>{}, kb);
            park_chunk(qq + FR_AHEAD, w0);
            load_chunk(qq + FR_AHEAD + 2, w0);
            iter(2 * qq + 2, 0, true, std::integral_constant<int, U0 + 2 * S>{}, kb);
            iter(2 * qq + 3, 1, true, std::integral_constant<int, U0 + 3 * S>{}, kb);
            park_chunk(qq + FR_AHEAD + 1, w1);
        };
        if constexpr (LOOPDX) {
            // 16 iterations k = 16 n + 2 .. 16 n + 17 per trip, unit p at iteration 16 n + 2 + p (p < NU <= 13; the ring is last
            // READ by unit 10 at D = 32, 5 at D = 16): block
            // kb = n - 1, whose rows were complete at iteration 16 n and stay in the ring until the chain wave starts iteration
            // 16 (n + 1) -- which needs e_u(16 n + 15), published two iterations after the last unit's reads.
            if (q + 7 < nfull) {                   // (the first trip: no block is complete yet)
                group(q, std::integral_constant<int, -1>{}, -1);
                group(q + 2, std::integral_constant<int, -1>{}, -1);
                group(q + 4, std::integral_constant<int, -1>{}, -1);
                group(q + 6, std::integral_constant<int, -1>{}, -1);
                q += 8;
            }
            for (; q + 7 < nfull; q += 8) {
                const int kb = ((2 * q - 2) >> 4) - 1;
                group(q, std::integral_constant<int, 0>{}, kb);
                group(q + 2, std::integral_constant<int, 4>{}, kb);
                group(q + 4, std::integral_constant<int, 8>{}, kb);
                group(q + 6, std::integral_constant<int, 12>{}, kb);     // (unit 12: D = 32's last tile; 13 .. 15 do not exist)
            }
            // the block that was complete when the loop ended, at once: up to 15 iterations follow, the last of which may
            // start overwriting it (a pause of ~2 steps for the chain wave, once per launch)
            if (q > 1) dx_block(((2 * q - 2) >> 4) - 1);
        }
        for (; q + 1 < nfull; q += 2) group(q, std::integral_constant<int, -1>{}, -1);
        if (q < nfull) {
            iter(2 * q, 0, true, no_unit, -1);
            iter(2 * q + 1, 1, true, no_unit, -1);
            park_chunk(q + FR_AHEAD, w0);
            q += 1;
        }
        if (nsteps & 1) iter(nsteps - 1, 0, nsteps > 1, no_unit, -1);
        // the last iteration's row: the chain wave reports "da_r of the last step is written" as dau_pub = nsteps+1
        {
            while (seen <= nsteps) {
                seen = lds_counter_peek(&dau_pub);
                if (seen <= nsteps) __builtin_amdgcn_s_sleep(1);
            }
            asm volatile("" ::: "memory");
            const float *row = dact[(nsteps - 1) & (NSLOT - 1)];
            dap[0] = row[l];
            dap[H] = row[H + l];
            dap[2 * H] = row[2 * H + l];
        }
        if constexpr (LOOPDX) {
            // the blocks the loop did not reach: all of their rows are still in the ring (at most the last 31 iterations)
            const int nblk = (nsteps + DXB - 1) / DXB;
            const int done = nfull >= 9 ? ((nfull - 9) / 8 + 1) - 1 : -1;      // trips of the 16-iteration loop - 1 = last block done
            for (int kb = done + 1; kb < nblk; ++kb) dx_block(kb);
        }
    } else {
    // ====================================================================== chain wave
    __builtin_amdgcn_s_setprio(3);
    f2 wcS[2][16], wrS[2][16];
    split_matvec_weights<2>(a.wc + (long)D * H, H, lane, wcS);
    split_matvec_weights<2>(a.wg + (long)D * 2 * H, 2 * H, lane, wrS);

    float dh = (t_hi == T) ? a.d_h_last[b * a.d_h_last_stride + l] : a.dh_carry[b * H + l];
    settle(dh);

    int fed_seen = 0, eu_seen = 0;
    auto wait_fed = [&](int need) {
        while (fed_seen < need) {
            fed_seen = lds_counter_peek(&fed);
            if (fed_seen < need) __builtin_amdgcn_s_sleep(1);
        }
        asm volatile("" ::: "memory");
    };
    wait_fed(1);
    v4f ca = ringA[0][l];
    f2 cb = ringB[0][l];

    auto step = [&](int k, int p) {
        float *row = dact[LOOPDX ? (k & (NSLOT - 1)) : p];
        const float dhin = dh + ca.x;
        const float dcp = dhin * ca.y;
        const float dau = dhin * ca.z;
        const float k3 = ca.w, r = cb.x, u = cb.y;
        row[H + l] = dau;
        row[2 * H + l] = dcp;
        lds_counter_set(&dau_pub, k + 1);                            // the feeder may start on e_u(k)
        wave_sync();
        const float drh = split_matvec<2>(row + 2 * H, wcS, lane);
        row[l] = drh * k3;
        wave_sync();
        // the next step's coefficients (parked several steps ago; the feeder parks past the end as well): issued
        // here so that their latency lies underneath the second product
        wait_fed(k + 2);
        const int slot = (k + 1) & (FR_STEPS - 1);
        ca = ringA[slot][l];
        cb = ringB[slot][l];
        const float er = split_matvec<2>(row, wrS, lane);
        const float part = fmaf(dhin, u, fmaf(drh, r, er));
        while (eu_seen <= k) {
            eu_seen = lds_counter_peek(&eu_pub);
            if (eu_seen <= k) __builtin_amdgcn_s_sleep(1);
        }
        asm volatile("" ::: "memory");
        dh = part + eU[p][l];
        wave_sync();
    };

    for (int q = 0; q < nfull; ++q) {
        step(2 * q, 0);
        step(2 * q + 1, 1);
    }
    if (nsteps & 1) step(nsteps - 1, 0);
    lds_counter_set(&dau_pub, nsteps + 1);                           // da_r of the last step is in LDS
    if (t_lo0 > 0) a.dh_carry[b * H + l] = dh;
    }

    if constexpr (DX && !LOOPDX) {
        // ================================================================== epilogue: the input gradient of this launch's steps
        __builtin_amdgcn_s_setprio(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // the feeders' d_act rows have reached memory
        __syncthreads();                                              // (waves of a sequence that does not exist left before the first barrier)
        const int j = lane & 15, g = lane >> 4;
        constexpr int NCT = DXD / 16;
        // A operands: W[col = 16 ct + j][f = 16 kq + 4 g + c], W = [wg[0:D] | wc[0:D]] (input rows x 3H gate columns)
        float wx[NCT][12][4];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int kq = 0; kq < 12; ++kq) {
                const long col = 16 * ct + j;
                const v4f v = kq < 8 ? *reinterpret_cast<const v4f *>(a.wg + col * 2 * H + 16 * kq + 4 * g)
                                     : *reinterpret_cast<const v4f *>(a.wc + col * H + 16 * (kq - 8) + 4 * g);
                wx[ct][kq][0] = v.x; wx[ct][kq][1] = v.y; wx[ct][kq][2] = v.z; wx[ct][kq][3] = v.w;
            }
        // blocks of 16 steps [t_lo0 + 16 q, ...), q = role, role + 2, ...: B operand of lane (j, g) = d_act[t][16 kq + 4 g ..+3]
        const int nblk = (nsteps + DXB - 1) / DXB;
        const float *src = a.d_act + (b * (long)T + t_lo0) * 3 * H + 4 * g;
        float *dst = a.d_x != nullptr ? a.d_x + (b * (long)T + t_lo0) * DXD + 4 * g : nullptr;
        auto fetch = [&](int q, v4f (&v)[12]) {
            int tr = DXB * q + j;
            tr = tr < nsteps ? tr : nsteps - 1;                       // (clamped: loaded, computed, not stored)
#pragma unroll
            for (int kq = 0; kq < 12; ++kq) v[kq] = *reinterpret_cast<const v4f *>(src + (long)tr * 3 * H + 16 * kq);
        };
        // Fused scatter (a.d_emb != NULL; E == 16, so column tile ct IS id column ct): a tile's 16 rows go into the table
        // gradient from here.  Lane (j, g) holds four elements of step 16 q + j's row; when the block's 16 steps share one id
        // -- the constant uid column, padding -- the rows are summed over j inside the 16-lane groups first, and runs that
        // continue over this wave's next block keep accumulating in registers: one atomic row add per run, not per step.
        constexpr bool scat = SCAT;       // (a template switch: the scatter's registers would cost the layers that do not
                                          //  use it their place beside a weight-gradient workgroup)
        int run_id[NCT];
        f4m run_acc[NCT];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) { run_id[ct] = -1; run_acc[ct] = f4m{0.f, 0.f, 0.f, 0.f}; }
        auto flush = [&](int ct) {
            if (run_id[ct] >= 0 && j == 0 && !((a.mask_id0 & HPMN_ID_MASK0) && run_id[ct] == 0)) {
                float *row = a.d_emb + (long)run_id[ct] * 16 + 4 * g;
                atomicAdd(row, run_acc[ct][0]); atomicAdd(row + 1, run_acc[ct][1]);
                atomicAdd(row + 2, run_acc[ct][2]); atomicAdd(row + 3, run_acc[ct][3]);
            }
            run_id[ct] = -1;
            run_acc[ct] = f4m{0.f, 0.f, 0.f, 0.f};
        };
        v4f cur[12], nxt[12];
        if (role < nblk) fetch(role, cur);
        for (int q = role; q < nblk; q += 2) {
            if (q + 2 < nblk) fetch(q + 2, nxt);
            const int tr = DXB * q + j;
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                f4m acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kq = 0; kq < 12; ++kq)
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wx[ct][kq][c], cur[kq][c], acc, 0, 0, 0);
                if (a.d_x != nullptr && tr < nsteps) *reinterpret_cast<f4m *>(dst + (long)tr * DXD + 16 * ct) = acc;
                if (scat) {
                    const int t = t_lo0 + tr;                          // scan step of this lane's row
                    if (a.d_last != nullptr && t == a.last_t && tr < nsteps) {
                        const f4m dl = *reinterpret_cast<const f4m *>(a.d_last + b * (long)DXD + 16 * ct + 4 * g);
                        acc += dl;
                    }
                    const int ti = t - a.front_zero;
                    const bool valid = tr < nsteps && ti >= 0;
                    const int id = valid ? reinterpret_cast<const int *>(a.scatter_ids)[(b * (long)a.Tids + ti) * a.F + ct] : -2;   // (int32 ids only: api.hip)
                    const int id0 = __builtin_amdgcn_readfirstlane(id);
                    const bool uniform = __builtin_amdgcn_ballot_w64(id != id0) == 0;     // (wave-uniform)
                    if (uniform && id0 >= 0) {
                        // sum over the block's 16 steps: xor-shuffles stay inside the 16-lane group of equal g
                        f4m sum = acc;
#pragma unroll
                        for (int m = 1; m < 16; m <<= 1) {
                            sum[0] += __shfl_xor(sum[0], m); sum[1] += __shfl_xor(sum[1], m);
                            sum[2] += __shfl_xor(sum[2], m); sum[3] += __shfl_xor(sum[3], m);
                        }
                        if (id0 != run_id[ct]) flush(ct);
                        run_id[ct] = id0;
                        run_acc[ct] += sum;
                    } else {
                        flush(ct);
                        if (valid && !((a.mask_id0 & HPMN_ID_MASK0) && id == 0)) {
                            float *row = a.d_emb + (long)id * 16 + 4 * g;
                            atomicAdd(row, acc[0]); atomicAdd(row + 1, acc[1]); atomicAdd(row + 2, acc[2]); atomicAdd(row + 3, acc[3]);
                        }
                    }
                }
            }
#pragma unroll
            for (int kq = 0; kq < 12; ++kq) cur[kq] = nxt[kq];
        }
        if (scat) {
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) flush(ct);
        }
    }
}

bool gru_scan_bwd_feed_dx_width(int D) { return D == 16 || D == 32 || D == 64; }
// the epilogue's fused scatter: id column f is column tile f of the input gradient
bool gru_scan_bwd_feed_scatter_ok(int D, int F, int E) { return E == 16 && D == F * 16 && gru_scan_bwd_feed_dx_width(D); }

template <bool CFH, int NP>
static int feed_launch(const HpmnGruBwd &a, hipStream_t st) {
    const dim3 grid((a.B + 1) / 2);
    if (a.d_emb != nullptr && a.d_x != nullptr && a.D <= 32 && (a.flags & HPMN_BWD_SCATTER_INLOOP)) {
        // the in-loop input gradient with the scatter fused into it (d_x: scratch for the rows that are not added)
        if (a.D == 16) hipLaunchKernelGGL((gru_scan_bwd_feed_kernel<16, true, true, CFH, NP>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((gru_scan_bwd_feed_kernel<32, true, true, CFH, NP>), grid, dim3(256), 0, st, a);
        return check_launch();
    }
    if (a.d_emb != nullptr) {
        if (a.D == 16) hipLaunchKernelGGL((gru_scan_bwd_feed_kernel<16, true, false, CFH>), grid, dim3(256), 0, st, a);
        else if (a.D == 32) hipLaunchKernelGGL((gru_scan_bwd_feed_kernel<32, true, false, CFH>), grid, dim3(256), 0, st, a);
        else if (a.D == 64) hipLaunchKernelGGL((gru_scan_bwd_feed_kernel<64, true, false, CFH>), grid, dim3(256), 0, st, a);
        else return HPMN_EUNSUPPORTED;
        return check_launch();
    }
    // HPMN_BWD_DX_INLOOP=0: the input gradient of D <= 32 as an epilogue too (the round-2/3 form)
    static const int inloop = [] { const char *e = getenv("HPMN_BWD_DX_INLOOP"); return e ? atoi(e) : 1; }();
    if (a.d_x == nullptr) hipLaunchKernelGGL((gru_scan_bwd_feed_kernel<0, false, false, CFH>), grid, dim3(256), 0, st, a);
    else if (a.D == 16 && inloop) hipLaunchKernelGGL((gru_scan_bwd_feed_kernel<16, false, true, CFH, NP>), grid, dim3(256), 0, st, a);
    else if (a.D == 32 && inloop) hipLaunchKernelGGL((gru_scan_bwd_feed_kernel<32, false, true, CFH, NP>), grid, dim3(256), 0, st, a);
    else if (a.D == 16) hipLaunchKernelGGL((gru_scan_bwd_feed_kernel<16, false, false, CFH>), grid, dim3(256), 0, st, a);
    else if (a.D == 32) hipLaunchKernelGGL((gru_scan_bwd_feed_kernel<32, false, false, CFH>), grid, dim3(256), 0, st, a);
    else if (a.D == 64) hipLaunchKernelGGL((gru_scan_bwd_feed_kernel<64, false, false, CFH>), grid, dim3(256), 0, st, a);
    else return HPMN_EUNSUPPORTED;
    return check_launch();
}

int gru_scan_bwd_feed_launch(const HpmnGruBwd &a, hipStream_t st) {
    static const int np = [] { const char *e = getenv("HPMN_DX_PLANES"); return (e && atoi(e) == 2) ? 2 : 3; }();
    if (np == 2) return (a.flags & HPMN_BWD_CANDIDATE_FROM_HS) ? feed_launch<true, 2>(a, st) : feed_launch<false, 2>(a, st);
    return (a.flags & HPMN_BWD_CANDIDATE_FROM_HS) ? feed_launch<true, 3>(a, st) : feed_launch<false, 3>(a, st);
}

}  // namespace hpmn
