// Weight / bias gradients of one GRU layer (H = 64) on the bf16 matrix pipe with SPLIT operands (round 4).
//
//   dWg += [x | h_prev]^T d_act[:, 0:2H]      dbg += sum_rows d_act[:, 0:2H]
//   dWc += [x | r*h_prev]^T d_act[:, 2H:3H]   dbc += sum_rows d_act[:, 2H:3H]
//
// Same decomposition as gru_wgrad.hip (a workgroup owns whole sequences, its waves split the OUTPUT rows -- wave w < DT: 32
// input columns, the others 32 state columns -- and keep their 3H/32 accumulator tiles resident; slabs + wgrad_reduce_kernel),
// but the products run as v_mfma_f32_32x32x16_bf16: every fp32 operand is split x = hi + lo (hi = bf16(x), lo = bf16(x - hi),
// |x - hi - lo| <= 2^-17 |x|, fp32's exponent range: no scaling) and a tile is three products, hi*hi + hi*lo + lo*hi, fp32
// accumulate: 3 x 32 cycles per 16 rows where the fp32 instruction needs 8 x 64 -- 5.3x less time on the matrix pipe that the
// reverse scan of layer 0 shares with six layers' weight gradients.
//
// The bf16 instruction wants 8 consecutive k (= rows of the [B*T, .] operands) per lane, the tensors are row-major with the
// features contiguous: a transpose.  It is done by the global LOADS, not in LDS: the lane that will hold column c, rows
// 8 kg .. 8 kg + 7 of a 32-column block loads exactly those eight dwords -- each of the eight load instructions covers 2 rows
// x 128 contiguous bytes across the wave, fully coalesced --, splits them in registers (v_cvt_pk_bf16_f32: ~22 VALU per 8
// values) and writes its two 16-byte fragments (hi, lo) to LDS IN LANE ORDER; a consumer wave reads a block's fragment with one
// linear ds_read_b128 per lane.  No bank conflicts on either side, and every element is converted once per workgroup (each
// wave stages ~1/NW of a tile's blocks), not once per consuming wave.
#include <cstdlib>

#include "common.h"

namespace hpmn {

typedef float f32x16b __attribute__((ext_vector_type(16)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));


constexpr int BR = 16;            // rows per tile == k of the instruction

// NP planes of an fp32 value: x = p0 + p1 (+ p2), p0 = bf16(x), p1 = bf16(x - p0), p2 = bf16(x - p0 - p1) (fp32's exponent
// range: no scaling).  Two planes leave 2^-17 |x| and the products h*h + h*l + l*h drop l*l (2^-18 of the product): measured
// 5e-6 of max|grad|.  THREE planes (round 6: VERDICT r5 #1) leave 2^-25 |x|, and the six products of order <= 2 -- hH, hM, mH,
// hL, lH, mM -- drop terms < 2^-24 of the product: fp32-equivalent (what read_path.hip's dense_bf does).
template <int NP>
struct Frag { bf8 p[NP]; };
template <int NP>
__device__ __forceinline__ Frag<NP> split8(const float (&v)[8]) {
    Frag<NP> f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float rest = v[j];
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            f.p[q][j] = (__bf16)rest;
            if (q + 1 < NP) rest -= (float)f.p[q][j];
        }
    }
    return f;
}
// acc += A * B over the planes' products of order <= NP - 1
template <int NP>
__device__ __forceinline__ f32x16b mma_planes(const bf8 (&a)[NP], const bf8 (&b)[NP], f32x16b acc) {
#pragma unroll
    for (int o = NP - 1; o >= 0; --o)           // smallest terms first
#pragma unroll
        for (int i = 0; i <= o; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[o - i], acc, 0, 0, 0);
    return acc;
}

__host__ __device__ inline long wgrad_slab_floats_bf(int D, int H) { return (long)(D + H) * 3 * H + 3 * H; }   // == gru_wgrad.hip

// blocks of a tile: [0, DT) x, [DT, DT+HT) h_prev (+ r*h_prev) -- each kept in its wave's registers (r6) --, then 3HT blocks of d_act in LDS.
// The body is instantiated PER WAVE (W is a template argument, the kernel dispatches once at the top): which blocks a wave
// stages and which source each comes from are then compile-time facts and the time loop is straight-line code -- with the
// wave index at run time every staging load sat inside a (uniform) branch and was waited for at its join, vmcnt(0), before
// the tile's matrix instructions could start (the lesson of gru_scan_fwd.hip and of scatter_sorted.hip again).  Loads are
// unconditional: rows beyond the range are clamped to a valid row and zeroed by a select.
// CS = column split (H = 128): blockIdx.y picks one of CS groups of 3 HT / CS consecutive 32-column tiles of d_act -- twelve
// accumulator tiles would be 192 registers; each group stages only ITS columns of d_act (plus x, h_prev, r: re-read per group,
// as in the fp32 kernel).
template <int HT, int DT, int CS, int W, int NP, bool PF2>
__device__ __forceinline__ void wgrad_bf16_wave(const HpmnGruWgrad &a, bf8 (*img)[3 * HT / CS][NP][64], const int bx,
                                                const int by, const int tsplit) {
    constexpr int H = 32 * HT;
    constexpr int NJ = 3 * HT / CS;            // 32-column tiles of d_act held by this workgroup
    constexpr int NW = HT + DT;                // waves
    constexpr int NBLK = DT + 2 * HT + NJ;
    const int jb = by * NJ;                    // first (global) column tile
    // staging TASKS: [0, DT) the x blocks, [DT, DT + HT) one per 32 state columns -- h_prev is loaded ONCE and parked twice, as
    // h_prev and as r * h_prev (two blocks of the image; r4: they were two tasks with two loads of the same rows) --, then the
    // NJ d_act blocks
    constexpr int NTASK = DT + HT + NJ;
    constexpr int MAXT = (NTASK - W + NW - 1) / NW;  // staging tasks of this wave: tasks W, W + NW, ...
    constexpr bool role_x = W < DT;
    constexpr int tile = role_x ? W : W - DT;

    const int lane = threadIdx.x & 63;
    const int c = lane & 31, kg = lane >> 5;
    const int T = a.T, D = a.D;

    f32x16b acc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    float bsum[MAXT];                           // column sums of the d_act blocks THIS wave stages (this lane's 8 rows)
#pragma unroll
    for (int i = 0; i < MAXT; ++i) bsum[i] = 0.f;

    // (tsplit > 1: the 16-row tiles of a sequence range are cut into `tsplit` consecutive pieces, one workgroup -- and one slab --
    //  each: more, shorter workgroups for a launch that has the chip to itself, see gru_wgrad_bf16_launch)
    const int piece = tsplit > 1 ? bx % tsplit : 0;
    const int bq = tsplit > 1 ? bx / tsplit : bx;
    const int b_begin = bq * a.seq_per_wg;
    const int b_end = (b_begin + a.seq_per_wg) < a.B ? (b_begin + a.seq_per_wg) : a.B;
    const int tb = a.t_begin, te = a.t_len > 0 ? a.t_begin + a.t_len : T;
    const int ipt_all = (te - tb + BR - 1) / BR;
    const int i_lo = tsplit > 1 ? (int)((long)ipt_all * piece / tsplit) : 0;
    const int ipt = (tsplit > 1 ? (int)((long)ipt_all * (piece + 1) / tsplit) : ipt_all) - i_lo;     // tiles per sequence, this workgroup
    const int niter = (b_end - b_begin) * ipt;

    // raw (unconverted) values of this wave's staging tasks for one tile.  Nothing is computed on them at load time -- not even
    // the zeroing of rows beyond the range (clamped addresses; the mask is recomputed from the tile index when the tile is
    // parked) -- so that two tiles' worth of them can be in flight without anything waiting on a load.
    constexpr bool has_rh = (W >= 0) && ([] { for (int i = 0; i < MAXT; ++i) { const int q = W + i * NW; if (q >= DT && q < DT + HT) return true; } return false; }());
    struct Raw { float v[MAXT][8]; float r2[has_rh ? 8 : 1]; };
    auto load_raw = [&](int it, Raw &g) {
        const int b = b_begin + it / ipt;
        const int t0 = tb + (i_lo + it % ipt) * BR + 8 * kg;
        long rowx[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) rowx[j] = (long)b * T + ((t0 + j) < te ? (t0 + j) : (te - 1));
#pragma unroll
        for (int i = 0; i < MAXT; ++i) {
            const int q = W + i * NW;           // (a constant after unrolling)
            if (q < DT) {
                const int col = 32 * q + c;
                const int colc = col < D ? col : D - 1;
#pragma unroll
                for (int j = 0; j < 8; ++j) g.v[i][j] = a.x[rowx[j] * D + colc];
            } else if (q < DT + HT) {
                const int col = 32 * (q - DT) + c;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    g.v[i][j] = a.hs[(rowx[j] + b) * H + col];       // hs has T + 1 rows per sequence
                    g.r2[j] = a.gates[rowx[j] * 3 * H + col];
                }
            } else {
                const int col = 32 * (jb + q - DT - HT) + c;
#pragma unroll
                for (int j = 0; j < 8; ++j) g.v[i][j] = a.d_act[rowx[j] * 3 * H + col];
            }
        }
    };
    // r6: task 0 of wave W is block W -- the wave's OWN operand block (x tile W for the input-column waves, h_prev tile W - DT
    // and r * h_prev for the others): staged and consumed by the same wave, lane for lane (the fragment order in LDS was lane
    // order).  It stays in REGISTERS (ap: x / h_prev, cp: x / r * h_prev); only the d_act blocks, which every wave reads, go
    // through LDS -- 5 of 11 blocks' writes and reads less at H = 64, D = 32, and the image shrinks from 11 to 6 blocks (with
    // three planes: 36 KB instead of 66).
    static_assert(MAXT >= 1, "every wave stages its own operand block");
    bf8 ap[NP], cp[NP];
    auto park = [&](int buf, const Raw &g, int it) {
        const int t0 = tb + (i_lo + it % ipt) * BR + 8 * kg;
#pragma unroll
        for (int i = 0; i < MAXT; ++i) {
            const int q = W + i * NW;
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                bool live = (t0 + j) < te;
                if (q < DT) live = live && (32 * q + c) < D;
                v[j] = live ? g.v[i][j] : 0.f;
            }
            const Frag<NP> f = split8<NP>(v);
            if (q < DT + HT) {                                                      // (i == 0: the wave's own block)
#pragma unroll
                for (int pl = 0; pl < NP; ++pl) ap[pl] = f.p[pl];
                if (q >= DT) {                                                      // r * h_prev (not stored by the forward)
                    float w[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) w[j] = v[j] * g.r2[j];
                    const Frag<NP> f2 = split8<NP>(w);
#pragma unroll
                    for (int pl = 0; pl < NP; ++pl) cp[pl] = f2.p[pl];
                } else {
#pragma unroll
                    for (int pl = 0; pl < NP; ++pl) cp[pl] = f.p[pl];
                }
            } else {
#pragma unroll
                for (int pl = 0; pl < NP; ++pl) img[buf][q - DT - HT][pl][lane] = f.p[pl];
                float s = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) s += v[j];
                bsum[i] += s;
            }
        }
    };

    // Prefetch distance ONE tile: the loads of tile it + 1 are issued in front of tile it's matrix instructions and parked
    // behind them.  (Distance two -- two tiles of raw values in registers -- was tried: 165-245 registers spilled beside the
    // 96 accumulators; the second workgroup on the CU is what overlaps the round trip instead.)
    auto compute = [&](int buf) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            bf8 bp[NP];
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) bp[pl] = img[buf][j][pl][lane];
            // gate columns (tile < 2 HT) pair with x / h_prev, candidate columns with x / r * h_prev
            const bool gate = (jb + j) < 2 * HT;               // (uniform per workgroup)
            bf8 xp[NP];
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) xp[pl] = gate ? ap[pl] : cp[pl];
            acc[j] = mma_planes<NP>(xp, bp, acc[j]);
        }
    };
    const int last = niter - 1;
    Raw sa;
    if (niter > 0) {
        load_raw(0, sa);
        park(0, sa, 0);
    }
    if constexpr (PF2) {
        // (r6) prefetch distance TWO: tile it + 2's loads are issued in front of tile it's matrix instructions and parked a whole
        // iteration later -- twice the bytes in flight per workgroup.  32 more registers: affordable since the H = 64 shapes run
        // at two workgroups per CU (256 registers per wave), which r4's attempt at three per CU (170) did not have.
        Raw sb;
        auto clampi = [&](int i) { return i < last ? i : (last > 0 ? last : 0); };
        load_raw(clampi(1), sa);
        __syncthreads();
        int it = 0;
        for (; it + 1 < niter; it += 2) {
            load_raw(clampi(it + 2), sb);
            compute(0);
            park(1, sa, it + 1);
            __syncthreads();
            load_raw(clampi(it + 3), sa);
            compute(1);
            if (it + 2 < niter) park(0, sb, it + 2);
            __syncthreads();
        }
        if (it < niter) compute(0);
    } else {
        __syncthreads();
        for (int it = 0; it < niter; ++it) {
            const int buf = it & 1;
            load_raw(it < last ? it + 1 : last, sa);          // (clamped: no branch around the loads)
            compute(buf);
            if (it < last) park(buf ^ 1, sa, it + 1);
            __syncthreads();
        }
    }

    // ---- epilogue: this workgroup's partial slab (summed by wgrad_reduce_kernel), C/D layout of the 32x32 instructions:
    //      rows (r & 3) + 8 (r >> 2) + 4 (lane >> 5), column lane & 31
    float *slab = a.workspace + (long)bx * wgrad_slab_floats_bf(D, H);
    float *s_wg = slab, *s_bg = slab + (long)(D + H) * 2 * H, *s_wc = s_bg + 2 * H;
    float *s_bc = s_wc + (long)(D + H) * H;
    const int row_base = role_x ? 32 * tile : D + 32 * tile;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int jg = jb + j;
        const bool gate_tile = jg < 2 * HT;
        float *dst = gate_tile ? s_wg : s_wc;
        const int ld = gate_tile ? 2 * H : H;
        const int col = gate_tile ? 32 * jg + c : 32 * (jg - 2 * HT) + c;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = (r & 3) + 8 * (r >> 2) + 4 * kg;
            if (!role_x || 32 * tile + i < D) dst[(long)(row_base + i) * ld + col] = acc[j][r];
        }
    }
#pragma unroll
    for (int i = 0; i < MAXT; ++i) {
        const int q = W + i * NW;
        if (q < DT + HT) continue;
        const int j = jb + q - DT - HT;
        const float tot = bsum[i] + __shfl_xor(bsum[i], 32);     // the two half-waves hold rows 0-7 / 8-15 of every tile
        if (kg == 0) {
            if (j < 2 * HT) s_bg[32 * j + c] = tot;
            else            s_bc[32 * (j - 2 * HT) + c] = tot;
        }
    }
}

// LB3: three workgroups per CU for the three-wave shape (H = 64, D <= 32): 168 registers, the rest spilled (12 dwords with two
// planes, 27 with three) -- against two workgroups per CU without spills (the default since r6, see launch_bf16).
template <int HT, int DT, int CS, bool XCD = true, int NP = 3, bool LB3 = true, bool PF2 = false>
__global__ __launch_bounds__(64 * (HT + DT), (HT + DT) == 3 ? (LB3 ? 3 : 2) : ((HT + DT) > 5 ? 1 : 2)) void gru_wgrad_bf16_kernel(const HpmnGruWgrad a) {
    static_assert((3 * HT) % CS == 0 && HT + DT <= 8, "column tiles split evenly; at most eight waves");
    __shared__ __attribute__((aligned(16))) bf8 img[2][3 * HT / CS][NP][64];      // (the d_act blocks only, see wgrad_bf16_wave)
    const int wave = threadIdx.x >> 6;          // (wave-uniform: one dispatch, then straight-line code per wave)
    constexpr int NW = HT + DT;
    // The CS column groups of a sequence range read the SAME x / h_prev / r rows.  Linear workgroup id L runs on XCD L % 8
    // (observed dispatch rule, used for speed only), and every XCD has its own L2: a 1-D grid with
    //     L = ((bx / 8) * CS + by) * 8 + bx % 8
    // puts a range's column groups on ONE XCD, 8 ids apart in dispatch order -- they start together, do the same work per tile
    // and stream the shared rows at the same pace, so the second and third reader of a row find it in that XCD's L2.
    // (The 2-D grid had them on linear ids bx, bx + nwg, bx + 2 nwg: different XCDs unless nwg % 8 == 0, and dispatched a
    //  whole grid row apart.)
    int bx = blockIdx.x, by = 0;
    if constexpr (CS > 1) {
        if constexpr (XCD) {
            const int L = blockIdx.x, xcd = L & 7, slot = L >> 3;
            by = slot % CS;
            bx = (slot / CS) * 8 + xcd;
        } else {                                 // (HPMN_WGRAD_XCD=0: column group after column group, the 2-D grid's order)
            const int n8 = gridDim.x / CS;
            by = blockIdx.x / n8;
            bx = blockIdx.x - by * n8;
        }
        if (bx * a.seq_per_wg >= a.B) return;    // (the grid is rounded up to whole groups of 8 ranges)
    }
    // (in the kernel's copy of the descriptor `whole_cu` carries the launch's time split: gru_wgrad_bf16_launch)
    const int tsplit = CS == 1 && a.whole_cu > 1 ? a.whole_cu : 1;
    if (wave == 0) wgrad_bf16_wave<HT, DT, CS, 0, NP, PF2>(a, img, bx, by, tsplit);
    else if (wave == 1) wgrad_bf16_wave<HT, DT, CS, 1, NP, PF2>(a, img, bx, by, tsplit);
    else if (wave == 2) wgrad_bf16_wave<HT, DT, CS, 2, NP, PF2>(a, img, bx, by, tsplit);
    else if constexpr (NW > 3) {
        if (wave == 3) wgrad_bf16_wave<HT, DT, CS, 3, NP, PF2>(a, img, bx, by, tsplit);
        else if constexpr (NW > 4) {
            if (wave == 4) wgrad_bf16_wave<HT, DT, CS, 4, NP, PF2>(a, img, bx, by, tsplit);
            else if constexpr (NW > 5) {
                if (wave == 5) wgrad_bf16_wave<HT, DT, CS, 5, NP, PF2>(a, img, bx, by, tsplit);
                else if (wave == 6) wgrad_bf16_wave<HT, DT, CS, 6, NP, PF2>(a, img, bx, by, tsplit);
                else wgrad_bf16_wave<HT, DT, CS, 7, NP, PF2>(a, img, bx, by, tsplit);
            }
        }
    }
}

// HPMN_WGRAD_PLANES=2: the round-4/5 arithmetic (two planes, three products: ~5e-6 of max|grad|); default 3 (fp32-equivalent)
static int wgrad_planes() {
    static const int np = [] { const char *e = getenv("HPMN_WGRAD_PLANES"); const int v = e ? atoi(e) : 3; return v == 2 ? 2 : 3; }();
    return np;
}

template <int HT, int DT, int CS, int NP, bool LB3, bool PF2 = false>
static void launch_bf16_np(const HpmnGruWgrad &k, int nwg, bool solo, hipStream_t st) {
    // (one workgroup per CU beside a reverse scan, as in gru_wgrad.hip: unused dynamic LDS caps the occupancy; by default for
    //  H <= 64 only -- the H = 128 form is shaped around its column split; HPMN_WGRAD_SOLO_ROWS reaches it too)
    size_t pad = 0;
    if (solo) {
        static const size_t p = [] {
            hipFuncAttributes fa = {};
            const void *fn = reinterpret_cast<const void *>(gru_wgrad_bf16_kernel<HT, DT, CS, true, NP, LB3, PF2>);
            if (hipFuncGetAttributes(&fa, fn) != hipSuccess) return (size_t)0;
            // (r6: 72 KB, was 82: layer 0's reverse scan holds 88 KB with the three-plane in-loop product's lo fragments in LDS,
            //  and the weight-gradient workgroup still has to fit beside it in the CU's 160 KB)
            const size_t want = 72 * 1024;
            const size_t q = fa.sharedSizeBytes < want ? want - fa.sharedSizeBytes : 0;
            (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)q);
            return q;
        }();
        pad = p;
    }
    const unsigned grid = CS > 1 ? (unsigned)((nwg + 7) / 8) * 8u * CS : (unsigned)nwg;
    static const int xcd_env = [] { const char *e = getenv("HPMN_WGRAD_XCD"); return e ? atoi(e) : 1; }();
    if (CS > 1 && !xcd_env)
        hipLaunchKernelGGL((gru_wgrad_bf16_kernel<HT, DT, CS, false, NP, LB3, PF2>), dim3(grid), dim3(64 * (HT + DT)), pad, st, k);
    else
        hipLaunchKernelGGL((gru_wgrad_bf16_kernel<HT, DT, CS, true, NP, LB3, PF2>), dim3(grid), dim3(64 * (HT + DT)), pad, st, k);
}

template <int HT, int DT, int CS>
static void launch_bf16(const HpmnGruWgrad &k, int nwg, bool solo, hipStream_t st) {
    // (r6, with the wave's own operand block in registers: two workgroups per CU without spills beat three with 12 / 27 spilled
    //  dwords -- alone on the chip 165 vs 188 us (two planes), 223 vs 257 (three); C3 step 2.415 vs 2.441 -- HPMN_WGRAD_OCC=3)
    static const int occ = [] { const char *e = getenv("HPMN_WGRAD_OCC"); return e ? atoi(e) : 2; }();
    // (prefetch distance two where the launch is not beside a scan: alone on the chip 218 -> 193 us with three planes, 162 -> 151
    //  with two (D = 32); 240 -> 217 at D = 64; C3 step 2.50 -> 2.47-2.49, C2 0.925 -> 0.915.  HPMN_WGRAD_PF=1: distance one)
    static const int pf = [] { const char *e = getenv("HPMN_WGRAD_PF"); return e ? atoi(e) : 2; }();
    if constexpr (HT == 2) {                      // (H = 64: two workgroups per CU, 256 registers per wave)
        // (never beside a reverse scan -- `solo`: that launch's waves must stay within 512 - 288 = 224 registers to share a SIMD
        //  with the scan's, which is what the r6 register cliff was about; distance two is 228-253)
        if (pf == 2 && !solo && (occ == 2 || HT + DT != 3)) {
            if (wgrad_planes() == 2) launch_bf16_np<HT, DT, CS, 2, HT + DT != 3, true>(k, nwg, solo, st);
            else launch_bf16_np<HT, DT, CS, 3, HT + DT != 3, true>(k, nwg, solo, st);
            return;
        }
    }
    if constexpr (HT + DT == 3) {
        if (occ == 2) {
            if (wgrad_planes() == 2) launch_bf16_np<HT, DT, CS, 2, false>(k, nwg, solo, st);
            else launch_bf16_np<HT, DT, CS, 3, false>(k, nwg, solo, st);
            return;
        }
    }
    if (wgrad_planes() == 2) launch_bf16_np<HT, DT, CS, 2, true>(k, nwg, solo, st);
    else launch_bf16_np<HT, DT, CS, 3, true>(k, nwg, solo, st);
}

// H = 64 with D <= 64, H = 128 with D = 32 / 128.  Returns false when the shape is not served (the caller keeps the fp32 kernel).
bool gru_wgrad_bf16_launch(const HpmnGruWgrad &k, int nwg, bool solo, hipStream_t st) {
    const int DT = (k.D + 31) / 32;
    if (k.H == 64 && DT == 1) launch_bf16<2, 1, 1>(k, nwg, solo, st);
    else if (k.H == 64 && DT == 2) launch_bf16<2, 2, 1>(k, nwg, solo, st);
    else if (k.H == 128 && DT == 1) launch_bf16<4, 1, 2>(k, nwg, solo, st);   // (two column groups: x, h_prev, r re-read twice, not 3x)
    else if (k.H == 128 && DT == 4) launch_bf16<4, 4, 3>(k, nwg, solo, st);
    else return false;
    return true;
}

}  // namespace hpmn
