// Row-wise dense transforms of one GRU layer for gfx950 -- the two time-parallel, HBM-bound
// GEMMs on either side of the serial scan, both as f32 MFMA with NO LDS staging:
//
//   input projection (forward):  xp[m, 0:3H] = x[m, 0:D] [Wg[0:D] | Wc[0:D]] + [bg | bc]
//       (layer 0 gathers x[m] from (ids, emb) on the fly: id-0 mask, zero prefix; optional x_out)
//   input gradient  (backward):  dx[m, 0:D]  = d_act[m, 0:3H] [Wg[0:D] | Wc[0:D]]^T
//
// out[M,N] = in[M,K] W[K,N] with v_mfma_f32_32x32x2_f32: A[i=row][k], B[k][n].  The MFMA k index
// is split across the two half-waves (lanes 0-31: k slot 0, lanes 32-63: k slot 1); because the
// reduction order is free we map half-wave p to the CONTIGUOUS half [p*K/2, (p+1)*K/2) of the row,
// so each lane's A operands are simply 16-byte loads of its own row half (every fetched line is
// fully used; a gathered 64-byte embedding row is exactly one lane's half row at D=32) and B is
// W[p*K/2 + ks][n] kept register-stationary for the whole launch.  The product is computed
// TRANSPOSED (the weights are the MFMA "A" operand, the rows the "B" operand): in the C/D layout a
// lane then owns one output row and 4 consecutive output columns per register quad, so the
// epilogue is 16-byte stores (4x fewer store instructions than the row-major product, whose
// lanes hold 4 consecutive ROWS of one column).  The bias rides in as one extra k-step.
// Persistent waves walk 32-row tiles; the next tile's rows are in flight under the current MFMAs.
#include <cstdlib>

#include "common.h"

namespace hpmn {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int RW_WAVES = 4;     // waves per workgroup (each owns its own 32-row tiles)

// K = reduction length; a wave computes NT 32-column output tiles, the NS column groups of a row tile
// go to NS different waves (NT*NS*32 = 3H): at H=64 the 6 tiles are split 3+3 so the stationary
// weights (NT*K/2 registers) + accumulators + double-buffered A fit without scratch.
template <int K, int NT, int NS, bool GATHER, bool XOUT>
__global__ __launch_bounds__(64 * RW_WAVES, 1) void input_proj_kernel(const HpmnInputProj a) {
    constexpr int KH = K / 2;        // k range of one half-wave
    constexpr int Q = KH / 4;        // float4 loads per lane per tile
    static_assert(K % 8 == 0, "K/2 must be a multiple of 4");
    const int lane = threadIdx.x & 63;
    const int c = lane & 31, p = lane >> 5;
    const int H = a.H, N = 3 * H;
    // rows of this launch: steps [t_begin, t_begin+TL) of every sequence; row r -> (b, t) -> flat row m = b*T + t
    const int TL = a.t_len > 0 ? a.t_len : a.T;
    // row arithmetic is 32-bit (the dispatcher rejects B*T >= 2^31): 64-bit divisions compile to branchy
    // code, and control flow inside the tile loop is what defeats the s_waitcnt bookkeeping (see below)
    const unsigned M = (unsigned)a.B * (unsigned)TL;
    const unsigned ntile = (M + 31u) / 32u;
    auto flat_row = [&](unsigned r) -> unsigned { const unsigned b = r / (unsigned)TL; return b * a.T + a.t_begin + (r - b * TL); };
    const unsigned gwave = blockIdx.x * RW_WAVES + (threadIdx.x >> 6);
    const int ns = (int)(gwave % NS);              // which column group
    const unsigned wave_id = gwave / NS;
    const unsigned nwave = gridDim.x * RW_WAVES / NS;
    const int n_base = ns * NT * 32;

    // B operand: Wcat[p*KH + ks][n_base + 32*nt + c], Wcat = [wg[0:D] | wc[0:D]]
    float wb[NT][KH];
    float bias[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n = n_base + 32 * nt + c;
        // xp is produced in the scan's exponent domain (hpmn_hip.h: HpmnInputProj.xp): gate columns times
        // -log2(e), candidate columns times -2 log2(e), folded into the stationary operand for free
        const bool gate = n < 2 * H;
        const float sc = gate ? NEG_LOG2E : 2.0f * NEG_LOG2E;
        // selected base / stride instead of a conditional per element (no branches, no waits at joins)
        const float *wcol = gate ? a.wg + n : a.wc + (n - 2 * H);
        const long ldw = gate ? 2 * H : H;
        const float *bcol = gate ? a.bg + n : a.bc + (n - 2 * H);
        bias[nt] = p == 0 ? sc * bcol[0] : 0.f;                                   // k slot 0 of the bias step
#pragma unroll
        for (int ks = 0; ks < KH; ++ks) wb[nt][ks] = sc * wcol[(long)(p * KH + ks) * ldw];
    }

    // No lane is ever "out of range": lanes past the last row (and prefetches past the last tile) are CLAMPED
    // to row M-1, so they load, compute and store exactly what that row's owner does (identical bytes to
    // the same address).  That keeps every load and store of the tile loop unconditional -- with stores
    // inside branches the s_waitcnt pass can no longer count them and turns the waits for the prefetched
    // ids/rows into waits for the previous tile's stores (seen in the ISA as vmcnt(3) at the loop top).
    auto tile_row = [&](unsigned tile) -> unsigned { const unsigned rr = tile * 32u + c; return rr < M ? rr : M - 1u; };

    // GATHER: the row of a tile is two DEPENDENT HBM reads (id, then its embedding row).  One tile of
    // MFMAs (~1.4 us) hides neither, so ids run three tiles ahead and rows two tiles ahead of the tile
    // being multiplied (measured with tools/micro/scan_ablate.py: 183 -> ~120 us at C3 layer 0, where the
    // kernel was latency-bound at 1.2 TB/s without its stores).
    // (load_ids must not LOOK at the ids it loads -- a compare would wait for them on the spot: the address
    // is clamped instead and "this row is all zero" travels in `live`; the id-0 mask is applied in load_rows)
    auto load_ids = [&](unsigned tile, long (&id)[Q], bool &live) {
        const unsigned m = flat_row(tile_row(tile));
        const unsigned b = m / (unsigned)a.T;
        const int t = (int)(m - b * a.T) - a.front_zero;
        live = t >= 0;
        const long base = ((long)b * a.Tids + (live ? t : 0)) * a.F;
#pragma unroll
        for (int q = 0; q < Q; ++q) id[q] = load_id(a.ids, base + (p * KH + 4 * q) / a.E, a.mask_id0);
    };
    // rows are loaded UNCONDITIONALLY (every id read from the clamped address is a valid row) and zeroed by
    // `keep` bits when they are consumed: conditional loads put branches and full vmcnt waits into the loop
    auto load_rows = [&](const long (&id)[Q], bool live, float4 (&v)[Q], unsigned &keep) {
        keep = 0;
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const int j = p * KH + 4 * q;
            const int f = j / a.E;
            v[q] = *reinterpret_cast<const float4 *>(a.emb + id[q] * a.E + (j - f * a.E));
            keep |= (live && !id_masked(id[q], a.mask_id0)) ? (1u << q) : 0u;
        }
    };
    auto load_x = [&](unsigned tile, float4 (&v)[Q]) {
        const long m = flat_row(tile_row(tile));
#pragma unroll
        for (int q = 0; q < Q; ++q) v[q] = *reinterpret_cast<const float4 *>(a.x + m * K + p * KH + 4 * q);
    };

    float4 cur[Q], nxt[Q], nx2[Q];
    long idn[Q];                        // ids of the tile whose rows are fetched next
    bool liven = false;
    unsigned kcur = ~0u, knxt = ~0u, knx2 = ~0u;   // per-stage keep bits (GATHER only)
    if constexpr (GATHER) {
        long id0[Q];
        bool live0;
        load_ids(wave_id, id0, live0);
        load_ids(wave_id + nwave, idn, liven);
        load_rows(id0, live0, cur, kcur);
        load_rows(idn, liven, nxt, knxt);
        load_ids(wave_id + 2 * nwave, idn, liven);
    } else {
        load_x(wave_id, cur);
    }
    // Land the prologue's loads before entering the loop (one exposed latency per launch): the s_waitcnt
    // pass merges the loop-entry state with the back-edge state, and loads still pending at entry (nothing
    // issued after them yet) would make every in-loop wait for a prefetched value a wait for (almost) all
    // outstanding memory operations, i.e. for the previous tile's stores.
    auto land4 = [](float4 &v) { settle(v.x); settle(v.y); settle(v.z); settle(v.w); };
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        land4(cur[q]);
        if constexpr (GATHER) {
            land4(nxt[q]);
            asm volatile("" : "+v"(idn[q]));
        }
    }
    for (unsigned tile = wave_id; tile < ntile; tile += nwave) {
        if constexpr (GATHER) {
            load_rows(idn, liven, nx2, knx2);        // rows of tile + 2
            load_ids(tile + 3 * nwave, idn, liven);
        } else {
            load_x(tile + nwave, nxt);
        }
        if constexpr (GATHER) {
#pragma unroll
            for (int q = 0; q < Q; ++q)
                if (!((kcur >> q) & 1u)) cur[q] = make_float4(0.f, 0.f, 0.f, 0.f);
            if constexpr (XOUT) {
                // the NS waves that share this row tile each store 1/NS of the gathered row (selects, not
                // a branch on ns)
                static_assert(Q % NS == 0, "x_out pieces split evenly over the column groups");
                float *xo = a.x_out + (long)flat_row(tile_row(tile)) * K + p * KH + 4 * (Q / NS) * ns;
#pragma unroll
                for (int i = 0; i < Q / NS; ++i) {
                    float4 v = cur[i];
#pragma unroll
                    for (int g = 1; g < NS; ++g)
                        if (ns == g) v = cur[g * (Q / NS) + i];
                    *reinterpret_cast<float4 *>(xo + 4 * i) = v;
                }
            }
        }
        f32x16 acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            f32x16 z;
#pragma unroll
            for (int r = 0; r < 16; ++r) z[r] = 0.f;
            acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(bias[nt], 1.0f, z, 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const float av[4] = {cur[q].x, cur[q].y, cur[q].z, cur[q].w};
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(wb[nt][4 * q + e], av[e], acc[nt], 0, 0, 0);
        }
        // C/D layout: lane (c, p), reg r -> D row (r&3) + 8*(r>>2) + 4*p = column n of row m = tile*32 + c
        {
            float *dst = a.xp + (long)flat_row(tile_row(tile)) * N + n_base + 4 * p;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<float4 *>(dst + 32 * nt + 8 * g) =
                        make_float4(acc[nt][4 * g], acc[nt][4 * g + 1], acc[nt][4 * g + 2], acc[nt][4 * g + 3]);
        }
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            cur[q] = nxt[q];
            if constexpr (GATHER) nxt[q] = nx2[q];
        }
        kcur = knxt;
        knxt = knx2;
    }
}

// dx[m, d] = sum_j d_act[m, j] Wx[d][j];  K = 3H (reduction); a wave computes NT 32-column output tiles
// (masked to D) and the NS column groups of a row tile go to NS waves.  DB = prefetch the next tile's
// rows under the MFMAs (two K/2-float row images in registers; off at K = 384 where one is 192 registers).
template <int K, int NT, int NS, bool DB>
__global__ __launch_bounds__(64 * RW_WAVES, 1) void gru_dx_kernel(const HpmnGruWgrad a) {
    constexpr int KH = K / 2;
    constexpr int Q = KH / 4;
    constexpr int H = K / 3;
    const int lane = threadIdx.x & 63;
    const int c = lane & 31, p = lane >> 5;
    const int D = a.D;
    const int TL = a.t_len > 0 ? a.t_len : a.T;
    const unsigned M = (unsigned)a.B * (unsigned)TL;          // 32-bit row arithmetic: see input_proj_kernel
    const unsigned ntile = (M + 31u) / 32u;
    auto flat_row = [&](unsigned r) -> unsigned { const unsigned b = r / (unsigned)TL; return b * a.T + a.t_begin + (r - b * TL); };
    const unsigned gwave = blockIdx.x * RW_WAVES + (threadIdx.x >> 6);
    const int n_base = (int)(gwave % NS) * NT * 32;
    const unsigned wave_id = gwave / NS;
    const unsigned nwave = gridDim.x * RW_WAVES / NS;

    // stationary operand: Wx[d][j], j = p*KH + ks;  row d of [wg[0:D] | wc[0:D]] is contiguous per source
    // Loaded as 16-byte pieces through SELECTED base pointers, no per-element conditionals (a branchy version
    // of this prologue -- ~190 single-dword gathers behind ~190 branches -- was a fixed ~30 us per launch, most
    // of the time of the small upper layers).  Half-wave p covers j in [p*KH, (p+1)*KH), KH = 1.5 H:
    //   ks <  H/2 : j < 2H for both halves            -> wg row, offset p*KH + ks
    //   ks >= H/2 : p = 0 still in the wg row; p = 1 is column ks - H/2 of the wc row
    // Rows past D are clamped (their output columns are never stored).
    float wb[NT][KH];
    static_assert(H % 8 == 0, "16-byte pieces");
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int d = n_base + 32 * nt + c;
        const long dr = d < D ? d : D - 1;
        const float *lo = a.wg + dr * 2 * H + p * KH;
        const float *hi = p == 0 ? a.wg + dr * 2 * H + H / 2 : a.wc + dr * H;
#pragma unroll
        for (int q = 0; q < KH / 4; ++q) {
            const float4 v = 4 * q < H / 2 ? *reinterpret_cast<const float4 *>(lo + 4 * q)
                                           : *reinterpret_cast<const float4 *>(hi + (4 * q - H / 2));
            wb[nt][4 * q] = v.x; wb[nt][4 * q + 1] = v.y; wb[nt][4 * q + 2] = v.z; wb[nt][4 * q + 3] = v.w;
        }
    }
    // rows past the end are clamped to M-1 (duplicate, identical work): no conditional loads or stores in
    // the tile loop -- see input_proj_kernel
    auto tile_row = [&](unsigned tile) -> unsigned { const unsigned rr = tile * 32u + c; return rr < M ? rr : M - 1u; };
    auto load_a = [&](unsigned tile, float4 (&v)[Q]) {
        const long m = flat_row(tile_row(tile));
#pragma unroll
        for (int q = 0; q < Q; ++q) v[q] = *reinterpret_cast<const float4 *>(a.d_act + m * K + p * KH + 4 * q);
    };
    float4 cur[Q], nxt[Q];
    load_a(wave_id, cur);
#pragma unroll
    for (int q = 0; q < Q; ++q) { settle(cur[q].x); settle(cur[q].y); settle(cur[q].z); settle(cur[q].w); }   // see input_proj_kernel
    for (unsigned tile = wave_id; tile < ntile; tile += nwave) {
        if constexpr (DB) load_a(tile + nwave, nxt);
        f32x16 acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const float av[4] = {cur[q].x, cur[q].y, cur[q].z, cur[q].w};
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(wb[nt][4 * q + e], av[e], acc[nt], 0, 0, 0);
        }
        // transposed product (see the file header): lane (c, p) owns row tile*32 + c, columns 32nt + 8g + 4p + 0..3
        {
            float *dst = a.d_x + (long)flat_row(tile_row(tile)) * D + n_base + 4 * p;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    if (n_base + 32 * nt + 8 * g + 4 * p < D)      // D is a multiple of 4: a quad is all in or all out
                        *reinterpret_cast<float4 *>(dst + 32 * nt + 8 * g) =
                            make_float4(acc[nt][4 * g], acc[nt][4 * g + 1], acc[nt][4 * g + 2], acc[nt][4 * g + 3]);
        }
        if constexpr (DB) {
#pragma unroll
            for (int q = 0; q < Q; ++q) cur[q] = nxt[q];
        } else {
            load_a(tile + nwave, cur);
        }
    }
}

// The same product on the bf16 matrix pipe with SPLIT operands (x = hi + lo in bf16, hi*hi + hi*lo + lo*hi, fp32 accumulate:
// gru_wgrad_bf16.hip's arithmetic, ~5e-6 of the result's max) for H = 128, where the fp32 form is bound by the matrix pipe:
// K = 384 is 192 v_mfma_f32_32x32x2_f32 per 32-row tile and wave, 12 288 cycles -- the launch of a 250 k-row layer took 322 us,
// all of it on the serial chain between two reverse scans (C4: 0.95 ms of 9.5).  Here a tile is 24 k-steps of 16: two 16-byte
// loads of the lane's own row (lane (row, p) holds columns 16 ks + 8 p .. + 7 -- the instruction's 8 consecutive k, no
// transpose), the split (24 VALU), three v_mfma_f32_32x32x16_bf16 (96 cycles): 2304 cycles of matrix pipe per tile, and the
// launch is bound by reading d_act once.  The weights' fragments (the "A" operand: lane (d, p) = Wx[d][16 ks + 8 p .. + 7],
// contiguous in memory) are stationary, split once: 192 registers.  Rows stream through a ring of PD k-steps of raw loads,
// across tile borders, clamped past the end (no branch around a load or store in the loop).
typedef __bf16 dbf8 __attribute__((ext_vector_type(8)));
// NP planes (round 6): 2 = x ~ hi + lo, three products (rounds 4/5, ~5e-6 of the result's max); 3 = hi + mid + lo (2^-25 |x| left)
// with the six products of order <= 2: fp32-equivalent (gru_wgrad_bf16.hip).  HPMN_DX_PLANES / HPMN_PROJ_PLANES = 2 select the
// old arithmetic.
template <int NP>
__device__ __forceinline__ void dx_split8(const float4 a, const float4 b, dbf8 (&pl)[NP]) {
    const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        float rest = v[i];
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            pl[q][i] = (__bf16)rest;
            if (q + 1 < NP) rest -= (float)pl[q][i];
        }
    }
}
// acc += W x over the planes' products of order <= NP - 1, smallest terms first
template <int NP>
__device__ __forceinline__ f32x16 dx_mma(const dbf8 (&w)[NP], const dbf8 (&x)[NP], f32x16 acc) {
    if constexpr (NP == 3) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[0], x[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[1], x[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[2], x[0], acc, 0, 0, 0);
    }
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[0], x[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[1], x[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[0], x[0], acc, 0, 0, 0);
    return acc;
}

template <int K, int NS, int NP>
__global__ __launch_bounds__(64 * RW_WAVES, 1) void gru_dx_bf16_kernel(const HpmnGruWgrad a) {
    constexpr int KS = K / 16;          // k-steps per tile
    constexpr int PD = 12;              // k-steps of raw row data in flight (24 x 16-byte loads per lane: 24 KB per wave)
    constexpr int H = K / 3;
    static_assert(KS % PD == 0 && (2 * H) % 16 == 0, "ring slots are compile-time; a k-step never straddles wg | wc");
    const int lane = threadIdx.x & 63;
    const int c = lane & 31, p = lane >> 5;
    const int D = a.D;
    const int TL = a.t_len > 0 ? a.t_len : a.T;
    const unsigned M = (unsigned)a.B * (unsigned)TL;
    const unsigned ntile = (M + 31u) / 32u;
    auto flat_row = [&](unsigned r) -> unsigned { const unsigned b = r / (unsigned)TL; return b * a.T + a.t_begin + (r - b * TL); };
    const unsigned gwave = blockIdx.x * RW_WAVES + (threadIdx.x >> 6);
    const int n_base = (int)(gwave % NS) * 32;
    const unsigned wave_id = gwave / NS;
    const unsigned nwave = gridDim.x * RW_WAVES / NS;

    dbf8 w[KS][NP];
    {
        const int d = n_base + c;
        const long dr = d < D ? d : D - 1;          // (columns past D are clamped: computed, never stored)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int j = 16 * ks + 8 * p;
            const float *src = j < 2 * H ? a.wg + dr * 2 * H + j : a.wc + dr * H + (j - 2 * H);
            dx_split8<NP>(*reinterpret_cast<const float4 *>(src), *reinterpret_cast<const float4 *>(src + 4), w[ks]);
        }
    }
    auto tile_row = [&](unsigned tile) -> unsigned { const unsigned rr = tile * 32u + c; return rr < M ? rr : M - 1u; };
    auto row_ptr = [&](unsigned tile) { return a.d_act + (long)flat_row(tile_row(tile)) * K + 8 * p; };
    float4 ring[PD][2];
    const float *cur = row_ptr(wave_id);
#pragma unroll
    for (int i = 0; i < PD; ++i) {
        ring[i][0] = *reinterpret_cast<const float4 *>(cur + 16 * i);
        ring[i][1] = *reinterpret_cast<const float4 *>(cur + 16 * i + 4);
    }
    // (land the prologue's loads before the loop, as in input_proj_kernel: pending loads at loop entry make the compiler
    //  merge entry and back-edge states into waits for nearly everything in flight -- vmcnt(3) where 10 are allowed)
#pragma unroll
    for (int i = 0; i < PD; ++i)
#pragma unroll
        for (int h = 0; h < 2; ++h) { settle(ring[i][h].x); settle(ring[i][h].y); settle(ring[i][h].z); settle(ring[i][h].w); }
    for (unsigned tile = wave_id; tile < ntile; tile += nwave) {
        const float *nxt = row_ptr(tile + nwave);
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            dbf8 xp[NP];
            {   // the k-step's consumption point, pinned in the order of the volatile statements (so are the reloads below):
                // left alone the compiler hoists every split of the tile to its top and sinks the reloads to their uses --
                // the ISA then shows the ring draining, vmcnt(22) .. vmcnt(0), then each k-step waiting for a load just issued
                float4 &r0 = ring[ks % PD][0], &r1 = ring[ks % PD][1];
                asm volatile("" : "+v"(r0.x), "+v"(r0.y), "+v"(r0.z), "+v"(r0.w), "+v"(r1.x), "+v"(r1.y), "+v"(r1.z), "+v"(r1.w));
            }
            dx_split8<NP>(ring[ks % PD][0], ring[ks % PD][1], xp);
            asm volatile("" ::: "memory");
            // the slot is free: k-step ks + PD of this tile, or the next tile's first ones
            const float *src = ks + PD < KS ? cur + 16 * (ks + PD) : nxt + 16 * (ks + PD - KS);
            ring[ks % PD][0] = *reinterpret_cast<const float4 *>(src);
            ring[ks % PD][1] = *reinterpret_cast<const float4 *>(src + 4);
            asm volatile("" ::: "memory");
            acc = dx_mma<NP>(w[ks], xp, acc);
        }
        // transposed product (file header): lane (c, p) owns row tile*32 + c, columns 8 g + 4 p + 0..3 of this wave's 32
        {
            float *dst = a.d_x + (long)flat_row(tile_row(tile)) * D + n_base + 4 * p;
#pragma unroll
            for (int g = 0; g < 4; ++g)
                if (n_base + 8 * g + 4 * p < D)
                    *reinterpret_cast<float4 *>(dst + 8 * g) = make_float4(acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]);
        }
        cur = nxt;
    }
}

// D = 128 (NS = 4 column slices = the four waves of a workgroup, all on the SAME row tile): the row-per-lane loads above are
// issued four times per tile -- every load instruction touches 32 rows (32 different 128-byte lines, 32 bytes of each), the four
// waves' rings together are three times the CU's vector L1, and every wave splits the same values again: the launch ran at
// ~9 us per 32-row tile, 1.85 TB/s on 0.5 GB (layer 1 of C4: 284 us on the serial chain between two reverse scans) where its
// matrix instructions need 1 us.  Here the WORKGROUP stages a tile once: 256 threads load its 48 KB fully coalesced (four
// adjacent lanes = one row's 128-byte line), split each 8-float fragment once, and park (hi, lo) in LDS in the B operand's
// lane order; the four waves read their k-steps from there (two ds_read_b128 per k-step) against their stationary weights.
// Two LDS buffers, the next tile's loads in flight underneath the current tile's 72 matrix instructions, one barrier per tile.
// LDS slot of fragment (row, ks, p): plane ks, slot ((row + 2 (2 (ks & 1) + p)) & 31) + 32 p.  A ds_write_b128 is serviced in
// groups of 8 contiguous lanes over 32 banks (MI355X_MICROARCH.md, LDS): here 2 rows x the 4 fragments of a line, and the
// rotation by 2 rows per fragment-of-a-line puts them on 8 different bank quads (a rotation by 4 was a 2-way conflict on every
// store: SQ_LDS_BANK_CONFLICT 26 % of the launch's LDS cycles); a ds_read_b128's 16-lane groups see 16 different slots mod 16
// under any rotation of the row.
template <int K, int NP>
__global__ __launch_bounds__(64 * RW_WAVES, 1) void gru_dx_bf16_lds_kernel(const HpmnGruWgrad a) {
    constexpr int KS = K / 16;                      // k-steps per tile (24)
    constexpr int FPT = 32 * KS * 2 / (64 * RW_WAVES);   // fragments a thread stages per tile (6)
    static_assert(RW_WAVES == 4 && (32 * KS * 2) % (64 * RW_WAVES) == 0 && KS % 2 == 0, "four column slices, whole passes");
    extern __shared__ __attribute__((aligned(16))) char dx_smem[];
    dbf8 (*img)[KS][NP][64] = reinterpret_cast<dbf8 (*)[KS][NP][64]>(dx_smem);    // [buffer][k-step][plane][slot]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int c = lane & 31, p = lane >> 5;
    const int D = a.D;
    const int TL = a.t_len > 0 ? a.t_len : a.T;
    const unsigned M = (unsigned)a.B * (unsigned)TL;
    const unsigned ntile = (M + 31u) / 32u;
    auto flat_row = [&](unsigned r) -> unsigned { const unsigned b = r / (unsigned)TL; return b * a.T + a.t_begin + (r - b * TL); };
    const int n_base = wv * 32;

    dbf8 w[KS][NP];
    {
        const int d = n_base + c;
        const long dr = d < D ? d : D - 1;          // (columns past D are clamped: computed, never stored)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int j = 16 * ks + 8 * p;
            const float *src = j < 2 * (K / 3) ? a.wg + dr * 2 * (K / 3) + j : a.wc + dr * (K / 3) + (j - 2 * (K / 3));
            dx_split8<NP>(*reinterpret_cast<const float4 *>(src), *reinterpret_cast<const float4 *>(src + 4), w[ks]);
        }
    }
    // staging role of this thread: row (tid >> 2) & 31 of the tile, fragments q = 8 j + 4 (tid >> 7) + (tid & 3), j < FPT
    const int s_row = (tid >> 2) & 31, s_sub = tid & 3, s_grp = tid >> 7;
    const int s_slot = ((s_row + 2 * s_sub) & 31) + 32 * (s_sub & 1);
    float4 raw[FPT][2];
    auto load_tile = [&](unsigned tile) {
        unsigned rr = tile * 32u + (unsigned)s_row;
        rr = rr < M ? rr : M - 1u;                  // (clamped: loaded, multiplied, never stored)
        const float *src = a.d_act + (long)flat_row(rr) * K + 8 * (4 * s_grp + s_sub);
#pragma unroll
        for (int j = 0; j < FPT; ++j) {
            raw[j][0] = *reinterpret_cast<const float4 *>(src + 64 * j);
            raw[j][1] = *reinterpret_cast<const float4 *>(src + 64 * j + 4);
        }
    };
    auto park_tile = [&](int buf) {
#pragma unroll
        for (int j = 0; j < FPT; ++j) {
            const int ks = 4 * j + 2 * s_grp + (s_sub >> 1);       // q / 2, q = 8 j + 4 s_grp + s_sub
            dbf8 pl[NP];
            dx_split8<NP>(raw[j][0], raw[j][1], pl);
#pragma unroll
            for (int q = 0; q < NP; ++q) img[buf][ks][q][s_slot] = pl[q];
        }
    };
    auto barrier = [] { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

    unsigned tile = blockIdx.x;
    if (tile >= ntile) return;
    load_tile(tile);
    park_tile(0);
    barrier();
    int buf = 0;
    for (; tile < ntile; tile += gridDim.x, buf ^= 1) {
        const unsigned nxt = tile + gridDim.x < ntile ? tile + gridDim.x : tile;    // (the last tile once more: no branch around loads)
        load_tile(nxt);
        asm volatile("" ::: "memory");              // (the loads stay HERE: left alone they sink below the matrix instructions, to their uses)
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int slot = ((c + 2 * (2 * (ks & 1) + p)) & 31) + 32 * p;
            dbf8 xp[NP];
#pragma unroll
            for (int q = 0; q < NP; ++q) xp[q] = img[buf][ks][q][slot];
            acc = dx_mma<NP>(w[ks], xp, acc);
        }
        {   // transposed product (file header): lane (c, p) owns row tile*32 + c, columns 8 g + 4 p + 0..3 of this wave's 32
            unsigned rr = tile * 32u + (unsigned)c;
            rr = rr < M ? rr : M - 1u;
            float *dst = a.d_x + (long)flat_row(rr) * D + n_base + 4 * p;
#pragma unroll
            for (int g = 0; g < 4; ++g)
                if (n_base + 8 * g + 4 * p < D)
                    *reinterpret_cast<float4 *>(dst + 8 * g) = make_float4(acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]);
        }
        park_tile(buf ^ 1);
        barrier();
    }
}

// The forward's input projection at H = 128 with a 128-wide input (the upper layers of configs[4]) the same way: the fp32 form
// is 195 matrix instructions of 64 cycles per 32-row tile and wave (12.5 k cycles; 268 us for the 250 k rows of layer 1, on the
// serial chain in front of that layer's scan), here 8 k-steps x 3 column tiles x 3 products of 32 cycles = 2304, and the launch
// is bound by writing xp.  Results within ~5e-6 of the row's largest pre-activation of the fp32 kernel's (the H = 128 forward
// parity tests hold 1e-4 on the logits).  Rows from memory only (layer 0 gathers: its launch is bound by its 0.83 GB of
// stores, not by the matrix pipe, and keeps the fp32 kernel).  The bias enters as one more k-step (hi and lo against 1.0).
template <int K, int NT, int NS, int NP>
__global__ __launch_bounds__(64 * RW_WAVES, 1) void input_proj_bf16_kernel(const HpmnInputProj a) {
    constexpr int KS = K / 16;
    const int lane = threadIdx.x & 63;
    const int c = lane & 31, p = lane >> 5;
    const int H = a.H, N = 3 * H;
    const int TL = a.t_len > 0 ? a.t_len : a.T;
    const unsigned M = (unsigned)a.B * (unsigned)TL;
    const unsigned ntile = (M + 31u) / 32u;
    auto flat_row = [&](unsigned r) -> unsigned { const unsigned b = r / (unsigned)TL; return b * a.T + a.t_begin + (r - b * TL); };
    const unsigned gwave = blockIdx.x * RW_WAVES + (threadIdx.x >> 6);
    const int ns = (int)(gwave % NS);
    const unsigned wave_id = gwave / NS;
    const unsigned nwave = gridDim.x * RW_WAVES / NS;
    const int n_base = ns * NT * 32;

    // A operand (stationary): lane (n = c, p) = sc * Wcat[16 ks + 8 p .. + 7][n_base + 32 nt + c], Wcat = [wg[0:D] | wc[0:D]]
    // Three planes: the weights' hi and mid planes are stationary in registers (192), the LO plane (only ever multiplied with
    // the rows' hi plane) lives in LDS in lane order -- one linear 16-byte read per lane and product: with all three planes in
    // registers (288 + two row images + 48 accumulators) the kernel ran out of the 512 and spilled (r6: twice the time).
    constexpr int NR = NP == 3 ? 2 : NP;               // planes kept in registers
    __shared__ __attribute__((aligned(16))) dbf8 wlo[NP == 3 ? RW_WAVES : 1][NP == 3 ? NT : 1][NP == 3 ? KS : 1][64];
    const int wv = threadIdx.x >> 6;
    dbf8 w[NT][KS][NR];
    float bias[NT];                     // (the bias step's planes are rebuilt per tile from this: 36 registers fewer)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n = n_base + 32 * nt + c;
        const bool gate = n < 2 * H;
        const float sc = gate ? NEG_LOG2E : 2.0f * NEG_LOG2E;          // (xp is produced in the scan's exponent domain)
        const float *wcol = gate ? a.wg + n : a.wc + (n - 2 * H);
        const long ldw = gate ? 2 * H : H;
        const float *bcol = gate ? a.bg + n : a.bc + (n - 2 * H);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            float4 v0, v1;
            const float *w0 = wcol + (long)(16 * ks + 8 * p) * ldw;
            v0.x = sc * w0[0]; v0.y = sc * w0[ldw]; v0.z = sc * w0[2 * ldw]; v0.w = sc * w0[3 * ldw];
            v1.x = sc * w0[4 * ldw]; v1.y = sc * w0[5 * ldw]; v1.z = sc * w0[6 * ldw]; v1.w = sc * w0[7 * ldw];
            dbf8 pl[NP];
            dx_split8<NP>(v0, v1, pl);
#pragma unroll
            for (int q = 0; q < NR; ++q) w[nt][ks][q] = pl[q];
            if constexpr (NP == 3) wlo[wv][nt][ks][lane] = pl[2];
        }
        // the bias step: k slot 0 of half-wave 0 carries the bias, everything else 0, against a row operand of 1.0 there
        bias[nt] = p == 0 ? sc * bcol[0] : 0.f;
    }
    dbf8 one;
#pragma unroll
    for (int i = 0; i < 8; ++i) one[i] = (__bf16)(i == 0 ? 1.0f : 0.0f);

    auto tile_row = [&](unsigned tile) -> unsigned { const unsigned rr = tile * 32u + c; return rr < M ? rr : M - 1u; };
    auto load_x = [&](unsigned tile, float4 (&v)[KS][2]) {
        const float *src = a.x + (long)flat_row(tile_row(tile)) * K + 8 * p;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            v[ks][0] = *reinterpret_cast<const float4 *>(src + 16 * ks);
            v[ks][1] = *reinterpret_cast<const float4 *>(src + 16 * ks + 4);
        }
    };
    // one tile: bias step, 8 k-steps x NT column tiles x 3 products, 16-byte stores (transposed product, file header)
    auto do_tile = [&](unsigned tile, const float4 (&v)[KS][2]) {
        f32x16 acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            f32x16 z;
#pragma unroll
            for (int r = 0; r < 16; ++r) z[r] = 0.f;
            acc[nt] = z;
            dbf8 bpl[NP];
            dx_split8<NP>(make_float4(bias[nt], 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f), bpl);
#pragma unroll
            for (int q = NP - 1; q >= 0; --q) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bpl[q], one, acc[nt], 0, 0, 0);
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            dbf8 xp[NP];
            dx_split8<NP>(v[ks][0], v[ks][1], xp);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                if constexpr (NP == 3) {
                    const dbf8 w2 = wlo[wv][nt][ks][lane];              // (written by this very lane: no barrier)
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[nt][ks][0], xp[2], acc[nt], 0, 0, 0);
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[nt][ks][1], xp[1], acc[nt], 0, 0, 0);
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2, xp[0], acc[nt], 0, 0, 0);
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[nt][ks][0], xp[1], acc[nt], 0, 0, 0);
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[nt][ks][1], xp[0], acc[nt], 0, 0, 0);
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[nt][ks][0], xp[0], acc[nt], 0, 0, 0);
                } else {
                    acc[nt] = dx_mma<NP>(w[nt][ks], xp, acc[nt]);
                }
            }
        }
        float *dst = a.xp + (long)flat_row(tile_row(tile)) * N + n_base + 4 * p;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4 *>(dst + 32 * nt + 8 * g) =
                    make_float4(acc[nt][4 * g], acc[nt][4 * g + 1], acc[nt][4 * g + 2], acc[nt][4 * g + 3]);
    };
    // The next tile's rows are issued in front of this tile's work and copied over behind it (the copy is where they are
    // waited for: one tile of work, ~1 us, after their issue).  (Two buffers and two tiles per trip, no copy, with the
    // consumption points pinned by volatile statements: 512 registers and 54 spills -- not kept.)
    float4 cur[KS][2], nxt[KS][2];
    load_x(wave_id, cur);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int h = 0; h < 2; ++h) { settle(cur[ks][h].x); settle(cur[ks][h].y); settle(cur[ks][h].z); settle(cur[ks][h].w); }
    for (unsigned tile = wave_id; tile < ntile; tile += nwave) {
        load_x(tile + nwave, nxt);
        asm volatile("" ::: "memory");
        do_tile(tile, cur);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) { cur[ks][0] = nxt[ks][0]; cur[ks][1] = nxt[ks][1]; }
    }
}

// persistent grid for a row-wise kernel whose row tile is shared by NS waves: one workgroup per CU at most,
// (grid * RW_WAVES) a multiple of NS
static unsigned rowwise_grid(long M, int NS) {
    const long ntile = (M + 31) / 32;
    long wg = (ntile * NS + RW_WAVES - 1) / RW_WAVES;
    if (wg > 256) wg = 256;            // one persistent workgroup per CU (1 wave per SIMD: 3H..6H weight registers)
    if (wg < 1) wg = 1;
    int step = NS;                     // smallest s with (s * RW_WAVES) % NS == 0
    for (int s2 = 1; s2 <= NS; ++s2)
        if ((s2 * RW_WAVES) % NS == 0) { step = s2; break; }
    wg = (wg + step - 1) / step * step;
    if (wg > 256) wg -= step;
    return (unsigned)wg;
}

template <int K, int NT, int NS>
static int launch_proj(const HpmnInputProj &a, hipStream_t st) {
    const unsigned grid = rowwise_grid((long)a.B * (a.t_len > 0 ? a.t_len : a.T), NS);
    if ((long)a.B * a.T >= (1L << 31) - 64) return HPMN_EUNSUPPORTED;   // 32-bit row arithmetic in the kernels
    constexpr bool can_gather = (K / 8) % NS == 0;       // the x_out pieces split evenly over the column groups
    if (a.x == nullptr) {
        if constexpr (can_gather) {
            if (a.x_out != nullptr)
                hipLaunchKernelGGL((input_proj_kernel<K, NT, NS, true, true>), dim3(grid), dim3(64 * RW_WAVES), 0, st, a);
            else
                hipLaunchKernelGGL((input_proj_kernel<K, NT, NS, true, false>), dim3(grid), dim3(64 * RW_WAVES), 0, st, a);
        } else {
            return HPMN_EUNSUPPORTED;
        }
    } else {
        hipLaunchKernelGGL((input_proj_kernel<K, NT, NS, false, false>), dim3(grid), dim3(64 * RW_WAVES), 0, st, a);
    }
    return check_launch();
}

// H = 128: 12 column tiles, 3 per wave over the 4 waves of a workgroup (D = 128: 192 stationary weights + two
// 64-float row images + 48 accumulators per lane -- fits the 512-register budget of one wave per SIMD)
bool input_proj_supported(int H, int D) {
    if (H == 128) return D == 32 || D == 128;
    return (H == 32 || H == 64) && (D == 16 || D == 32 || D == 48 || D == 64);
}

int input_proj_dispatch(const HpmnInputProj &a, hipStream_t st) {
    if (a.H == 128) {
        // HPMN_PROJ_BF16=0: the fp32 kernel for the 128-wide layers too
        static const int pbf = [] { const char *e = getenv("HPMN_PROJ_BF16"); return e ? atoi(e) : 1; }();
        if (a.D == 128 && a.x != nullptr && pbf && (long)a.B * a.T < (1L << 31) - 64) {
            const unsigned grid = rowwise_grid((long)a.B * (a.t_len > 0 ? a.t_len : a.T), 4);
            static const int np = [] { const char *e = getenv("HPMN_PROJ_PLANES"); return (e && atoi(e) == 2) ? 2 : 3; }();
            if (np == 2) hipLaunchKernelGGL((input_proj_bf16_kernel<128, 3, 4, 2>), dim3(grid), dim3(64 * RW_WAVES), 0, st, a);
            else hipLaunchKernelGGL((input_proj_bf16_kernel<128, 3, 4, 3>), dim3(grid), dim3(64 * RW_WAVES), 0, st, a);
            return check_launch();
        }
        if (a.D == 32) return launch_proj<32, 3, 4>(a, st);
        if (a.D == 128) return launch_proj<128, 3, 4>(a, st);
        return HPMN_EUNSUPPORTED;
    }
#define X(d) \
    if (a.D == d) return a.H == 32 ? launch_proj<d, 3, 1>(a, st) : launch_proj<d, 3, 2>(a, st);
    X(16) X(32) X(48) X(64)
#undef X
    return HPMN_EUNSUPPORTED;
}

// HPMN_DX_BF16=0: the fp32 kernel for H = 128 as well
static bool dx_bf16() {
    static const int on = [] { const char *e = getenv("HPMN_DX_BF16"); return e ? atoi(e) : 1; }();
    return on != 0;
}

static int dx_planes() {
    static const int np = [] { const char *e = getenv("HPMN_DX_PLANES"); return (e && atoi(e) == 2) ? 2 : 3; }();
    return np;
}

// HPMN_DX_LDS=0: D = 128 through the row-per-lane kernel as well; HPMN_DX_LDS_GRID: workgroups of the staged form (default: CUs)
static bool dx_lds() {
    static const int on = [] { const char *e = getenv("HPMN_DX_LDS"); return e ? atoi(e) : 1; }();
    return on != 0;
}
static long dx_lds_grid() {
    static const long n = [] {
        const char *e = getenv("HPMN_DX_LDS_GRID");
        if (e && atol(e) > 0) return atol(e);
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess) return 256L;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) return 256L;
        return (long)cus;
    }();
    return n;
}

int gru_dx_dispatch(const HpmnGruWgrad &a, hipStream_t st) {
    const long rows = (long)a.B * (a.t_len > 0 ? a.t_len : a.T);
    const int DT = (a.D + 31) / 32;
    if ((long)a.B * a.T >= (1L << 31) - 64) return HPMN_EUNSUPPORTED;   // 32-bit row arithmetic in the kernel
    const dim3 blk(64 * RW_WAVES);
    if (a.H == 32 && DT == 1) hipLaunchKernelGGL((gru_dx_kernel<96, 1, 1, true>), dim3(rowwise_grid(rows, 1)), blk, 0, st, a);
    else if (a.H == 32 && DT == 2) hipLaunchKernelGGL((gru_dx_kernel<96, 2, 1, true>), dim3(rowwise_grid(rows, 1)), blk, 0, st, a);
    else if (a.H == 64 && DT == 1) hipLaunchKernelGGL((gru_dx_kernel<192, 1, 1, true>), dim3(rowwise_grid(rows, 1)), blk, 0, st, a);
    else if (a.H == 64 && DT == 2) hipLaunchKernelGGL((gru_dx_kernel<192, 2, 1, true>), dim3(rowwise_grid(rows, 1)), blk, 0, st, a);
    else if (a.H == 128 && DT == 1 && dx_bf16()) {
        if (dx_planes() == 2) hipLaunchKernelGGL((gru_dx_bf16_kernel<384, 1, 2>), dim3(rowwise_grid(rows, 1)), blk, 0, st, a);
        else hipLaunchKernelGGL((gru_dx_bf16_kernel<384, 1, 3>), dim3(rowwise_grid(rows, 1)), blk, 0, st, a);
    }
    else if (a.H == 128 && DT == 4 && dx_bf16() && dx_lds()) {
        // (the workgroup-staged form: one tile per workgroup at a time, persistent over the CUs)
        const int np = dx_planes();
        const int lds = 2 * 24 * np * 64 * 16;
        static const bool attr = [] {
            return hipFuncSetAttribute(reinterpret_cast<const void *>(gru_dx_bf16_lds_kernel<384, 2>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 24 * 2 * 64 * 16) == hipSuccess &&
                   hipFuncSetAttribute(reinterpret_cast<const void *>(gru_dx_bf16_lds_kernel<384, 3>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 24 * 3 * 64 * 16) == hipSuccess;
        }();
        (void)attr;
        const long ntile = (rows + 31) / 32;
        const unsigned grid = (unsigned)(ntile < dx_lds_grid() ? ntile : dx_lds_grid());
        if (np == 2) hipLaunchKernelGGL((gru_dx_bf16_lds_kernel<384, 2>), dim3(grid), blk, lds, st, a);
        else hipLaunchKernelGGL((gru_dx_bf16_lds_kernel<384, 3>), dim3(grid), blk, lds, st, a);
    }
    else if (a.H == 128 && DT == 4 && dx_bf16()) {
        if (dx_planes() == 2) hipLaunchKernelGGL((gru_dx_bf16_kernel<384, 4, 2>), dim3(rowwise_grid(rows, 4)), blk, 0, st, a);
        else hipLaunchKernelGGL((gru_dx_bf16_kernel<384, 4, 3>), dim3(rowwise_grid(rows, 4)), blk, 0, st, a);
    }
    else if (a.H == 128 && DT == 1) hipLaunchKernelGGL((gru_dx_kernel<384, 1, 1, false>), dim3(rowwise_grid(rows, 1)), blk, 0, st, a);
    else if (a.H == 128 && DT == 4) hipLaunchKernelGGL((gru_dx_kernel<384, 1, 4, false>), dim3(rowwise_grid(rows, 4)), blk, 0, st, a);
    else return HPMN_EUNSUPPORTED;
    return check_launch();
}

}  // namespace hpmn
