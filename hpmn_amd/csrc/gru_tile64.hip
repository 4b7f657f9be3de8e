// build_memory forward for EVALUATION at H = 64, second generation (r5): gru_tile128.hip's decomposition at the width where
// everything fits in registers.  One layer per launch, 16-sequence tiles, FOUR waves per workgroup: wave w owns units
// [16 w, 16 w + 16) of ALL THREE gates (3 tiles: 48 registers of recurrent weights, up to 48 of input weights), so r, u and the
// state never leave its registers; a step is two operand images through LDS (h before the gates, r*h before the candidate) and
// two barriers; the input product of step t + 1 rides in phase 2 of step t.  Arithmetic as gru_pipe_fwd.hip / pipe_common.h:
// every operand x = hi + lo in f16, three products per tile, fp32 accumulate.
//
// Against gru_pipe_fwd_kernel (twelve role-specialised waves, one tile per CU, ~1830 cycles per tile-step): ~150 registers
// and 20 KB of LDS per workgroup, so TWO OR THREE tiles share a CU and interleave their issue slots, and nothing is handed
// between roles.  Reference: tf.nn.dynamic_rnn(GRUCell(64)) + the every-p-th-output gather, code/hpmn.py:118-128.
#include "pipe_common.h"

namespace hpmn {

struct Tile128Args {                       // (shared with gru_tile128.hip / pipe_api.hip)
    int B, T, D, period;
    const float *x, *xp;
    const float *wg, *bg, *wc, *bc;
    float *y;
    float *h_last;
    long h_last_stride;
};

#define MF64(A, Bv, C) __builtin_amdgcn_mfma_f32_16x16x32_f16(A, Bv, C, 0, 0, 0)

// NKS: k-steps of the input product (1: D <= 32, 2: D <= 64)
// (two workgroups per CU: 180 / 212 registers.  Cut for three or four -- 168 / 128 -- the loop spills and a pass of 12 288
//  rows takes 10.3 ms instead of 6.1: r5, measured)
template <int NKS>
__global__ __launch_bounds__(256, 2) void gru_tile64_fwd_kernel(const Tile128Args a) {
    __shared__ __attribute__((aligned(16))) char smem[8 * IMG];     // h hi/lo, r*h hi/lo, x ring 2 x hi/lo
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, g = lane >> 4, n = lane & 15;
    const int tile = blockIdx.x;
    const int B = a.B, T = a.T, D = a.D;
    const bool live = tile * TS + n < B;
    const long b = live ? (long)tile * TS + n : (long)B - 1;
    char *Hhi = smem, *Hlo = smem + IMG, *Rhi = smem + 2 * IMG, *Rlo = smem + 3 * IMG, *X = smem + 4 * IMG;
    const int wr = img_wr_off(w, g, n);                             // the lane's four units of a row (8 bytes)
    const int rd0 = img_rd_off(0, g, n), rd1 = img_rd_off(1, g, n);
    const int u0 = 16 * w + 4 * g;                                  // units u0 + 0..3 (outputs) / features (inputs)
    const bool has_x = 16 * w < D;                                  // this unit block's feature block exists (D % 16 == 0)

    h8 Ah_hi[3][2], Ah_lo[3][2], Ai_hi[3][NKS], Ai_lo[3][NKS];
    f4 bias[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const float sc = q < 2 ? NEG_LOG2E : 2.0f * NEG_LOG2E;
        const float *W = q < 2 ? a.wg + q * PH : a.wc;
        const int ld = q < 2 ? 2 * PH : PH;
        const float *bp = q < 2 ? a.bg + q * PH : a.bc;
        const int col = 16 * w + n;                                 // A's row m = lane % 16 -> output unit of the tile
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            float vh[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) vh[e] = W[(long)(D + slot_unit(s, g, e)) * ld + col] * sc;
            split8(vh, Ah_hi[q][s], Ah_lo[q][s]);
        }
#pragma unroll
        for (int s = 0; s < NKS; ++s) {
            float vi[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int unit = slot_unit(s, g, e);
                const float wi = W[(long)(unit < D ? unit : 0) * ld + col] * sc;
                vi[e] = unit < D ? wi : 0.f;
            }
            split8(vi, Ai_hi[q][s], Ai_lo[q][s]);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) bias[q][k] = bp[u0 + k] * sc;
    }

    const int period = a.period;
    const bool has_y = a.y != nullptr;
    float *yp = has_y ? a.y + b * (long)(T / period) * PH + u0 : nullptr;
    int next_fire = period - 1;

    for (int i = tid; i < 8 * IMG / 16; i += 256) reinterpret_cast<uint4 *>(smem)[i] = uint4{0u, 0u, 0u, 0u};
    __syncthreads();

    // ---- input rows, four steps in flight: lane (w, g, n) brings features u0 .. u0 + 3 of sequence n (where they exist)
    f4 R[4];
    const float *xrow = a.x + b * (long)T * D + (has_x ? u0 : 0);
    auto load_row = [&](int rho, f4 &dst) {
        const int rc = rho < T ? rho : T - 1;
        if (has_x) dst = *reinterpret_cast<const f4 *>(xrow + (long)rc * D);
    };
    auto park = [&](const f4 src, int slot) {
        if (has_x) {
            uint2 hi, lo;
            split4(src, hi, lo);
            *reinterpret_cast<uint2 *>(X + 2 * slot * IMG + wr) = hi;
            *reinterpret_cast<uint2 *>(X + (2 * slot + 1) * IMG + wr) = lo;
        }
    };
    f4 xp[3];
    auto project = [&](int slot) {                                  // bias + x W[:D] of the row parked in `slot`
#pragma unroll
        for (int q = 0; q < 3; ++q) xp[q] = bias[q];
#pragma unroll
        for (int s = 0; s < NKS; ++s) {
            const h8 xh = *reinterpret_cast<const h8 *>(X + 2 * slot * IMG + (s == 0 ? rd0 : rd1));
            const h8 xl = *reinterpret_cast<const h8 *>(X + (2 * slot + 1) * IMG + (s == 0 ? rd0 : rd1));
#pragma unroll
            for (int q = 0; q < 3; ++q) xp[q] = MF64(Ai_hi[q][s], xh, xp[q]);
#pragma unroll
            for (int q = 0; q < 3; ++q) xp[q] = MF64(Ai_hi[q][s], xl, xp[q]);
#pragma unroll
            for (int q = 0; q < 3; ++q) xp[q] = MF64(Ai_lo[q][s], xh, xp[q]);
        }
    };
#pragma unroll
    for (int k = 0; k < 4; ++k) load_row(k, R[k]);
    park(R[0], 0);
    park(R[1], 1);
    lds_barrier();
    project(0);
    lds_barrier();

    f4 h = {0.f, 0.f, 0.f, 0.f};
    auto step = [&](const int t, f4 &Rt, const f4 &Rt2) {
        park(Rt2, t & 1);                                           // row t + 2 -> the slot row t has left
        load_row(t + 4, Rt);
        // ---------------- phase 1: reset and update gates of the wave's units
        f4 zr = xp[0], zu = xp[1], zc = xp[2];
        {
            const h8 b0h = *reinterpret_cast<const h8 *>(Hhi + rd0), b1h = *reinterpret_cast<const h8 *>(Hhi + rd1);
            const h8 b0l = *reinterpret_cast<const h8 *>(Hlo + rd0), b1l = *reinterpret_cast<const h8 *>(Hlo + rd1);
            zr = MF64(Ah_hi[0][0], b0h, zr); zu = MF64(Ah_hi[1][0], b0h, zu);
            zr = MF64(Ah_hi[0][1], b1h, zr); zu = MF64(Ah_hi[1][1], b1h, zu);
            zr = MF64(Ah_hi[0][0], b0l, zr); zu = MF64(Ah_hi[1][0], b0l, zu);
            zr = MF64(Ah_hi[0][1], b1l, zr); zu = MF64(Ah_hi[1][1], b1l, zu);
            zr = MF64(Ah_lo[0][0], b0h, zr); zu = MF64(Ah_lo[1][0], b0h, zu);
            zr = MF64(Ah_lo[0][1], b1h, zr); zu = MF64(Ah_lo[1][1], b1h, zu);
        }
        f4 u;
        {
            f4 rh;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                rh[k] = sigmoid_scaled(zr[k]) * h[k];
                u[k] = sigmoid_scaled(zu[k]);
            }
            uint2 hi, lo;
            split4(rh, hi, lo);
            *reinterpret_cast<uint2 *>(Rhi + wr) = hi;
            *reinterpret_cast<uint2 *>(Rlo + wr) = lo;
        }
        lds_barrier();                                              // A: r*h of every unit is in LDS
        // ---------------- phase 2: candidate and state update; the input product of step t + 1 rides along
        {
            const h8 b0h = *reinterpret_cast<const h8 *>(Rhi + rd0), b1h = *reinterpret_cast<const h8 *>(Rhi + rd1);
            const h8 b0l = *reinterpret_cast<const h8 *>(Rlo + rd0), b1l = *reinterpret_cast<const h8 *>(Rlo + rd1);
            f4 zq = {0.f, 0.f, 0.f, 0.f};
            zc = MF64(Ah_hi[2][0], b0h, zc); zq = MF64(Ah_hi[2][1], b1h, zq);
            zc = MF64(Ah_hi[2][0], b0l, zc); zq = MF64(Ah_hi[2][1], b1l, zq);
            zc = MF64(Ah_lo[2][0], b0h, zc); zq = MF64(Ah_lo[2][1], b1h, zq);
            zc = zc + zq;
        }
        project((t + 1) & 1);                                       // (xp's old values live on in zr, zu, zc)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float c = tanh_scaled(zc[k]);
            h[k] = fmaf(u[k], h[k] - c, c);                         // u h + (1 - u) c
        }
        {
            uint2 hi, lo;
            split4(h, hi, lo);
            *reinterpret_cast<uint2 *>(Hhi + wr) = hi;
            *reinterpret_cast<uint2 *>(Hlo + wr) = lo;
        }
        const bool fire = t == next_fire;
        if (fire && has_y && live) *reinterpret_cast<f4 *>(yp) = h;
        next_fire += fire ? period : 0;
        if (has_y) yp += fire ? PH : 0;
        lds_barrier();                                              // B: h' (and the parked row t + 2) are in LDS
    };

    int t = 0;
    for (; t + 3 < T; t += 4) {
        step(t, R[0], R[2]);
        step(t + 1, R[1], R[3]);
        step(t + 2, R[2], R[0]);
        step(t + 3, R[3], R[1]);
    }
    if (t < T) step(t, R[0], R[2]);
    if (t + 1 < T) step(t + 1, R[1], R[3]);
    if (t + 2 < T) step(t + 2, R[2], R[0]);

    if (live) *reinterpret_cast<f4 *>(a.h_last + b * a.h_last_stride + u0) = h;
}
#undef MF64

bool tile64_supported(int H, int D) { return H == PH && D >= 16 && D <= 64 && D % 16 == 0; }

int tile64_fwd_launch(const Tile128Args &a, hipStream_t st) {
    const int ntiles = (a.B + TS - 1) / TS;
    if (a.D <= 32) hipLaunchKernelGGL((gru_tile64_fwd_kernel<1>), dim3(ntiles), dim3(256), 0, st, a);
    else           hipLaunchKernelGGL((gru_tile64_fwd_kernel<2>), dim3(ntiles), dim3(256), 0, st, a);
    return check_launch();
}

}  // namespace hpmn
