// Weight and bias gradients of ALL K layers of an H = 32 graph in ONE launch + one reduction (gru32_all.hip's companion).
//
//   dWg_i += [x | h_prev]^T d_act[:, 0:64]     dbg_i += sum_rows d_act[:, 0:64]
//   dWc_i += [x | r*h_prev]^T d_act[:, 64:96]  dbc_i += sum_rows d_act[:, 64:96]
//
// At the reference shapes (B = 128, T <= 100) one layer's reduction is 100 MFLOP -- a few microseconds of matrix-core work
// -- and the per-layer kernels of gru_wgrad.hip (staged tiles, two launches per layer, 20-40 us each, one after the other on
// the helper stream) had become the longest part of the step once the scans were a single launch.  Here blockIdx.y is the
// layer, blockIdx.x a group of sequences; the reduction index (b, t) is the MFMA k (v_mfma_f32_32x32x2_f32, as in
// gru_wgrad.hip: both operands are then plain 128-byte row segments), operands straight from L2 -- the rows were written a
// few microseconds ago by the scan launches.  The bias rides along as one more "feature" whose value is 1.
// Workgroup partial sums go to slabs, gru32_wgrad_reduce_kernel adds them in a fixed order (deterministic).
#include "gru32_all.h"

namespace hpmn {

typedef float f32x16w __attribute__((ext_vector_type(16)));

struct Wgrad32Args {
    int32_t B, K, spw, nwg;                 // sequences per workgroup, workgroups per layer
    int32_t D[AMAXK], T[AMAXK];
    const float *x[AMAXK], *hs[AMAXK], *gates[AMAXK], *d_act[AMAXK];
    float *d_wg[AMAXK], *d_bg[AMAXK], *d_wc[AMAXK], *d_bc[AMAXK];
    float *slab[AMAXK];                     // [nwg][slab_floats(D)] per layer
};

// slab of one workgroup: rows (D + 32 features + 1 bias row) x 96 columns, row-major
__host__ __device__ inline long wgrad32_slab_floats(int D) { return (long)(D + 33) * 96; }

// r5: EIGHT waves -- two halves of the time axis x four tile waves.  A workgroup is one sequence (B < 256) whose 100 steps
// were 13 groups of 8 steps, one memory round trip per group, one behind the other: 40 us for 100 MFLOP on the tail of the
// Amazon step.  More workgroups per sequence would mean more slabs (31 KB each, written and read back); here the second half's
// waves hand their accumulators to the first half's through LDS and the workgroup still writes ONE slab.
constexpr int W32_PIECES = 2;
__global__ __launch_bounds__(256 * W32_PIECES) void gru32_wgrad_all_kernel(const Wgrad32Args a) {
    constexpr int H = 32;
    const int L = blockIdx.y;
    const int D = a.D[L], T = a.T[L];
    const int NR = D + H + 1;                           // operand rows ("features"): x | h_prev | 1
    const int nrt = (NR + 31) / 32;                     // 32-row tiles of the output
    const int wave = (threadIdx.x >> 6) & 3, piece = threadIdx.x >> 8, lane = threadIdx.x & 63;
    const int c = lane & 31, kk = lane >> 5;            // column within a tile, k index (row parity)
    // this half's steps: whole groups of 8
    const int ngrp = (T + 7) / 8, g0 = piece == 0 ? 0 : (ngrp + 1) / 2, g1 = piece == 0 ? (ngrp + 1) / 2 : ngrp;
    const int t_begin = 8 * g0, t_stop = 8 * g1 < T ? 8 * g1 : T;
    const int b0 = blockIdx.x * a.spw;
    const int b1 = b0 + a.spw < a.B ? b0 + a.spw : a.B;
    const float *X = a.x[L], *HS = a.hs[L], *G = a.gates[L], *DA = a.d_act[L];
    // this wave's output tiles: (row tile rt, column tile ct) = tile index wave, wave + 4, ... of the nrt x 3 grid
    constexpr int MAXT = 3;                             // <= 12 tiles / 4 waves
    f32x16w acc[MAXT];
#pragma unroll
    for (int q = 0; q < MAXT; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    const int ntile = nrt * 3;
    for (int b = b0; b < b1; ++b) {
        const float *xb = X + (long)b * T * D, *hb = HS + (long)b * (T + 1) * H;
        const float *gb = G + (long)b * T * 3 * H, *db = DA + (long)b * T * 3 * H;
        // four 2-step MFMA operand pairs per tile in flight: the loads of a group are issued before its matrix instructions
        // (one dependent L2 round trip per k-step left the launch at 69 us for 100 MFLOP)
        constexpr int UN = 4;
        for (int t0 = t_begin; t0 < t_stop; t0 += 2 * UN) {
            float av[MAXT][UN], bv[MAXT][UN];
#pragma unroll
            for (int q = 0; q < MAXT; ++q) {
                const int tile = wave + 4 * q;
                const bool on = tile < ntile;            // (wave-uniform)
                const int rt = on ? tile / 3 : 0, ct = on ? tile - 3 * rt : 0;
                const int f = 32 * rt + c;               // operand feature of this lane
#pragma unroll
                for (int s2 = 0; s2 < UN; ++s2) {
                    const int t = t0 + 2 * s2 + kk;
                    const bool live = on && t < t_stop;
                    const int tc = t < T ? t : T - 1;
                    // BRANCH-FREE (r5): which operand a lane feeds -- an x column, a state column (times r for the candidate
                    // columns), the bias row's 1 -- differs from lane to lane inside a tile, and as branches around the loads
                    // every one of the group's 12 operand fetches was waited for at its join: 12 round trips per group
                    // instead of one, 37 us for 100 MFLOP.  Addresses are selected, every load is issued, values are selected.
                    const bool is_x = f < D, is_h = !is_x && f < D + H;
                    const int fh = is_h ? f - D : 0;
                    const float *src = is_x ? xb + (long)tc * D + f : hb + (long)tc * H + fh;
                    const float raw = *src;
                    const float rg = gb[(long)tc * 3 * H + fh];
                    float v = is_x ? raw : (is_h ? raw * (ct == 2 ? rg : 1.f) : (f == D + H ? 1.f : 0.f));
                    av[q][s2] = live ? v : 0.f;
                    const float w = db[(long)tc * 3 * H + 32 * ct + c];
                    bv[q][s2] = live ? w : 0.f;
                }
            }
#pragma unroll
            for (int q = 0; q < MAXT; ++q) {
                if (wave + 4 * q < ntile) {
#pragma unroll
                    for (int s2 = 0; s2 < UN; ++s2)
                        acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q][s2], bv[q][s2], acc[q], 0, 0, 0);
                }
            }
        }
    }
    // the second half's accumulators -> the first half's, lane for lane
    __shared__ float red[4][MAXT * 16][64];
    if (piece == 1) {
#pragma unroll
        for (int q = 0; q < MAXT; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) red[wave][q * 16 + r][lane] = acc[q][r];
    }
    __syncthreads();
    if (piece == 1) return;
#pragma unroll
    for (int q = 0; q < MAXT; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] += red[wave][q * 16 + r][lane];
    // C/D layout of 32x32x2: rows (r & 3) + 8 (r >> 2) + 4 (lane >> 5), column lane & 31
    float *slab = a.slab[L] + (long)blockIdx.x * wgrad32_slab_floats(D);
#pragma unroll
    for (int q = 0; q < MAXT; ++q) {
        const int tile = wave + 4 * q;
        if (tile < ntile) {
            const int rt = tile / 3, ct = tile - 3 * rt;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = 32 * rt + (r & 3) + 8 * (r >> 2) + 4 * kk;
                if (row < NR) slab[(long)row * 96 + 32 * ct + c] = acc[q][r];
            }
        }
    }
}

// d_w += sum over workgroups of the slabs; blockIdx.y = layer.  r5: 32 elements x 8 groups per workgroup -- group g adds slabs
// g, g + 8, ... (four running sums), the groups' sums are added in order: deterministic as before (one thread per element walked
// all 128 slabs: 32 dependent round trips, 17 us on the Amazon step's tail).
constexpr int RG32 = 8;
__global__ __launch_bounds__(32 * RG32) void gru32_wgrad_reduce_kernel(const Wgrad32Args a) {
    constexpr int H = 32;
    __shared__ float part[RG32][32];
    const int L = blockIdx.y;
    const int D = a.D[L];
    const long n = wgrad32_slab_floats(D);
    const int c = threadIdx.x & 31, g = threadIdx.x >> 5;
    const long e = (long)blockIdx.x * 32 + c;
    const long ec = e < n ? e : n - 1;
    const float *s = a.slab[L] + ec;
    float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
    int w = g;
    for (; w + 3 * RG32 < a.nwg; w += 4 * RG32) {
        t0 += s[(long)w * n]; t1 += s[(long)(w + RG32) * n]; t2 += s[(long)(w + 2 * RG32) * n]; t3 += s[(long)(w + 3 * RG32) * n];
    }
    for (; w < a.nwg; w += RG32) t0 += s[(long)w * n];
    part[g][c] = (t0 + t1) + (t2 + t3);
    __syncthreads();
    if (g != 0 || e >= n) return;
    float tot = part[0][c];
#pragma unroll
    for (int k = 1; k < RG32; ++k) tot += part[k][c];
    const int row = (int)(e / 96), col = (int)(e - (long)row * 96);
    if (row < D + H) {
        if (col < 2 * H) a.d_wg[L][(long)row * 2 * H + col] += tot;
        else a.d_wc[L][(long)row * H + (col - 2 * H)] += tot;
    } else {
        if (col < 2 * H) a.d_bg[L][col] += tot;
        else a.d_bc[L][col - 2 * H] += tot;
    }
}

static int wgrad32_spw(int B) { return B >= 256 ? 2 : 1; }

size_t gru32_wgrad_all_workspace_bytes(int B, int K, const int *D) {
    const int spw = wgrad32_spw(B), nwg = (B + spw - 1) / spw;
    size_t n = 0;
    for (int i = 0; i < K; ++i) n += (size_t)nwg * wgrad32_slab_floats(D[i]);
    return n * sizeof(float);
}

// x[i]: layer i's input rows [B, T_i, D_i]; workspace: gru32_wgrad_all_workspace_bytes
int gru32_wgrad_all_launch(int B, int K, const int *D, const int *T, const float *const *x, const float *const *hs,
                           const float *const *gates, const float *const *d_act, float *const *d_wg, float *const *d_bg,
                           float *const *d_wc, float *const *d_bc, float *workspace, hipStream_t st) {
    if (K < 1 || K > AMAXK) return HPMN_EUNSUPPORTED;
    Wgrad32Args a = {};
    a.B = B; a.K = K; a.spw = wgrad32_spw(B); a.nwg = (B + a.spw - 1) / a.spw;
    long off = 0, nmax = 0;
    for (int i = 0; i < K; ++i) {
        if (D[i] < 1 || D[i] > 64) return HPMN_EUNSUPPORTED;
        a.D[i] = D[i]; a.T[i] = T[i];
        a.x[i] = x[i]; a.hs[i] = hs[i]; a.gates[i] = gates[i]; a.d_act[i] = d_act[i];
        a.d_wg[i] = d_wg[i]; a.d_bg[i] = d_bg[i]; a.d_wc[i] = d_wc[i]; a.d_bc[i] = d_bc[i];
        a.slab[i] = workspace + off;
        const long n = wgrad32_slab_floats(D[i]);
        off += (long)a.nwg * n;
        nmax = n > nmax ? n : nmax;
    }
    hipLaunchKernelGGL(gru32_wgrad_all_kernel, dim3(a.nwg, K), dim3(256 * W32_PIECES), 0, st, a);
    int rc = check_launch();
    if (rc != HPMN_OK) return rc;
    hipLaunchKernelGGL(gru32_wgrad_reduce_kernel, dim3((unsigned)((nmax + 31) / 32), K), dim3(32 * RG32), 0, st, a);
    return check_launch();
}

}  // namespace hpmn
