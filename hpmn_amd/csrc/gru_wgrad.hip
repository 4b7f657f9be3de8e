// Weight / bias / input gradients of one GRU layer from the reverse scan's pre-activation
// gradients d_act [B,T,3H] (the time-parallel half of BPTT) for gfx950.
//
//   dWg += [x | h_prev]^T d_act[:, 0:2H]      dbg += sum_rows d_act[:, 0:2H]
//   dWc += [x | r*h_prev]^T d_act[:, 2H:3H]   dbc += sum_rows d_act[:, 2H:3H]
//   dx   = d_act [Wg[0:D] | Wc[0:D]]^T
//
// (1) gru_wgrad_kernel -- an fp32 MFMA reduction over the B*T rows.  The GEMM is extremely
//     skinny (<=128 x 192 outputs, up to 10^6 reduction rows), which is exactly the shape
//     v_mfma_f32_32x32x2_f32 wants when the REDUCTION index is the MFMA k: lane l supplies
//     A[i=l&31][k=l>>5] = Z[row+k][i0+i] and B[k][j] = d_act[row+k][j0+j], i.e. both operands
//     are plain 128-byte row segments -- no transposes, and row-major LDS images are read
//     conflict-free.  16-row tiles of (d_act | x | h_prev | r) are staged once per workgroup
//     with 16-byte coalesced loads, double-buffered in LDS (every wave needs all of d_act, so
//     staging cuts the global load instructions 12x vs operand loads straight from HBM --
//     measured 0.52-0.65 ms -> see DESIGN_HISTORY.md 3.2).  A workgroup owns whole sequences;
//     its waves split the OUTPUT tiles (wave w<DT: 32 input columns x all 3H gate columns;
//     the others: 32 state columns of h_prev x 2H and of r*h_prev x H) and keep their
//     3H/32 accumulator tiles (<=96 registers) resident over all rows; the next tile's HBM
//     loads are in flight under the current tile's MFMAs.  Each workgroup stores its
//     partial result as a slab in a caller-provided workspace and wgrad_reduce_kernel adds the
//     slabs into dW (single writer per element: deterministic, no atomics).
// (2) dx = d_act Wx^T is a row-wise transform: gru_dx_kernel in input_proj.hip.
#include <cstdlib>

#include "common.h"

namespace hpmn {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int WU = 8;          // 2-row MFMA steps per iteration (16 rows)
constexpr int WG_MIN_ROWS = 128;  // a workgroup takes whole sequences, at least this many rows

// Slab layout of one workgroup's partial result (floats): [d_wg (D+H)x2H][d_bg 2H][d_wc (D+H)xH][d_bc H]
__host__ __device__ inline long wgrad_slab_floats(int D, int H) { return (long)(D + H) * 3 * H + 3 * H; }

// CS = column split: blockIdx.y picks one of CS groups of 3*HT/CS consecutive 32-column tiles of d_act, so a
// wave holds 3*HT/CS accumulator tiles.  H = 128 uses CS = 3: twelve tiles are 192 accumulator registers,
// and with 5..8 waves per workgroup a wave only has 256 -- the unsplit kernel spilled 124 registers to
// scratch and ran at 27 TF/s; four tiles per wave also leave room for two workgroups per CU.  Each group
// stages only ITS columns of d_act (plus x, h_prev, r).
template <int HT, int DT, int CS>
__global__ __launch_bounds__(64 * (HT + DT), 2) void gru_wgrad_kernel(const HpmnGruWgrad a) {
    constexpr int H = 32 * HT;
    static_assert((3 * HT) % CS == 0, "column tiles split evenly");
    constexpr int NJ = 3 * HT / CS;       // 32-column tiles of d_act held by this workgroup
    constexpr int NC = 32 * NJ;           // ... = this many columns, starting at column jb*32
    const int jb = blockIdx.y * NJ;       // first (global) column tile
    constexpr int NT = 64 * (HT + DT);    // threads
    constexpr int R = 2 * WU;             // rows per staged tile
    constexpr int XS = 32 * DT;           // row stride of the x image (zero-filled beyond D)
    // float4 items of one tile image and per-thread staging registers
    constexpr int N_DA = R * NC / 4, N_X = R * XS / 4, N_H = R * H / 4;
    constexpr int P_DA = (N_DA + NT - 1) / NT, P_X = (N_X + NT - 1) / NT, P_H = (N_H + NT - 1) / NT;
    __shared__ __attribute__((aligned(16))) float l_da[2][R * NC];
    __shared__ __attribute__((aligned(16))) float l_x[2][R * XS];
    __shared__ __attribute__((aligned(16))) float l_h[2][R * H];
    __shared__ __attribute__((aligned(16))) float l_r[2][R * H];

    const int tid = threadIdx.x;
    const int wave = tid >> 6;
    const int lane = tid & 63;
    const int c = lane & 31;          // column within a tile
    const int rp = lane >> 5;         // row parity (MFMA k index)
    const int T = a.T, D = a.D;

    f32x16 acc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    float bsum[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) bsum[j] = 0.f;

    const bool role_x = wave < DT;
    const int tile = role_x ? wave : wave - DT;

    const int b_begin = blockIdx.x * a.seq_per_wg;
    const int b_end = (b_begin + a.seq_per_wg) < a.B ? (b_begin + a.seq_per_wg) : a.B;
    // rows of this launch: steps [tb, te) of every sequence (the whole sequence unless the caller splits the
    // reduction in time to start it before the reverse scan has finished)
    const int tb = a.t_begin, te = a.t_len > 0 ? a.t_begin + a.t_len : T;
    const int ipt = (te - tb + R - 1) / R;                // tiles per sequence
    const int niter = (b_end - b_begin) * ipt;

    struct Stage { float4 da[P_DA], x[P_X], h[P_H], r[P_H]; };
    // tile `it` = rows t0..t0+R-1 of sequence b; rows >= T are zero.  All loads are 16-byte, coalesced.
    auto load_tile = [&](int it, Stage &g) {
        const int b = b_begin + it / ipt;
        const int t0 = tb + (it % ipt) * R;
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int p = 0; p < P_DA; ++p) {
            const int i = p * NT + tid;
            const int row = i / (NC / 4), q = i % (NC / 4);
            g.da[p] = z;
            if (i < N_DA && t0 + row < te)
                g.da[p] = *reinterpret_cast<const float4 *>(a.d_act + ((long)b * T + t0 + row) * 3 * H + jb * 32 + 4 * q);
        }
#pragma unroll
        for (int p = 0; p < P_X; ++p) {
            const int i = p * NT + tid;
            const int row = i / (XS / 4), q = i % (XS / 4);
            g.x[p] = z;
            if (i < N_X && t0 + row < te && 4 * q < D)
                g.x[p] = *reinterpret_cast<const float4 *>(a.x + ((long)b * T + t0 + row) * D + 4 * q);
        }
#pragma unroll
        for (int p = 0; p < P_H; ++p) {
            const int i = p * NT + tid;
            const int row = i / (H / 4), q = i % (H / 4);
            g.h[p] = z;
            g.r[p] = z;
            if (i < N_H && t0 + row < te) {
                g.h[p] = *reinterpret_cast<const float4 *>(a.hs + ((long)b * (T + 1) + t0 + row) * H + 4 * q);
                g.r[p] = *reinterpret_cast<const float4 *>(a.gates + ((long)b * T + t0 + row) * 3 * H + 4 * q);
            }
        }
    };
    auto park_tile = [&](int buf, const Stage &g) {
#pragma unroll
        for (int p = 0; p < P_DA; ++p) {
            const int i = p * NT + tid;
            if (i < N_DA) reinterpret_cast<float4 *>(l_da[buf])[i] = g.da[p];
        }
#pragma unroll
        for (int p = 0; p < P_X; ++p) {
            const int i = p * NT + tid;
            if (i < N_X) reinterpret_cast<float4 *>(l_x[buf])[i] = g.x[p];
        }
#pragma unroll
        for (int p = 0; p < P_H; ++p) {
            const int i = p * NT + tid;
            if (i < N_H) {
                reinterpret_cast<float4 *>(l_h[buf])[i] = g.h[p];
                reinterpret_cast<float4 *>(l_r[buf])[i] = g.r[p];
            }
        }
    };

    Stage st;
    if (niter > 0) {
        load_tile(0, st);
        park_tile(0, st);
    }
    __syncthreads();
    for (int it = 0; it < niter; ++it) {
        const int buf = it & 1;
        if (it + 1 < niter) load_tile(it + 1, st);       // HBM loads in flight under this tile's MFMAs
        const float *pda = &l_da[buf][rp * NC + c];
        if (role_x) {
            const float *px = &l_x[buf][rp * XS + 32 * tile + c];
#pragma unroll
            for (int s = 0; s < WU; ++s) {
                const float av = px[2 * s * XS];
                float bv[NJ];
#pragma unroll
                for (int j = 0; j < NJ; ++j) bv[j] = pda[2 * s * NC + 32 * j];
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[j], acc[j], 0, 0, 0);
                if (tile == 0) {
#pragma unroll
                    for (int j = 0; j < NJ; ++j) bsum[j] += bv[j];
                }
                if (s & 1) asm volatile("" ::: "memory");   // bound the LDS reads hoisted ahead (VGPR budget)
            }
        } else {
            const float *ph = &l_h[buf][rp * H + 32 * tile + c];
            const float *pr = &l_r[buf][rp * H + 32 * tile + c];
#pragma unroll
            for (int s = 0; s < WU; ++s) {
                const float hv = ph[2 * s * H];
                const float rh = pr[2 * s * H] * hv;      // r * h_prev (not stored by the forward)
                float bv[NJ];
#pragma unroll
                for (int j = 0; j < NJ; ++j) bv[j] = pda[2 * s * NC + 32 * j];
                // gate columns (global tile < 2 HT) pair with h_prev, candidate columns with r * h_prev
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32((jb + j) < 2 * HT ? hv : rh, bv[j], acc[j], 0, 0, 0);
                if (s & 1) asm volatile("" ::: "memory");
            }
        }
        if (it + 1 < niter) park_tile(buf ^ 1, st);
        __syncthreads();
    }

    // ---- epilogue: plain coalesced stores of this workgroup's partial slab (summed by
    //      wgrad_reduce_kernel).  C/D layout of 32x32x2: rows (r&3) + 8*(r>>2) + 4*(lane>>5), col lane&31
    float *slab = a.workspace + (long)blockIdx.x * wgrad_slab_floats(D, H);
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
            const int i = (r & 3) + 8 * (r >> 2) + 4 * rp;
            if (!role_x || 32 * tile + i < D) dst[(long)(row_base + i) * ld + col] = acc[j][r];
        }
    }
    if (role_x && tile == 0) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            // the two half-waves hold the even / odd rows' sums of the same column
            const float tot = bsum[j] + __shfl_xor(bsum[j], 32);
            if (rp == 0) {
                const int jg = jb + j;
                if (jg < 2 * HT) s_bg[32 * jg + c] = tot;
                else             s_bc[32 * (jg - 2 * HT) + c] = tot;
            }
        }
    }
}

// dst[e] += sum over workgroups of slab[w][e]  (single writer per element, fixed summation order:
// deterministic, no atomics).  A block owns 32 consecutive elements; its 8 groups of 32 lanes each sum
// every 8th slab (four independent partial sums per thread, 128-byte coalesced reads) and the groups are
// combined through LDS in a fixed order -- an element's slab column is a few hundred values deep, one
// thread per element left the launch latency-bound (60 us for 25 k elements).
constexpr int RED_G = 8;
__global__ __launch_bounds__(32 * RED_G) void wgrad_reduce_kernel(const float *__restrict__ ws, int nwg, long n,
                                                                  float *d_wg, float *d_bg, float *d_wc, float *d_bc,
                                                                  long n_wg, long n_bg, long n_wc) {
    __shared__ float part[RED_G][32];
    const int c = threadIdx.x & 31, g = threadIdx.x >> 5;
    const long e = (long)blockIdx.x * 32 + c;
    const long ec = e < n ? e : n - 1;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int w = g;
    for (; w + 3 * RED_G < nwg; w += 4 * RED_G) {
        s0 += ws[(long)w * n + ec];
        s1 += ws[(long)(w + RED_G) * n + ec];
        s2 += ws[(long)(w + 2 * RED_G) * n + ec];
        s3 += ws[(long)(w + 3 * RED_G) * n + ec];
    }
    for (; w < nwg; w += RED_G) s0 += ws[(long)w * n + ec];
    part[g][c] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (g != 0 || e >= n) return;
    float tot = part[0][c];
#pragma unroll
    for (int k = 1; k < RED_G; ++k) tot += part[k][c];
    if (e < n_wg) d_wg[e] += tot;
    else if (e < n_wg + n_bg) d_bg[e - n_wg] += tot;
    else if (e < n_wg + n_bg + n_wc) d_wc[e - n_wg - n_bg] += tot;
    else d_bc[e - n_wg - n_bg - n_wc] += tot;
}

int gru_dx_dispatch(const HpmnGruWgrad &a, hipStream_t st);   // input_proj.hip (row-wise MFMA)
bool gru_wgrad_bf16_launch(const HpmnGruWgrad &k, int nwg, bool solo, hipStream_t st);   // gru_wgrad_bf16.hip

static int wgrad_seq_per_wg(int T, int H) {   // T = steps per sequence in this launch
    // (HPMN_WGRAD_MIN_ROWS: every workgroup writes a slab of (D + H + 1) 3H floats -- 98 KB at H = D = 64 -- and the reduction
    //  reads it back; for the short top layers the slabs outweigh the rows they summarise.  H = 128: a slab is 395 KB and the
    //  launch has three column groups' worth of workgroups anyway: 1024 rows -- C4 7.44 -> 7.34 ms/step)
    static const int env_rows = [] { const char *e = getenv("HPMN_WGRAD_MIN_ROWS"); return e && atoi(e) > 0 ? atoi(e) : 0; }();
    const int min_rows = env_rows > 0 ? env_rows : (H >= 128 ? 1024 : WG_MIN_ROWS);
    int spw = (min_rows + T - 1) / T;
    return spw < 1 ? 1 : spw;
}

// A launch that has the chip to itself (layer 0's, in the step's tail: whole_cu) is bound by ONE memory round trip per 16-row
// tile with whatever workgroups a CU holds to overlap it: 500 three-wave workgroups are two per CU.  HPMN_WGRAD_TSPLIT=n cuts
// its tiles into n consecutive pieces per sequence range -- n times the workgroups, three resident per CU (the kernel fits 168
// registers for that), a slab each.  Measured at C3 (n = 1 / 2 / 3 / 4): 2.504-2.53 / 2.53 / 2.555 / 2.55-2.57 ms per step --
// the launch itself shrinks (292 -> 248 us) but the scatter and the late table-Adam pass beside it lose what it gains (both
// branches of the tail are bandwidth kernels) and the reduction reads n times the slabs: default 1.
static int wgrad_tsplit(int D, int H) {
    static const int env = [] { const char *e = getenv("HPMN_WGRAD_TSPLIT"); return e && atoi(e) > 0 ? atoi(e) : 1; }();
    return (H == 64 && D <= 32) ? (env > 8 ? 8 : env) : 1;
}

size_t gru_wgrad_workspace_bytes(int B, int T, int D, int H) {
    const int spw = wgrad_seq_per_wg(T, H);
    const long nwg = (B + spw - 1) / spw;
    return (size_t)nwg * (size_t)wgrad_tsplit(D, H) * (size_t)wgrad_slab_floats(D, H) * sizeof(float);
}

// The slab reduction of a launch may go to ANOTHER stream (gru_wgrad_reduce_aside: set by hpmn_scan_bwd around its calls, per
// thread): the next layer's weight-gradient launch then does not queue behind a reduction that is starving beside a reverse scan
// (H = 128, r5: the scans own every register of the chip, a 197 MB reduction took 811 us beside one and the whole chain of
// weight gradients fell behind it).  The caller gives every layer its own slab buffer then.
static thread_local hipStream_t g_reduce_stream = nullptr;
static thread_local hipEvent_t g_reduce_event = nullptr;
void gru_wgrad_reduce_aside(hipStream_t reduce_stream, hipEvent_t ev) { g_reduce_stream = reduce_stream; g_reduce_event = ev; }

template <int HT, int DT, int CS = 1>
static int launch_wgrad(const HpmnGruWgrad &a, hipStream_t st) {
    HpmnGruWgrad k = a;
    k.seq_per_wg = wgrad_seq_per_wg(a.t_len > 0 ? a.t_len : a.T, a.H);
    // One weight-gradient workgroup per CU while the launch shares the chip with a reverse scan (every layer but the
    // longest): two of them take every register of a CU, and the single-wave workgroups of the next reverse scan --
    // the serial chain -- then cannot even be dispatched until they retire.  Padding the workgroup's LDS to 82 KiB with
    // (unused) dynamic LDS caps the occupancy at one per CU and still leaves the 76 KiB a reverse-scan workgroup with
    // its input-gradient ring needs beside it; HPMN_WGRAD_SOLO_ROWS = largest B*T that gets this treatment (0 = never).
    // Measured (C3 / C2 / C4 steps, ms): 3.985 -> 3.846 / 1.655 -> 1.591 / 10.29 -> 10.47: on for H <= 64 (the H = 128
    // kernels are already shaped for two workgroups per CU around their column split).
    static const long solo_env = [] { const char *e = getenv("HPMN_WGRAD_SOLO_ROWS"); return e ? atol(e) : -1L; }();
    const long solo_rows = solo_env >= 0 ? solo_env : (a.H <= 64 ? (1L << 62) : 0L);
    const long rows = (long)a.B * (a.t_len > 0 ? a.t_len : a.T);
    // A launch that is capped at one workgroup per CU anyway (beside a reverse scan) gains nothing from more workgroups than
    // CUs: let each take ceil(B / CUs) sequences -- half the slabs at B = 500 (their write + the reduction's read: 0.15 GB of
    // the step's 8.3, and the reduce launches on the helper stream's chain shrink with them).
    if (rows <= solo_rows && !a.whole_cu) {
        static const int cus = [] {
            int dev = 0, n = 0;
            if (hipGetDevice(&dev) != hipSuccess) return 256;
            if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) return 256;
            return n;
        }();
        const int per_cu = (a.B + cus - 1) / cus;
        if (per_cu > k.seq_per_wg) k.seq_per_wg = per_cu;
    }
    const int nwg = (a.B + k.seq_per_wg - 1) / k.seq_per_wg;
    size_t lds_pad = 0;
    if (rows <= solo_rows && !a.whole_cu) {
        static const size_t pad = [] {
            hipFuncAttributes fa = {};
            const void *fn = reinterpret_cast<const void *>(gru_wgrad_kernel<HT, DT, CS>);
            if (hipFuncGetAttributes(&fa, fn) != hipSuccess) return (size_t)0;
            const size_t want = 82 * 1024;
            const size_t p = fa.sharedSizeBytes < want ? want - fa.sharedSizeBytes : 0;
            (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p);
            return p;
        }();
        lds_pad = pad;
    }
    // H = 64: the same reduction on the bf16 matrix pipe with split operands (gru_wgrad_bf16.hip), HPMN_WGRAD_BF16=0: fp32
    static const int bf16_env = [] { const char *e = getenv("HPMN_WGRAD_BF16"); return e ? atoi(e) : 1; }();
    // (the time split of a whole-CU launch travels in the kernel's copy of `whole_cu`; the fp32 kernel does not know it)
    int nslab = nwg;
    bool done = false;
    if (bf16_env && HT >= 2) {
        HpmnGruWgrad kb = k;
        const long tiles = ((a.t_len > 0 ? a.t_len : a.T) + 15) / 16;
        const int ts = a.whole_cu && tiles >= 8 ? wgrad_tsplit(a.D, a.H) : 1;
        kb.whole_cu = ts > 1 ? ts : 0;
        done = gru_wgrad_bf16_launch(kb, nwg * ts, rows <= solo_rows && !a.whole_cu, st);
        if (done) nslab = nwg * ts;
    }
    if (!done)
        hipLaunchKernelGGL((gru_wgrad_kernel<HT, DT, CS>), dim3((unsigned)nwg, CS), dim3(64 * (HT + DT)), lds_pad, st, k);
    int rc = check_launch();
    if (rc != HPMN_OK) return rc;
    const int H = a.H, D = a.D;
    const long n = wgrad_slab_floats(D, H);
    hipStream_t rst = st;
    if (g_reduce_stream != nullptr && g_reduce_event != nullptr && g_reduce_stream != st) {
        if (hipEventRecord(g_reduce_event, st) != hipSuccess || hipStreamWaitEvent(g_reduce_stream, g_reduce_event, 0) != hipSuccess) {
            set_last_hip_error((int)hipGetLastError());
            return HPMN_EHIP;
        }
        rst = g_reduce_stream;
    }
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((n + 31) / 32)), dim3(32 * RED_G), 0, rst, a.workspace, nslab, n,
                       a.d_wg, a.d_bg, a.d_wc, a.d_bc, (long)(D + H) * 2 * H, (long)2 * H, (long)(D + H) * H);
    return check_launch();
}

int gru_wgrad_dispatch(const HpmnGruWgrad &a, hipStream_t st) {
    const int DT = (a.D + 31) / 32;
    int rc = HPMN_EUNSUPPORTED;
    if (a.H == 32 && DT == 1) rc = launch_wgrad<1, 1>(a, st);
    else if (a.H == 32 && DT == 2) rc = launch_wgrad<1, 2>(a, st);
    else if (a.H == 64 && DT == 1) rc = launch_wgrad<2, 1>(a, st);
    else if (a.H == 64 && DT == 2) rc = launch_wgrad<2, 2>(a, st);
    else if (a.H == 128 && DT == 1) rc = launch_wgrad<4, 1, 3>(a, st);  // 3 column groups x 4 accumulator tiles,
    else if (a.H == 128 && DT == 4) rc = launch_wgrad<4, 4, 3>(a, st);  // two workgroups per CU
    if (rc != HPMN_OK || a.d_x == nullptr) return rc;
    return gru_dx_dispatch(a, st);
}

}  // namespace hpmn
