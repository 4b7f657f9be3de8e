// Memory read path for gfx950: covariance regulariser + multi-hop attention over the K memory
// slots + prediction head + log-loss, forward and (fused) forward+backward.
//
// Reference: Hpmn_Basic.get_covreg / query_memory / attention / build_fc_net,
// code/hpmn.py:161-170, 172-182, 133-146, 184-207.
//
// The work is tiny (K <= 12 slots, ~0.55 MFLOP-pairs per sample) but the reference graph -- and a
// library implementation -- spends it in ~150 launches per step.  Here one workgroup owns a tile of
// RS samples and keeps every activation of the tile in LDS:
//   * inference: one pass, writes prediction / first-hop attention weights / covariance loss;
//   * training : the same forward (activations stay in LDS), then the backward in the same launch:
//     d_memory, d_last and the tile's contribution to every read-path weight gradient, written as a
//     slab that mirrors the contiguous read-path range of the flat parameter buffer and summed over
//     tiles by read_reduce_kernel (single writer per element, deterministic).
// Dense layers with at least 16 outputs run on f32 MFMA tiles (the tile's <= 24 rows fill one 32-row
// tile; the four waves split the output columns); the single-column logit layers stay on VALU loops.
#include "common.h"

namespace hpmn {

#ifndef HPMN_READ_RS
#define HPMN_READ_RS 2
#endif
constexpr int RS = HPMN_READ_RS;          // samples per workgroup
// threads per workgroup: a RUN-TIME quantity in the device code (r5) -- the inference launch and the fp32 training launch run
// four waves, the training launch on bf16 fragments eight (two per SIMD: its phases are instruction streams of a few hundred
// instructions per wave, and a lone wave issues one instruction per ~5 cycles; C3 -19 us, C1 -10 us per launch against four)
#ifndef HPMN_READ_RT
#define HPMN_READ_RT 256
#endif
#ifndef HPMN_READ_RT_BF
#define HPMN_READ_RT_BF 512
#endif
constexpr int RT_BASE = HPMN_READ_RT, RT_BF = HPMN_READ_RT_BF, RT_MAX = RT_BF > RT_BASE ? RT_BF : RT_BASE;
#define RT ((int)blockDim.x)
constexpr int A1 = 80, A2 = 40;      // attention MLP widths (code/hpmn.py:137-138)
constexpr int F1 = 200, F2 = 80;     // head widths (code/hpmn.py:191,193)
// Row strides in LDS: every activation row is padded by 4 floats.  The products read an operand row per lane (16 bytes
// at lane_row * stride): with the natural strides (256, 80, 200 ... floats) the rows of a tile start in the same few
// banks; stride = 4 * odd spreads 16 rows over all 64 banks (worth ~3 % of the launch: the operand reads are a small
// part of a layer's time).
constexpr int PADF = 4;
constexpr int A1P = A1 + PADF, A2P = A2 + PADF, F1P = F1 + PADF, F2P = F2 + PADF;
constexpr int MAXK = HPMN_MAX_LAYERS;
constexpr int MAXHOP = 4;

#ifdef READ_CLOCK      // (timing-only build, tools/read_clock.sh: cycles of workgroup 0 per phase, printed at the end of the launch;
                       //  phases inside the hop loop are SUMS over the hops -- three in every bench configuration)
__shared__ unsigned long long rck_acc[48], rck_last;
#define RCLK(i)                                                                                          \
    do {                                                                                                 \
        __syncthreads();                                                                                 \
        if (threadIdx.x == 0 && blockIdx.x == 0) {                                                       \
            const unsigned long long t_ = clock64();                                                     \
            rck_acc[i] += t_ - rck_last;                                                                 \
            rck_last = t_;                                                                               \
        }                                                                                                \
    } while (0)
// (inside dense_bf, no barrier: cycles of wave 0 of workgroup 0 between the marks, summed over all calls)
#define BFCLK(i)                                                                                         \
    do {                                                                                                 \
        if (threadIdx.x == 0 && blockIdx.x == 0) {                                                       \
            const unsigned long long t_ = clock64();                                                     \
            rck_acc[i] += t_ - rck_last;                                                                 \
            rck_last = t_;                                                                               \
        }                                                                                                \
    } while (0)
#else
#define RCLK(i)
#define BFCLK(i)
#endif


// The barriers of these kernels order LDS traffic only: nothing a workgroup writes to global memory (the tape, d_memory, d_last,
// the predictions, the loss atomics) is read again inside the launch.  __syncthreads() also waits for every outstanding global
// STORE of the wave (s_waitcnt vmcnt(0) in front of s_barrier): each tape_store in front of a barrier cost the workgroup a round
// trip to memory -- ~20 of them per training launch, 2-4 k cycles each (tools/read_clock.sh: the two 128-float tape rows of mark
// 39 took 3.9 k cycles per hop).  r5: wait for the LDS counter alone.
__device__ __forceinline__ void rsync() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ float elu(float x) { return x > 0.f ? x : __expf(x) - 1.f; }

// Dropout keep factor (0 or 1) of unit j of sample b in head layer `layer`: the caller's mask when one is
// given, else a counter-based draw -- splitmix64 of (seed, layer, b, j) -- so forward and backward of the
// same launch see the same mask without materialising it (saves six framework launches per step).
__device__ __noinline__ float keep_factor(const float *mask, uint64_t seed, int layer, long b, int j, int width,
                                          float keep_prob) {
    if (mask != nullptr) return mask[b * width + j];
    if (seed == 0) return 1.f;
    // distinct odd multipliers per coordinate (the caller's seed is itself a mixed value, not a counter)
    uint64_t z = seed + (uint64_t)(b + 1) * 0x9E3779B97F4A7C15ull + (uint64_t)(j + 1) * 0xC2B2AE3D27D4EB4Full +
                 (uint64_t)layer * 0xD1B54A32D192ED03ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    const float u = (float)(z >> 40) * (1.0f / 16777216.0f);          // [0, 1)
    return u < keep_prob ? 1.f : 0.f;
}

// Y[r][n] = act(b[n] + sum_i X[r][i] W[i][n]) for r < R, n < N.  X, Y in LDS; W [I,N] row-major in
// global (coalesced over n).  ACT: 0 none, 1 relu, 2 elu.
// Register blocking: a work item = (column n, block of RB rows) carries RB accumulators, so one
// weight load feeds RB FMAs (the x operands are LDS broadcasts).  Single owner per output:
// deterministic, no atomics, no internal barriers.
constexpr int RB = 8;

template <int ACT>
__device__ __forceinline__ void dense_fwd_valu(const float *X, int ldx, int R, int I, const float *W, const float *b,
                                               int N, float *Y, int ldy) {
    const int nrb = (R + RB - 1) / RB;
    for (int o = threadIdx.x; o < nrb * N; o += RT) {
        const int rb = o / N, n = o - rb * N;
        const int r0 = rb * RB;
        float acc[RB];
#pragma unroll
        for (int j = 0; j < RB; ++j) acc[j] = b != nullptr ? b[n] : 0.f;
        const float *xr[RB];
#pragma unroll
        for (int j = 0; j < RB; ++j) xr[j] = X + ((r0 + j) < R ? (r0 + j) : (R - 1)) * ldx;
        // the weight column walk is a chain of L2 round trips: keep WU loads in flight
        constexpr int WU = 8;
        int i = 0;
        for (; i + WU <= I; i += WU) {
            float wv[WU];
#pragma unroll
            for (int q = 0; q < WU; ++q) wv[q] = W[(long)(i + q) * N + n];
#pragma unroll
            for (int q = 0; q < WU; ++q)
#pragma unroll
                for (int j = 0; j < RB; ++j) acc[j] = fmaf(xr[j][i + q], wv[q], acc[j]);
        }
        for (; i < I; ++i) {
            const float w = W[(long)i * N + n];
#pragma unroll
            for (int j = 0; j < RB; ++j) acc[j] = fmaf(xr[j][i], w, acc[j]);
        }
#pragma unroll
        for (int j = 0; j < RB; ++j) {
            if (r0 + j < R) {
                float v = acc[j];
                if (ACT == 1) v = fmaxf(v, 0.f);
                if (ACT == 2) v = elu(v);
                Y[(r0 + j) * ldy + n] = v;
            }
        }
    }
}

// dX[r][i] (+)= sum_n dY[r][n] W[i][n]      thread = input unit i, walks its weight row once per RB rows
template <bool ACCUM>
__device__ __forceinline__ void dense_bwd_x_valu(const float *dY, int ldy, int R, int N, const float *W, int I, float *dX,
                                                 int ldx) {
    for (int i = threadIdx.x; i < I; i += RT) {
        const float *w = W + (long)i * N;
        for (int r0 = 0; r0 < R; r0 += RB) {
            float acc[RB];
#pragma unroll
            for (int j = 0; j < RB; ++j) acc[j] = 0.f;
            const float *dr[RB];
#pragma unroll
            for (int j = 0; j < RB; ++j) dr[j] = dY + ((r0 + j) < R ? (r0 + j) : (R - 1)) * ldy;
            int n = 0;
            if ((N & 3) == 0) {              // weight rows are 16-byte aligned (N % 4 == 0): 4 columns per load
                for (; n + 8 <= N; n += 8) {
                    const float4 w0 = *reinterpret_cast<const float4 *>(w + n);
                    const float4 w1 = *reinterpret_cast<const float4 *>(w + n + 4);
#pragma unroll
                    for (int j = 0; j < RB; ++j) {
                        float a = acc[j];
                        a = fmaf(dr[j][n + 0], w0.x, a); a = fmaf(dr[j][n + 1], w0.y, a);
                        a = fmaf(dr[j][n + 2], w0.z, a); a = fmaf(dr[j][n + 3], w0.w, a);
                        a = fmaf(dr[j][n + 4], w1.x, a); a = fmaf(dr[j][n + 5], w1.y, a);
                        a = fmaf(dr[j][n + 6], w1.z, a); a = fmaf(dr[j][n + 7], w1.w, a);
                        acc[j] = a;
                    }
                }
            }
            for (; n < N; ++n) {
                const float wv = w[n];
#pragma unroll
                for (int j = 0; j < RB; ++j) acc[j] = fmaf(dr[j][n], wv, acc[j]);
            }
#pragma unroll
            for (int j = 0; j < RB; ++j)
                if (r0 + j < R) {
                    if (ACCUM) dX[(r0 + j) * ldx + i] += acc[j];
                    else dX[(r0 + j) * ldx + i] = acc[j];
                }
        }
    }
}

// ---- MFMA versions (v_mfma_f32_32x32x2_f32) ---------------------------------------------------------
// The tile has R <= RS*MAXK = 24 rows, so ONE 32-row MFMA tile holds all of them (lanes past R
// repeat row R-1; their outputs are dropped) and the four waves of the workgroup split the output
// column tiles.  As in input_proj.hip the MFMA k index goes to the half-waves and half-wave p takes
// the contiguous half [p*Kd/2, (p+1)*Kd/2) of the reduction, so LDS / weight-row operands are
// 16-byte reads.  Layers with a single output column (the logits) and reductions that are not a
// multiple of 8 stay on the VALU loops above.
typedef float f32x16 __attribute__((ext_vector_type(16)));

// Y[r][n] = act(b[n] + sum_i X[r][i] W[i][n]):  A = X (LDS rows), B = W[k][n] (global, coalesced over n)
// N == 1 (the logit layers): wave per row, lanes over the reduction -- the weight column is ONE coalesced
// L2 round trip instead of a serial chain of them in a single thread
template <int ACT>
__device__ __forceinline__ void dense_fwd_col(const float *X, int ldx, int R, int I, const float *W, const float *b,
                                              float *Y, int ldy) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int r = wave; r < R; r += RT / 64) {
        float acc = 0.f;
        for (int i = lane; i < I; i += 64) acc = fmaf(X[r * ldx + i], W[i], acc);
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m);
        if (lane == 0) {
            float v = acc + (b != nullptr ? b[0] : 0.f);
            if (ACT == 1) v = fmaxf(v, 0.f);
            if (ACT == 2) v = elu(v);
            Y[r * ldy] = v;
        }
    }
}

template <int ACT>
__device__ __forceinline__ void dense_fwd_g(const float *X, int ldx, int R, int I, const float *W, const float *b,
                                          int N, float *Y, int ldy) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int c = lane & 31, p = lane >> 5;
    const int KH = I >> 1;
    const float *xr = X + (c < R ? c : R - 1) * ldx + p * KH;
    for (int nt = wave; nt * 32 < N; nt += RT / 64) {
        const int n = nt * 32 + c, nc = n < N ? n : N - 1;
        const float *wp = W + (long)(p * KH) * N + nc;
        const float bn = b != nullptr ? b[nc] : 0.f;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = bn;
        // weights stream from L2 one 16-step chunk ahead (a chunk of MFMAs ~ 1000 cycles covers the
        // round trip; one 8-step block ahead did not: the kernel was waiting on every block)
        constexpr int CH = 16;
        float wv[CH], wn[CH];
#pragma unroll
        for (int e = 0; e < CH; ++e) wv[e] = wp[(long)(e < KH ? e : KH - 1) * N];
        for (int k0 = 0; k0 < KH; k0 += CH) {
            const int kn = k0 + CH < KH ? k0 + CH : k0;                  // harmless reload on the last chunk
            const int rem = KH - kn;                                     // >= 4 (KH % 4 == 0)
#pragma unroll
            for (int e = 0; e < CH; ++e) wn[e] = wp[(long)(kn + (e < rem ? e : rem - 1)) * N];
#pragma unroll
            for (int j = 0; j < CH / 4; ++j) {
                if (k0 + 4 * j < KH) {                                   // wave-uniform
                    const float4 x = *reinterpret_cast<const float4 *>(xr + k0 + 4 * j);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x.x, wv[4 * j], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x.y, wv[4 * j + 1], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x.z, wv[4 * j + 2], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x.w, wv[4 * j + 3], acc, 0, 0, 0);
                }
            }
#pragma unroll
            for (int e = 0; e < CH; ++e) wv[e] = wn[e];
        }
        // C/D layout: lane (c, p), reg r -> row (r&3) + 8*(r>>2) + 4*p, column n
        if (n < N) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * p;
                if (row < R) {
                    float v = acc[r];
                    if (ACT == 1) v = fmaxf(v, 0.f);
                    if (ACT == 2) v = elu(v);
                    Y[row * ldy + n] = v;
                }
            }
        }
    }
}

// dX[r][i] (+)= sum_n dY[r][n] W[i][n]:  transposed product dX^T = W dY^T, A = W rows (global, 16-byte
// pieces of the lane's own row), B = dY (LDS rows); a lane ends up with 4 consecutive i of row r = c.
template <bool ACCUM>
__device__ __forceinline__ void dense_bwd_x_g(const float *dY, int ldy, int R, int N, const float *W, int I, float *dX,
                                            int ldx) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int c = lane & 31, p = lane >> 5;
    const int KH = N >> 1;
    const float *yr = dY + (c < R ? c : R - 1) * ldy + p * KH;
    for (int it = wave; it * 32 < I; it += RT / 64) {
        const int i = it * 32 + c;
        const float *wr = W + (long)(i < I ? i : I - 1) * N + p * KH;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        // the lane's weight-row pieces stream from L2 four 16-byte pieces (16 k-steps) ahead
        constexpr int PQ = 4;
        float4 wv[PQ], wn[PQ];
        const int nq = KH / 4;
#pragma unroll
        for (int e = 0; e < PQ; ++e) wv[e] = *reinterpret_cast<const float4 *>(wr + 4 * (e < nq ? e : nq - 1));
        for (int q0 = 0; q0 < nq; q0 += PQ) {
            const int qn = q0 + PQ < nq ? q0 + PQ : q0;
            const int rem = nq - qn;
#pragma unroll
            for (int e = 0; e < PQ; ++e) wn[e] = *reinterpret_cast<const float4 *>(wr + 4 * (qn + (e < rem ? e : rem - 1)));
#pragma unroll
            for (int e = 0; e < PQ; ++e) {
                if (q0 + e < nq) {                                       // wave-uniform
                    const float4 y = *reinterpret_cast<const float4 *>(yr + 4 * (q0 + e));
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[e].x, y.x, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[e].y, y.y, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[e].z, y.z, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[e].w, y.w, acc, 0, 0, 0);
                }
            }
#pragma unroll
            for (int e = 0; e < PQ; ++e) wv[e] = wn[e];
        }
        // D[i_local][r]: lane (c = r, p), regs 4g..4g+3 -> i = it*32 + 8g + 4p + 0..3
        if (c < R) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int i0 = it * 32 + 8 * g + 4 * p;
                if (i0 < I) {                       // I % 4 == 0: a quad is all in or all out
                    float4 *dst = reinterpret_cast<float4 *>(dX + c * ldx + i0);
                    float4 v = make_float4(acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]);
                    if (ACCUM) { const float4 o = *dst; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
                    *dst = v;
                }
            }
        }
    }
}

template <int ACT>
__device__ __forceinline__ void dense_fwd(const float *X, int ldx, int R, int I, const float *W, const float *b, int N, float *Y,
                                          int ldy) {
#ifdef READ_ABLATE_FWD          // (timing-only builds, tools/read_ablate.sh: where the kernel's time goes; results are wrong)
    return;
#endif
    if (N == 1) { dense_fwd_col<ACT>(X, ldx, R, I, W, b, Y, ldy); return; }
    if (N < 16 || (I & 7) != 0) { dense_fwd_valu<ACT>(X, ldx, R, I, W, b, N, Y, ldy); return; }
    dense_fwd_g<ACT>(X, ldx, R, I, W, b, N, Y, ldy);
}

template <bool ACCUM>
__device__ __forceinline__ void dense_bwd_x(const float *dY, int ldy, int R, int N, const float *W, int I, float *dX, int ldx) {
#ifdef READ_ABLATE_BWD_X
    return;
#endif
    if ((N & 7) != 0 || (I & 3) != 0 || I < 16) { dense_bwd_x_valu<ACCUM>(dY, ldy, R, N, W, I, dX, ldx); return; }
    dense_bwd_x_g<ACCUM>(dY, ldy, R, N, W, I, dX, ldx);
}

// ---- r5: the TRAINING launch's dense layers on the bf16 matrix pipe with split operands ------------------------------------
// tools/read_clock.sh on the r4 kernel (C3, cycles of one workgroup, three hops summed): the 4H -> 80 layer 54 k of 364 k,
// 80 -> 40 22 k, q Hmap 19 k, the head's two layers 19 + 16 k, the transposed products another ~90 k -- and HALF of each is
// the matrix pipe itself: a 32x32x2 fp32 instruction holds it for 64 cycles and carries TWO k-steps, 128 of them one behind
// the other for I = 256, on a tile whose 32 rows hold 14 (two samples x seven slots).  v_mfma_f32_16x16x32_bf16 takes 32
// k-steps in 16 cycles and its 16 rows fit the 14; with every fp32 operand split into THREE bf16 planes (x = h + m + l,
// |x - h - m - l| <= 2^-25 |x|) and the six products of order <= 2 (hH hM mH hL lH mM; the dropped ones are below 2^-24 of
// the product) the result is fp32-equivalent -- no precision claim changes -- for 96 cycles of pipe per 32 k-steps instead of
// 1024.  The other half of a layer's time was the weight stream: a column of W per lane, one dword per k-step.  Here the
// weights arrive as OPERAND FRAGMENTS: read_wimg_kernel (one small launch in front of the training launch; the weights change
// every step) writes, for every dense layer and both directions (Y = X W: B[k][n] = W[k][n];  dX = dY W^T: B[k][i] = W[i][k]),
// the three planes of every (16-column tile, 32-k chunk) in lane order -- a wave reads a fragment plane with ONE linear
// 16-byte load per lane, zero padding included.
typedef __bf16 rbf8 __attribute__((ext_vector_type(8)));
typedef float rf4 __attribute__((ext_vector_type(4)));
#ifndef HPMN_READ_NSF
#define HPMN_READ_NSF 3          // planes of the forward products (3: fp32-equivalent; 2: 2^-17)
#endif
#ifndef HPMN_READ_NSB
#define HPMN_READ_NSB 3          // planes of the transposed (input-gradient) products (2: 8.5e-6 of max|grad| instead of 7e-7, same speed)
#endif
constexpr int IMG_FRAG = 3 * 64;          // uint4 per fragment: [plane][lane]

template <int NS>
__device__ __forceinline__ void split_planes(const float (&v)[8], rbf8 (&out)[NS]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float r = v[j];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const __bf16 h = (__bf16)r;
            out[s][j] = h;
            r -= (float)h;
        }
    }
}

struct ImgLayer { int w_off, I, N, fwd, bwd, first; };      // offsets of the two images in uint4; first fragment item
constexpr int IMG_MAXL = 2 * (2 + 2 * MAXHOP) + 2;
struct ImgArgs {
    ImgLayer L[IMG_MAXL];
    int nl, items;
    const float *P;
    uint4 *img;
};
// where the training launch finds each layer's images (uint4 offsets from `base`; [..][0] forward, [..][1] transposed)
struct ReadImg {
    const uint4 *base;
    int total;                            // uint4 in all images
    int wq[2][2], map[2][2];
    int att[2][MAXHOP][2][2];             // [branch][hop][4H -> A1 | A1 -> A2][direction]
    int fc[2][2];                         // [fc1 | fc2][direction]
};

__global__ __launch_bounds__(256) void read_wimg_kernel(const ImgArgs a) {
    const int item = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= a.items) return;
    const int lane = threadIdx.x & 63, n = lane & 15, kg = lane >> 4;
    int li = 0;
#pragma unroll 1
    for (int i = 1; i < a.nl; ++i) if (a.L[i].first <= item) li = i;
    const int w_off = a.L[li].w_off, I = a.L[li].I, N = a.L[li].N;
    int j = item - a.L[li].first;
    const int nf = ((N + 15) >> 4) * ((I + 31) >> 5);
    const float *W = a.P + w_off;
    float v[8];
    uint4 *dst;
    if (j < nf) {                                     // forward image: fragment (column tile t, k chunk c), k = input unit
        const int nch = (I + 31) >> 5, t = j / nch, c = j - t * nch;
        const int col = 16 * t + n, k0 = 32 * c + 8 * kg;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (col < N && k0 + e < I) ? W[(long)(k0 + e) * N + col] : 0.f;
        dst = a.img + a.L[li].fwd + (long)j * IMG_FRAG + lane;
    } else {                                          // transposed image: fragment (input tile t, chunk c of the OUTPUT units)
        j -= nf;
        const int nch = (N + 31) >> 5, t = j / nch, c = j - t * nch;
        const int row = 16 * t + n, k0 = 32 * c + 8 * kg;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (row < I && k0 + e < N) ? W[(long)row * N + k0 + e] : 0.f;
        dst = a.img + a.L[li].bwd + (long)j * IMG_FRAG + lane;
    }
    rbf8 p[3];
    split_planes<3>(v, p);
#pragma unroll
    for (int s = 0; s < 3; ++s) dst[s * 64] = *reinterpret_cast<const uint4 *>(&p[s]);
}

// the layers that have images, in image order; fills `im` (offsets) and, when L != nullptr, the builder's table.
// Returns the images' size in uint4.
__host__ __device__ inline long img_layout(const HpmnReadDesc &d0, const HpmnReadDesc &d1, int nb, ReadImg *im, ImgLayer *L,
                                           int *nl_out, int *items_out) {
    long off = 0;
    int nl = 0, items = 0;
    auto add = [&](int w_off, int I, int N, int (&slot)[2]) {
        const int nf = ((N + 15) >> 4) * ((I + 31) >> 5), nb_ = ((I + 15) >> 4) * ((N + 31) >> 5);
        if (im) { slot[0] = (int)off; slot[1] = (int)(off + (long)nf * IMG_FRAG); }
        if (L) { L[nl].w_off = w_off; L[nl].I = I; L[nl].N = N; L[nl].fwd = (int)off; L[nl].bwd = (int)(off + (long)nf * IMG_FRAG); L[nl].first = items; }
        off += (long)(nf + nb_) * IMG_FRAG;
        items += nf + nb_;
        ++nl;
    };
    int dummy[2];
    int W = 0;
    for (int b = 0; b < nb; ++b) {
        const HpmnReadDesc &x = b == 0 ? d0 : d1;
        add(x.off_wq, x.D0, x.H, im ? im->wq[b] : dummy);
        add(x.off_map, x.H, x.H, im ? im->map[b] : dummy);
        for (int h = 0; h < x.hop; ++h) {
            add(x.off_att[h][0], 4 * x.H, A1, im ? im->att[b][h][0] : dummy);
            add(x.off_att[h][2], A1, A2, im ? im->att[b][h][1] : dummy);
        }
        W += x.H + x.D0;
    }
    add(d0.off_fc[0], W, F1, im ? im->fc[0] : dummy);
    add(d0.off_fc[2], F1, F2, im ? im->fc[1] : dummy);
    if (nl_out) *nl_out = nl;
    if (items_out) *items_out = items;
    return off;
}

// Y[r][n] = act(b[n] + sum_k X[r][k] B[k][n]) (ACCUM: added to what Y holds), r < R <= 16, n < N <= 512; X, Y in LDS, B as the
// image of Kd k-steps x N columns.  Chunk-outer, tile-inner: per 32-k chunk a wave requests the fragment planes of all ITS
// column tiles, reads its 8 k-steps of the row it feeds (two 16-byte LDS reads), splits them ONCE, and issues the products
// into one accumulator per tile.  Who does what:
//   * SHORT reductions (< 4 chunks): the waves split the column tiles (wave, wave + 4, ...: at most 8 each), every wave walks
//     all chunks and stores its tiles itself;
//   * LONG reductions (4 chunks or more: 4H -> 80, 200 -> 80, 200 -> W; at most 8 column tiles) split the K axis instead: the
//     operand split of the X rows -- ~45 VALU per chunk and lane -- is then done once per chunk in the WORKGROUP instead of
//     once per wave, and five column tiles no longer mean one wave with two of them.  Each wave leaves its 16 x N partial sums
//     in LDS and after one barrier all threads add the four partials in wave order (deterministic), bias, activation.
// (The first version walked tile by tile with a three-chunk ring of fragments: every tile began with an exposed L2 round trip
//  and repeated the split -- 4H -> 80 took 14 k cycles per hop at C3 where the fp32 form took 18 k, its transpose 10 k against 9 k.)
constexpr int BF_MAXT = 4;                 // column tiles per wave
constexpr int BF_WAVES = RT_BF / 64;
static_assert(BF_WAVES * BF_MAXT >= 32, "4H <= 512 columns over the waves' tiles");
constexpr int KS_PART_FLOATS = BF_WAVES * 16 * (16 * 8 + 4);
template <int ACT, bool ACCUM, int NS>
__device__ __forceinline__ void dense_bf(const float *X, int ldx, int R, int Kd, const uint4 *img, const float *bias, int N,
                                         float *Y, int ldy, float *part) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, n = lane & 15, kg = lane >> 4;
    const int nw = RT / 64;
    const int nch = (Kd + 31) >> 5, ntile = (N + 15) >> 4;
    const bool ks = nch >= 4 && ntile <= 8;                           // (workgroup-uniform)
    // K-split: G groups of four column tiles x S = nw / G shares of the chunks; otherwise the tiles go round the waves
    const int G = ks ? (ntile + BF_MAXT - 1) / BF_MAXT : 1, S = ks ? nw / G : 1;
    const int cstep = S, c0 = ks ? wave % S : 0;
    const int tstep = ks ? 1 : nw, t0 = ks ? (wave / S) * BF_MAXT : wave;
    const float *xr = X + (n < R ? n : R - 1) * ldx;
    rf4 acc[BF_MAXT];
    // (the bias rides in the accumulators of the wave that owns the tile -- K-split: the first share's partial -- requested
    //  here, in front of the chunks: asked for in the epilogue it was a round trip of its own per call)
    const bool addb = bias != nullptr && (!ks || c0 == 0);
#pragma unroll
    for (int j = 0; j < BF_MAXT; ++j) {
        const int col = 16 * (t0 + j * tstep) + n;
        const float bn = (addb && col < N) ? bias[col] : 0.f;
        acc[j] = rf4{bn, bn, bn, bn};
    }
    BFCLK(41);
    // fragment (t, c), plane s, of this lane: byte ((t nch + c) IMG_FRAG + 64 s + lane) 16 from the image -- 32-bit offsets from a
    // uniform base (one add per tile and chunk; the planes are immediate offsets): with 64-bit pointer arithmetic per load the
    // requests of a chunk were ~150 instructions of a wave that issues one per five cycles
    const char *ibase = reinterpret_cast<const char *>(img);
    const unsigned tstride = (unsigned)(tstep * nch * IMG_FRAG * 16);
    unsigned coff = (unsigned)(((t0 * nch + c0) * IMG_FRAG + lane) * 16);
    for (int c = c0; c < nch; c += cstep, coff += (unsigned)(cstep * IMG_FRAG * 16)) {
        uint4 bq[BF_MAXT][NS];
        unsigned o = coff;
#pragma unroll
        for (int j = 0; j < BF_MAXT; ++j, o += tstride)
            if (t0 + j * tstep < ntile) {                             // (wave-uniform)
#pragma unroll
                for (int s = 0; s < NS; ++s) bq[j][s] = *reinterpret_cast<const uint4 *>(ibase + o + s * 1024);
            }
        const int k0 = 32 * c + 8 * kg;
        const bool live = k0 < Kd;                                    // (Kd % 8 == 0: a lane's eight k-steps are all in or out)
        const int ko = live ? k0 : 0;
        const float4 a0 = *reinterpret_cast<const float4 *>(xr + ko), a1 = *reinterpret_cast<const float4 *>(xr + ko + 4);
        // (a lane beyond Kd reads the row's first eight values -- finite activations -- against fragment entries that are zero)
        float v[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
        rbf8 ap[NS];
        split_planes<NS>(v, ap);
        BFCLK(42);
#pragma unroll
        for (int j = 0; j < BF_MAXT; ++j)
            if (t0 + j * tstep < ntile) {
                rbf8 bp[NS];
#pragma unroll
                for (int s = 0; s < NS; ++s) bp[s] = *reinterpret_cast<const rbf8 *>(&bq[j][s]);
                // smallest terms first
                if constexpr (NS >= 3) {
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap[1], bp[1], acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap[0], bp[2], acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap[2], bp[0], acc[j], 0, 0, 0);
                }
                if constexpr (NS >= 2) {
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap[0], bp[1], acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap[1], bp[0], acc[j], 0, 0, 0);
                }
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap[0], bp[0], acc[j], 0, 0, 0);
            }
        BFCLK(43);
    }
    // C layout: lane (n, kg) holds rows 4 kg .. 4 kg + 3 of column 16 t + n
    if (ks) {
        const int NP = 16 * ntile + 4;
#pragma unroll
        for (int j = 0; j < BF_MAXT; ++j)
            if (t0 + j < ntile) {
#pragma unroll
                for (int i = 0; i < 4; ++i) part[(c0 * 16 + 4 * kg + i) * NP + 16 * (t0 + j) + n] = acc[j][i];
            }
        rsync();
        BFCLK(44);
        // thread = (column, row): N <= 128; the shares added in order
        const int col = threadIdx.x & 127;
        if (col < N) {
            for (int r = threadIdx.x >> 7; r < R; r += RT / 128) {
                float v = part[r * NP + col];
                for (int w = 1; w < S; ++w) v += part[(w * 16 + r) * NP + col];
                if (ACT == 1) v = fmaxf(v, 0.f);
                if (ACT == 2) v = elu(v);
                if (ACCUM) v += Y[r * ldy + col];
                Y[r * ldy + col] = v;
            }
        }
        BFCLK(45);
    } else {
#pragma unroll
        for (int j = 0; j < BF_MAXT; ++j) {
            const int col = 16 * (t0 + j * tstep) + n;
            if (t0 + j * tstep < ntile && col < N) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = 4 * kg + i;
                    if (row < R) {
                        float v = acc[j][i];
                        if (ACT == 1) v = fmaxf(v, 0.f);
                        if (ACT == 2) v = elu(v);
                        if (ACCUM) v += Y[row * ldy + col];
                        Y[row * ldy + col] = v;
                    }
                }
            }
        }
        BFCLK(46);
    }
}

// the layers of the kernels below: BF = the training launch with images (img = the layer's image in the direction of the call)
template <int ACT, bool BF>
__device__ __forceinline__ void dense_fwd_i(const float *X, int ldx, int R, int I, const float *W, const float *b, int N, float *Y,
                                            int ldy, const uint4 *img, float *part) {
    if constexpr (BF) dense_bf<ACT, false, HPMN_READ_NSF>(X, ldx, R, I, img, b, N, Y, ldy, part);
    else dense_fwd<ACT>(X, ldx, R, I, W, b, N, Y, ldy);
}
template <bool ACCUM, bool BF>
__device__ __forceinline__ void dense_bwd_x_i(const float *dY, int ldy, int R, int N, const float *W, int I, float *dX, int ldx,
                                              const uint4 *img, float *part) {
    if constexpr (BF) dense_bf<0, ACCUM, HPMN_READ_NSB>(dY, ldy, R, N, img, nullptr, I, dX, ldx, part);
    else dense_bwd_x<ACCUM>(dY, ldy, R, N, W, I, dX, ldx);
}

// ---- the tape: what the read-path WEIGHT gradients are made of.  Only the optimiser needs them, BPTT waits for d_memory /
// d_last alone -- so the training kernel does not form them (each workgroup used to write a 260 KB slab of per-tile
// products: a third of its time, 65 MB per launch at the reference batch).  It leaves the operand rows of every product
// in `workspace` instead, row-major per layer over the WHOLE batch, and read_wgrad_kernel (any stream behind it) forms
// gW = X^T dY with the batch rows as the reduction index.  Offsets in floats; hop h of the attention layers: + h * hop_stride.
struct TapeBranch {
    long inp, dt1;          // [B*K][4H], [B*K][A1]     first attention layer: input rows, gradient behind the relu
    long x1, dt2;           // [B*K][A1], [B*K][A2]     second
    long x2, dsc;           // [B*K][A2], [B*K]         third (single column)
    long hop_stride;
    long q, dqn;            // [hop][B][H] each         Hmap (shared by the hops): query entering hop h, gradient wrt the one leaving it
    long last, dq0;         // [B][D0], [B][H]          q0 = last Wq + bq: its input rows (a copy), the gradient wrt q0
};
struct Tape {
    TapeBranch br[2];
    long rep, dt1;          // [B][W], [B][F1]          fc1
    long h1, dt2;           // [B][F1], [B][F2]         fc2 (h1 behind the dropout)
    long h2, dlg;           // [B][F2], [B]             fc3
    long v, drep;           // [B][W], [B][W]           batch-norm affine: its input, the gradient wrt its output
    long total;
};

__host__ __device__ inline Tape tape_layout(const HpmnReadDesc &d0, const HpmnReadDesc &d1, int nb) {
    Tape t{};
    long off = 0;
    auto take = [&](long n) { const long r = off; off += (n + 3) / 4 * 4; return r; };
    const long B = d0.B;
    int W = 0;
    for (int b = 0; b < nb; ++b) {
        const HpmnReadDesc &d = b == 0 ? d0 : d1;
        TapeBranch &x = t.br[b];
        const long BK = B * d.K;
        const long h0 = off;
        x.inp = take(BK * 4 * d.H); x.dt1 = take(BK * A1);
        x.x1 = take(BK * A1); x.dt2 = take(BK * A2);
        x.x2 = take(BK * A2); x.dsc = take(BK);
        x.hop_stride = off - h0;
        off = h0 + x.hop_stride * d.hop;
        x.q = take(B * d.H * d.hop); x.dqn = take(B * d.H * d.hop);
        x.last = take(B * d.D0); x.dq0 = take(B * d.H);
        W += d.H + d.D0;
    }
    t.rep = take(B * W); t.dt1 = take(B * F1);
    t.h1 = take(B * F1); t.dt2 = take(B * F2);
    t.h2 = take(B * F2); t.dlg = take(B);
    t.v = take(B * W); t.drep = take(B * W);
    t.total = off;
    return t;
}

constexpr int WG_NCH = 16;          // row chunks per layer = slabs
constexpr int WG_MAXL = 48;         // layers: 2 branches x (4 hops x 4 + 1) + 3 head + 1 affine
struct WgLayer {
    long x_off, d_off;              // tape offsets
    int rows, I, N;
    int w_off, b_off;               // parameter offsets of the kernel / the bias (b_off < 0: none)
    int first;                      // first work item of the layer (items = tiles * WG_NCH)
    int kind;                       // 0 product; 1 batch-norm affine: g_gamma[i] = scale sum d v, g_beta[i] = sum d
    int ntn;                        // column tiles
};
struct WgArgs {
    const WgLayer *L;               // the layer table: written by the training launch into the workspace (it knows the
    int nl, items, n_params;        // descriptors; 2.4 KB of kernel arguments per launch stalled the host's queue instead)
    const float *tape;
    float *slabs;                   // [WG_NCH][n_params]
};
constexpr size_t WG_TABLE_FLOATS = (WG_MAXL * sizeof(WgLayer) + 15) / 16 * 4;

// the products of a launch, in slab order; returns the number of work items.  L == nullptr: count only.
__host__ __device__ inline int wg_layers(const HpmnReadDesc &d0, const HpmnReadDesc &d1, int nb, const Tape &t, WgLayer *L,
                                         int *nl_out) {
    int items = 0, nl = 0;
    auto add = [&](long x_off, long d_off, long rows, int I, int N, int w_off, int b_off, int kind) {
        const int ntn = (N + 31) / 32;
        if (L != nullptr) {
            WgLayer &l = L[nl];
            l.x_off = x_off; l.d_off = d_off; l.rows = (int)rows; l.I = I; l.N = N; l.w_off = w_off; l.b_off = b_off;
            l.kind = kind; l.first = items; l.ntn = ntn;
        }
        ++nl;
        const int tiles = kind == 1 ? (I + 63) / 64 : ((I + 31) / 32) * ntn;
        items += tiles * WG_NCH;
    };
    const long B = d0.B;
    int W = 0;
    for (int b = 0; b < nb; ++b) {
        const HpmnReadDesc &x = b == 0 ? d0 : d1;
        const TapeBranch &tb = t.br[b];
        const long BK = B * x.K;
        for (int h = 0; h < x.hop; ++h) {
            const long hs = (long)h * tb.hop_stride;
            const int *oa = x.off_att[h];
            add(tb.inp + hs, tb.dt1 + hs, BK, 4 * x.H, A1, oa[0], oa[1], 0);
            add(tb.x1 + hs, tb.dt2 + hs, BK, A1, A2, oa[2], oa[3], 0);
            add(tb.x2 + hs, tb.dsc + hs, BK, A2, 1, oa[4], oa[5], 0);
        }
        // Hmap is shared by the hops of a branch: one product over the rows of every hop (contiguous on the tape)
        add(tb.q, tb.dqn, B * x.hop, x.H, x.H, x.off_map, -1, 0);
        add(tb.last, tb.dq0, B, x.D0, x.H, x.off_wq, x.off_bq, 0);
        W += x.H + x.D0;
    }
    add(t.rep, t.dt1, B, W, F1, d0.off_fc[0], d0.off_fc[1], 0);
    add(t.h1, t.dt2, B, F1, F2, d0.off_fc[2], d0.off_fc[3], 0);
    add(t.h2, t.dlg, B, F2, 1, d0.off_fc[4], d0.off_fc[5], 0);
    add(t.v, t.drep, B, W, W, d0.off_gamma, d0.off_beta, 1);
    *nl_out = nl;
    return items;
}

// rows of an LDS array (stride ld) -> rows [row0, row0 + rows) of a tape array of width `width`
__device__ __forceinline__ void tape_store(float *g, long row0, const float *lds, int ld, int rows, int width) {
    float *dst = g + row0 * width;
    for (int o = threadIdx.x; o < rows * width; o += RT) {
        const int r = o / width, i = o - r * width;
        dst[o] = lds[r * ld + i];
    }
}

// ---- one or two branches ("User" alone in every reference configuration; "User" + "item" in dual mode, or "item"
//      alone: code/hpmn.py:452-462 / :307-317).  Each branch has its own memory, query row, attention stacks and covariance
//      regulariser; the head sees repre = concat over branches of [query_b, last_b], memory_loss = sum of the branches'.
struct ReadArgs {
    HpmnReadDesc d[2];          // per branch: K, H, D0, hop, off_wq/bq/map/att; head offsets, n_params, seed, B: d[0]'s
    const float *memory[2], *last[2];
    float *d_memory[2], *d_last[2];
    float *att_w0[2];
    int nb, W;                  // branches; head input width = sum_b (H_b + D0_b)
    int rs;                     // samples per workgroup: RS (training; latency per workgroup is what counts), 4 for large
                                // inference batches (4 K <= 32 rows fill the matrix tile: half the workgroups, the same work each)
    float *tape;                // training: the operand rows of the weight-gradient products
    WgLayer *table;             //           the layer table of read_wgrad_kernel (workgroup 0 writes it)
    Tape tp;                    //           and their layout (tape_layout, filled in by the host)
    ReadImg im;                 //           the dense layers' operand-fragment images (r5; base == NULL: none)
};

struct BranchSmem {
    // (row strides: the width + PADF)
    float *mem;      // [RS*K][H+]       memory slots of the tile
    float *last;     // [RS][D0+]
    float *q;        // [hop+1][RS][H+]  query before each hop and after the last
    float *x1;       // [hop][RS*K][A1P]
    float *x2;       // [hop][RS*K][A2P]
    float *sc;       // [hop][RS*K]      softmax scores
    float *cmean;    // [RS*K]        slot means (covariance regulariser)
    float *ccov;     // [RS*K*K]      off-diagonal covariance
    float *cnorm;    // [RS]          Frobenius norms
};
struct ReadSmem {
    // sizes depend on the branches' (K, H, D0, hop); carved from dynamic LDS
    BranchSmem br[2];
    float *inp;      // [RS*Kmax][4Hmax]  attention MLP input of the current hop (one branch at a time)
    float *dmem;     // [RS*Kmax][Hmax]   gradient wrt memory of the branch being differentiated (training)
    float *rep;      // [RS][W]           head input (bn output after the affine map)
    float *h1;       // [RS][F1]
    float *h2;       // [RS][F2]
    float *t1;       // scratch [RS*Kmax][A1] (d of x1) / [RS][F1]
    float *t2;       // scratch [RS*Kmax][A2] / [RS][F2]
    float *t3;       // scratch [RS*Kmax] / [RS]
    float *dq;       // [RS][Hmax]
    float *tq;       // [RS][Hmax]
    float *drep;     // [RS][W]
    float *zero;     // [max(H, D0)] zeros (bias of the bias-free products)
    float *mk1;      // [RS][F1]  dropout factor mask/keep_prob of the tile (training)
    float *mk2;      // [RS][F2]
    float *part;     // (training) the K-split layers' partial sums, one 16 x N block per wave (dense_bf_ks)
    float *wcol;     // the single-column layers' weights + bias: [branch][hop][WCOL] (A2 + 1 used), then the head's [F2 + 1]
    int rs;          // samples per workgroup (the "RS" of the comments above)
};

constexpr int WCOL = A2 + 4;                 // floats per attention logit layer in s.wcol: 40 weights, the bias, pad
constexpr int WCOL_HEAD = 2 * MAXHOP * WCOL; // offset of the head's logit layer (F2 weights + bias)
struct ReadDims { int Kmax, Hmax, Zmax, W; };
__host__ __device__ inline ReadDims read_dims(const HpmnReadDesc &d0, const HpmnReadDesc &d1, int nb) {
    ReadDims m;
    m.Kmax = d0.K; m.Hmax = d0.H; m.Zmax = d0.H > d0.D0 ? d0.H : d0.D0; m.W = d0.H + d0.D0;
    if (nb > 1) {
        m.Kmax = d1.K > m.Kmax ? d1.K : m.Kmax;
        m.Hmax = d1.H > m.Hmax ? d1.H : m.Hmax;
        const int z = d1.H > d1.D0 ? d1.H : d1.D0;
        m.Zmax = z > m.Zmax ? z : m.Zmax;
        m.W += d1.H + d1.D0;
    }
    return m;
}

// carve of the dynamic LDS (straight-line on purpose: pointer tables indexed at run time put the kernel's argument
// struct into scratch memory).  Returns the float count.
__host__ __device__ inline size_t carve_branch(BranchSmem &x, float *base, size_t off, const HpmnReadDesc &d, int RS) {
    auto take = [&](size_t n) { float *r = base ? base + off : nullptr; off += (n + 3) / 4 * 4; return r; };
    const size_t RK = (size_t)RS * d.K;
    x.mem = take(RK * (d.H + PADF));
    x.last = take((size_t)RS * (d.D0 + PADF));
    x.q = take((size_t)(d.hop + 1) * RS * (d.H + PADF));
    x.x1 = take((size_t)d.hop * RK * A1P);
    x.x2 = take((size_t)d.hop * RK * A2P);
    x.sc = take((size_t)d.hop * RK);
    x.cmean = take(RK);
    x.ccov = take(RK * d.K);
    x.cnorm = take(RS);
    return off;
}

// train: 0 inference, 1 training, 2 training on bf16 fragments (+ the K-split layers' partial sums)
__host__ __device__ inline size_t carve_all(ReadSmem &s, float *base, const HpmnReadDesc &d0, const HpmnReadDesc &d1, int nb,
                                            int train, int RS) {
    s.rs = RS;
    size_t off = 0;
    auto take = [&](size_t n) { float *r = base ? base + off : nullptr; off += (n + 3) / 4 * 4; return r; };
    const ReadDims m = read_dims(d0, d1, nb);
    const size_t RKm = (size_t)RS * m.Kmax;
    off = carve_branch(s.br[0], base, off, d0, RS);
    if (nb > 1) off = carve_branch(s.br[1], base, off, d1, RS);
    s.inp = take(RKm * (4 * m.Hmax + PADF));
    s.rep = take((size_t)RS * (m.W + PADF));
    s.h1 = take((size_t)RS * F1P);
    s.h2 = take((size_t)RS * F2P);
    s.t3 = take(RKm > 64 ? RKm + 64 : 128);        // (+ the per-sample scalars behind entries 32 / 48)
    s.zero = take((size_t)m.Zmax);
    s.mk1 = take((size_t)RS * F1P);
    s.mk2 = take((size_t)RS * F2P);
    s.wcol = take((size_t)WCOL_HEAD + F2 + 4);
    s.dmem = s.t1 = s.t2 = s.dq = s.tq = s.drep = s.part = nullptr;
    if (train) {
        if (train == 2) s.part = take((size_t)KS_PART_FLOATS);
        s.dmem = take(RKm * (m.Hmax + PADF));
        s.t1 = take(RKm * A1P > (size_t)RS * F1P ? RKm * A1P : (size_t)RS * F1P);
        s.t2 = take(RKm * A2P > (size_t)RS * F2P ? RKm * A2P : (size_t)RS * F2P);
        s.dq = take((size_t)RS * (m.Hmax + PADF));
        s.tq = take((size_t)RS * (m.Hmax + PADF));
        s.drep = take((size_t)RS * (m.W + PADF));
    }
    return off + 16;
}

inline size_t read_smem_floats(const HpmnReadDesc *d, int nb, int train, int rs) {
    ReadSmem s;
    return carve_all(s, nullptr, d[0], d[nb > 1 ? 1 : 0], nb, train, rs);
}

__device__ inline void carve(ReadSmem &s, float *base, const ReadArgs &a, int train) {
    carve_all(s, base, a.d[0], a.d[1], a.nb, train, a.rs);
    const ReadDims m = read_dims(a.d[0], a.d[1], a.nb);
    for (int o = threadIdx.x; o < m.Zmax; o += RT) s.zero[o] = 0.f;   // visible after the caller's first barrier
}

// dot products of rows with a per-sample vector: out[row] = <v[row / K], m[row]> over H, wave per row (a thread per row
// walked its H terms serially)
__device__ __forceinline__ void rows_dot(const float *v, int ldv, const float *m, int ldm, int RK, int K, int H, float *out) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int row = wave; row < RK; row += RT / 64) {
        float acc = 0.f;
        for (int i = lane; i < H; i += 64) acc = fmaf(v[(row / K) * ldv + i], m[row * ldm + i], acc);
#pragma unroll
        for (int x = 32; x >= 1; x >>= 1) acc += __shfl_xor(acc, x);
        if (lane == 0) out[row] = acc;
    }
}

// ---- forward of one branch of one tile: query, hops, covariance regulariser; leaves every activation in LDS and the
//      per-sample Frobenius norm in x.cnorm --------------------------------------------------------------------------
template <bool BF>
__device__ __forceinline__ void read_forward_branch(const HpmnReadDesc &d, const float *P, const ReadSmem &s, const BranchSmem &x, int R,
                                                    const ReadImg &im, int bi) {
    const uint4 *ib = im.base;
    const int K = d.K, H = d.H, D0 = d.D0, RK = R * K;
    const int HP = H + PADF, D0P = D0 + PADF, IP = 4 * H + PADF;
    const int tid = threadIdx.x;
    // q0 = last Wq + bq  (code/hpmn.py:173)
    dense_fwd_i<0, BF>(x.last, D0P, R, D0, P + d.off_wq, P + d.off_bq, H, x.q, HP, ib + im.wq[bi][0], s.part);
    rsync();
    RCLK(2);
    for (int hop = 0; hop < d.hop; ++hop) {
        const float *q = x.q + (size_t)hop * s.rs * HP;
        // inp = [q, m, q-m, q*m]  (code/hpmn.py:135-136)
        for (int o = tid; o < RK * H; o += RT) {
            const int row = o / H, i = o - row * H;
            const float qv = q[(row / K) * HP + i], mv = x.mem[row * HP + i];
            float *xi = s.inp + (size_t)row * IP;
            xi[i] = qv; xi[H + i] = mv; xi[2 * H + i] = qv - mv; xi[3 * H + i] = qv * mv;
        }
        rsync();
        RCLK(3);
        float *x1 = x.x1 + (size_t)hop * s.rs * K * A1P, *x2 = x.x2 + (size_t)hop * s.rs * K * A2P;
        float *sc = x.sc + (size_t)hop * s.rs * K;
        const int *oa = d.off_att[hop];
        dense_fwd_i<1, BF>(s.inp, IP, RK, 4 * H, P + oa[0], P + oa[1], A1, x1, A1P, ib + im.att[bi][hop][0][0], s.part);
        rsync();
        RCLK(4);
        dense_fwd_i<1, BF>(x1, A1P, RK, A1, P + oa[2], P + oa[3], A2, x2, A2P, ib + im.att[bi][hop][1][0], s.part);
        rsync();
        RCLK(5);
        const float *wc = s.wcol + (bi * MAXHOP + hop) * WCOL;
        dense_fwd_col<0>(x2, A2P, RK, A2, wc, wc + A2, sc, 1);
        rsync();
        RCLK(6);
        // softmax over the K slots of each sample (code/hpmn.py:141): a lane per (sample, slot) -- every lane walks its sample's
        // K scores (independent LDS reads) for the maximum and the denominator; the scores are rewritten behind a barrier
        // (r4: one thread per sample, three dependent passes over the slots)
        {
            float mine = 0.f, mx = -3.4e38f, den = 0.f;
            const int r = tid / K;
            if (tid < RK) {
                for (int k = 0; k < K; ++k) mx = fmaxf(mx, sc[r * K + k]);
                for (int k = 0; k < K; ++k) den += __expf(sc[r * K + k] - mx);
                mine = __expf(sc[tid] - mx) / den;
            }
            rsync();
            if (tid < RK) sc[tid] = mine;
        }
        rsync();
        RCLK(7);
        // q' = q Hmap + sum_k score_k m_k   (code/hpmn.py:143-144, 179)
        float *qn = x.q + (size_t)(hop + 1) * s.rs * HP;
        dense_fwd_i<0, BF>(q, HP, R, H, P + d.off_map, nullptr, H, qn, HP, ib + im.map[bi][0], s.part);        // q Hmap (no bias)
        rsync();
        RCLK(8);
        for (int o = tid; o < R * H; o += RT) {
            const int r = o / H, n = o - r * H;
            float acc = qn[r * HP + n];
            for (int k = 0; k < K; ++k) acc = fmaf(sc[r * K + k], x.mem[(r * K + k) * HP + n], acc);
            qn[r * HP + n] = acc;
        }
        rsync();
        RCLK(9);
    }
    // covariance regulariser (code/hpmn.py:161-170): per-sample Frobenius norm of the off-diagonal cov.
    // Parallel over (sample, slot[, slot]); means / covariances stay in LDS for the backward.
    for (int o = tid; o < RK; o += RT) {
        float a = 0.f;
        for (int i = 0; i < H; ++i) a += x.mem[o * HP + i];
        x.cmean[o] = a / H;
    }
    rsync();
    for (int o = tid; o < RK * K; o += RT) {
        const int rk = o / K, j = o - rk * K;       // rk = r*K + k
        const int r = rk / K, k = rk - r * K;
        float cv = 0.f;
        if (j != k) {
            const float *mk = x.mem + (size_t)rk * HP, *mj = x.mem + (size_t)(r * K + j) * HP;
            const float ak = x.cmean[rk], aj = x.cmean[r * K + j];
            for (int i = 0; i < H; ++i) cv = fmaf(mk[i] - ak, mj[i] - aj, cv);
            cv /= H;
        }
        x.ccov[o] = cv;
    }
    rsync();
    {   // (a wave per sample, lanes over the K x K entries; r4: one thread walked them)
        const int wv = tid >> 6, ln = tid & 63;
        for (int r = wv; r < R; r += RT / 64) {
            float ss = 0.f;
            for (int o = ln; o < K * K; o += 64) { const float cv = x.ccov[r * K * K + o]; ss = fmaf(cv, cv, ss); }
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) ss += __shfl_xor(ss, m);
            if (ln == 0) x.cnorm[r] = sqrtf(ss);
        }
    }
    rsync();
    RCLK(10);
}

// ---- forward of one tile: every branch, then the head.  Returns (in s.t3[0..R)) the logits and (in cov_sum[0..R)) the
//      samples' covariance losses summed over the branches ---------------------------------------------------------
template <bool BF>
__device__ __forceinline__ void read_forward_tile(const ReadArgs &a, const float *P, const ReadSmem &s, int R,
                                  const float *mask1, const float *mask2, float keep_prob, long b0, float *cov_sum) {
    const HpmnReadDesc &d0 = a.d[0];
    const int tid = threadIdx.x, W = a.W, WP = W + PADF;
    read_forward_branch<BF>(a.d[0], P, s, s.br[0], R, a.im, 0);
    if (a.nb > 1) read_forward_branch<BF>(a.d[1], P, s, s.br[1], R, a.im, 1);
    if (tid < R) cov_sum[tid] = s.br[0].cnorm[tid] + (a.nb > 1 ? s.br[1].cnorm[tid] : 0.f);
    // head (code/hpmn.py:190-199): repre = concat_b [q_b, last_b]; bn (inference affine); fc1 elu; dropout; fc2 elu;
    // dropout; fc3
    const float bn_scale = rsqrtf(1.f + 1e-3f);
    int off = 0;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        if (b >= a.nb) break;
        const int H = a.d[b].H, D0 = a.d[b].D0;
        const float *qf = s.br[b].q + (size_t)a.d[b].hop * s.rs * (H + PADF);
        for (int o = tid; o < R * (H + D0); o += RT) {
            const int r = o / (H + D0), i = o - r * (H + D0);
            const float v = i < H ? qf[r * (H + PADF) + i] : s.br[b].last[r * (D0 + PADF) + (i - H)];
            s.rep[r * WP + off + i] = v * (P[d0.off_gamma + off + i] * bn_scale) + P[d0.off_beta + off + i];
        }
        off += H + D0;
    }
    rsync();
    dense_fwd_i<2, BF>(s.rep, WP, R, W, P + d0.off_fc[0], P + d0.off_fc[1], F1, s.h1, F1P, a.im.base + a.im.fc[0][0], s.part);
    rsync();
    RCLK(12);
    const bool drop = mask1 != nullptr || mask2 != nullptr || (d0.dropout_seed != 0 && keep_prob < 1.f);
    if (drop) {
        // the tile's dropout factors, once, into LDS (the hash is 64-bit integer math: kept out of line and out
        // of the layer loops -- inlined at its four use sites it doubled the kernel's registers and spilled)
#pragma unroll 1
        for (int o = tid; o < R * F1; o += RT)
            s.mk1[(o / F1) * F1P + o % F1] = keep_factor(mask1, d0.dropout_seed, 1, b0 + o / F1, o % F1, F1, keep_prob) / keep_prob;
#pragma unroll 1
        for (int o = tid; o < R * F2; o += RT)
            s.mk2[(o / F2) * F2P + o % F2] = keep_factor(mask2, d0.dropout_seed, 2, b0 + o / F2, o % F2, F2, keep_prob) / keep_prob;
        rsync();
        for (int o = tid; o < R * F1P; o += RT) s.h1[o] *= s.mk1[o];       // (the pad columns: finite garbage, never read)
        rsync();
    }
    RCLK(13);
    dense_fwd_i<2, BF>(s.h1, F1P, R, F1, P + d0.off_fc[2], P + d0.off_fc[3], F2, s.h2, F2P, a.im.base + a.im.fc[1][0], s.part);
    rsync();
    RCLK(14);
    if (drop) {
        for (int o = tid; o < R * F2P; o += RT) s.h2[o] *= s.mk2[o];
        rsync();
    }
    dense_fwd_col<0>(s.h2, F2P, R, F2, s.wcol + WCOL_HEAD, s.wcol + WCOL_HEAD + F2, s.t3, 1);
    rsync();
    RCLK(15);
}

__device__ inline void load_tile_inputs(const ReadArgs &a, const float *P, const ReadSmem &s, long b0, int R) {
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        if (b >= a.nb) break;
        const int K = a.d[b].K, H = a.d[b].H, D0 = a.d[b].D0;
        for (int o = threadIdx.x; o < R * K * H; o += RT) s.br[b].mem[(o / H) * (H + PADF) + o % H] = a.memory[b][b0 * K * H + o];
        for (int o = threadIdx.x; o < R * D0; o += RT) s.br[b].last[(o / D0) * (D0 + PADF) + o % D0] = a.last[b][b0 * D0 + o];
        // the logit layers' weight column and bias (r5): every use of them from global memory was a round trip of its own on
        // the workgroup's only path (the forward's wave-per-row reduction: one for the column, one for the bias behind it)
        for (int o = threadIdx.x; o < a.d[b].hop * (A2 + 1); o += RT) {
            const int h = o / (A2 + 1), i = o - h * (A2 + 1);
            s.wcol[(b * MAXHOP + h) * WCOL + i] = P[i < A2 ? a.d[b].off_att[h][4] + i : a.d[b].off_att[h][5]];
        }
    }
    for (int o = threadIdx.x; o < F2 + 1; o += RT) s.wcol[WCOL_HEAD + o] = P[o < F2 ? a.d[0].off_fc[4] + o : a.d[0].off_fc[5]];
    // pad columns that elementwise loops sweep: defined values (mask products over whole padded rows)
    for (int o = threadIdx.x; o < s.rs * F1P; o += RT) { s.h1[o] = 0.f; s.mk1[o] = 0.f; }
    for (int o = threadIdx.x; o < s.rs * F2P; o += RT) { s.h2[o] = 0.f; s.mk2[o] = 0.f; }
    rsync();
}

__global__ __launch_bounds__(RT_BASE) void read_fwd_kernel(const ReadArgs a, const float *__restrict__ P, float *pred,
                                                      float *logit, float *mem_loss) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    ReadSmem s;
    carve(s, smem, a, 0);
    const long b0 = (long)blockIdx.x * a.rs;
    const int B = a.d[0].B;
    const int R = (B - b0) < a.rs ? (int)(B - b0) : a.rs;
    load_tile_inputs(a, P, s, b0, R);
    float *cov = s.t3 + 32;
    read_forward_tile<false>(a, P, s, R, nullptr, nullptr, 1.f, b0, cov);
    const int tid = threadIdx.x;
    if (tid < R) {
        const float lg = s.t3[tid];
        if (logit) logit[b0 + tid] = lg;
        pred[b0 + tid] = 1.f / (1.f + __expf(-lg));
        atomicAdd(mem_loss, cov[tid]);
    }
#pragma unroll
    for (int b = 0; b < 2; ++b)         // first hop (code/hpmn.py:182)
        if (b < a.nb && a.att_w0[b]) for (int o = tid; o < R * a.d[b].K; o += RT) a.att_w0[b][b0 * a.d[b].K + o] = s.br[b].sc[o];
}

// ---- backward of one branch: covariance regulariser, hops in reverse, q0; needs s.dq = gradient wrt the branch's final
//      query and s.drep[:, doff .. doff + D0) = the head's gradient wrt the branch's `last` row; writes d_memory / d_last of
//      the tile and the operand rows of the branch's weight-gradient products onto the tape ---------------------------
template <bool BF>
__device__ __forceinline__ void read_backward_branch(const HpmnReadDesc &d, const float *P, const ReadSmem &s, const BranchSmem &x, int R,
                                     float memory_reg, float *tape, const TapeBranch &tb, float *d_memory, float *d_last, long b0,
                                     int doff, int W, const ReadImg &im, int bi) {
    const uint4 *ib = im.base;
    const int K = d.K, H = d.H, D0 = d.D0, RK = R * K;
    const int HP = H + PADF, D0P = D0 + PADF, IP = 4 * H + PADF, WP = W + PADF;
    const int tid = threadIdx.x;
    // covariance regulariser backward into dmem (code/hpmn.py:161-170): loss_b = ||C_off||_F,
    // C = cc^T / H with c = m - mean_H(m):  d m = (2/(H*norm)) * (C_off c) projected off the mean
    // dL/dC_kj = C_kj / nrm (off-diagonal); dL/dc_k = (2/H) sum_j dC_kj c_j; the mean subtraction is a
    // projection that leaves it unchanged because sum_i c_j[i] = 0
    for (int o = tid; o < RK * H; o += RT) {
        const int rk = o / H, i = o - rk * H;
        const int r = rk / K;
        float acc = 0.f;
        if (memory_reg != 0.f) {
            const float nrm = x.cnorm[r];
            if (nrm > 0.f) {
                for (int j = 0; j < K; ++j)
                    acc = fmaf(x.ccov[rk * K + j], x.mem[(size_t)(r * K + j) * HP + i] - x.cmean[r * K + j], acc);
                acc *= memory_reg * 2.f / (H * nrm);
            }
        }
        s.dmem[rk * HP + i] = acc;
    }
    rsync();
    RCLK(25);

    // ---- hops backward (reverse order) ---------------------------------------------------------
    for (int hop = d.hop - 1; hop >= 0; --hop) {
        const float *q = x.q + (size_t)hop * s.rs * HP;           // query entering this hop
        float *x1 = x.x1 + (size_t)hop * s.rs * K * A1P, *x2 = x.x2 + (size_t)hop * s.rs * K * A2P;
        float *sc = x.sc + (size_t)hop * s.rs * K;
        const int *oa = d.off_att[hop];
        // q' = q Hmap + sum_k sc_k m_k : d Hmap += q^T dq' (tape);  d sc_k = <dq', m_k>;  d m_k += sc_k dq'
        float *th = tape + (long)hop * tb.hop_stride;
        tape_store(tape + tb.q + (long)hop * d.B * H, b0, q, HP, R, H);
        tape_store(tape + tb.dqn + (long)hop * d.B * H, b0, s.dq, HP, R, H);
        RCLK(39);
        float *dsc = s.t3;          // [RK]
        rows_dot(s.dq, HP, x.mem, HP, RK, K, H, dsc);
        RCLK(40);
        for (int o = tid; o < RK * H; o += RT) {
            const int row = o / H, i = o - row * H;
            s.dmem[row * HP + i] = fmaf(sc[row], s.dq[(row / K) * HP + i], s.dmem[row * HP + i]);
        }
        rsync();
        RCLK(26);
        // softmax backward: d s_k = sc_k (d sc_k - sum_j sc_j d sc_j)
        {
            float mine = 0.f;
            if (tid < RK) {
                const int r = tid / K;
                float dot = 0.f;
                for (int k = 0; k < K; ++k) dot = fmaf(sc[r * K + k], dsc[r * K + k], dot);
                mine = sc[tid] * (dsc[tid] - dot);
            }
            rsync();
            if (tid < RK) dsc[tid] = mine;
        }
        rsync();
        RCLK(27);
        // rebuild inp of this hop (the forward overwrote it hop by hop)
        for (int o = tid; o < RK * H; o += RT) {
            const int row = o / H, i = o - row * H;
            const float qv = q[(row / K) * HP + i], mv = x.mem[row * HP + i];
            float *xi = s.inp + (size_t)row * IP;
            xi[i] = qv; xi[H + i] = mv; xi[2 * H + i] = qv - mv; xi[3 * H + i] = qv * mv;
        }
        // fc3 (A2 -> 1, no activation)
        tape_store(th + tb.x2, b0 * K, x2, A2P, RK, A2);
        tape_store(th + tb.dsc, b0 * K, dsc, 1, RK, 1);
        // d x2 = d sc (x) w3, through the relu in the same pass (r4: an outer-product call that fetched the column from global
        // memory, a barrier, then the relu pass)
        {
            const float *wc = s.wcol + (bi * MAXHOP + hop) * WCOL;
            for (int o = tid; o < RK * A2; o += RT) {
                const int row = o / A2, i = o - row * A2;
                s.t2[row * A2P + i] = x2[row * A2P + i] > 0.f ? dsc[row] * wc[i] : 0.f;
            }
        }
        rsync();
        RCLK(28);
        RCLK(29);
        tape_store(th + tb.x1, b0 * K, x1, A1P, RK, A1);
        tape_store(th + tb.dt2, b0 * K, s.t2, A2P, RK, A2);
        RCLK(30);
        dense_bwd_x_i<false, BF>(s.t2, A2P, RK, A2, P + oa[2], A1, s.t1, A1P, ib + im.att[bi][hop][1][1], s.part);
        rsync();
        RCLK(31);
        for (int o = tid; o < RK * A1P; o += RT) s.t1[o] = x1[o] > 0.f ? s.t1[o] : 0.f;      // relu
        rsync();
        RCLK(32);
        tape_store(th + tb.inp, b0 * K, s.inp, IP, RK, 4 * H);
        tape_store(th + tb.dt1, b0 * K, s.t1, A1P, RK, A1);
        // d inp [RK, 4H] -> reuse s.inp AFTER the tape has its copy
        rsync();
        RCLK(33);
        dense_bwd_x_i<false, BF>(s.t1, A1P, RK, A1, P + oa[0], 4 * H, s.inp, IP, ib + im.att[bi][hop][0][1], s.part);
        rsync();
        RCLK(34);
        // inp = [q, m, q-m, q*m]:  dq_row = d0 + d2 + d3*m ; dm += d1 - d2 + d3*q
        // new dq (gradient wrt the query entering the hop) = dq' Hmap^T + sum_k dq_row
        float *dqn = s.tq;          // [R][H+]
        dense_bwd_x_i<false, BF>(s.dq, HP, R, H, P + d.off_map, H, dqn, HP, ib + im.map[bi][1], s.part);     // dq' Hmap^T
        rsync();
        RCLK(35);
        for (int o = tid; o < R * H; o += RT) {
            const int r = o / H, i = o - r * H;
            float acc = dqn[r * HP + i];
            for (int k = 0; k < K; ++k) {
                const float *di = s.inp + (size_t)(r * K + k) * IP;
                acc += di[i] + di[2 * H + i] + di[3 * H + i] * x.mem[(r * K + k) * HP + i];
            }
            s.dq[r * HP + i] = acc;            // (dq' is no longer needed: its Hmap^T product sits in dqn, its tape copy is out)
        }
        for (int o = tid; o < RK * H; o += RT) {
            const int row = o / H, i = o - row * H;
            const float *di = s.inp + (size_t)row * IP;
            s.dmem[row * HP + i] += di[H + i] - di[2 * H + i] + di[3 * H + i] * q[(row / K) * HP + i];
        }
        rsync();
        RCLK(36);
    }
    // q0 = last Wq + bq
    tape_store(tape + tb.last, b0, x.last, D0P, R, D0);
    tape_store(tape + tb.dq0, b0, s.dq, HP, R, H);
    dense_bwd_x_i<true, BF>(s.dq, HP, R, H, P + d.off_wq, D0, s.drep + doff, WP, ib + im.wq[bi][1], s.part);      // += dq Wq^T onto the head part
    rsync();
    RCLK(37);
    for (int o = tid; o < R * D0; o += RT) {
        const int r = o / D0, i = o - r * D0;
        d_last[b0 * D0 + o] = s.drep[r * WP + doff + i];
    }
    for (int o = tid; o < RK * H; o += RT) d_memory[b0 * K * H + o] = s.dmem[(o / H) * HP + o % H];
    rsync();
    RCLK(38);
}

// Training: forward + loss + backward of the tile in one launch.
//   loss = sum_b ll_b * inv_global_batch + memory_reg * sum_b cov_b        (code/hpmn.py:202-207)
// outputs: pred [B]; loss_out[0] += sum ll_b, loss_out[1] += sum cov_b (atomics); d_memory [B,K,H] and d_last [B,D0] of
// every branch; a.tape: the operand rows of the read-path weight-gradient products (read_wgrad_kernel forms them).
template <bool BF>
__global__ __launch_bounds__(BF ? RT_BF : RT_BASE) void read_fwd_bwd_kernel(const ReadArgs a, const float *__restrict__ P,
                                                          const int32_t *__restrict__ label,
                                                          const float *__restrict__ mask1,
                                                          const float *__restrict__ mask2, float keep_prob,
                                                          float inv_global_batch, float memory_reg, float *pred,
                                                          float *loss_out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    ReadSmem s;
    const HpmnReadDesc &d = a.d[0];
    const int W = a.W, WP = W + PADF;
    carve(s, smem, a, BF ? 2 : 1);
    const int tid = threadIdx.x;
    const long b0 = (long)blockIdx.x * a.rs;
    const int R = (d.B - b0) < a.rs ? (int)(d.B - b0) : a.rs;
    const Tape &tp = a.tp;
    float *tape = a.tape;
    if (blockIdx.x == 0 && tid == 0) {
        int nl;
        wg_layers(a.d[0], a.d[1], a.nb, tp, a.table, &nl);
    }
#ifdef READ_CLOCK
    if (tid < 48) rck_acc[tid] = 0;
    if (tid == 0) rck_last = clock64();
#endif
    // The images were written by the launch in front of this one: they sit in memory, not in the L2 of the XCD this workgroup
    // runs on, and every fragment is read exactly once per workgroup -- each dense call would begin with a round trip to memory
    // (~2.5 k cycles; measured: q Hmap, two chunks one behind the other, 5.3 k per call).  The workgroups of an XCD (linear id
    // % 8, observed dispatch rule, used for speed only) run in step and read the same images, so each of them TOUCHES one
    // share of the lines now -- four loads per thread, nothing waits for them until the kernel's last instruction -- and by
    // the time the first dense call asks, the XCD's L2 has (most of) them.
    unsigned warm = 0;
    if constexpr (BF) {
        const long lines = ((long)a.im.total * 16 + 127) / 128;
        const int slot = blockIdx.x >> 3, nslot = (gridDim.x + 7) >> 3;
        const long per = (lines + nslot - 1) / nslot, l0 = (long)slot * per + tid;
        const unsigned *wb = reinterpret_cast<const unsigned *>(a.im.base);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            long l = l0 + (long)e * RT;
            l = (e * RT + tid < per && l < lines) ? l : lines - 1;
            warm ^= wb[l * 32];
        }
    }
    load_tile_inputs(a, P, s, b0, R);
    RCLK(1);
    float *cov = s.t3 + 32;
    read_forward_tile<BF>(a, P, s, R, mask1, mask2, keep_prob, b0, cov);
    const bool drop = mask1 != nullptr || mask2 != nullptr || (d.dropout_seed != 0 && keep_prob < 1.f);

    // ---- loss and d logit -------------------------------------------------------------------
    float *dlg = s.t3 + 48;       // [R]
    if (tid < R) {
        const float lg = s.t3[tid];
        const float p = 1.f / (1.f + __expf(-lg));
        const float y = (float)label[b0 + tid];
        const float eps = 1e-7f;
        pred[b0 + tid] = p;
        atomicAdd(loss_out, -y * __logf(p + eps) - (1.f - y) * __logf(1.f - p + eps));
        atomicAdd(loss_out + 1, cov[tid]);
        // d ll / d p, then through the sigmoid
        const float dp = (-y / (p + eps) + (1.f - y) / (1.f - p + eps)) * inv_global_batch;
        dlg[tid] = dp * p * (1.f - p);
    }
    rsync();
    RCLK(16);

    // ---- head backward ------------------------------------------------------------------------
    // fc3: logit = h2 W3 + b3
    tape_store(tape + tp.h2, b0, s.h2, F2P, R, F2);
    tape_store(tape + tp.dlg, b0, dlg, 1, R, 1);
    RCLK(17);
    // d h2 (post-dropout) = d logit (x) w3, then through dropout2 and elu2 in the same pass: h2 = elu(a2) * mask/keep.
    // elu'(a) = a>0 ? 1 : elu(a)+1; recover from h2.
    for (int o = tid; o < R * F2; o += RT) {
        const int r = o / F2, i = o - r * F2;
        const int oo = r * F2P + i;
        float mk = 1.f;
        if (drop) mk = s.mk2[oo];
        const float hv = mk != 0.f ? s.h2[oo] / mk : 0.f;                    // elu(a2); irrelevant where mask==0
        s.t2[oo] = dlg[r] * s.wcol[WCOL_HEAD + i] * mk * (hv > 0.f ? 1.f : hv + 1.f);
    }
    rsync();
    RCLK(18);
    tape_store(tape + tp.h1, b0, s.h1, F1P, R, F1);
    tape_store(tape + tp.dt2, b0, s.t2, F2P, R, F2);
    RCLK(19);
    dense_bwd_x_i<false, BF>(s.t2, F2P, R, F2, P + d.off_fc[2], F1, s.t1, F1P, a.im.base + a.im.fc[1][1], s.part);     // d h1 (post-dropout)
    rsync();
    RCLK(20);
    for (int o = tid; o < R * F1; o += RT) {
        const int oo = (o / F1) * F1P + o % F1;
        float mk = 1.f;
        if (drop) mk = s.mk1[oo];
        const float hv = mk != 0.f ? s.h1[oo] / mk : 0.f;
        s.t1[oo] = s.t1[oo] * mk * (hv > 0.f ? 1.f : hv + 1.f);
    }
    rsync();
    RCLK(21);
    tape_store(tape + tp.rep, b0, s.rep, WP, R, W);
    tape_store(tape + tp.dt1, b0, s.t1, F1P, R, F1);
    RCLK(22);
    dense_bwd_x_i<false, BF>(s.t1, F1P, R, F1, P + d.off_fc[0], W, s.drep, WP, a.im.base + a.im.fc[0][1], s.part);   // d bn-output
    rsync();
    RCLK(23);
    // bn affine: rep = v*gamma*scale + beta  ->  d gamma, d beta (tape: v and the gradient wrt rep), d v;
    // v = concat_b [q_final_b, last_b]
    const float bn_scale = rsqrtf(1.f + 1e-3f);
    {
        int off = 0;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            if (b >= a.nb) break;
            const int H = a.d[b].H, D0 = a.d[b].D0;
            const float *qf = s.br[b].q + (size_t)a.d[b].hop * s.rs * (H + PADF);
            for (int o = tid; o < R * (H + D0); o += RT) {
                const int r = o / (H + D0), i = o - r * (H + D0);
                const float v = i < H ? qf[r * (H + PADF) + i] : s.br[b].last[r * (D0 + PADF) + (i - H)];
                tape[tp.v + (b0 + r) * W + off + i] = v;
                tape[tp.drep + (b0 + r) * W + off + i] = s.drep[r * WP + off + i];
            }
            off += H + D0;
        }
    }
    rsync();          // the loop below rescales s.drep in place
    for (int o = tid; o < R * W; o += RT) {
        const int r = o / W, i = o - r * W;
        s.drep[r * WP + i] *= P[d.off_gamma + i] * bn_scale;            // now: gradient wrt [q_final_b, last_b] of every branch
    }
    rsync();
    RCLK(24);
    // ---- the branches ---------------------------------------------------------------------------
    int off = 0;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        if (b >= a.nb) break;
        const int H = a.d[b].H, D0 = a.d[b].D0;
        for (int o = tid; o < R * H; o += RT) s.dq[(o / H) * (H + PADF) + o % H] = s.drep[(o / H) * WP + off + (o % H)];
        rsync();
        read_backward_branch<BF>(a.d[b], P, s, s.br[b], R, memory_reg, tape, tp.br[b], a.d_memory[b], a.d_last[b], b0, off + H, W,
                                 a.im, b);
        off += H + D0;
    }
    if constexpr (BF) {
        if (warm == 0x9e3779b9u && blockIdx.x == 0x7fffffffu) pred[0] = 0.f;      // (never: keeps the touch loads alive)
    }
#ifdef READ_CLOCK
    if (tid == 0 && blockIdx.x == 0) {
        unsigned long long tot = 0;
        for (int i = 0; i < 48; ++i) tot += rck_acc[i];
        printf("RCLK total %llu :", tot);
        for (int i = 0; i < 48; ++i) printf(" %d=%llu", i, rck_acc[i]);
        printf("\n");
    }
#endif
}


// ---- the weight gradients, from the tape ---------------------------------------------------------------------------------
// One product gW[I,N] = X^T dY (+ gb[N] = column sums of dY) per dense layer, X [rows, I] and dY [rows, N] row-major on the
// tape (or, for Wq, the `last` rows of the call).  A work item = (layer, 32x32 output tile, one of WG_NCH row chunks) on ONE
// wave: v_mfma_f32_32x32x2_f32 with the batch rows as the reduction index -- lane (c, p) feeds A = X[row + p][i0 + c],
// B = dY[row + p][n0 + c], both 128-byte coalesced reads.  The item writes its tile into slab `chunk` of the workspace
// (layout == parameter range, like the slabs the training kernel used to write per workgroup -- 16 now, not B/2), and
// read_reduce_kernel adds the slabs to the gradient buffer in a fixed order.
__global__ __launch_bounds__(256) void read_wgrad_kernel(const WgArgs a) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int item = blockIdx.x * 4 + wave;
    if (item >= a.items) return;
    int l = 0;
    while (l + 1 < a.nl && a.L[l + 1].first <= item) ++l;
    const WgLayer &L = a.L[l];
    const int local = item - L.first;
    const int chunk = local % WG_NCH, tile = local / WG_NCH;
    const int per = (L.rows + WG_NCH - 1) / WG_NCH;
    const int r0 = chunk * per, r1 = (r0 + per) < L.rows ? (r0 + per) : L.rows;
    float *slab = a.slabs + (long)chunk * a.n_params;
    const float *X = a.tape + L.x_off;
    const float *D = a.tape + L.d_off;
    if (L.kind == 1) {
        // affine: X = v, D = gradient wrt the output; 64 features per item
        const int i = tile * 64 + lane;
        if (i >= L.I) return;
        const float bn_scale = rsqrtf(1.f + 1e-3f);
        float gg = 0.f, gb = 0.f;
        for (int r = r0; r < r1; ++r) {
            const float dy = D[(long)r * L.I + i];
            gg = fmaf(dy, X[(long)r * L.I + i] * bn_scale, gg);
            gb += dy;
        }
        slab[L.w_off + i] = gg;
        slab[L.b_off + i] = gb;
        return;
    }
    const int c = lane & 31, p = lane >> 5;
    const int ti = tile / L.ntn, tn = tile - ti * L.ntn;
    const int i = ti * 32 + c, n = tn * 32 + c;
    const bool iv = i < L.I, nv = n < L.N;
    const float *xp = X + (iv ? i : 0), *dp = D + (nv ? n : 0);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float bsum = 0.f;
    constexpr int UN = 8;           // products per group: 16 loads in flight
    for (int r = r0; r < r1; r += 2 * UN) {
        float av[UN], bv[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int row = r + 2 * u + p;
            const bool in = row < r1;
            av[u] = (in && iv) ? xp[(long)row * L.I] : 0.f;
            bv[u] = (in && nv) ? dp[(long)row * L.N] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u], acc, 0, 0, 0);
            bsum += bv[u];
        }
    }
    // C/D layout: lane (c, p), reg r -> row (r&3) + 8*(r>>2) + 4*p (the i index), column c (the n index)
    if (nv) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ii = ti * 32 + (r & 3) + 8 * (r >> 2) + 4 * p;
            if (ii < L.I) slab[L.w_off + (long)ii * L.N + n] = acc[r];
        }
    }
    if (ti == 0 && L.b_off >= 0) {
        bsum += __shfl_xor(bsum, 32);
        if (p == 0 && nv) slab[L.b_off + n] = bsum;
    }
}


// grad[e] += sum over tiles of slabs[w][e].  A block owns 32 consecutive elements; its 8 groups of 32 lanes
// each sum every 8th slab (4 independent partial sums, 128-byte coalesced reads) and are combined through LDS
// in a fixed order (deterministic) -- same scheme as wgrad_reduce_kernel: one thread walking 250 slabs per
// element was latency-bound.
constexpr int RRED_G = 8;
// Optionally (loss_acc != nullptr) the step's loss scalars ride along: loss3 = {sum of log-losses, sum of covariance losses,
// cross_entropy = inv_global_batch * the first + memory_reg * the second}, and the two accumulators the training launch added
// into are cleared for the next step -- four framework launches of a few bytes each otherwise (23 us of a 0.34 ms C1 step).
__global__ __launch_bounds__(32 * RRED_G) void read_reduce_kernel(const float *__restrict__ slabs, int ntile, int n,
                                                                  float *grad, float *loss_acc, float inv_global_batch,
                                                                  float memory_reg, float *loss3) {
    __shared__ float part[RRED_G][32];
    const int c = threadIdx.x & 31, g = threadIdx.x >> 5;
    if (loss_acc != nullptr && blockIdx.x == 0 && threadIdx.x == 0) {
        const float ll = loss_acc[0], ml = loss_acc[1];
        loss3[0] = ll; loss3[1] = ml;
        loss3[2] = ll * inv_global_batch + memory_reg * ml;
        loss_acc[0] = 0.f; loss_acc[1] = 0.f;
    }
    const int e = blockIdx.x * 32 + c;
    const int ec = e < n ? e : n - 1;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int w = g;
    for (; w + 3 * RRED_G < ntile; w += 4 * RRED_G) {
        s0 += slabs[(long)w * n + ec];
        s1 += slabs[(long)(w + RRED_G) * n + ec];
        s2 += slabs[(long)(w + 2 * RRED_G) * n + ec];
        s3 += slabs[(long)(w + 3 * RRED_G) * n + ec];
    }
    for (; w < ntile; w += RRED_G) s0 += slabs[(long)w * n + ec];
    part[g][c] = (s0 + s1) + (s2 + s3);
    rsync();
    if (g != 0 || e >= n) return;
    float tot = part[0][c];
#pragma unroll
    for (int k = 1; k < RRED_G; ++k) tot += part[k][c];
    grad[e] += tot;
}

static size_t read_slab_floats(const HpmnReadDesc &d) { return ((size_t)WG_NCH * (size_t)d.n_params + 3) / 4 * 4; }

static bool read_desc_ok(const HpmnReadDesc &d) {
    return d.B >= 0 && d.K >= 1 && d.K <= MAXK && d.H >= 1 && d.D0 >= 1 && d.hop >= 1 && d.hop <= MAXHOP &&
           d.n_params > 0 && RS * d.K <= 48;
}

// workspace = [WG_NCH slabs of n_params | layer table | tape]  (the slabs first: their place must not move with the batch size -- positions
// of the parameter range that belong to no read-path variable are never written and rely on the caller's one-time zero fill)
// the shapes dense_bf serves: every k extent a multiple of 8 (a lane's eight k-steps are all in or all out), one 16-row tile
// (K <= 8 slots: every reference configuration), at most 4 x 8 column tiles (4H <= 512)
static bool read_bf_shapes_ok(const HpmnReadDesc *const *d, int nb) {
    int W = 0;
    for (int b = 0; b < nb; ++b) {
        if (d[b]->H % 8 != 0 || d[b]->H < 16 || d[b]->H > 128 || d[b]->D0 % 8 != 0 || d[b]->D0 > 512 || RS * d[b]->K > 16) return false;
        W += d[b]->H + d[b]->D0;
    }
    return W % 8 == 0;
}
// HPMN_READ_BF16=0: the r4 launch (fp32 matrix instructions, weights streamed column by column)
static bool read_bf_enabled() {
    static const int on = [] { const char *e = getenv("HPMN_READ_BF16"); return e ? atoi(e) : 1; }();
    return on != 0;
}

// r5: [... | tape | images]  (the images behind the tape: rebuilt by every training launch)
size_t read_workspace_bytes_n(const HpmnReadDesc *const *d, int nb) {
    const Tape t = tape_layout(*d[0], *d[nb > 1 ? 1 : 0], nb);
    const long img = img_layout(*d[0], *d[nb > 1 ? 1 : 0], nb, nullptr, nullptr, nullptr, nullptr);
    return ((size_t)t.total + read_slab_floats(*d[0]) + WG_TABLE_FLOATS) * sizeof(float) + (size_t)img * sizeof(uint4) + 16;
}
size_t read_workspace_bytes(const HpmnReadDesc &d) {
    const HpmnReadDesc *dp[1] = {&d};
    return read_workspace_bytes_n(dp, 1);
}

// nb = 1: d[0] alone (the "User"-only graph); nb = 2: d[0], d[1] in the order of the head's concat (user, item).
static int read_args(ReadArgs &a, const HpmnReadDesc *const *d, int nb, const float *const *memory,
                     const float *const *last, float *const *d_memory, float *const *d_last, float *const *att_w0) {
    if (nb < 1 || nb > 2) return HPMN_EINVAL;
    a = ReadArgs{};
    a.nb = nb;
    for (int b = 0; b < nb; ++b) {
        if (!d[b] || !read_desc_ok(*d[b]) || d[b]->B != d[0]->B) return d[b] ? HPMN_EUNSUPPORTED : HPMN_EINVAL;
        a.d[b] = *d[b];
        a.memory[b] = memory[b]; a.last[b] = last[b];
        a.d_memory[b] = d_memory ? d_memory[b] : nullptr;
        a.d_last[b] = d_last ? d_last[b] : nullptr;
        a.att_w0[b] = att_w0 ? att_w0[b] : nullptr;
        a.W += d[b]->H + d[b]->D0;
    }
    return HPMN_OK;
}

// the second half of the training call on its own: d_params += the read-path weight gradients, formed from the tape the
// training kernel left in `workspace` (two launches: the products per row chunk, the fixed-order sum of the chunks)
int read_param_grads_launch_n(const HpmnReadDesc *const *d, int nb, float *d_params, float *workspace, hipStream_t st,
                              float *loss_acc, float inv_global_batch, float memory_reg, float *loss3) {
    if (nb < 1 || nb > 2) return HPMN_EINVAL;
    for (int b = 0; b < nb; ++b)
        if (!d[b] || !read_desc_ok(*d[b]) || d[b]->B != d[0]->B) return d[b] ? HPMN_EUNSUPPORTED : HPMN_EINVAL;
    const HpmnReadDesc &d0 = *d[0];
    const Tape t = tape_layout(d0, *d[nb > 1 ? 1 : 0], nb);
    WgArgs a{};
    const int items = wg_layers(d0, *d[nb > 1 ? 1 : 0], nb, t, nullptr, &a.nl);
    a.items = items;
    a.n_params = d0.n_params;
    a.slabs = workspace;
    a.L = reinterpret_cast<const WgLayer *>(workspace + read_slab_floats(d0));
    a.tape = workspace + read_slab_floats(d0) + WG_TABLE_FLOATS;
    hipLaunchKernelGGL(read_wgrad_kernel, dim3((unsigned)((items + 3) / 4)), dim3(256), 0, st, a);
    int rc = check_launch();
    if (rc != HPMN_OK) return rc;
    hipLaunchKernelGGL(read_reduce_kernel, dim3((unsigned)((d0.n_params + 31) / 32)), dim3(32 * RRED_G), 0, st,
                       workspace, WG_NCH, d0.n_params, d_params, loss_acc, inv_global_batch, memory_reg, loss3);
    return check_launch();
}

int read_fwd_launch_n(const HpmnReadDesc *const *d, int nb, const float *P, const float *const *memory,
                      const float *const *last, float *pred, float *logit, float *const *att_w0, float *mem_loss,
                      hipStream_t st) {
    ReadArgs a;
    int rc = read_args(a, d, nb, memory, last, nullptr, nullptr, att_w0);
    if (rc != HPMN_OK) return rc;
    // Inference is a throughput problem once the batch covers the chip several times: four samples per workgroup fill the
    // 32-row matrix tile (4 K rows, K <= 8) -- half the workgroups, the same matrix instructions each.  HPMN_READ_INFER_RS.
    static const int rs_env = [] { const char *e = getenv("HPMN_READ_INFER_RS"); return e ? atoi(e) : 0; }();
    int kmax = a.d[0].K;
    if (nb > 1 && a.d[1].K > kmax) kmax = a.d[1].K;
    a.rs = RS;
    if (rs_env == 4 || (rs_env == 0 && a.d[0].B >= 1024)) a.rs = 4;
    if (a.rs * kmax > 32) a.rs = RS;
    size_t lds = read_smem_floats(a.d, nb, 0, a.rs) * sizeof(float);
    if (lds > 160 * 1024 && a.rs != RS) { a.rs = RS; lds = read_smem_floats(a.d, nb, 0, a.rs) * sizeof(float); }
    if (lds > 160 * 1024) return HPMN_EUNSUPPORTED;
    hipError_t e = hipFuncSetAttribute((const void *)read_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { set_last_hip_error((int)e); return HPMN_EHIP; }
    const unsigned grid = (unsigned)((a.d[0].B + a.rs - 1) / a.rs);
    hipLaunchKernelGGL(read_fwd_kernel, dim3(grid), dim3(RT_BASE), lds, st, a, P, pred, logit, mem_loss);
    return check_launch();
}

int read_fwd_bwd_launch_n(const HpmnReadDesc *const *d, int nb, const float *P, const float *const *memory,
                          const float *const *last, const int32_t *label, const float *mask1, const float *mask2,
                          float keep_prob, float inv_global_batch, float memory_reg, float *pred, float *loss_out,
                          float *const *d_memory, float *const *d_last, float *d_params, float *workspace, hipStream_t st) {
    ReadArgs a;
    int rc = read_args(a, d, nb, memory, last, d_memory, d_last, nullptr);
    if (rc != HPMN_OK) return rc;
    a.rs = RS;
    bool bf = read_bf_enabled() && read_bf_shapes_ok(d, nb);
    size_t lds = read_smem_floats(a.d, nb, bf ? 2 : 1, a.rs) * sizeof(float);
    if (bf && lds > 160 * 1024) { bf = false; lds = read_smem_floats(a.d, nb, 1, a.rs) * sizeof(float); }
    if (lds > 160 * 1024) return HPMN_EUNSUPPORTED;
    const void *fn = bf ? (const void *)read_fwd_bwd_kernel<true> : (const void *)read_fwd_bwd_kernel<false>;
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { set_last_hip_error((int)e); return HPMN_EHIP; }
    const unsigned grid = (unsigned)((a.d[0].B + RS - 1) / RS);
    a.table = reinterpret_cast<WgLayer *>(workspace + read_slab_floats(a.d[0]));
    a.tape = workspace + read_slab_floats(a.d[0]) + WG_TABLE_FLOATS;
    a.tp = tape_layout(a.d[0], a.d[nb > 1 ? 1 : 0], nb);
    a.im = ReadImg{};
    if (bf) {
        // the weights' operand-fragment images, behind the tape (16-byte aligned: every part before is a multiple of 4 floats)
        ImgArgs ia{};
        a.im.total = (int)img_layout(a.d[0], a.d[nb > 1 ? 1 : 0], nb, &a.im, ia.L, &ia.nl, &ia.items);
        ia.P = P;
        ia.img = reinterpret_cast<uint4 *>(a.tape + a.tp.total);
        a.im.base = ia.img;
        hipLaunchKernelGGL(read_wimg_kernel, dim3((unsigned)((ia.items + 3) / 4)), dim3(256), 0, st, ia);
        rc = check_launch();
        if (rc != HPMN_OK) return rc;
        hipLaunchKernelGGL(read_fwd_bwd_kernel<true>, dim3(grid), dim3(RT_BF), lds, st, a, P, label, mask1, mask2, keep_prob,
                           inv_global_batch, memory_reg, pred, loss_out);
    } else {
        hipLaunchKernelGGL(read_fwd_bwd_kernel<false>, dim3(grid), dim3(RT_BASE), lds, st, a, P, label, mask1, mask2, keep_prob,
                           inv_global_batch, memory_reg, pred, loss_out);
    }
    rc = check_launch();
    if (rc != HPMN_OK || d_params == nullptr) return rc;      // (NULL: the caller forms them later, read_param_grads_launch_n)
    return read_param_grads_launch_n(d, nb, d_params, workspace, st, nullptr, 0.f, 0.f, nullptr);
}

int read_fwd_launch(const HpmnReadDesc &d, const float *P, const float *memory, const float *last, float *pred,
                    float *logit, float *att_w0, float *mem_loss, hipStream_t st) {
    const HpmnReadDesc *dp[1] = {&d};
    return read_fwd_launch_n(dp, 1, P, &memory, &last, pred, logit, &att_w0, mem_loss, st);
}

int read_fwd_bwd_launch(const HpmnReadDesc &d, const float *P, const float *memory, const float *last,
                        const int32_t *label, const float *mask1, const float *mask2, float keep_prob,
                        float inv_global_batch, float memory_reg, float *pred, float *loss_out, float *d_memory,
                        float *d_last, float *d_params, float *workspace, hipStream_t st) {
    const HpmnReadDesc *dp[1] = {&d};
    return read_fwd_bwd_launch_n(dp, 1, P, &memory, &last, label, mask1, mask2, keep_prob, inv_global_batch, memory_reg,
                                 pred, loss_out, &d_memory, &d_last, d_params, workspace, st);
}

int read_reduce_launch(const HpmnReadDesc &d, float *d_params, float *workspace, hipStream_t st) {
    const HpmnReadDesc *dp[1] = {&d};
    return read_param_grads_launch_n(dp, 1, d_params, workspace, st, nullptr, 0.f, 0.f, nullptr);
}

}  // namespace hpmn
