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
//     are plain coalesced 128-byte row segments loaded straight from HBM into the MFMA
//     register layout -- no LDS staging, no transposes.  A workgroup owns whole sequences;
//     its waves split the OUTPUT tiles (wave w<DT: 32 input columns x all 3H gate columns;
//     the others: 32 state columns of h_prev x 2H and of r*h_prev x H) and keep their
//     3H/32 accumulator tiles (<=96 VGPRs) resident over all rows, finishing with one fp32
//     atomic add per accumulator element (dW must be zeroed by the caller).
// (2) gru_dx_kernel -- also fp32 MFMA: d_act tile staged in LDS as the A operand, the input
//     rows of the kernels as register-stationary B operands, coalesced stores.
#include "common.h"

namespace hpmn {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int WU = 8;  // 2-row MFMA steps per unrolled block (16 rows)

template <int HT, int DT>
__global__ __launch_bounds__(64 * (HT + DT)) void gru_wgrad_kernel(const HpmnGruWgrad a) {
    constexpr int H = 32 * HT;
    constexpr int NJ = 3 * HT;        // 32-column tiles of d_act
    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
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
    const int xcol = 32 * tile + c;
    const bool xcol_ok = xcol < D;

    const int b_begin = blockIdx.x * a.seq_per_wg;
    const int b_end = (b_begin + a.seq_per_wg) < a.B ? (b_begin + a.seq_per_wg) : a.B;
    for (int b = b_begin; b < b_end; ++b) {
        const float *dab = a.d_act + (long)b * T * 3 * H + c;
        const float *xb = a.x + (long)b * T * D + xcol;
        const float *hb = a.hs + (long)b * (T + 1) * H + 32 * tile + c;
        const float *gb = a.gates + (long)b * T * 3 * H + 32 * tile + c;   // r column of this tile
        for (int t0 = 0; t0 < T; t0 += 2 * WU) {
            float A1[WU], A2[WU], Bv[WU][NJ];
#pragma unroll
            for (int s = 0; s < WU; ++s) {
                const int t = t0 + 2 * s + rp;
                const bool ok = t < T;
#pragma unroll
                for (int j = 0; j < NJ; ++j) Bv[s][j] = ok ? dab[(long)t * 3 * H + 32 * j] : 0.f;
                if (role_x) {
                    A1[s] = (ok && xcol_ok) ? xb[(long)t * D] : 0.f;
                    A2[s] = 0.f;
                } else {
                    A1[s] = ok ? hb[(long)t * H] : 0.f;
                    A2[s] = ok ? gb[(long)t * 3 * H] : 0.f;          // r; becomes r*h_prev below
                }
            }
            if (role_x) {
#pragma unroll
                for (int s = 0; s < WU; ++s) {
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(A1[s], Bv[s][j], acc[j], 0, 0, 0);
                    if (tile == 0) {
#pragma unroll
                        for (int j = 0; j < NJ; ++j) bsum[j] += Bv[s][j];
                    }
                }
            } else {
#pragma unroll
                for (int s = 0; s < WU; ++s) A2[s] *= A1[s];
#pragma unroll
                for (int s = 0; s < WU; ++s) {
#pragma unroll
                    for (int j = 0; j < 2 * HT; ++j)
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(A1[s], Bv[s][j], acc[j], 0, 0, 0);
#pragma unroll
                    for (int j = 2 * HT; j < NJ; ++j)
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(A2[s], Bv[s][j], acc[j], 0, 0, 0);
                }
            }
        }
    }

    // ---- epilogue: C/D layout of 32x32x2: lane holds rows (r&3) + 8*(r>>2) + 4*(lane>>5), col lane&31
    const int row_base = role_x ? 32 * tile : D + 32 * tile;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const bool gate_tile = j < 2 * HT;
        float *dst = gate_tile ? a.d_wg : a.d_wc;
        const int ld = gate_tile ? 2 * H : H;
        const int col = gate_tile ? 32 * j + c : 32 * (j - 2 * HT) + c;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = (r & 3) + 8 * (r >> 2) + 4 * rp;
            if (!role_x || 32 * tile + i < D) atomicAdd(dst + (long)(row_base + i) * ld + col, acc[j][r]);
        }
    }
    if (role_x && tile == 0) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            if (j < 2 * HT) atomicAdd(a.d_bg + 32 * j + c, bsum[j]);
            else            atomicAdd(a.d_bc + 32 * (j - 2 * HT) + c, bsum[j]);
        }
    }
}

// ------------------------------------------------------------------------------------------
// dx[m][d] = sum_j d_act[m][j] * Wx[d][j],  Wx = [Wg[0:D] | Wc[0:D]]  ([D, 3H], row d contiguous
// per source).  MFMA with the d_act tile as the A operand read from LDS (row stride 3H+1 floats
// -> the 32 rows of a half-wave hit 32 different banks) and the weights as register-stationary B
// operands; the C/D layout then puts 32 consecutive d of one row in 32 consecutive lanes, so the
// stores are coalesced.  One wave = 32 rows; a workgroup = 2 waves = one contiguous 64-row tile.
constexpr int XR = 64;

template <int HT, int DT>
__global__ __launch_bounds__(128) void gru_dx_kernel(const HpmnGruWgrad a) {
    constexpr int H = 32 * HT;
    constexpr int N = 3 * H;
    constexpr int LD = N + 1;
    constexpr int KS = N / 2;                 // MFMA k-steps
    __shared__ float tile[XR * LD];
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int c = lane & 31, kp = lane >> 5;
    const int D = a.D;
    const long M = (long)a.B * a.T;
    const long ntiles = (M + XR - 1) / XR;

    // B operand (loaded once per persistent workgroup): lane (n = c -> d, k = kp): Wx[32*dt + c][2*ks + kp]
    float wb[DT][KS];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
        const int d = 32 * dt + c;
        const bool ok = d < D;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int j = 2 * ks + kp;
            float v = 0.f;
            if (ok) v = (j < 2 * H) ? a.wg[(long)d * 2 * H + j] : a.wc[(long)d * H + (j - 2 * H)];
            wb[dt][ks] = v;
        }
    }

    for (long ti = blockIdx.x; ti < ntiles; ti += gridDim.x) {
        const long m0 = ti * XR;
        __syncthreads();   // previous tile fully consumed
        // stage the contiguous [64 x 3H] d_act tile (coalesced float4), scalar LDS stores (odd stride)
        for (int i = tid; i < XR * (N / 4); i += 128) {
            const int r = i / (N / 4);
            const int q = i % (N / 4);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (m0 + r < M) v = *reinterpret_cast<const float4 *>(a.d_act + (m0 + r) * N + 4 * q);
            float *t = &tile[r * LD + 4 * q];
            t[0] = v.x; t[1] = v.y; t[2] = v.z; t[3] = v.w;
        }
        __syncthreads();

        f32x16 acc[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[dt][r] = 0.f;
        const float *arow = &tile[(wave * 32 + c) * LD + kp];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const float av = arow[2 * ks];
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
                acc[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, wb[dt][ks], acc[dt], 0, 0, 0);
        }
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const int d = 32 * dt + c;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long m = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * kp;
                if (d < D && m < M) a.d_x[m * D + d] = acc[dt][r];
            }
        }
    }
}

template <int HT, int DT>
static int launch_wgrad(const HpmnGruWgrad &a, hipStream_t st) {
    HpmnGruWgrad k = a;
    int spw = (512 + a.T - 1) / a.T;          // >= ~512 rows per workgroup amortise the atomic epilogue
    if (spw < 1) spw = 1;
    k.seq_per_wg = spw;
    const unsigned grid = (unsigned)((a.B + spw - 1) / spw);
    hipLaunchKernelGGL((gru_wgrad_kernel<HT, DT>), dim3(grid), dim3(64 * (HT + DT)), 0, st, k);
    return check_launch();
}

template <int HT, int DT>
static int launch_dx(const HpmnGruWgrad &a, hipStream_t st) {
    const long M = (long)a.B * a.T;
    long grid = (M + XR - 1) / XR;
    if (grid > 256 * 3) grid = 256 * 3;      // persistent: 3 workgroups per CU (LDS-limited), tile-stride loop
    hipLaunchKernelGGL((gru_dx_kernel<HT, DT>), dim3((unsigned)grid), dim3(128), 0, st, a);
    return check_launch();
}

int gru_wgrad_dispatch(const HpmnGruWgrad &a, hipStream_t st) {
    const int DT = (a.D + 31) / 32;
    int rc = HPMN_EUNSUPPORTED;
    if (a.H == 32 && DT == 1) rc = launch_wgrad<1, 1>(a, st);
    else if (a.H == 32 && DT == 2) rc = launch_wgrad<1, 2>(a, st);
    else if (a.H == 64 && DT == 1) rc = launch_wgrad<2, 1>(a, st);
    else if (a.H == 64 && DT == 2) rc = launch_wgrad<2, 2>(a, st);
    if (rc != HPMN_OK || a.d_x == nullptr) return rc;
    if (a.H == 32 && DT == 1) return launch_dx<1, 1>(a, st);
    if (a.H == 32 && DT == 2) return launch_dx<1, 2>(a, st);
    if (a.H == 64 && DT == 1) return launch_dx<2, 1>(a, st);
    if (a.H == 64 && DT == 2) return launch_dx<2, 2>(a, st);
    return HPMN_EUNSUPPORTED;
}

}  // namespace hpmn
