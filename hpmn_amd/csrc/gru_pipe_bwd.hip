// BPTT through build_memory, ALL layers in ONE launch: the reverse of gru_pipe_fwd.hip (same tiles, same operand
// layout, same hand-off protocol with the direction of the pipeline reversed: layer i+1 produces, a step at a
// time, the gradient wrt its input rows = the gradient wrt every period-th output of layer i).
//
// One reverse step of a 16-sequence tile (TF autodiff of code/util.py:95-109 through the while_loop):
//     dh    += d_y[t]                                    (firing steps)
//     dcp    = dh (1-u) (1-c^2)          dau = dh (h_prev - c) u (1-u)
//     d(rh)  = Wc[D:] dcp                                [64 x 64] x [64 x 16]   6 MFMAs per wave
//     dar    = d(rh) h_prev r (1-r)
//     dh     = dh u + d(rh) r + Wg[D:] [dar; dau]        [64 x 128] x [128 x 16] 12 MFMAs per wave
//     d_act[t] = (dar, dau, dcp)                         -> weight gradients (gru_wgrad.hip)
//     d_x[t] = W[:D] [dar; dau; dcp]                     [D x 192] x [192 x 16]  18 MFMAs per wave (layers >= 1:
//              this IS the d_y the layer below is waiting for; layer 0's goes through gru_dx to the scatter)
// Wave w owns hidden units (and input features) [16w, 16w+16): everything elementwise happens on the four
// values per column a lane's MFMA results land on; dcp/dau and dar cross waves through LDS operand images
// (double-buffered by step parity: the d_x product still reads step t's images while step t-1 writes).
#include "pipe_common.h"

// r4: the tile kernel's forward is the EVALUATION kernel now (ops.tiled_forward_inference); its training backward was
// measured slower than the per-sequence kernels at the reference batch (DESIGN_HISTORY.md 3.7) and is compiled only with
// -DHPMN_LEGACY_KERNELS; the default library answers hpmn_pipe_bwd with HPMN_EUNSUPPORTED.
#ifndef HPMN_LEGACY_KERNELS
namespace hpmn {
int pipe_bwd_launch(const PipeArgs &, int, hipStream_t) { return HPMN_EUNSUPPORTED; }
}  // namespace hpmn
#else
namespace hpmn {

constexpr int BWD_IMGS = 12;                      // (dcp, dau, dar) x (hi, lo) x 2 parities
constexpr int BWD_LDS = BWD_IMGS * IMG + 64;
constexpr int BPUB_DELAY = 3;

template <bool DX, bool DEP>
__device__ __forceinline__ void pipe_bwd_body(const PipeArgs &a, const PipeLayer &L, const int layer, const int tile,
                                              char *smem) {
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, g = lane >> 4, n = lane & 15;
    const int B = a.B, T = L.T, D = L.D;
    const bool live = tile * TS + n < B;
    const long b = live ? (long)tile * TS + n : (long)B - 1;
    const int u0 = 16 * w + 4 * g;
    const bool has_dx = DX && 16 * w < D;        // this wave's block of input features exists
    const int wr = img_wr_off(w, g, n);
    const int rd0 = img_rd_off(0, g, n), rd1 = img_rd_off(1, g, n);
    // image q of parity p at smem + (6 p + q) IMG: q = 0/1 dcp hi/lo, 2/3 dau, 4/5 dar

    // ---- stationary A operands: ROW k = 16w + m of the recurrent blocks (transposed products), and for d_x
    //      row d = 16w + m of the input blocks
    h8 Ac_hi[2], Ac_lo[2], Ag_hi[4], Ag_lo[4], Ax_hi[6], Ax_lo[6];
    {
        const float *wc_row = L.wc + (long)(D + 16 * w + n) * PH;
        const float *wg_row = L.wg + (long)(D + 16 * w + n) * 2 * PH;
        const int dr = 16 * w + n < D ? 16 * w + n : 0;
        const float *xc_row = L.wc + (long)dr * PH;
        const float *xg_row = L.wg + (long)dr * 2 * PH;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            float vc[8], vr[8], vu[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int unit = slot_unit(s, g, e);
                vc[e] = wc_row[unit];
                vr[e] = wg_row[unit];
                vu[e] = wg_row[PH + unit];
            }
            split8(vc, Ac_hi[s], Ac_lo[s]);
            split8(vr, Ag_hi[s], Ag_lo[s]);
            split8(vu, Ag_hi[2 + s], Ag_lo[2 + s]);
            if constexpr (DX) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int unit = slot_unit(s, g, e);
                    vr[e] = xg_row[unit];
                    vu[e] = xg_row[PH + unit];
                    vc[e] = xc_row[unit];
                }
                split8(vr, Ax_hi[s], Ax_lo[s]);
                split8(vu, Ax_hi[2 + s], Ax_lo[2 + s]);
                split8(vc, Ax_hi[4 + s], Ax_lo[4 + s]);
            }
        }
    }

    const int period = L.period;
    float *dump = a.dump + tid * 4;
    const float *gb = L.gates + b * (long)T * 3 * PH + u0;
    const float *hb = L.hs + b * (long)(T + 1) * PH + u0;
    float *dap = live ? L.d_act + (b * (long)T + (T - 1)) * 3 * PH + u0 : dump;
    const int da_adv = live ? 3 * PH : 0;
    float *dxp = dump;
    int dx_adv = 0;
    if constexpr (DX) {
        if (live && has_dx) {
            dxp = L.d_x + (b * (long)T + (T - 1)) * D + u0;
            dx_adv = D;
        }
    }
    const int ny = T / period;                       // rows of d_y
    const float *dyb = DEP ? L.d_y + b * (long)ny * PH + u0 : nullptr;

    unsigned *my_flag = a.sync + 2 + ((long)layer * a.ntiles + tile) * 4 + w;
    const unsigned *dep_flag = a.sync + 2 + ((long)(layer + 1) * a.ntiles + tile) * 4 + w;
    int avail = 0;
    // (hysteresis: a poll is a round trip through memory, ~1.5 us; once the consumer has caught up with the
    //  producer it would pay one per step -- more than the producer needs for a row.  When it has to wait it
    //  waits for WAIT_AHEAD rows beyond the one it needs, and then runs that many steps without polling.)
    auto wait_rows = [&](int need, int limit) {
        if constexpr (DEP) {
            if (need > avail) {
                const int want = need + WAIT_AHEAD < limit ? need + WAIT_AHEAD : limit;
                unsigned spins = 0;
                do {
                    const unsigned v = __hip_atomic_load(dep_flag, RLX_AGENT);
                    avail = __builtin_amdgcn_readfirstlane((int)v);
                    if (avail >= want) break;
                    __builtin_amdgcn_s_sleep(2);
                    if (++spins > PIPE_SPIN_LIMIT) {
                        if (lane == 0) __hip_atomic_store(a.sync + 1, 0x100u + (unsigned)layer, RLX_AGENT);
                        avail = 0x7fffffff;
                    }
                } while (avail < want);
            }
        }
    };

    for (int i = tid; i < BWD_IMGS * IMG / 16; i += 256) reinterpret_cast<uint4 *>(smem)[i] = uint4{0u, 0u, 0u, 0u};
    __syncthreads();

    // saved activations of step t: (r, u, c) and h_{t-1}; rows below 0 are clamped (loaded, never used)
    struct Saved { f4 r, u, c, hp; };
    auto load_saved = [&](int t) -> Saved {
        const int tc = t > 0 ? t : 0;
        Saved s;
        s.r = *reinterpret_cast<const f4 *>(gb + (long)tc * 3 * PH);
        s.u = *reinterpret_cast<const f4 *>(gb + (long)tc * 3 * PH + PH);
        s.c = *reinterpret_cast<const f4 *>(gb + (long)tc * 3 * PH + 2 * PH);
        s.hp = *reinterpret_cast<const f4 *>(hb + (long)tc * PH);
        return s;
    };
    // incoming output gradient rows, from the end: row j belongs to step (j+1)*period - 1; walking backwards
    // iteration k = T-1-t fires iff k % period == 0 and consumes row ny-1 - k/period.  Loaded two iterations
    // ahead, every iteration (a non-firing iteration re-reads a clamped row and masks it out: no control flow).
    int pf_phase = 0, pf_cnt = 0;        // phase of iteration (k + 2) within the period, rows handed out so far
    auto load_dy = [&](bool &is_row) -> f4 {
        if constexpr (DEP) {
            is_row = pf_phase == 0 && pf_cnt < ny;
            if (is_row) wait_rows(pf_cnt + 1, ny);
            const int j = ny - 1 - pf_cnt;
            const f4 v = load4_agent(dyb + (long)(j > 0 ? j : 0) * PH);
            pf_cnt += is_row ? 1 : 0;
            pf_phase = pf_phase + 1 == period ? 0 : pf_phase + 1;
            return v;
        } else {
            is_row = false;
            return f4{0.f, 0.f, 0.f, 0.f};
        }
    };

    f4 dh = *reinterpret_cast<const f4 *>(L.d_h_last + b * L.h_last_stride + u0);
    Saved s_cur = load_saved(T - 1), s_nxt = load_saved(T - 2);
    bool m2, m1;
    f4 dy2 = load_dy(m2);          // for iteration 0
    f4 dy1 = load_dy(m1);          // for iteration 1
    int done = 0, q1 = 0, q2 = 0, q3 = 0;

#define MF(A, Bv, C) C = __builtin_amdgcn_mfma_f32_16x16x32_f16(A, Bv, C, 0, 0, 0)
    for (int t = T - 1; t >= 0; --t) {
        char *img = smem + ((t & 1) ? 6 * IMG : 0);
        const Saved s_new = load_saved(t - 2);
        bool mn;
        const f4 dyn = load_dy(mn);
        const f4 r = s_cur.r, u = s_cur.u, c = s_cur.c, hp = s_cur.hp;
        f4 dcp, dau;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            dh[j] += m2 ? dy2[j] : 0.f;
            const float omu = 1.f - u[j];
            dcp[j] = dh[j] * omu * (1.f - c[j] * c[j]);
            dau[j] = dh[j] * (hp[j] - c[j]) * u[j] * omu;
        }
        {
            uint2 hi, lo;
            split4(dcp, hi, lo);
            *reinterpret_cast<uint2 *>(img + 0 * IMG + wr) = hi;
            *reinterpret_cast<uint2 *>(img + 1 * IMG + wr) = lo;
            split4(dau, hi, lo);
            *reinterpret_cast<uint2 *>(img + 2 * IMG + wr) = hi;
            *reinterpret_cast<uint2 *>(img + 3 * IMG + wr) = lo;
        }
        lds_barrier();                                                   // 1: dcp, dau of all 64 units in LDS
        const h8 c0h = *reinterpret_cast<const h8 *>(img + 0 * IMG + rd0), c1h = *reinterpret_cast<const h8 *>(img + 0 * IMG + rd1);
        const h8 c0l = *reinterpret_cast<const h8 *>(img + 1 * IMG + rd0), c1l = *reinterpret_cast<const h8 *>(img + 1 * IMG + rd1);
        const h8 u0h = *reinterpret_cast<const h8 *>(img + 2 * IMG + rd0), u1h = *reinterpret_cast<const h8 *>(img + 2 * IMG + rd1);
        const h8 u0l = *reinterpret_cast<const h8 *>(img + 3 * IMG + rd0), u1l = *reinterpret_cast<const h8 *>(img + 3 * IMG + rd1);
        f4 d1 = {0.f, 0.f, 0.f, 0.f}, d2 = d1, e1 = d1, e2 = d1;
        MF(Ac_hi[0], c0h, d1); MF(Ac_hi[1], c1h, d2);
        MF(Ac_hi[0], c0l, d1); MF(Ac_hi[1], c1l, d2);
        MF(Ac_lo[0], c0h, d1); MF(Ac_lo[1], c1h, d2);
        // the update-gate half of the second product does not need dar: under the d(rh) chain
        MF(Ag_hi[2], u0h, e1); MF(Ag_hi[3], u1h, e2);
        MF(Ag_hi[2], u0l, e1); MF(Ag_hi[3], u1l, e2);
        MF(Ag_lo[2], u0h, e1); MF(Ag_lo[3], u1h, e2);
        f4 drh, dar;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            drh[j] = d1[j] + d2[j];
            dar[j] = drh[j] * hp[j] * r[j] * (1.f - r[j]);
        }
        {
            uint2 hi, lo;
            split4(dar, hi, lo);
            *reinterpret_cast<uint2 *>(img + 4 * IMG + wr) = hi;
            *reinterpret_cast<uint2 *>(img + 5 * IMG + wr) = lo;
        }
        lds_barrier();                                                   // 2: dar of all 64 units in LDS
        const h8 r0h = *reinterpret_cast<const h8 *>(img + 4 * IMG + rd0), r1h = *reinterpret_cast<const h8 *>(img + 4 * IMG + rd1);
        const h8 r0l = *reinterpret_cast<const h8 *>(img + 5 * IMG + rd0), r1l = *reinterpret_cast<const h8 *>(img + 5 * IMG + rd1);
        MF(Ag_hi[0], r0h, e1); MF(Ag_hi[1], r1h, e2);
        MF(Ag_hi[0], r0l, e1); MF(Ag_hi[1], r1l, e2);
        MF(Ag_lo[0], r0h, e1); MF(Ag_lo[1], r1h, e2);
        // d_act (for the weight gradients)
        *reinterpret_cast<f4 *>(dap) = dar;
        *reinterpret_cast<f4 *>(dap + PH) = dau;
        *reinterpret_cast<f4 *>(dap + 2 * PH) = dcp;
        dap -= da_adv;
        if constexpr (DX) {
            // gradient wrt the input rows: matrix-pipe work off the chain (issued behind the chain's MFMAs)
            f4 x1 = {0.f, 0.f, 0.f, 0.f}, x2 = x1;
            if (has_dx) {
                MF(Ax_hi[0], r0h, x1); MF(Ax_hi[1], r1h, x2);
                MF(Ax_hi[2], u0h, x1); MF(Ax_hi[3], u1h, x2);
                MF(Ax_hi[4], c0h, x1); MF(Ax_hi[5], c1h, x2);
                MF(Ax_hi[0], r0l, x1); MF(Ax_hi[1], r1l, x2);
                MF(Ax_hi[2], u0l, x1); MF(Ax_hi[3], u1l, x2);
                MF(Ax_hi[4], c0l, x1); MF(Ax_hi[5], c1l, x2);
                MF(Ax_lo[0], r0h, x1); MF(Ax_lo[1], r1h, x2);
                MF(Ax_lo[2], u0h, x1); MF(Ax_lo[3], u1h, x2);
                MF(Ax_lo[4], c0h, x1); MF(Ax_lo[5], c1h, x2);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) dh[j] = fmaf(dh[j], u[j], fmaf(drh[j], r[j], e1[j] + e2[j]));
            store4_agent(dxp, x1 + x2);
            dxp -= dx_adv;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) dh[j] = fmaf(dh[j], u[j], fmaf(drh[j], r[j], e1[j] + e2[j]));
        }
        s_cur = s_nxt;
        s_nxt = s_new;
        dy2 = dy1; m2 = m1;
        dy1 = dyn; m1 = mn;
        done += 1;
        if constexpr (DX) {
            constexpr int PMIN = 3 + 2;
#ifndef HPMN_DBG_NOWAIT
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(BPUB_DELAY * PMIN) : "memory");
#endif
            if (lane == 0) __hip_atomic_store(my_flag, (unsigned)q3, RLX_AGENT);
            q3 = q2; q2 = q1; q1 = done;
        }
    }
#undef MF
    if constexpr (DX) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_store(my_flag, (unsigned)done, RLX_AGENT);
    }
}

__global__ __launch_bounds__(256, 1) void gru_pipe_bwd_kernel(const PipeArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int *role = reinterpret_cast<int *>(smem + BWD_IMGS * IMG);
    if (threadIdx.x == 0) role[0] = (int)atomicAdd(a.sync, 1u);
    __syncthreads();
    const int ticket = __builtin_amdgcn_readfirstlane(role[0]);
    __syncthreads();
    // the producer of a tile's hand-offs is the layer ABOVE: tickets run top layer first within a tile
    const int layer = a.K - 1 - ticket % a.K, tile = ticket / a.K;
    PipeLayer L = a.L[0];
#pragma unroll
    for (int i = 1; i < HPMN_MAX_LAYERS; ++i)
        if (i == layer) L = a.L[i];
    const bool top = layer == a.K - 1;
    if (layer == 0) {
        if (top) pipe_bwd_body<false, false>(a, L, layer, tile, smem);
        else     pipe_bwd_body<false, true>(a, L, layer, tile, smem);
    } else {
        if (top) pipe_bwd_body<true, false>(a, L, layer, tile, smem);
        else     pipe_bwd_body<true, true>(a, L, layer, tile, smem);
    }
}

size_t pipe_sync_bytes(int K, int ntiles);

int pipe_bwd_launch(const PipeArgs &a, int num_cus, hipStream_t st) {
    hipError_t e = hipMemsetAsync(a.sync, 0, pipe_sync_bytes(a.K, a.ntiles), st);
    if (e != hipSuccess) { set_last_hip_error((int)e); return HPMN_EHIP; }
    const int grid = a.K * a.ntiles;
    const size_t lds = grid <= num_cus ? (size_t)96 * 1024 : (size_t)BWD_LDS;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(gru_pipe_bwd_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL(gru_pipe_bwd_kernel, dim3(grid), dim3(256), lds, st, a);
    return check_launch();
}

}  // namespace hpmn

#endif  // HPMN_LEGACY_KERNELS
