// Whole-graph TRAINING entry points of the C ABI: build_memory with saved states and its BPTT as two
// calls, hpmn_scan_fwd_train / hpmn_scan_bwd, over one caller-provided workspace.  Everything the Python host used
// to orchestrate -- the per-layer launches, the weight-gradient reductions running on a helper stream underneath
// the serial chain (fork / join with events on the caller's stream), the row add of d_last, the embedding scatter
// -- happens inside the library, so a non-Python host can run a training step through the ABI alone.
#include <cstdlib>
#include <mutex>

#include "common.h"

namespace hpmn {
int embed_gather_launch(const void *ids, int64_t ids_stride, const float *emb, float *out, int64_t N, int32_t F,
                        int32_t E, int32_t mask_id0, hipStream_t st);
bool gru_fused_fwd_supported(int H, int D, int gather);
bool gru_fused_fwd_writes_last();
bool gru_pair_fwd_supported(int H, int D_lo, int gather);
size_t gru_proj_image_floats(int D);
bool gru_pair_bwd_supported(int H, int D_lo);
}  // namespace hpmn
#include "gru32_all.h"
namespace hpmn {
size_t gru_wgrad_workspace_bytes(int B, int T, int D, int H);
void gru_wgrad_reduce_aside(hipStream_t reduce_stream, hipEvent_t ev);
bool gru_scan_bwd_fuses_dx(int H, int B);
bool gru_candidate_elision(int H, int B);
bool gru_scan_bwd_dx_width_ok(int D);
bool gru_scan_bwd_fuses_scatter(int H, int B, int D, int F, int E);
bool gru_scan_bwd_scatter_inloop(int D);
int embed_grad_scatter_launch(const void *ids, const float *d_x, float *d_emb, int32_t B, int32_t T,
                              int32_t F, int32_t E, int32_t front_zero, int32_t mask_id0, int32_t t_lo, int32_t t_hi,
                              hipStream_t st, const float *d_last = nullptr, int32_t t_last = 0);

struct TrainCtx {
    int device = -1, cus = 256;
    hipStream_t side = nullptr, side2 = nullptr;         // (side2: the second of two small weight-gradient launches at the end)
    hipEvent_t fork = nullptr, join = nullptr, join2 = nullptr, scat = nullptr, red = nullptr;
    bool pending2 = false;
    hipEvent_t probe0 = nullptr, probe1 = nullptr;      // (timing events around layer 0's reverse-scan launch, on request)
    bool pending = false, probe = false, probed = false;
    hipEvent_t l0_start = nullptr;                       // (recorded in front of layer 0's reverse launch, on request)
    bool mark_l0 = false, l0_marked = false;
    HpmnScatterPlan plan = {};                           // (one-shot: the next hpmn_scan_bwd scatters through it, n > 0)
};

int embed_grad_segsum_launch(const HpmnScatterPlan &p, const float *d_x, float *d_emb, int32_t B, int32_t T, int32_t F,
                             int32_t E, int32_t front_zero, int32_t id_flags, const float *d_last, int32_t t_last,
                             hipStream_t st);

// Does layer i's saved `gates` tensor go WITHOUT the candidate (include/hpmn_hip.h, ABI v11)?  hpmn_scan_fwd_train and
// hpmn_scan_bwd both decide with this: the layer's forward runs on a fused kernel that honours HPMN_FWD_NO_CANDIDATE (single
// layer or pair) and its reverse scan on a chain + feeder kernel that honours HPMN_BWD_CANDIDATE_FROM_HS.
// (All layers or none: the two-layer reverse launch takes the switch as one template argument for both of its layers.)
static bool drops_candidate(const TrainCtx *c, const HpmnScanDesc *d, int) {
    if (!(2.0 * d->B <= 1.1 * 4 * c->cus) || !gru_fused_fwd_writes_last() || !gru_candidate_elision(d->H, d->B)) return false;
    for (int i = 0; i < d->K; ++i) {
        const int D = i == 0 ? d->F * d->E : d->H;
        if (!(gru_fused_fwd_supported(d->H, D, i == 0) && (i > 0 || 64 % d->E == 0))) return false;
    }
    return true;
}

// The whole-range scatter of a step: through the context's plan (deterministic segmented reduction) when one is set.
static int scatter_all(TrainCtx *c, const HpmnScanDesc *d, const void *ids, const float *d_x0, float *d_emb,
                       const float *d_last, hipStream_t st) {
    const int t_last = d->T + d->last_index;
    if (c->plan.n > 0) {
        const HpmnScatterPlan p = c->plan;
        c->plan = HpmnScatterPlan{};
        if (p.n != (int64_t)d->B * d->T * d->F) return HPMN_EINVAL;
        return embed_grad_segsum_launch(p, d_x0, d_emb, d->B, d->T, d->F, d->E, d->front_zero, d->mask_id0, d_last, t_last, st);
    }
    return embed_grad_scatter_launch(ids, d_x0, d_emb, d->B, d->T, d->F, d->E, d->front_zero, d->mask_id0, 0, d->T, st, d_last,
                                     t_last);
}

static size_t up256(size_t x) { return (x + 255) / 256 * 256; }

// HPMN_WGRAD_REDUCE_ASIDE=1 (H = 128 only; default 0): the weight gradients' slab reductions on the second helper stream.
// Built, parity-green (the 26 H = 128 tests), measured NEUTRAL at C4 (r5: 7.25 / 7.27 vs 7.17 / 7.33 ms per step): the chain of
// weight-gradient launches does run earlier -- layer 0's starts the moment its scan ends instead of 320 us later -- but what
// runs earlier runs beside layer 0's reverse scan, which stretches from 1.22 to 1.44 ms: who is resident decides (DESIGN_HISTORY 3.9).
static bool reduce_aside_enabled() {
    static const int on = [] { const char *e = getenv("HPMN_WGRAD_REDUCE_ASIDE"); return e ? atoi(e) : 0; }();
    return on != 0;
}

static bool lengths(const HpmnScanDesc &d, int32_t *len) {
    if (d.K < 1 || d.K > HPMN_MAX_LAYERS) return false;
    long t = (long)d.T + d.front_zero;
    for (int i = 0; i < d.K; ++i) {
        len[i] = (int32_t)t;
        if (d.periods[i] < 1 || t % d.periods[i] != 0) return false;
        t /= d.periods[i];
    }
    return true;
}

static bool layout(const HpmnScanDesc &d, HpmnTrainLayout &L) {
    int32_t len[HPMN_MAX_LAYERS];
    if (d.B < 1 || d.H < 1 || d.F < 1 || d.E < 1 || !lengths(d, len)) return false;
    const size_t B = d.B, H = d.H, D0 = (size_t)d.F * d.E;
    size_t off = 0, wmax = 0;
    auto take = [&](size_t floats) { const size_t o = off; off += up256(floats * sizeof(float)); return o; };
    L = HpmnTrainLayout{};
    L.K = d.K;
    L.x0 = take(B * len[0] * D0);
    for (int i = 0; i < d.K; ++i) {
        const size_t T = len[i], D = i == 0 ? D0 : H;
        L.T[i] = len[i];
        L.xp[i] = take(B * T * 3 * H);
        L.hs[i] = take(B * (T + 1) * H);
        L.gates[i] = take(B * T * 3 * H);
        L.y[i] = i + 1 < d.K ? take(B * (T / d.periods[i]) * H) : 0;
        L.d_act[i] = take(B * T * 3 * H);
        L.d_x[i] = take(B * T * D);
        const size_t w = gru_wgrad_workspace_bytes(d.B, len[i], (int)D, d.H);
        wmax = w > wmax ? w : wmax;
    }
    L.wgrad_ws = off;
    off += up256(wmax);
    if (d.H == 128 && reduce_aside_enabled()) {    // r5: a slab buffer PER LAYER (the reductions run on another stream)
        for (int i = 0; i < d.K; ++i) {
            L.wgrad_ws_layer[i] = off;
            off += up256(gru_wgrad_workspace_bytes(d.B, len[i], (int)(i == 0 ? D0 : H), d.H));
        }
    } else if (d.H != 32) {                        // a second slab buffer: two small weight-gradient launches side by side
        L.wgrad_ws_layer[1] = off;
        off += up256(wmax);
    }
    if (d.H == 32 && d.K <= AMAXK) {               // H = 32 (gru32_wgrad.hip): the slabs of all layers, one launch
        int Ds[HPMN_MAX_LAYERS];
        for (int i = 0; i < d.K; ++i) Ds[i] = (int)(i == 0 ? D0 : H);
        L.wgrad_ws_layer[0] = off;
        off += up256(gru32_wgrad_all_workspace_bytes(d.B, d.K, Ds));
    }
    L.pair_ws = off;
    off += up256((size_t)d.K * gru_proj_image_floats(64) * sizeof(float));
    L.total_bytes = off + 256;
    return true;
}

__global__ void add_rows_kernel(float *dst, long dst_stride, const float *src, int B, int D) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (long)B * D) {
        const long b = i / D;
        dst[b * dst_stride + (i - b * D)] += src[i];
    }
}
}  // namespace hpmn

using namespace hpmn;

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { set_last_hip_error((int)e_); return HPMN_EHIP; } } while (0)

extern "C" {

int hpmn_train_ctx_create(HpmnTrainCtx **out) {
    (void)hipGetLastError();
    if (!out) return HPMN_EINVAL;
    TrainCtx *c = new TrainCtx();
    if (hipGetDevice(&c->device) != hipSuccess) { delete c; return HPMN_ENODEVICE; }
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, c->device) == hipSuccess && n > 0) c->cus = n;
    // r6: the helper streams are created at the device's HIGHEST priority (HPMN_SIDE_PRIORITY=0: plain streams, rounds 2-5).
    // (1) The runtime keeps a pool of hardware queues PER PRIORITY and hands streams of one priority their queues round-robin:
    //     the step's other streams (the caller's, the auxiliary, the plan's, RCCL's) are plain ones, so the helper stream can
    //     no longer land on the CALLER's queue -- which is what happened to the data-parallel rows step on the default
    //     communicator (one rank on RCCL, C3, profiles/r06_timeline_rows_sg0.txt: the weight gradients of layers 1-6 sat IN
    //     FRONT of layer 0's reverse scan on one queue instead of underneath it: 3.23 ms per step against 2.86 with a second
    //     communicator, whose extra stream merely shifted the round-robin).
    // (2) When a CU frees up, pending workgroups of the helper stream (weight gradients, slab reductions) are dispatched in
    //     front of a reverse scan's next round (H = 128: 500 one-CU workgroups on 256 CUs run in two rounds; the 3 084 small
    //     workgroups of a slab reduction found a dozen free CUs: 454 / 793 us beside the scans against 5-20 alone).
    static const int side_prio = [] { const char *e = getenv("HPMN_SIDE_PRIORITY"); return e ? atoi(e) : 1; }();
    int prio_lo = 0, prio_hi = 0;
    if (side_prio) (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
    auto mk = [&](hipStream_t *st) {
        return side_prio ? hipStreamCreateWithPriority(st, hipStreamNonBlocking, prio_hi) : hipStreamCreateWithFlags(st, hipStreamNonBlocking);
    };
    if (mk(&c->side) != hipSuccess ||
        mk(&c->side2) != hipSuccess ||
        hipEventCreateWithFlags(&c->join2, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->join, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->scat, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->red, hipEventDisableTiming) != hipSuccess) {
        set_last_hip_error((int)hipGetLastError());
        delete c;
        return HPMN_EHIP;
    }

    *out = reinterpret_cast<HpmnTrainCtx *>(c);
    return HPMN_OK;
}

void hpmn_train_ctx_destroy(HpmnTrainCtx *ctx) {
    TrainCtx *c = reinterpret_cast<TrainCtx *>(ctx);
    if (!c) return;
    if (c->side) (void)hipStreamSynchronize(c->side);
    if (c->side2) (void)hipStreamSynchronize(c->side2);
    if (c->join2) (void)hipEventDestroy(c->join2);
    if (c->side2) (void)hipStreamDestroy(c->side2);
    if (c->l0_start) (void)hipEventDestroy(c->l0_start);
    if (c->probe0) (void)hipEventDestroy(c->probe0);
    if (c->probe1) (void)hipEventDestroy(c->probe1);
    if (c->fork) (void)hipEventDestroy(c->fork);
    if (c->join) (void)hipEventDestroy(c->join);
    if (c->scat) (void)hipEventDestroy(c->scat);
    if (c->red) (void)hipEventDestroy(c->red);
    if (c->side) (void)hipStreamDestroy(c->side);
    delete c;
}

int hpmn_scan_train_layout(const HpmnScanDesc *d, HpmnTrainLayout *out) {
    if (!d || !out) return HPMN_EINVAL;
    return layout(*d, *out) ? HPMN_OK : HPMN_EINVAL;
}

size_t hpmn_scan_train_workspace_bytes(const HpmnScanDesc *d) {
    HpmnTrainLayout L;
    if (!d || !layout(*d, L)) return 0;
    return L.total_bytes;
}

int hpmn_scan_fwd_train(HpmnTrainCtx *ctx, const HpmnScanDesc *d, const void *ids, const float *emb,
                        const float *const *wg, const float *const *bg, const float *const *wc,
                        const float *const *bc, float *memory, float *last, void *workspace, void *stream) {
    (void)hipGetLastError();
    TrainCtx *c = reinterpret_cast<TrainCtx *>(ctx);
    if (!c || !d || !ids || !emb || !wg || !bg || !wc || !bc || !memory || !workspace) return HPMN_EINVAL;
    if (d->B < 0 || d->T < 1 || d->F < 1 || d->E < 4 || d->H < 1 || d->V < 1) return HPMN_EINVAL;
    if (d->B == 0) return HPMN_OK;
    HpmnTrainLayout L;
    if (!layout(*d, L)) return HPMN_EINVAL;
    if (d->last_index >= 0 || -d->last_index > L.T[0]) return HPMN_EINVAL;
    const int D0 = d->F * d->E;
    if (!hpmn_gru_shape_supported(d->H, D0) || (d->K > 1 && !hpmn_gru_shape_supported(d->H, d->H))) return HPMN_EUNSUPPORTED;
    char *ws = reinterpret_cast<char *>((reinterpret_cast<size_t>(workspace) + 255) / 256 * 256);
    auto F = [&](size_t off) { return reinterpret_cast<float *>(ws + off); };
    if (gru32_all_enabled() && gru32_all_supported(d->H, D0, d->K, d->E)) {
        // H = 32: every layer and the gather in one launch (gru32_all.hip)
        All32Args a = {};
        gru32_all_fill(a, *d, L.T, ids, emb, wg, bg, wc, bc, memory, last);
        a.x0 = F(L.x0);
        for (int i = 0; i < d->K; ++i) {
            a.hs[i] = F(L.hs[i]); a.gates[i] = F(L.gates[i]);
            a.y[i] = i + 1 < d->K ? F(L.y[i]) : nullptr;
        }
        return gru32_fwd_all_launch(a, D0, true, (hipStream_t)stream);
    }
    // the fused layer spends a second wave per sequence on a SIMD that would otherwise idle: a win while 2 B waves
    // still find (about) a SIMD each (measured at C3: +3.7 % at B=500, -6.6 % at B=750 / 1000)
    const bool room = 2.0 * d->B <= 1.1 * 4 * c->cus;
    // Two layers per launch (hpmn_gru_pair_fwd): one 8-wave workgroup per CU and two sequences.  A pair launch holds
    // every register of the CUs it runs on, so NOTHING can share the chip with it: whatever is queued on another stream
    // either starves (the 5 us gradient clear took 486 us beside the pair of layers 0+1) or, worse, takes CUs away from
    // it (the early table-Adam pass beside the pair of layers 2+3: 593 instead of 165 us, the whole step 3.01 instead of
    // 2.89 ms).  HPMN_PAIR_FWD: 0 off; 1 pairs (0,1), (2,3), ...; 2 (default) layer 0 on its own -- its four-wave
    // workgroups leave half of every CU to the caller's early optimiser pass -- and pairs (1,2), (3,4), ...
    static const int pair_env = [] { const char *e = getenv("HPMN_PAIR_FWD"); return e ? atoi(e) : 2; }();
    const bool pair_room = pair_env > 0 && (d->B + 1) / 2 <= c->cus;
    // (a batch that leaves half of the CUs empty has room for the optimiser pass whatever the pairs hold: from layer 0)
    const bool half_chip = (d->B + 1) / 2 <= c->cus / 2;
    const int pair_first = (pair_env == 1 || half_chip) ? 0 : 1;
    bool last_done = false;
    const size_t img_stride = gru_proj_image_floats(64);
    auto image = [&](int i) { return F(L.pair_ws) + (size_t)i * img_stride; };
    bool images_built = false;
    auto fused_args = [&](int i) {
        const int D = i == 0 ? D0 : d->H;
        HpmnGruFusedFwd a = {};
        a.B = d->B; a.T = L.T[i]; a.D = D; a.H = d->H;
        if (i == 0) {
            if (last && gru_fused_fwd_writes_last()) {      // uinp[:, last_index, :] straight out of the launch
                a.last = last; a.last_t = L.T[0] + d->last_index;
                last_done = true;
            }
            a.ids = ids; a.emb = emb; a.Tids = d->T; a.F = d->F; a.E = d->E; a.front_zero = d->front_zero;
            a.mask_id0 = d->mask_id0; a.V = d->V; a.x_out = F(L.x0);
        } else {
            a.x = F(L.y[i - 1]);
        }
        a.wg = wg[i]; a.bg = bg[i]; a.wc = wc[i]; a.bc = bc[i];
        a.h_last = memory + (size_t)i * d->H; a.h_last_stride = (int64_t)d->K * d->H;
        a.y = i + 1 < d->K ? F(L.y[i]) : nullptr;
        a.period = d->periods[i]; a.hs = F(L.hs[i]); a.gates = F(L.gates[i]);
        a.flags = drops_candidate(c, d, i) ? HPMN_FWD_NO_CANDIDATE : 0;
        return a;
    };
    for (int i = 0; i < d->K; ++i) {
        const int D = i == 0 ? D0 : d->H;
        const bool fused = room && gru_fused_fwd_supported(d->H, D, i == 0) && (i > 0 || 64 % d->E == 0);
        float *y = i + 1 < d->K ? F(L.y[i]) : nullptr;
        int rc;
        // (a pair pays while the layers are long: the upper one ends 16-32 of ITS steps behind the lower, which for layers
        //  of 32 + 16 steps is more than the second launch costs -- measured 54 us paired vs 26 + 16 apart at C3)
        if (fused && pair_room && i >= pair_first && i + 1 < d->K && L.T[i + 1] >= 64 &&
            gru_pair_fwd_supported(d->H, D, i == 0)) {
            if (!images_built) {
                // every layer's projection weights as MFMA operand images, one launch in front of the first pair
                const float *iwg[HPMN_MAX_LAYERS], *ibg[HPMN_MAX_LAYERS], *iwc[HPMN_MAX_LAYERS], *ibc[HPMN_MAX_LAYERS];
                float *img[HPMN_MAX_LAYERS];
                int32_t iD[HPMN_MAX_LAYERS];
                int n = 0;
                for (int j = i; j < d->K; ++j) {
                    const int Dj = j == 0 ? D0 : d->H;
                    if (Dj != 64) continue;
                    iwg[n] = wg[j]; ibg[n] = bg[j]; iwc[n] = wc[j]; ibc[n] = bc[j]; img[n] = image(j); iD[n] = Dj;
                    ++n;
                }
                rc = hpmn_gru_proj_images(n, iwg, ibg, iwc, ibc, iD, img, stream);
                if (rc != HPMN_OK) return rc;
                images_built = true;
            }
            HpmnGruPairFwd p = {};
            p.lo = fused_args(i);
            p.up = fused_args(i + 1);
            p.img_lo = D == 64 ? image(i) : nullptr;
            p.img_up = image(i + 1);
            rc = hpmn_gru_pair_fwd(&p, stream);
            if (rc != HPMN_OK) return rc;
            ++i;
            continue;
        }
        if (fused) {
            HpmnGruFusedFwd a = fused_args(i);
            rc = hpmn_gru_fused_fwd(&a, stream);
        } else {
            HpmnInputProj p = {};
            p.B = d->B; p.T = L.T[i]; p.D = D; p.H = d->H;
            if (i == 0) {
                p.ids = ids; p.emb = emb; p.Tids = d->T; p.F = d->F; p.E = d->E; p.front_zero = d->front_zero;
                p.mask_id0 = d->mask_id0; p.V = d->V; p.x_out = F(L.x0);
            } else {
                p.x = F(L.y[i - 1]);
            }
            p.wg = wg[i]; p.bg = bg[i]; p.wc = wc[i]; p.bc = bc[i]; p.xp = F(L.xp[i]);
            rc = hpmn_gru_input_proj(&p, stream);
            if (rc != HPMN_OK) return rc;
            HpmnGruFwd a = {};
            a.B = d->B; a.T = L.T[i]; a.D = D; a.H = d->H;
            a.xp = F(L.xp[i]); a.wg = wg[i]; a.wc = wc[i];
            a.h_last = memory + (size_t)i * d->H; a.h_last_stride = (int64_t)d->K * d->H;
            a.y = y; a.period = d->periods[i]; a.hs = F(L.hs[i]); a.gates = F(L.gates[i]);
            rc = hpmn_gru_scan_fwd(&a, stream);
        }
        if (rc != HPMN_OK) return rc;
    }
    if (last && !last_done) {
        // uinp[:, last_index, :] (code/hpmn.py:439 / :292): a row of the materialised layer-0 input
        const size_t row = (size_t)(L.T[0] + d->last_index) * D0;
        HIPCHK(hipMemcpy2DAsync(last, (size_t)D0 * sizeof(float), F(L.x0) + row, (size_t)L.T[0] * D0 * sizeof(float),
                                (size_t)D0 * sizeof(float), d->B, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    }
    return HPMN_OK;
}

int hpmn_scan_bwd(HpmnTrainCtx *ctx, const HpmnScanDesc *d, const void *ids, const float *const *wg,
                  const float *const *wc, const float *d_memory, const float *d_last, float *const *d_wg,
                  float *const *d_bg, float *const *d_wc, float *const *d_bc, float *d_emb, void *workspace,
                  int32_t defer_join, void *stream) {
    (void)hipGetLastError();
    TrainCtx *c = reinterpret_cast<TrainCtx *>(ctx);
    if (!c || !d || !ids || !wg || !wc || !d_memory || !d_wg || !d_bg || !d_wc || !d_bc || !workspace) return HPMN_EINVAL;
    // A scatter plan (hpmn_train_set_scatter_plan) is consumed by the whole-range scatter ONLY: the paths that scatter on
    // their own -- fused into layer 0's reverse scan, the two time-cut forms -- are switched off while one is set (ADVICE r4:
    // they used to ignore it silently and leave plan.out_rows unwritten).  With a plan that writes compact rows the dense
    // table gradient is optional (ABI v12: hpmn_rows_sum_adam consumes the compact rows).
    const bool has_plan = c->plan.n > 0;
    if (!d_emb && !(has_plan && c->plan.out_rows)) return HPMN_EINVAL;
    if (d->B == 0) return HPMN_OK;
    HpmnTrainLayout L;
    if (!layout(*d, L)) return HPMN_EINVAL;
    if (64 % d->E != 0) return HPMN_EUNSUPPORTED;     // (the scatter's lane mapping)
    const int D0 = d->F * d->E;
    char *ws = reinterpret_cast<char *>((reinterpret_cast<size_t>(workspace) + 255) / 256 * 256);
    auto F = [&](size_t off) { return reinterpret_cast<float *>(ws + off); };
    hipStream_t st = (hipStream_t)stream;
    // Layer 0 in two time halves (HPMN_L0_SPLIT=1; built, parity-green, measured NEUTRAL at C3 -- 3.282 vs 3.272 ms/step:
    // the early half of the scan runs 340 instead of 262 us beside the late half's gradient kernels, which is what the
    // shorter tail gains -- default off).  Everything behind the layer-0
    // reverse scan -- its input gradient, the scatter into the table gradient, the dense table Adam of the caller --
    // is serial, and half of it does not need the early steps: the late half's input gradient, the d_last row add and
    // its scatter run on the helper stream underneath the scan of the early half, together with the late half's
    // weight gradient.  (The scatter is an atomic row add, so the two halves commute.)
    static const int split_env = [] { const char *e = getenv("HPMN_L0_SPLIT"); return e ? atoi(e) : 0; }();
    int cut = 0;
    if (split_env && !has_plan && L.T[0] >= 256 && !gru_scan_bwd_fuses_dx(d->H, d->B) && !(d->H == 128 && reduce_aside_enabled())) {
        // (not with the reductions on another stream: the two time halves of layer 0 share one slab buffer)
        const int p0 = d->periods[0], q = (p0 % 2 == 0) ? p0 : 2 * p0;
        cut = (L.T[0] / 2) / q * q;
        if (cut <= d->front_zero || cut >= L.T[0] + d->last_index) cut = 0;
    }
    if (cut == 0 && gru32_all_enabled() && gru32_all_supported(d->H, D0, d->K, d->E)) {
        // H = 32: every layer's reverse scan and input gradient in one launch (gru32_all.hip); the weight gradients follow
        // on the helper stream, the scatter on this one
        All32Args a = {};
        gru32_all_fill(a, *d, L.T, ids, nullptr, wg, nullptr, wc, nullptr, nullptr, nullptr);
        a.d_memory = d_memory;
        a.d_x0 = F(L.d_x[0]);
        for (int i = 0; i < d->K; ++i) { a.hs[i] = F(L.hs[i]); a.gates[i] = F(L.gates[i]); a.d_act[i] = F(L.d_act[i]); }
        int rc = gru32_bwd_all_launch(a, D0, st);
        if (rc != HPMN_OK) return rc;
        // every layer's weight and bias gradient: one launch + one reduction on the helper stream (gru32_wgrad.hip)
        HIPCHK(hipEventRecord(c->fork, st));
        HIPCHK(hipStreamWaitEvent(c->side, c->fork, 0));
        {
            int Ds[HPMN_MAX_LAYERS], Ts[HPMN_MAX_LAYERS];
            const float *xs[HPMN_MAX_LAYERS], *hss[HPMN_MAX_LAYERS], *gs[HPMN_MAX_LAYERS], *das[HPMN_MAX_LAYERS];
            for (int i = 0; i < d->K; ++i) {
                Ds[i] = i == 0 ? D0 : d->H; Ts[i] = L.T[i];
                xs[i] = i == 0 ? F(L.x0) : F(L.y[i - 1]);
                hss[i] = F(L.hs[i]); gs[i] = F(L.gates[i]); das[i] = F(L.d_act[i]);
            }
            rc = gru32_wgrad_all_launch(d->B, d->K, Ds, Ts, xs, hss, gs, das, d_wg, d_bg, d_wc, d_bc,
                                        F(L.wgrad_ws_layer[0]), c->side);
            if (rc != HPMN_OK) return rc;
        }
        c->pending = true;
        const bool in_scatter = d_last && d->T + d->last_index >= 0;
        if (d_last && !in_scatter) {
            const long n = (long)d->B * D0;
            hipLaunchKernelGGL(add_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st,
                               F(L.d_x[0]) + (size_t)(L.T[0] + d->last_index) * D0, (long)L.T[0] * D0, d_last, d->B, D0);
            rc = check_launch();
            if (rc != HPMN_OK) return rc;
        }
        rc = scatter_all(c, d, ids, F(L.d_x[0]), d_emb, in_scatter ? d_last : nullptr, st);
        if (rc != HPMN_OK) return rc;
        if (!defer_join) return hpmn_train_join(ctx, stream);
        return HPMN_OK;
    }
    // H = 128 (r5): every weight gradient's slab reduction goes to the second helper stream, out of a slab buffer of the
    // layer's own (layout()) -- the chain of weight-gradient launches on the first no longer waits for reductions that starve
    // beside the scans (C4's timeline: 467 and 811 us for reductions that take 20-50 us on a free chip).
    const bool aside = d->H == 128 && reduce_aside_enabled() && L.wgrad_ws_layer[0] != 0;
    struct AsideGuard { bool on; ~AsideGuard() { if (on) gru_wgrad_reduce_aside(nullptr, nullptr); } } aside_guard{aside};
    if (aside) gru_wgrad_reduce_aside(c->side2, c->red);
    bool scatter_pending = false, scatter_fused = false;
    int scat_cut = cut;                                  // first scan step whose scatter is already under way
    HpmnGruWgrad held[4], late[HPMN_MAX_LAYERS];
    int nheld = 0, nlate = 0;
    auto scan_args = [&](int i) {
        HpmnGruBwd a = {};
        a.B = d->B; a.T = L.T[i]; a.D = i == 0 ? D0 : d->H; a.H = d->H;
        a.wg = wg[i]; a.wc = wc[i]; a.hs = F(L.hs[i]); a.gates = F(L.gates[i]);
        a.d_h_last = d_memory + (size_t)i * d->H; a.d_h_last_stride = (int64_t)d->K * d->H;
        a.d_y = i + 1 < d->K ? F(L.d_x[i + 1]) : nullptr;
        a.period = d->periods[i];
        a.d_act = F(L.d_act[i]);
        a.flags = drops_candidate(c, d, i) ? HPMN_BWD_CANDIDATE_FROM_HS : 0;
        return a;
    };
    auto wgrad_args = [&](int i) {
        HpmnGruWgrad w = {};
        w.B = d->B; w.T = L.T[i]; w.D = i == 0 ? D0 : d->H; w.H = d->H;
        w.x = i == 0 ? F(L.x0) : F(L.y[i - 1]);
        w.hs = F(L.hs[i]); w.gates = F(L.gates[i]); w.d_act = F(L.d_act[i]);
        w.wg = wg[i]; w.wc = wc[i];
        w.d_wg = d_wg[i]; w.d_bg = d_bg[i]; w.d_wc = d_wc[i]; w.d_bc = d_bc[i];
        w.workspace = aside ? F(L.wgrad_ws_layer[i]) : F(L.wgrad_ws);
        // (layer 0's weight gradient runs beside the scatter and the table update, which are bandwidth kernels, and
        //  bounds the step's tail: it may fill the CUs -- 2.985 -> 2.905 ms/step at C3)
        w.whole_cu = i == 0 && d->H <= 64 ? 1 : 0;
        return w;
    };
    // two layers per launch (hpmn_gru_pair_bwd).  HPMN_PAIR_BWD: 0 off; 1 pairs (1,0), (3,2), ...; 2 (default) layer 0
    // alone and pairs (2,1), (4,3), ...
    // Measured at C3 (BPTT incl. weight gradients and scatter, us): off 1624-1631; 1: 1631-1669 -- the chain of reverse
    // scans drops from 1315 to 961, but a pair launch leaves the weight gradients no registers to run beside it, and
    // 630 us of them end up exposed behind the last scan; 2: 1575-1586 -- the upper pairs save 166 us and layer 0's
    // single-layer launch still hides the weight gradients (at 686 instead of 589 us).
    static const int pair_env = [] { const char *e = getenv("HPMN_PAIR_BWD"); return e ? atoi(e) : 2; }();
    // (as in the forward: a batch on half of the CUs leaves the weight gradients the other half -- pairs from layer 0)
    const int pair_mode = pair_env <= 0 ? 0 : ((pair_env == 1 || (d->B + 1) / 2 <= c->cus / 2) ? 1 : 2);
    const bool pair_ok = pair_mode > 0 && cut == 0 && (d->B + 1) / 2 <= c->cus && gru_scan_bwd_fuses_dx(d->H, d->B);
    for (int i = d->K - 1; i >= 0; --i) {
        const int D = i == 0 ? D0 : d->H;
        if (pair_ok && i >= 1 && (i - 1) % 2 == (pair_mode == 1 ? 0 : 1) && L.T[i] >= 8 &&
            gru_pair_bwd_supported(d->H, i - 1 == 0 ? D0 : d->H)) {
            HpmnGruPairBwd p = {};
            p.up = scan_args(i);
            p.lo = scan_args(i - 1);
            p.lo.d_x = F(L.d_x[i - 1]);
            int rc = hpmn_gru_pair_bwd(&p, stream);
            if (rc != HPMN_OK) return rc;
            for (int h = 0; h < nheld; ++h) late[nlate++] = held[h];      // (of a single-layer launch in front of this pair)
            nheld = 0;
            // A pair launch holds every register of its CUs (eight waves of up to 256): a weight-gradient kernel forked
            // beside it only gets the few CUs the launch leaves over (measured: 665 us instead of 150).  They are kept
            // until the pairs are through (the queue of them is flushed where a single-layer launch follows: that one
            // leaves half of every CU free) and may fill the CUs then.
            if ((d->B + 1) / 2 <= c->cus / 2) {
                // (... unless the batch leaves half of the CUs empty: then they start now, on those)
                HIPCHK(hipEventRecord(c->fork, st));
                HIPCHK(hipStreamWaitEvent(c->side, c->fork, 0));
                for (int h = 0; h < nlate; ++h) {
                    const int rc0 = hpmn_gru_param_grads(&late[h], c->side);
                    if (rc0 != HPMN_OK) return rc0;
                }
                nlate = 0;
                // (the pair's two weight gradients side by side: each is a few dozen workgroups; the lower layer's on a
                //  second stream with its own slab buffer)
                static const int two_env = [] { const char *e = getenv("HPMN_WGRAD_TWO_STREAMS"); return e ? atoi(e) : 1; }();
                const bool two = two_env && L.wgrad_ws_layer[1] != 0;
                if (two) HIPCHK(hipStreamWaitEvent(c->side2, c->fork, 0));
                for (int l = i; l >= i - 1; --l) {
                    HpmnGruWgrad w = wgrad_args(l);
                    w.whole_cu = 0;
                    const bool second = two && l == i - 1;
                    if (second) w.workspace = F(L.wgrad_ws_layer[1]);
                    const int rc0 = hpmn_gru_param_grads(&w, second ? c->side2 : c->side);
                    if (rc0 != HPMN_OK) return rc0;
                    if (second) c->pending2 = true;
                }
                c->pending = true;
            } else {
                late[nlate++] = wgrad_args(i);
                late[nlate++] = wgrad_args(i - 1);
            }
            --i;
            continue;
        }
        if (nlate > 0) {                       // a single-layer launch follows pairs: their weight gradients run beside it
            HIPCHK(hipEventRecord(c->fork, st));
            HIPCHK(hipStreamWaitEvent(c->side, c->fork, 0));
            for (int h = 0; h < nlate; ++h) {
                const int rc0 = hpmn_gru_param_grads(&late[h], c->side);
                if (rc0 != HPMN_OK) return rc0;
            }
            nlate = 0;
            c->pending = true;
        }
        HpmnGruBwd a = scan_args(i);
        HpmnGruWgrad w = wgrad_args(i);
        if (i == 0 && cut > 0) {
            const int T0 = L.T[0];
            if (nheld) {                                   // (weight gradients of short layers still waiting for a fork)
                HIPCHK(hipEventRecord(c->fork, st));
                HIPCHK(hipStreamWaitEvent(c->side, c->fork, 0));
                for (int h = 0; h < nheld; ++h) {
                    const int rc0 = hpmn_gru_param_grads(&held[h], c->side);
                    if (rc0 != HPMN_OK) return rc0;
                }
                nheld = 0;
                c->pending = true;
            }
            a.t_begin = cut; a.t_end = T0; a.dh_carry = F(L.xp[0]);      // (xp is free once the forward is done)
            int rc = hpmn_gru_scan_bwd(&a, stream);
            if (rc != HPMN_OK) return rc;
            HIPCHK(hipEventRecord(c->fork, st));
            HIPCHK(hipStreamWaitEvent(c->side, c->fork, 0));
            rc = hpmn_gru_input_grad(F(L.d_act[0]), wg[0], wc[0], F(L.d_x[0]), d->B, T0, D, d->H, cut, T0 - cut, c->side);
            if (rc != HPMN_OK) return rc;
            if (d_last) {
                const long n = (long)d->B * D0;
                hipLaunchKernelGGL(add_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->side,
                                   F(L.d_x[0]) + (size_t)(T0 + d->last_index) * D0, (long)T0 * D0, d_last, d->B, D0);
                rc = check_launch();
                if (rc != HPMN_OK) return rc;
            }
            rc = embed_grad_scatter_launch(ids, F(L.d_x[0]), d_emb, d->B, d->T, d->F, d->E, d->front_zero, d->mask_id0,
                                           cut - d->front_zero, d->T, c->side);
            if (rc != HPMN_OK) return rc;
            HIPCHK(hipEventRecord(c->scat, c->side));
            scatter_pending = true;
            w.t_begin = cut; w.t_len = T0 - cut; w.whole_cu = 0;      // (beside the early half's scan)
            rc = hpmn_gru_param_grads(&w, c->side);
            if (rc != HPMN_OK) return rc;
            c->pending = true;
            a.t_begin = 0; a.t_end = cut;
            rc = hpmn_gru_scan_bwd(&a, stream);
            if (rc != HPMN_OK) return rc;
            HIPCHK(hipEventRecord(c->fork, st));
            HIPCHK(hipStreamWaitEvent(c->side, c->fork, 0));
            w.t_begin = 0; w.t_len = cut; w.whole_cu = d->H <= 64 ? 1 : 0;
            rc = hpmn_gru_param_grads(&w, c->side);
            if (rc != HPMN_OK) return rc;
            rc = hpmn_gru_input_grad(F(L.d_act[0]), wg[0], wc[0], F(L.d_x[0]), d->B, T0, D, d->H, 0, cut, stream);
            if (rc != HPMN_OK) return rc;
            continue;
        }
        if (i == 0 && c->mark_l0 && c->l0_start != nullptr) {
            HIPCHK(hipEventRecord(c->l0_start, st));
            c->l0_marked = true;
        }
        const bool probing = i == 0 && c->probe && c->probe0 != nullptr;
        if (probing) HIPCHK(hipEventRecord(c->probe0, st));
        const bool fused_dx = gru_scan_bwd_fuses_dx(d->H, d->B) && gru_scan_bwd_dx_width_ok(D);
        if (fused_dx) a.d_x = F(L.d_x[i]);       // the input gradient comes out of the scan launch itself
        if (i == 0 && fused_dx && !has_plan && !(d->mask_id0 & HPMN_ID_I64) &&
            gru_scan_bwd_fuses_scatter(d->H, d->B, D, d->F, d->E)) {
            // ... and goes straight into the table gradient: no d_x buffer, no scatter launch behind layer 0
            if (gru_scan_bwd_scatter_inloop(D)) a.flags |= HPMN_BWD_SCATTER_INLOOP;     // (d_x: the kernel's scratch)
            else a.d_x = nullptr;
            a.scatter_ids = ids; a.d_emb = d_emb; a.Tids = d->T; a.F = d->F; a.E = d->E;
            a.front_zero = d->front_zero; a.mask_id0 = d->mask_id0; a.last_t = L.T[0] + d->last_index;
            a.d_last = d->T + d->last_index >= 0 ? d_last : nullptr;
            scatter_fused = true;
        }
        // Layer 0's launch in two time halves (HPMN_L0_CUT, with the fused input gradient): the late half's weight gradient
        // then runs beside the early half's scan instead of waiting for the whole scan -- the weight gradient of layer 0 is
        // the step's tail (326 us at C3), half of it moves under the chain.
        static const int l0_cut_env = [] { const char *e = getenv("HPMN_L0_CUT"); return e ? atoi(e) : 0; }();
        int cut0 = 0;
        if (i == 0 && l0_cut_env && fused_dx && !scatter_fused && L.T[0] >= 512) {
            const int p0 = d->periods[0], q = (p0 % 2 == 0) ? p0 : 2 * p0;
            cut0 = (L.T[0] / 2) / q * q;
        }
        int rc;
        if (cut0 > 0) {
            a.t_begin = cut0; a.t_end = L.T[0]; a.dh_carry = F(L.xp[0]);      // (xp is free once the forward is done)
            rc = hpmn_gru_scan_bwd(&a, stream);
            if (rc != HPMN_OK) return rc;
            HpmnGruWgrad wa = w;
            wa.t_begin = cut0; wa.t_len = L.T[0] - cut0; wa.whole_cu = 0;
            held[nheld++] = wa;
            HIPCHK(hipEventRecord(c->fork, st));
            HIPCHK(hipStreamWaitEvent(c->side, c->fork, 0));
            if (l0_cut_env >= 2 && !has_plan && cut0 > d->front_zero && d->T + d->last_index >= cut0 - d->front_zero) {
                // ... and the late half's scatter (with the read path's d_last row, which lies in it): the step's other tail
                rc = embed_grad_scatter_launch(ids, F(L.d_x[0]), d_emb, d->B, d->T, d->F, d->E, d->front_zero, d->mask_id0,
                                               cut0 - d->front_zero, d->T, c->side, d_last, d->T + d->last_index);
                if (rc != HPMN_OK) return rc;
                HIPCHK(hipEventRecord(c->scat, c->side));
                scatter_pending = true;
                scat_cut = cut0;
            }
            for (int h = 0; h < nheld; ++h) {
                rc = hpmn_gru_param_grads(&held[h], c->side);
                if (rc != HPMN_OK) return rc;
            }
            nheld = 0;
            c->pending = true;
            a.t_begin = 0; a.t_end = cut0;
            w.t_begin = 0; w.t_len = cut0;
        }
        rc = hpmn_gru_scan_bwd(&a, stream);
        if (rc != HPMN_OK) return rc;
        if (probing) { HIPCHK(hipEventRecord(c->probe1, st)); c->probed = true; }
        // the weight gradient of this layer: an MFMA reduction over d_act, off the serial chain, on the helper stream.
        // Every fork (event record on the launch stream) costs the chain ~6 us of queue processing, so the short top
        // layers (<= 128 steps: 30-80 us of weight-gradient work each) share the fork of the layer below them.
        // (Only where a long chain is still ahead to hide them under: with short sequences -- Amazon: 100 steps -- holding
        // them back just moves them into the tail.)
        // H = 128 (r4): the scans fill every register of the chip now (two workgroups per CU), so a weight gradient forked beside
        // one mostly waits for it and the big ones -- layers 1 and 0 -- end up one behind the other in the step's tail, each
        // latency-bound on its own at ~2.3 TB/s.  HPMN_WGRAD_ALTERNATE=1: layers alternate between the two helper streams (each
        // with its own slab buffer, the reduction behind its launch on the same stream) so that neighbours overlap -- measured
        // NEUTRAL (C4 7.52 vs 7.39-7.68 ms/step: two weight gradients resident when a scan launches take its CUs, layer 0's scan
        // 1430 instead of 1100 us), default off.
        static const int alt_env = [] { const char *e = getenv("HPMN_WGRAD_ALTERNATE"); return e ? atoi(e) : 0; }();
        const bool alternate = alt_env && !aside && d->H == 128 && L.wgrad_ws_layer[1] != 0 && cut0 == 0;
        if (alternate && (i & 1) == 0) w.workspace = F(L.wgrad_ws_layer[1]);
        held[nheld++] = w;
        if (L.T[i] > 128 || L.T[0] < 512 || i == 0 || nheld == 4) {
            HIPCHK(hipEventRecord(c->fork, st));
            HIPCHK(hipStreamWaitEvent(c->side, c->fork, 0));
            if (alternate) HIPCHK(hipStreamWaitEvent(c->side2, c->fork, 0));
            for (int h = 0; h < nheld; ++h) {
                const bool second = alternate && held[h].workspace == F(L.wgrad_ws_layer[1]);
                rc = hpmn_gru_param_grads(&held[h], second ? c->side2 : c->side);
                if (rc != HPMN_OK) return rc;
                if (second || aside) c->pending2 = true;
            }
            nheld = 0;
            c->pending = true;
        }
        if (!fused_dx) {
            rc = hpmn_gru_input_grad(F(L.d_act[i]), wg[i], wc[i], F(L.d_x[i]), d->B, L.T[i], D, d->H, 0, 0, stream);
            if (rc != HPMN_OK) return rc;
        }
    }
    if (nlate > 0) {
        // the pairs' weight gradients: nothing latency-critical is left on the chip, the largest (the last pair's) first
        HIPCHK(hipEventRecord(c->fork, st));
        HIPCHK(hipStreamWaitEvent(c->side, c->fork, 0));
        // (batches that fill half of the CUs or less: the launches are small, every other one goes to a second stream with
        //  its own slab buffer -- C2: 30 + 63 us one after the other at the end of the step)
        static const int two_env = [] { const char *e = getenv("HPMN_WGRAD_TWO_STREAMS"); return e ? atoi(e) : 1; }();
        const bool two = two_env && nlate >= 2 && d->H != 32 && L.wgrad_ws_layer[1] != 0 && 2 * d->B <= c->cus;
        if (two) HIPCHK(hipStreamWaitEvent(c->side2, c->fork, 0));
        for (int h = nlate - 1; h >= 0; --h) {
            late[h].whole_cu = d->H <= 64 ? 1 : 0;
            const bool second = two && ((nlate - 1 - h) & 1);
            if (second) late[h].workspace = F(L.wgrad_ws_layer[1]);
            const int rc0 = hpmn_gru_param_grads(&late[h], second ? c->side2 : c->side);
            if (rc0 != HPMN_OK) return rc0;
            if (second) c->pending2 = true;
        }
        nlate = 0;
        c->pending = true;
    }
    if (scatter_fused) {
        if (!defer_join) return hpmn_train_join(ctx, stream);
        return HPMN_OK;
    }
    // (d_last, the read path's gradient wrt uinp[:, last_index, :], rides into the scatter: ids step T + last_index)
    const bool last_in_scatter = d_last && !scatter_pending && d->T + d->last_index >= 0;
    if (d_last && !scatter_pending && !last_in_scatter) {
        const long n = (long)d->B * D0;
        hipLaunchKernelGGL(add_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st,
                           F(L.d_x[0]) + (size_t)(L.T[0] + d->last_index) * D0, (long)L.T[0] * D0, d_last, d->B, D0);
        int rc = check_launch();
        if (rc != HPMN_OK) return rc;
    }
    int rc = scatter_pending
                 ? embed_grad_scatter_launch(ids, F(L.d_x[0]), d_emb, d->B, d->T, d->F, d->E, d->front_zero, d->mask_id0, 0,
                                             scat_cut - d->front_zero, st, last_in_scatter ? d_last : nullptr,
                                             d->T + d->last_index)
                 : scatter_all(c, d, ids, F(L.d_x[0]), d_emb, last_in_scatter ? d_last : nullptr, st);
    if (rc != HPMN_OK) return rc;
    if (scatter_pending) HIPCHK(hipStreamWaitEvent(st, c->scat, 0));   // the caller's table update needs both halves
    if (!defer_join) return hpmn_train_join(ctx, stream);
    return HPMN_OK;
}

int hpmn_train_set_scatter_plan(HpmnTrainCtx *ctx, const HpmnScatterPlan *plan) {
    TrainCtx *c = reinterpret_cast<TrainCtx *>(ctx);
    if (!c) return HPMN_EINVAL;
    if (!plan || plan->n <= 0) { c->plan = HpmnScatterPlan{}; return HPMN_OK; }
    if (!plan->perm || !plan->seg || !plan->start || !plan->rows || !plan->count || !plan->partials) return HPMN_EINVAL;
    c->plan = *plan;
    return HPMN_OK;
}

int hpmn_train_probe(HpmnTrainCtx *ctx, int32_t enable) {
    TrainCtx *c = reinterpret_cast<TrainCtx *>(ctx);
    if (!c) return HPMN_EINVAL;
    if (enable && c->probe0 == nullptr) {
        HIPCHK(hipEventCreate(&c->probe0));
        HIPCHK(hipEventCreate(&c->probe1));
    }
    c->probe = enable != 0;
    c->probed = false;
    return HPMN_OK;
}

int hpmn_train_mark_layer0_reverse(HpmnTrainCtx *ctx, int32_t enable) {
    TrainCtx *c = reinterpret_cast<TrainCtx *>(ctx);
    if (!c) return HPMN_EINVAL;
    if (enable && c->l0_start == nullptr) HIPCHK(hipEventCreateWithFlags(&c->l0_start, hipEventDisableTiming));
    c->mark_l0 = enable != 0;
    c->l0_marked = false;
    return HPMN_OK;
}

int hpmn_train_wait_layer0_reverse(HpmnTrainCtx *ctx, void *stream) {
    TrainCtx *c = reinterpret_cast<TrainCtx *>(ctx);
    if (!c) return HPMN_EINVAL;
    if (!c->l0_marked) return HPMN_OK;                   // (a path without a separate layer-0 launch: nothing to wait for)
    HIPCHK(hipStreamWaitEvent((hipStream_t)stream, c->l0_start, 0));
    c->l0_marked = false;
    return HPMN_OK;
}

int hpmn_train_probe_ms(HpmnTrainCtx *ctx, float *ms) {
    TrainCtx *c = reinterpret_cast<TrainCtx *>(ctx);
    if (!c || !ms || !c->probed) return HPMN_EINVAL;
    HIPCHK(hipEventSynchronize(c->probe1));
    HIPCHK(hipEventElapsedTime(ms, c->probe0, c->probe1));
    return HPMN_OK;
}

int hpmn_train_step(HpmnTrainCtx *ctx, const HpmnTrainStep *s, void *stream) {
    if (!ctx || !s || !s->ids || !s->label || !s->param || !s->grad || !s->m || !s->v || !s->memory || !s->last || !s->pred ||
        !s->d_memory || !s->d_last || !s->scan_workspace || !s->read_workspace || !s->loss_acc || !s->loss3)
        return HPMN_EINVAL;
    const int K = s->scan.K;
    if (K < 1 || K > HPMN_MAX_LAYERS || s->scan.B < 1 || s->read.B != s->scan.B || s->n_emb < 0 || s->n_total < s->n_emb ||
        s->off_read < s->n_emb || s->off_read + s->read.n_params > s->n_total)
        return HPMN_EINVAL;
    const float *wg[HPMN_MAX_LAYERS], *bg[HPMN_MAX_LAYERS], *wc[HPMN_MAX_LAYERS], *bc[HPMN_MAX_LAYERS];
    float *dwg[HPMN_MAX_LAYERS], *dbg[HPMN_MAX_LAYERS], *dwc[HPMN_MAX_LAYERS], *dbc[HPMN_MAX_LAYERS];
    for (int i = 0; i < K; ++i) {
        for (int j = 0; j < 4; ++j)
            if (s->off_gru[i][j] < s->n_emb || s->off_gru[i][j] >= s->n_total) return HPMN_EINVAL;
        wg[i] = s->param + s->off_gru[i][0]; bg[i] = s->param + s->off_gru[i][1];
        wc[i] = s->param + s->off_gru[i][2]; bc[i] = s->param + s->off_gru[i][3];
        dwg[i] = s->grad + s->off_gru[i][0]; dbg[i] = s->grad + s->off_gru[i][1];
        dwc[i] = s->grad + s->off_gru[i][2]; dbc[i] = s->grad + s->off_gru[i][3];
    }
    hipStream_t st = (hipStream_t)stream;
    if (s->clear_grad_first) HIPCHK(hipMemsetAsync(s->grad, 0, (size_t)s->n_total * sizeof(float), st));
    int rc = hpmn_scan_fwd_train(ctx, &s->scan, s->ids, s->param, wg, bg, wc, bc, s->memory, s->last, s->scan_workspace, stream);
    if (rc != HPMN_OK) return rc;
    // read path: forward + loss + backward; its weight gradients stay in the workspace as partial sums
    rc = hpmn_read_fwd_bwd(&s->read, s->param + s->off_read, s->memory, s->last, s->label, s->mask1, s->mask2, s->keep_prob,
                           s->inv_global_batch, s->memory_reg, s->pred, s->loss_acc, s->d_memory, s->d_last, nullptr,
                           s->read_workspace, stream);
    if (rc != HPMN_OK) return rc;
    // The read path's weight gradients + the loss scalars need only the tape the launch above left.  HPMN_READ_GRADS_ASIDE=1 puts
    // them on the context's helper stream underneath the reverse scans instead of behind them on the caller's stream -- measured
    // (r6, C1): 0.259 against 0.252 ms/step; at 512 sequences 0.357 against 0.350: the helper stream's first launch then waits
    // for a cross-queue event instead of being queued behind the scatter, and the join at the end has two launches more in
    // front of it.  Off (r5 found the same for a Python-side auxiliary stream).
    TrainCtx *c = reinterpret_cast<TrainCtx *>(ctx);
    static const int rd_aside = [] { const char *e = getenv("HPMN_READ_GRADS_ASIDE"); return e ? atoi(e) : 0; }();
    const HpmnReadDesc *rd = &s->read;
    if (rd_aside) {
        HIPCHK(hipEventRecord(c->fork, st));
        HIPCHK(hipStreamWaitEvent(c->side, c->fork, 0));
        rc = hpmn_read_param_grads_loss_n(1, &rd, s->grad + s->off_read, s->read_workspace, s->loss_acc, s->inv_global_batch,
                                          s->memory_reg, s->loss3, c->side);
        if (rc != HPMN_OK) return rc;
        c->pending = true;
    }
    rc = hpmn_scan_bwd(ctx, &s->scan, s->ids, wg, wc, s->d_memory, s->d_last, dwg, dbg, dwc, dbc, s->grad, s->scan_workspace,
                       /*defer_join=*/1, stream);
    if (rc != HPMN_OK) return rc;
    if (!rd_aside) {
        rc = hpmn_read_param_grads_loss_n(1, &rd, s->grad + s->off_read, s->read_workspace, s->loss_acc, s->inv_global_batch,
                                          s->memory_reg, s->loss3, stream);
        if (rc != HPMN_OK) return rc;
    }
    if (s->n_emb > 0) {
        rc = hpmn_adam_step_clear(s->param, s->grad, s->m, s->v, s->n_emb, s->lr_t, s->beta1, s->beta2, s->eps, s->clip, 1.0f, stream);
        if (rc != HPMN_OK) return rc;
    }
    rc = hpmn_train_join(ctx, stream);
    if (rc != HPMN_OK) return rc;
    return hpmn_adam_step_clear(s->param + s->n_emb, s->grad + s->n_emb, s->m + s->n_emb, s->v + s->n_emb, s->n_total - s->n_emb,
                                s->lr_t, s->beta1, s->beta2, s->eps, s->clip, 1.0f, stream);
}

int hpmn_train_join(HpmnTrainCtx *ctx, void *stream) {
    TrainCtx *c = reinterpret_cast<TrainCtx *>(ctx);
    if (!c) return HPMN_EINVAL;
    if (c->pending2) {
        HIPCHK(hipEventRecord(c->join2, c->side2));
        HIPCHK(hipStreamWaitEvent((hipStream_t)stream, c->join2, 0));
        c->pending2 = false;
    }
    if (!c->pending) return HPMN_OK;
    HIPCHK(hipEventRecord(c->join, c->side));
    HIPCHK(hipStreamWaitEvent((hipStream_t)stream, c->join, 0));
    c->pending = false;
    return HPMN_OK;
}

}  // extern "C"
