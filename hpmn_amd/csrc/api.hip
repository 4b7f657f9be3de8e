// extern "C" surface of libhpmn_hip.so (declared in include/hpmn_hip.h): argument
// validation, shape dispatch, and the multi-layer build_memory forward chain.
#include "common.h"
#include "gru32_all.h"

namespace hpmn {

static thread_local int g_last_hip_error = 0;
void set_last_hip_error(int e) { g_last_hip_error = e; }

bool gru_shape_supported(int H, int D);
int gru_scan_fwd_dispatch(const HpmnGruFwd &a, hipStream_t st);
int gru_scan_bwd_dispatch(const HpmnGruBwd &a, hipStream_t st);
bool gru_scan_bwd_fuses_dx(int H, int B);
bool gru_candidate_elision(int H, int B);
bool gru_scan_bwd_dx_width_ok(int D);
bool gru_scan_bwd_fuses_scatter(int H, int B, int D, int F, int E);
bool input_proj_supported(int H, int D);
int memory_update_launch(const HpmnOnlineUpdate &a, hipStream_t st);
bool gru_fused_fwd_supported(int H, int D, int gather);
bool gru_fused_fwd_writes_last();
int gru_fused_fwd_dispatch(const HpmnGruFusedFwd &a, hipStream_t st);
bool gru_pair_fwd_supported(int H, int D_lo, int gather);
bool gru_pair_bwd_supported(int H, int D_lo);
int gru_pair_bwd_launch(const HpmnGruBwd &lo, const HpmnGruBwd &up, int flags, hipStream_t st);
size_t gru_pair_fwd_scratch_bytes();
int gru_pair_fwd_launch(const HpmnGruFusedFwd &lo, const HpmnGruFusedFwd &up, int flags, float *scratch,
                        const float *img_lo, const float *img_up, hipStream_t st);
size_t gru_proj_image_floats(int D);
int gru_proj_images_launch(int n, const float *const *wg, const float *const *bg, const float *const *wc,
                           const float *const *bc, const int *D, float *const *img, hipStream_t st);
int input_proj_dispatch(const HpmnInputProj &a, hipStream_t st);
int gru_wgrad_dispatch(const HpmnGruWgrad &a, hipStream_t st);
int gru_dx_dispatch(const HpmnGruWgrad &a, hipStream_t st);
size_t gru_wgrad_workspace_bytes(int B, int T, int D, int H);
size_t read_workspace_bytes(const HpmnReadDesc &d);
size_t read_workspace_bytes_n(const HpmnReadDesc *const *d, int nb);
int read_param_grads_launch_n(const HpmnReadDesc *const *d, int nb, float *d_params, float *workspace, hipStream_t st,
                              float *loss_acc, float inv_global_batch, float memory_reg, float *loss3);
int read_fwd_launch(const HpmnReadDesc &d, const float *P, const float *memory, const float *last, float *pred,
                    float *logit, float *att_w0, float *mem_loss, hipStream_t st);
int read_fwd_bwd_launch(const HpmnReadDesc &d, const float *P, const float *memory, const float *last,
                        const int32_t *label, const float *mask1, const float *mask2, float keep_prob,
                        float inv_global_batch, float memory_reg, float *pred, float *loss_out, float *d_memory,
                        float *d_last, float *d_params, float *workspace, hipStream_t st);
int read_reduce_launch(const HpmnReadDesc &d, float *d_params, float *workspace, hipStream_t st);
int read_fwd_launch_n(const HpmnReadDesc *const *d, int nb, const float *P, const float *const *memory,
                      const float *const *last, float *pred, float *logit, float *const *att_w0, float *mem_loss,
                      hipStream_t st);
int read_fwd_bwd_launch_n(const HpmnReadDesc *const *d, int nb, const float *P, const float *const *memory,
                          const float *const *last, const int32_t *label, const float *mask1, const float *mask2,
                          float keep_prob, float inv_global_batch, float memory_reg, float *pred, float *loss_out,
                          float *const *d_memory, float *const *d_last, float *d_params, float *workspace, hipStream_t st);
int embed_gather_launch(const void *ids, int64_t ids_stride, const float *emb, float *out, int64_t N,
                        int32_t F, int32_t E, int32_t mask_id0, hipStream_t st);
int embed_grad_scatter_launch(const void *ids, const float *d_x, float *d_emb, int32_t B, int32_t T,
                              int32_t F, int32_t E, int32_t front_zero, int32_t mask_id0, int32_t t_lo, int32_t t_hi,
                              hipStream_t st, const float *d_last = nullptr, int32_t t_last = 0);
int adam_launch(float *p, const float *g, float *m, float *v, int64_t n, float lr_t, float b1, float b2,
                float eps, float clip, float gs, hipStream_t st);
int adam_clear_launch(float *p, float *g, float *m, float *v, int64_t n, float lr_t, float b1, float b2,
                      float eps, float clip, float gs, hipStream_t st);
int table_mark_launch(const void *ids, int64_t n, uint8_t *flags, int64_t V, int32_t id_flags, hipStream_t st);
int scatter_plan_launch(const void *sorted_ids, int32_t id_flags, int64_t n, const int32_t *seg, int32_t *start, void *rows,
                        int32_t *count, hipStream_t st);
size_t segsum_partials_floats(int64_t n, int32_t E);
int segsum_chunk_entries();
int embed_grad_segsum_launch(const HpmnScatterPlan &p, const float *d_x, float *d_emb, int32_t B, int32_t T, int32_t F,
                             int32_t E, int32_t front_zero, int32_t id_flags, const float *d_last, int32_t t_last,
                             hipStream_t st);
int adam_table_launch(float *p, float *g, float *m, float *v, uint8_t *flags, int64_t V, int E, int pass, float lr_t,
                      float b1, float b2, float eps, float clip, float gs, hipStream_t st);
int rows_sum_adam_launch(const HpmnRowsAdam &h, hipStream_t st);
size_t scatter_plan_build_workspace_bytes(int64_t n, int32_t id_flags, int64_t V);
int scatter_plan_build_launch(const void *ids, int32_t id_flags, int64_t n, int64_t V, void *workspace, size_t workspace_bytes,
                              int32_t *perm, int32_t *seg, int32_t *start, void *rows, int32_t *count,
                              const int64_t *row_bounds, int32_t nb, int32_t *counts, hipStream_t st);
int table_mark_ranks_launch(const void *ids, int64_t ids_stride, int32_t world, const int32_t *counts, int32_t counts_stride,
                            int64_t cap, uint8_t *flags, int64_t V, int32_t id_flags, int32_t *bstart, int64_t bstride,
                            int32_t bshift, hipStream_t st);
int adam_rows_launch(float *p, const float *g, float *m, float *v, const int64_t *row_ids, int64_t n_rows, int E,
                     float lr_t, float b1, float b2, float eps, float clip, float gs, hipStream_t st);

static size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Layer lengths of code/hpmn.py:122-128; false if some reshape there would not divide.
static bool layer_lengths(const HpmnScanDesc &d, int32_t *len) {
    if (d.K < 1 || d.K > HPMN_MAX_LAYERS) return false;
    long t = (long)d.T + d.front_zero;
    for (int i = 0; i < d.K; ++i) {
        len[i] = (int32_t)t;
        if (d.periods[i] < 1 || t % d.periods[i] != 0) return false;
        t /= d.periods[i];
    }
    return true;
}

}  // namespace hpmn

using namespace hpmn;

// hipGetLastError() is sticky per thread: an error left behind by an unrelated earlier HIP call of the
// host application (seen: 100 from a device probe made before the runtime was initialised) would be
// reported by the first launch check of this library.  Every launching entry point drops it first.
static inline void drop_stale_hip_error() { (void)hipGetLastError(); }

extern "C" {

int hpmn_abi_version(void) { return HPMN_ABI_VERSION; }

int hpmn_has_legacy_kernels(void) {
#ifdef HPMN_LEGACY_KERNELS
    return 1;
#else
    return 0;
#endif
}

const char *hpmn_strerror(int code) {
    switch (code) {
        case HPMN_OK: return "ok";
        case HPMN_EINVAL: return "invalid argument";
        case HPMN_EUNSUPPORTED: return "unsupported shape for the gfx950 kernels";
        case HPMN_EHIP: return "HIP runtime error (see hpmn_last_hip_error)";
        case HPMN_ENODEVICE: return "no gfx950 device";
        default: return "unknown error";
    }
}

int hpmn_last_hip_error(void) { return g_last_hip_error; }

int hpmn_gru_shape_supported(int32_t H, int32_t D) {
    return (gru_shape_supported(H, D) && input_proj_supported(H, D)) ? 1 : 0;
}

int hpmn_embed_gather(const void *ids, const float *emb, float *out, int64_t N, int32_t F, int32_t E,
                      int64_t V, int32_t mask_id0, void *stream) {
    drop_stale_hip_error();
    if (N < 0 || F < 1 || E < 4 || V < 1) return HPMN_EINVAL;
    if (E % 4 != 0) return HPMN_EUNSUPPORTED;
    if (N == 0) return HPMN_OK;
    if (!ids || !emb || !out) return HPMN_EINVAL;
    return embed_gather_launch(ids, F, emb, out, N, F, E, mask_id0, (hipStream_t)stream);
}

int hpmn_gru_input_proj(const HpmnInputProj *a, void *stream) {
    drop_stale_hip_error();
    if (!a) return HPMN_EINVAL;
    if (a->B < 0 || a->T < 1 || a->D < 1 || a->H < 1) return HPMN_EINVAL;
    if (!a->wg || !a->bg || !a->wc || !a->bc || !a->xp) return HPMN_EINVAL;
    if (a->x == nullptr) {
        if (!a->ids || !a->emb) return HPMN_EINVAL;
        if (a->F < 1 || a->E < 4 || a->F * a->E != a->D || a->front_zero < 0 || a->Tids < 1 ||
            a->front_zero + a->Tids != a->T)
            return HPMN_EINVAL;
        if (a->E % 4 != 0) return HPMN_EUNSUPPORTED;
    }
    if (a->t_begin < 0 || a->t_len < 0 || a->t_begin + a->t_len > a->T) return HPMN_EINVAL;
    if (!input_proj_supported(a->H, a->D)) return HPMN_EUNSUPPORTED;
    if (a->B == 0) return HPMN_OK;
    return input_proj_dispatch(*a, (hipStream_t)stream);
}

int hpmn_gru_scan_fwd(const HpmnGruFwd *a, void *stream) {
    drop_stale_hip_error();
    if (!a) return HPMN_EINVAL;
    if (a->B < 0 || a->T < 1 || a->D < 1 || a->H < 1) return HPMN_EINVAL;
    if (!a->xp || !a->wg || !a->wc || !a->h_last) return HPMN_EINVAL;
    if ((a->hs == nullptr) != (a->gates == nullptr)) return HPMN_EINVAL;
    if (a->y && (a->period < 1 || a->T % a->period != 0)) return HPMN_EINVAL;
    if (a->t_begin < 0 || a->t_end < 0 || a->t_end > a->T || (a->t_end > 0 && a->t_end <= a->t_begin)) return HPMN_EINVAL;
    if (a->t_begin % 2 != 0 || (a->period > 0 && a->t_begin % a->period != 0)) return HPMN_EINVAL;
    if (!gru_shape_supported(a->H, a->D)) return HPMN_EUNSUPPORTED;
    if (a->B == 0) return HPMN_OK;
    HpmnGruFwd k = *a;
    if (k.period < 1) k.period = 1;
    return gru_scan_fwd_dispatch(k, (hipStream_t)stream);
}

int hpmn_gru_scan_bwd(const HpmnGruBwd *a, void *stream) {
    drop_stale_hip_error();
    if (!a) return HPMN_EINVAL;
    if (a->B < 0 || a->T < 1 || a->D < 1 || a->H < 1) return HPMN_EINVAL;
    if (!a->wg || !a->wc || !a->hs || !a->gates || !a->d_h_last || !a->d_act) return HPMN_EINVAL;
    if (a->d_y && (a->period < 1 || a->T % a->period != 0)) return HPMN_EINVAL;
    {
        const int t_hi = a->t_end > 0 ? a->t_end : a->T;
        if (a->t_begin < 0 || t_hi > a->T || t_hi <= a->t_begin) return HPMN_EINVAL;
        if ((t_hi < a->T || a->t_begin > 0) && !a->dh_carry) return HPMN_EINVAL;
        if (a->d_y && t_hi % a->period != 0) return HPMN_EINVAL;
    }
    if (!gru_shape_supported(a->H, a->D)) return HPMN_EUNSUPPORTED;
    if (a->d_x && (!gru_scan_bwd_fuses_dx(a->H, a->B) || !gru_scan_bwd_dx_width_ok(a->D))) return HPMN_EUNSUPPORTED;
    if ((a->flags & HPMN_BWD_CANDIDATE_FROM_HS) && !gru_candidate_elision(a->H, a->B)) return HPMN_EUNSUPPORTED;
    if (a->d_emb) {
        if (!a->scatter_ids || a->Tids < 1 || a->F < 1 || a->front_zero < 0 || a->front_zero + a->Tids != a->T) return HPMN_EINVAL;
        if (a->t_begin != 0 || (a->t_end != 0 && a->t_end != a->T)) return HPMN_EINVAL;
        if (!gru_scan_bwd_fuses_scatter(a->H, a->B, a->D, a->F, a->E)) return HPMN_EUNSUPPORTED;
        if (a->mask_id0 & HPMN_ID_I64) return HPMN_EUNSUPPORTED;   // the fused scatter reads int32 ids (hpmn_embed_grad_scatter takes both)
    }
    if (a->B == 0) return HPMN_OK;
    HpmnGruBwd k = *a;
    if (k.period < 1) k.period = 1;
    return gru_scan_bwd_dispatch(k, (hipStream_t)stream);
}

int hpmn_gru_scan_bwd_fuses_dx(int32_t H, int32_t B) { return gru_scan_bwd_fuses_dx(H, B) ? 1 : 0; }
int hpmn_gru_candidate_elision(int32_t H, int32_t B) { return gru_candidate_elision(H, B) ? 1 : 0; }
int hpmn_gru_scan_bwd_fuses_scatter(int32_t H, int32_t B, int32_t D, int32_t F, int32_t E) {
    return gru_scan_bwd_fuses_scatter(H, B, D, F, E) ? 1 : 0;
}

size_t hpmn_gru_param_grads_workspace_bytes(int32_t B, int32_t T, int32_t D, int32_t H) {
    if (B < 1 || T < 1 || D < 1 || H < 1) return 0;
    return gru_wgrad_workspace_bytes(B, T, D, H);
}

int hpmn_gru_param_grads(const HpmnGruWgrad *a, void *stream) {
    drop_stale_hip_error();
    if (!a) return HPMN_EINVAL;
    if (a->B < 0 || a->T < 1 || a->D < 1 || a->H < 1) return HPMN_EINVAL;
    if (!gru_shape_supported(a->H, a->D) || !input_proj_supported(a->H, a->D)) return HPMN_EUNSUPPORTED;
    if (a->B == 0) return HPMN_OK;
    if (!a->x || !a->hs || !a->gates || !a->d_act || !a->wg || !a->wc || !a->d_wg || !a->d_bg || !a->d_wc ||
        !a->d_bc || !a->workspace)
        return HPMN_EINVAL;
    return gru_wgrad_dispatch(*a, (hipStream_t)stream);
}

int hpmn_gru_input_grad(const float *d_act, const float *wg, const float *wc, float *d_x, int32_t B, int32_t T,
                        int32_t D, int32_t H, int32_t t_begin, int32_t t_len, void *stream) {
    drop_stale_hip_error();
    if (B < 0 || T < 1 || D < 1 || H < 1 || t_begin < 0 || t_len < 0 || t_begin + t_len > T) return HPMN_EINVAL;
    if (!gru_shape_supported(H, D) || !input_proj_supported(H, D)) return HPMN_EUNSUPPORTED;
    if (B == 0) return HPMN_OK;
    if (!d_act || !wg || !wc || !d_x) return HPMN_EINVAL;
    HpmnGruWgrad a = {};
    a.B = B; a.T = T; a.D = D; a.H = H;
    a.d_act = d_act; a.wg = wg; a.wc = wc; a.d_x = d_x;
    a.t_begin = t_begin; a.t_len = t_len;
    return gru_dx_dispatch(a, (hipStream_t)stream);
}

// workspace carve of hpmn_scan_fwd: [xp: B*T0*3H] [y0: B*(T0/p0)*H] [y1: same]
static void scan_ws_sizes(const HpmnScanDesc &d, const int32_t *len, size_t &xp_bytes, size_t &y_bytes) {
    xp_bytes = align_up((size_t)d.B * (size_t)len[0] * 3 * d.H * sizeof(float), 256);
    y_bytes = d.K > 1 ? align_up((size_t)d.B * (size_t)(len[0] / d.periods[0]) * d.H * sizeof(float), 256) : 0;
}

// operand images of the two-layer launches (the paired inference path below), behind the buffers above
static size_t scan_img_bytes(const HpmnScanDesc &d) {
    return align_up((size_t)d.K * gru_proj_image_floats(64) * sizeof(float), 256);
}

size_t hpmn_scan_workspace_bytes(const HpmnScanDesc *d) {
    int32_t len[HPMN_MAX_LAYERS];
    if (!d || d->B < 0 || d->H < 1 || !layer_lengths(*d, len)) return 0;
    size_t xp_bytes, y_bytes;
    scan_ws_sizes(*d, len, xp_bytes, y_bytes);
    return xp_bytes + 2 * y_bytes + scan_img_bytes(*d) + 256;
}

// Inference with two layers per launch (H = 64): pairs (0,1), (2,3), ... of hpmn_gru_pair_fwd without saved states, an odd
// last layer through hpmn_gru_fused_fwd.  The rows between the layers of a pair never reach memory; nothing else runs on
// the chip during an evaluation pass, so the pairs may start at layer 0, and a batch larger than two sequences per CU just
// queues (the workgroups of a launch are independent).  HPMN_PAIR_INFER=0: the two-kernel layers below.
static bool scan_fwd_pairs_ok(const HpmnScanDesc &d, int D0) {
    static const int on = [] { const char *e = getenv("HPMN_PAIR_INFER"); return e ? atoi(e) : 1; }();
    return on && d.K >= 2 && d.E % 4 == 0 && gru_pair_fwd_supported(d.H, D0, 1);
}

static int scan_fwd_pairs(const HpmnScanDesc &d, const int32_t *len, const void *ids, const float *emb,
                          const float *const *wg, const float *const *bg, const float *const *wc, const float *const *bc,
                          float *memory, float *last, float *const *ybuf, float *img_base, hipStream_t st) {
    const int D0 = d.F * d.E, K = d.K, H = d.H;
    const size_t img_stride = gru_proj_image_floats(64);
    {
        const float *iwg[HPMN_MAX_LAYERS], *ibg[HPMN_MAX_LAYERS], *iwc[HPMN_MAX_LAYERS], *ibc[HPMN_MAX_LAYERS];
        float *img[HPMN_MAX_LAYERS];
        int iD[HPMN_MAX_LAYERS], n = 0;
        for (int j = 0; j < K; ++j) {
            if ((j == 0 ? D0 : H) != 64) continue;
            iwg[n] = wg[j]; ibg[n] = bg[j]; iwc[n] = wc[j]; ibc[n] = bc[j]; img[n] = img_base + j * img_stride; iD[n] = 64;
            ++n;
        }
        const int rc = gru_proj_images_launch(n, iwg, ibg, iwc, ibc, iD, img, st);
        if (rc != HPMN_OK) return rc;
    }
    int in_buf = 0;                                  // which of the two row buffers the next launch reads its input from
    auto layer = [&](int i) {
        HpmnGruFusedFwd a = {};
        a.B = d.B; a.T = len[i]; a.D = i == 0 ? D0 : H; a.H = H;
        if (i == 0) {
            a.ids = ids; a.emb = emb; a.Tids = d.T; a.F = d.F; a.E = d.E; a.front_zero = d.front_zero;
            a.mask_id0 = d.mask_id0; a.V = d.V;
            if (last) { a.last = last; a.last_t = len[0] + d.last_index; }
        } else {
            a.x = ybuf[in_buf];
        }
        a.wg = wg[i]; a.bg = bg[i]; a.wc = wc[i]; a.bc = bc[i];
        a.h_last = memory + (size_t)i * H; a.h_last_stride = (int64_t)K * H;
        a.period = d.periods[i];
        return a;
    };
    for (int i = 0; i < K; i += 2) {
        if (i + 1 < K) {
            HpmnGruFusedFwd lo = layer(i), up = layer(i + 1);
            up.x = nullptr;
            // only a following launch reads rows from memory -- and not from the buffer this launch's lower layer is
            // still reading ITS input rows from
            const int out_buf = i == 0 ? 0 : in_buf ^ 1;
            up.y = i + 2 < K ? ybuf[out_buf] : nullptr;
            const int rc = gru_pair_fwd_launch(lo, up, 0, nullptr, i == 0 && D0 != 64 ? nullptr : img_base + i * img_stride,
                                               img_base + (i + 1) * img_stride, st);
            if (rc != HPMN_OK) return rc;
            in_buf = out_buf;
        } else {
            const HpmnGruFusedFwd a = layer(i);
            const int rc = gru_fused_fwd_dispatch(a, st);
            if (rc != HPMN_OK) return rc;
        }
    }
    return HPMN_OK;
}

int hpmn_scan_fwd(const HpmnScanDesc *d, const void *ids, const float *emb, const float *const *wg,
                  const float *const *bg, const float *const *wc, const float *const *bc, float *memory,
                  float *last, void *workspace, void *stream) {
    drop_stale_hip_error();
    int32_t len[HPMN_MAX_LAYERS];
    if (!d || !ids || !emb || !wg || !bg || !wc || !bc || !memory || !workspace) return HPMN_EINVAL;
    if (d->B < 0 || d->T < 1 || d->F < 1 || d->E < 4 || d->H < 1 || d->V < 1) return HPMN_EINVAL;
    if (!layer_lengths(*d, len)) return HPMN_EINVAL;
    if (d->last_index >= 0 || -d->last_index > len[0]) return HPMN_EINVAL;
    const int D0 = d->F * d->E;
    if (!hpmn_gru_shape_supported(d->H, D0) || (d->K > 1 && !hpmn_gru_shape_supported(d->H, d->H)))
        return HPMN_EUNSUPPORTED;
    if (d->B == 0) return HPMN_OK;
    hipStream_t st = (hipStream_t)stream;

    size_t xp_bytes, y_bytes;
    scan_ws_sizes(*d, len, xp_bytes, y_bytes);
    char *ws = reinterpret_cast<char *>(align_up(reinterpret_cast<size_t>(workspace), 256));
    float *xp = reinterpret_cast<float *>(ws);
    float *ybuf[2] = {reinterpret_cast<float *>(ws + xp_bytes), reinterpret_cast<float *>(ws + xp_bytes + y_bytes)};

    if (gru32_all_enabled() && gru32_all_supported(d->H, D0, d->K, d->E)) {
        All32Args a = {};
        gru32_all_fill(a, *d, len, ids, emb, wg, bg, wc, bc, memory, last);
        return gru32_fwd_all_launch(a, D0, false, st);
    }
    if (scan_fwd_pairs_ok(*d, D0) && gru_fused_fwd_writes_last())
        return scan_fwd_pairs(*d, len, ids, emb, wg, bg, wc, bc, memory, last, ybuf,
                              reinterpret_cast<float *>(ws + xp_bytes + 2 * y_bytes), st);

    for (int i = 0; i < d->K; ++i) {
        // (inference keeps the two-kernel layer: without the saved-state stores the fused layer's scan wave,
        //  slowed by sharing its CU's LDS pipe with the projection wave, gains nothing -- measured 473 k vs
        //  534 k sequences/s forward-only at C3)
        HpmnInputProj p = {};
        p.B = d->B; p.T = len[i]; p.H = d->H;
        p.wg = wg[i]; p.bg = bg[i]; p.wc = wc[i]; p.bc = bc[i];
        p.xp = xp;
        if (i == 0) {
            p.D = D0; p.ids = ids; p.emb = emb;
            p.Tids = d->T; p.F = d->F; p.E = d->E; p.front_zero = d->front_zero;
            p.mask_id0 = d->mask_id0; p.V = d->V;
        } else {
            p.D = d->H; p.x = ybuf[(i - 1) & 1];
        }
        int rc = hpmn_gru_input_proj(&p, stream);
        if (rc != HPMN_OK) return rc;

        HpmnGruFwd a = {};
        a.B = d->B; a.T = len[i]; a.H = d->H; a.D = p.D;
        a.xp = xp; a.wg = wg[i]; a.wc = wc[i];
        a.h_last = memory + (size_t)i * d->H;
        a.h_last_stride = (int64_t)d->K * d->H;
        a.period = d->periods[i];
        a.y = (i + 1 < d->K) ? ybuf[i & 1] : nullptr;
        rc = hpmn_gru_scan_fwd(&a, stream);
        if (rc != HPMN_OK) return rc;
    }
    if (last) {
        // uinp[:, last_index, :]: scan position len0+last_index -> loader position minus the zero prefix
        const long tid = (long)len[0] + d->last_index - d->front_zero;
        if (tid < 0) {
            hipError_t e = hipMemsetAsync(last, 0, (size_t)d->B * D0 * sizeof(float), st);
            if (e != hipSuccess) { set_last_hip_error((int)e); return HPMN_EHIP; }
        } else {
            const size_t idw = (d->mask_id0 & HPMN_ID_I64) ? 8 : 4;
            int rc = embed_gather_launch(static_cast<const char *>(ids) + (size_t)(tid * d->F) * idw, (int64_t)d->T * d->F, emb, last, d->B, d->F, d->E,
                                         d->mask_id0, st);
            if (rc != HPMN_OK) return rc;
        }
    }
    return HPMN_OK;
}

size_t hpmn_read_workspace_bytes(const HpmnReadDesc *d) {
    if (!d || d->B < 1 || d->n_params < 1) return 0;
    return read_workspace_bytes(*d);
}

size_t hpmn_read_workspace_bytes_n(int32_t nb, const HpmnReadDesc *const *desc) {
    if (nb < 1 || nb > 2 || !desc || !desc[0] || (nb > 1 && !desc[1])) return 0;
    if (desc[0]->B < 1 || desc[0]->n_params < 1) return 0;
    return read_workspace_bytes_n(desc, nb);
}

int hpmn_read_fwd(const HpmnReadDesc *d, const float *params, const float *memory, const float *last, float *pred,
                  float *logit, float *att_w0, float *mem_loss, void *stream) {
    drop_stale_hip_error();
    if (!d) return HPMN_EINVAL;
    if (d->B < 0) return HPMN_EINVAL;
    if (d->B == 0) return HPMN_OK;
    if (!params || !memory || !last || !pred || !mem_loss) return HPMN_EINVAL;
    return read_fwd_launch(*d, params, memory, last, pred, logit, att_w0, mem_loss, (hipStream_t)stream);
}

int hpmn_read_fwd_bwd(const HpmnReadDesc *d, const float *params, const float *memory, const float *last,
                      const int32_t *label, const float *mask1, const float *mask2, float keep_prob,
                      float inv_global_batch, float memory_reg, float *pred, float *loss_out, float *d_memory,
                      float *d_last, float *d_params, float *workspace, void *stream) {
    drop_stale_hip_error();
    if (!d) return HPMN_EINVAL;
    if (d->B < 0 || !(keep_prob > 0.f)) return HPMN_EINVAL;
    if (d->B == 0) return HPMN_OK;
    if (!params || !memory || !last || !label || !pred || !loss_out || !d_memory || !d_last || !workspace)
        return HPMN_EINVAL;                       // (d_params may be NULL: see hpmn_read_param_grads)
    if ((mask1 == nullptr) != (mask2 == nullptr)) return HPMN_EINVAL;
    return read_fwd_bwd_launch(*d, params, memory, last, label, mask1, mask2, keep_prob, inv_global_batch,
                               memory_reg, pred, loss_out, d_memory, d_last, d_params, workspace,
                               (hipStream_t)stream);
}

int hpmn_read_fwd_n(int32_t nb, const HpmnReadDesc *const *desc, const float *params, const float *const *memory,
                    const float *const *last, float *pred, float *logit, float *const *att_w0, float *mem_loss,
                    void *stream) {
    drop_stale_hip_error();
    if (nb < 1 || nb > 2 || !desc || !desc[0] || (nb > 1 && !desc[1])) return HPMN_EINVAL;
    if (desc[0]->B < 0) return HPMN_EINVAL;
    if (desc[0]->B == 0) return HPMN_OK;
    if (!params || !memory || !last || !pred || !mem_loss) return HPMN_EINVAL;
    for (int b = 0; b < nb; ++b)
        if (!memory[b] || !last[b]) return HPMN_EINVAL;
    return read_fwd_launch_n(desc, nb, params, memory, last, pred, logit, att_w0, mem_loss, (hipStream_t)stream);
}

int hpmn_read_fwd_bwd_n(int32_t nb, const HpmnReadDesc *const *desc, const float *params, const float *const *memory,
                        const float *const *last, const int32_t *label, const float *mask1, const float *mask2,
                        float keep_prob, float inv_global_batch, float memory_reg, float *pred, float *loss_out,
                        float *const *d_memory, float *const *d_last, float *d_params, float *workspace, void *stream) {
    drop_stale_hip_error();
    if (nb < 1 || nb > 2 || !desc || !desc[0] || (nb > 1 && !desc[1])) return HPMN_EINVAL;
    if (desc[0]->B < 0) return HPMN_EINVAL;
    if (desc[0]->B == 0) return HPMN_OK;
    if (!params || !memory || !last || !label || !pred || !loss_out || !d_memory || !d_last || !workspace) return HPMN_EINVAL;
    if (!(keep_prob > 0.f && keep_prob <= 1.f)) return HPMN_EINVAL;
    for (int b = 0; b < nb; ++b)
        if (!memory[b] || !last[b] || !d_memory[b] || !d_last[b]) return HPMN_EINVAL;
    return read_fwd_bwd_launch_n(desc, nb, params, memory, last, label, mask1, mask2, keep_prob, inv_global_batch,
                                 memory_reg, pred, loss_out, d_memory, d_last, d_params, workspace, (hipStream_t)stream);
}

int hpmn_read_param_grads(const HpmnReadDesc *d, float *d_params, float *workspace, void *stream) {
    drop_stale_hip_error();
    if (!d || d->B < 0) return HPMN_EINVAL;
    if (d->B == 0) return HPMN_OK;
    if (!d_params || !workspace) return HPMN_EINVAL;
    return read_reduce_launch(*d, d_params, workspace, (hipStream_t)stream);
}

int hpmn_read_param_grads_n(int32_t nb, const HpmnReadDesc *const *desc, float *d_params, float *workspace, void *stream) {
    drop_stale_hip_error();
    if (nb < 1 || nb > 2 || !desc || !desc[0] || (nb > 1 && !desc[1])) return HPMN_EINVAL;
    if (desc[0]->B < 0) return HPMN_EINVAL;
    if (desc[0]->B == 0) return HPMN_OK;
    if (!d_params || !workspace) return HPMN_EINVAL;
    return read_param_grads_launch_n(desc, nb, d_params, workspace, (hipStream_t)stream, nullptr, 0.f, 0.f, nullptr);
}

int hpmn_read_param_grads_loss_n(int32_t nb, const HpmnReadDesc *const *desc, float *d_params, float *workspace,
                                 float *loss_acc, float inv_global_batch, float memory_reg, float *loss3, void *stream) {
    drop_stale_hip_error();
    if (nb < 1 || nb > 2 || !desc || !desc[0] || (nb > 1 && !desc[1])) return HPMN_EINVAL;
    if (desc[0]->B < 0) return HPMN_EINVAL;
    if (!loss_acc || !loss3) return HPMN_EINVAL;
    if (desc[0]->B == 0) return HPMN_OK;
    if (!d_params || !workspace) return HPMN_EINVAL;
    return read_param_grads_launch_n(desc, nb, d_params, workspace, (hipStream_t)stream, loss_acc, inv_global_batch,
                                     memory_reg, loss3);
}

int hpmn_embed_grad_scatter(const void *ids, const float *d_x, float *d_emb, int32_t B, int32_t T,
                            int32_t F, int32_t E, int32_t front_zero, int64_t V, int32_t mask_id0,
                            void *stream) {
    drop_stale_hip_error();
    if (B < 0 || T < 1 || F < 1 || E < 1 || front_zero < 0 || V < 1) return HPMN_EINVAL;
    if (64 % E != 0) return HPMN_EUNSUPPORTED;
    if (B == 0) return HPMN_OK;
    if (!ids || !d_x || !d_emb) return HPMN_EINVAL;
    return embed_grad_scatter_launch(ids, d_x, d_emb, B, T, F, E, front_zero, mask_id0, 0, T, (hipStream_t)stream);
}

int hpmn_scatter_plan(const void *sorted_ids, int32_t id_flags, int64_t n, const int32_t *seg, int32_t *start, void *rows,
                      int32_t *count, void *stream) {
    drop_stale_hip_error();
    if (n < 0 || n > 0x7fffffffLL) return HPMN_EINVAL;
    if (n == 0) return HPMN_OK;
    if (!sorted_ids || !seg || !start || !rows || !count) return HPMN_EINVAL;
    return scatter_plan_launch(sorted_ids, id_flags, n, seg, start, rows, count, (hipStream_t)stream);
}

size_t hpmn_scatter_plan_build_workspace_bytes(int64_t n, int32_t id_flags, int64_t V) {
    if (n <= 0 || n > 0x7fffffffLL || V < 1) return 0;
    return scatter_plan_build_workspace_bytes(n, id_flags, V);
}

int hpmn_scatter_plan_build(const void *ids, int32_t id_flags, int64_t n, int64_t V, void *workspace, size_t workspace_bytes,
                            int32_t *perm, int32_t *seg, int32_t *start, void *rows, int32_t *count,
                            const int64_t *row_bounds, int32_t nb, int32_t *counts, void *stream) {
    drop_stale_hip_error();
    if (n < 0 || n > 0x7fffffffLL || V < 1 || nb < 0 || nb > HPMN_MAX_CHUNKS) return HPMN_EINVAL;
    if (counts && (nb < 1 || (!row_bounds && nb != 1))) return HPMN_EINVAL;
    if (n == 0) return HPMN_OK;
    if (!ids || !workspace || !perm || !seg || !start || !rows || !count) return HPMN_EINVAL;
    return scatter_plan_build_launch(ids, id_flags, n, V, workspace, workspace_bytes, perm, seg, start, rows, count, row_bounds,
                                     nb, counts, (hipStream_t)stream);
}

size_t hpmn_embed_grad_segsum_partials_floats(int64_t n, int32_t E) { return n > 0 && E > 0 ? segsum_partials_floats(n, E) : 0; }
int hpmn_embed_grad_segsum_chunk(void) { return segsum_chunk_entries(); }

int hpmn_embed_grad_segsum(const HpmnScatterPlan *plan, const float *d_x, float *d_emb, int32_t B, int32_t T, int32_t F,
                           int32_t E, int32_t front_zero, int32_t id_flags, const float *d_last, int32_t t_last,
                           void *stream) {
    drop_stale_hip_error();
    if (!plan || B < 0 || T < 1 || F < 1 || E < 4 || front_zero < 0) return HPMN_EINVAL;
    if (E % 4 != 0 || (E / 4 & (E / 4 - 1)) != 0 || E / 4 > 64) return HPMN_EUNSUPPORTED;   // (64 / (E/4) lane groups per wave)
    if (plan->n != (int64_t)B * T * F || plan->n > 0x7fffffffLL) return HPMN_EINVAL;     // (perm / start are int32)
    if (B == 0) return HPMN_OK;
    if (!plan->perm || !plan->seg || !plan->start || !plan->rows || !plan->count || !plan->partials || !d_x) return HPMN_EINVAL;
    if (!plan->out_rows && !d_emb) return HPMN_EINVAL;
    return embed_grad_segsum_launch(*plan, d_x, d_emb, B, T, F, E, front_zero, id_flags, d_last, t_last, (hipStream_t)stream);
}

int hpmn_adam_step(float *param, const float *grad, float *m, float *v, int64_t n, float lr_t, float beta1,
                   float beta2, float eps, float clip, float grad_scale, void *stream) {
    drop_stale_hip_error();
    if (n < 0) return HPMN_EINVAL;
    if (n == 0) return HPMN_OK;
    if (!param || !grad || !m || !v) return HPMN_EINVAL;
    return adam_launch(param, grad, m, v, n, lr_t, beta1, beta2, eps, clip, grad_scale, (hipStream_t)stream);
}

int hpmn_adam_step_clear(float *param, float *grad, float *m, float *v, int64_t n, float lr_t, float beta1,
                         float beta2, float eps, float clip, float grad_scale, void *stream) {
    drop_stale_hip_error();
    if (n < 0) return HPMN_EINVAL;
    if (n == 0) return HPMN_OK;
    if (!param || !grad || !m || !v) return HPMN_EINVAL;
    return adam_clear_launch(param, grad, m, v, n, lr_t, beta1, beta2, eps, clip, grad_scale, (hipStream_t)stream);
}

int hpmn_adam_step_rows(float *param, const float *grad_rows, float *m, float *v, const int64_t *row_ids,
                        int64_t n_rows, int32_t E, float lr_t, float beta1, float beta2, float eps, float clip,
                        float grad_scale, void *stream) {
    drop_stale_hip_error();
    if (n_rows < 0 || E < 4) return HPMN_EINVAL;
    if (E % 4 != 0) return HPMN_EUNSUPPORTED;
    if (n_rows == 0) return HPMN_OK;
    if (!param || !grad_rows || !m || !v || !row_ids) return HPMN_EINVAL;
    return adam_rows_launch(param, grad_rows, m, v, row_ids, n_rows, E, lr_t, beta1, beta2, eps, clip, grad_scale,
                            (hipStream_t)stream);
}

int hpmn_table_mark_rows(const void *ids, int64_t n_ids, uint8_t *flags, int64_t V, int32_t id_flags, void *stream) {
    drop_stale_hip_error();
    if (n_ids < 0 || V < 1) return HPMN_EINVAL;
    if (n_ids == 0) return HPMN_OK;
    if (!ids || !flags) return HPMN_EINVAL;
    return table_mark_launch(ids, n_ids, flags, V, id_flags, (hipStream_t)stream);
}

int hpmn_adam_step_table(float *param, float *grad, float *m, float *v, uint8_t *flags, int64_t V, int32_t E,
                         int32_t pass, float lr_t, float beta1, float beta2, float eps, float clip, float grad_scale,
                         void *stream) {
    drop_stale_hip_error();
    if (V < 0 || E < 4 || (pass != 0 && pass != 1)) return HPMN_EINVAL;
    // (a row's E/4 lanes must sit in ONE wave: the lane that clears the row's flag does so after every lane of the row has
    //  read it only then)
    if (E % 4 != 0 || (E / 4 & (E / 4 - 1)) != 0 || E / 4 > 64) return HPMN_EUNSUPPORTED;
    if (V == 0) return HPMN_OK;
    if (!param || (pass == 1 && !grad) || !m || !v || !flags) return HPMN_EINVAL;     // (pass 0 never reads the gradient)
    if ((reinterpret_cast<uintptr_t>(param) | reinterpret_cast<uintptr_t>(grad) | reinterpret_cast<uintptr_t>(m) |
         reinterpret_cast<uintptr_t>(v)) & 15)
        return HPMN_EINVAL;
    return adam_table_launch(param, grad, m, v, flags, V, E, pass, lr_t, beta1, beta2, eps, clip, grad_scale,
                             (hipStream_t)stream);
}

int hpmn_rows_sum_adam(const HpmnRowsAdam *a, void *stream) {
    drop_stale_hip_error();
    if (!a || a->world < 1 || a->world > HPMN_MAX_RANKS || a->E < 4 || a->V < 1 || a->ids_stride < 0 || a->rows_stride < 0)
        return HPMN_EINVAL;
    if (a->E % 4 != 0 || (a->E / 4 & (a->E / 4 - 1)) != 0 || a->E / 4 > 64) return HPMN_EUNSUPPORTED;
    int64_t total = 0;
    for (int r = 0; r < a->world; ++r) {
        if (a->first[r] < 0 || a->n[r] < 0 || (!a->counts && a->len[r] < 0)) return HPMN_EINVAL;
        if (a->n[r] > a->rows_stride && a->world > 1) return HPMN_EINVAL;
        if (a->first[r] + a->n[r] > a->ids_stride && a->world > 1) return HPMN_EINVAL;
        total += a->n[r];
    }
    if (total == 0) return HPMN_OK;
    if (!a->ids || !a->rows || !a->flags || !a->param || !a->m || !a->v) return HPMN_EINVAL;
    if (a->counts && a->counts_stride < 1) return HPMN_EINVAL;
    if (a->bucket_start && (a->bucket_shift < 0 || a->bucket_shift > 62 || a->bucket_stride < ((a->V - 1) >> a->bucket_shift) + 2))
        return HPMN_EINVAL;
    if ((reinterpret_cast<uintptr_t>(a->param) | reinterpret_cast<uintptr_t>(a->m) | reinterpret_cast<uintptr_t>(a->v) |
         reinterpret_cast<uintptr_t>(a->rows)) & 15)
        return HPMN_EINVAL;
    return rows_sum_adam_launch(*a, (hipStream_t)stream);
}

int hpmn_table_mark_ranks(const void *ids, int64_t ids_stride, int32_t world, const int32_t *counts, int32_t counts_stride,
                          int64_t cap, uint8_t *flags, int64_t V, int32_t id_flags, int32_t *bucket_start, int64_t bucket_stride,
                          int32_t bucket_shift, void *stream) {
    drop_stale_hip_error();
    if (world < 1 || world > HPMN_MAX_RANKS || cap < 0 || V < 1 || ids_stride < cap) return HPMN_EINVAL;
    if (bucket_start && (bucket_shift < 0 || bucket_shift > 62 || bucket_stride < ((V - 1) >> bucket_shift) + 2)) return HPMN_EINVAL;
    if (cap == 0) return HPMN_OK;
    if (!ids || !flags || (counts && counts_stride < 1)) return HPMN_EINVAL;
    if (reinterpret_cast<uintptr_t>(flags) & 3) return HPMN_EINVAL;
    return table_mark_ranks_launch(ids, ids_stride, world, counts, counts_stride, cap, flags, V, id_flags, bucket_start,
                                   bucket_stride, bucket_shift, (hipStream_t)stream);
}

int hpmn_gru_fused_fwd_writes_last(void) { return gru_fused_fwd_writes_last() ? 1 : 0; }

int hpmn_gru_fused_fwd_supported(int32_t H, int32_t D, int32_t gather) {
    return gru_fused_fwd_supported(H, D, gather) ? 1 : 0;
}

int hpmn_gru_fused_fwd(const HpmnGruFusedFwd *a, void *stream) {
    drop_stale_hip_error();
    if (a == nullptr || a->B < 0 || a->T < 1 || a->period < 1) return HPMN_EINVAL;
    if (!a->wg || !a->bg || !a->wc || !a->bc || !a->h_last) return HPMN_EINVAL;
    if ((a->hs == nullptr) != (a->gates == nullptr)) return HPMN_EINVAL;
    if (a->x == nullptr) {
        if (!a->ids || !a->emb || a->F < 1 || a->E < 1 || a->F * a->E != a->D || a->Tids + a->front_zero != a->T ||
            a->front_zero < 0)
            return HPMN_EINVAL;
    }
    if (a->y != nullptr && a->T % a->period != 0) return HPMN_EINVAL;
    if (!gru_fused_fwd_supported(a->H, a->D, a->x == nullptr)) return HPMN_EUNSUPPORTED;
    if (a->B == 0) return HPMN_OK;
    return gru_fused_fwd_dispatch(*a, (hipStream_t)stream);
}

int hpmn_gru_pair_fwd_supported(int32_t H, int32_t D_lo, int32_t gather) {
    return gru_pair_fwd_supported(H, D_lo, gather) ? 1 : 0;
}

size_t hpmn_gru_pair_fwd_scratch_bytes(void) { return gru_pair_fwd_scratch_bytes(); }

size_t hpmn_gru_proj_image_floats(int32_t D) { return (D == 32 || D == 64) ? gru_proj_image_floats(D) : 0; }

int hpmn_gru_proj_images(int32_t n, const float *const *wg, const float *const *bg, const float *const *wc,
                         const float *const *bc, const int32_t *D, float *const *img, void *stream) {
    drop_stale_hip_error();
    if (n < 0 || n > HPMN_MAX_LAYERS || (n > 0 && (!wg || !bg || !wc || !bc || !D || !img))) return HPMN_EINVAL;
    if (n == 0) return HPMN_OK;
    for (int i = 0; i < n; ++i)
        if (!wg[i] || !bg[i] || !wc[i] || !bc[i] || !img[i] || (reinterpret_cast<size_t>(img[i]) & 15) != 0) return HPMN_EINVAL;
    return gru_proj_images_launch(n, wg, bg, wc, bc, D, img, (hipStream_t)stream);
}

static int fused_args_ok(const HpmnGruFusedFwd *a, bool needs_input) {
    if (a->B < 0 || a->T < 1 || a->period < 1) return HPMN_EINVAL;
    if (!a->wg || !a->bg || !a->wc || !a->bc || !a->h_last) return HPMN_EINVAL;
    if ((a->hs == nullptr) != (a->gates == nullptr)) return HPMN_EINVAL;
    if (needs_input && a->x == nullptr) {
        if (!a->ids || !a->emb || a->F < 1 || a->E < 1 || a->F * a->E != a->D || a->Tids + a->front_zero != a->T ||
            a->front_zero < 0)
            return HPMN_EINVAL;
    }
    if (a->y != nullptr && a->T % a->period != 0) return HPMN_EINVAL;
    return HPMN_OK;
}

int hpmn_gru_pair_fwd(const HpmnGruPairFwd *p, void *stream) {
    drop_stale_hip_error();
    if (p == nullptr || (reinterpret_cast<size_t>(p->scratch) & 15) != 0) return HPMN_EINVAL;
    if ((reinterpret_cast<size_t>(p->img_lo) & 15) != 0 || (reinterpret_cast<size_t>(p->img_up) & 15) != 0) return HPMN_EINVAL;
    if (p->scratch == nullptr && (p->img_up == nullptr || (p->lo.D > 32 && p->img_lo == nullptr))) return HPMN_EINVAL;
    int rc = fused_args_ok(&p->lo, true);
    if (rc != HPMN_OK) return rc;
    rc = fused_args_ok(&p->up, false);
    if (rc != HPMN_OK) return rc;
    if (p->lo.T % p->lo.period != 0 || p->up.T != p->lo.T / p->lo.period || p->up.B != p->lo.B) return HPMN_EINVAL;
    if ((p->lo.hs == nullptr) != (p->up.hs == nullptr)) return HPMN_EINVAL;
    if (p->up.D != p->lo.H || p->up.H != p->lo.H || p->up.last != nullptr) return HPMN_EINVAL;
    if (!gru_pair_fwd_supported(p->lo.H, p->lo.D, p->lo.x == nullptr)) return HPMN_EUNSUPPORTED;
    if (p->lo.x == nullptr && p->lo.E % 4 != 0) return HPMN_EUNSUPPORTED;
    if (p->lo.B == 0) return HPMN_OK;
    return gru_pair_fwd_launch(p->lo, p->up, p->flags, reinterpret_cast<float *>(p->scratch), p->img_lo, p->img_up,
                               (hipStream_t)stream);
}

int hpmn_gru_pair_bwd_supported(int32_t H, int32_t D_lo) { return gru_pair_bwd_supported(H, D_lo) ? 1 : 0; }

int hpmn_gru_pair_bwd(const HpmnGruPairBwd *p, void *stream) {
    drop_stale_hip_error();
    if (p == nullptr) return HPMN_EINVAL;
    const HpmnGruBwd *both[2] = {&p->lo, &p->up};
    for (const HpmnGruBwd *a : both) {
        if (a->B < 0 || a->T < 1 || a->D < 1 || a->H < 1 || a->period < 1) return HPMN_EINVAL;
        if (!a->wg || !a->wc || !a->hs || !a->gates || !a->d_h_last || !a->d_act) return HPMN_EINVAL;
        if (a->t_begin != 0 || a->t_end != 0) return HPMN_EINVAL;
    }
    if (p->up.B != p->lo.B || p->up.H != p->lo.H || p->up.D != p->lo.H) return HPMN_EINVAL;
    if (p->lo.T % p->lo.period != 0 || p->up.T != p->lo.T / p->lo.period) return HPMN_EINVAL;
    if (p->up.d_y != nullptr && p->up.T % p->up.period != 0) return HPMN_EINVAL;
    if (!gru_pair_bwd_supported(p->lo.H, p->lo.D)) return HPMN_EUNSUPPORTED;
    if (((p->lo.flags | p->up.flags) & HPMN_BWD_CANDIDATE_FROM_HS) && !gru_candidate_elision(p->lo.H, p->lo.B)) return HPMN_EUNSUPPORTED;
    if (p->lo.B == 0) return HPMN_OK;
    return gru_pair_bwd_launch(p->lo, p->up, p->flags, (hipStream_t)stream);
}

int hpmn_memory_update(const HpmnOnlineUpdate *a, void *stream) {
    drop_stale_hip_error();
    if (a == nullptr || a->B < 0 || a->K < 1 || a->K > HPMN_MAX_LAYERS || a->D < 1 || a->D > 128) return HPMN_EINVAL;
    if (a->B == 0) return HPMN_OK;
    if (a->user == nullptr || a->x == nullptr || a->state == nullptr || a->count == nullptr) return HPMN_EINVAL;
    for (int i = 0; i < a->K; ++i) {
        if (a->wg[i] == nullptr || a->bg[i] == nullptr || a->wc[i] == nullptr || a->bc[i] == nullptr) return HPMN_EINVAL;
        if (i + 1 < a->K && a->periods[i] < 1) return HPMN_EINVAL;
    }
    if (a->H != 32 && a->H != 64 && a->H != 128) return HPMN_EUNSUPPORTED;
    return memory_update_launch(*a, (hipStream_t)stream);
}

}  // extern "C"
