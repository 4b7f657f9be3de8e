// Internal interface of gru32_all.hip (H = 32: all layers of build_memory in one launch per direction).
#pragma once
#include "common.h"

namespace hpmn {

constexpr int AMAXK = 7;         // layer waves per workgroup (+ 1 loader = 8 waves)

struct All32Args {
    int32_t B, Tids, F, E, K, front_zero, mask_id0, last_t;
    int64_t V;
    int32_t period[AMAXK], len[AMAXK];
    const void *ids;
    const float *emb;
    const float *wg[AMAXK], *bg[AMAXK], *wc[AMAXK], *bc[AMAXK];
    float *memory;               // [B, K, 32]
    float *last;                 // [B, D0] or NULL
    // training (all NULL in inference)
    float *x0;                   // [B, T0, D0]
    float *hs[AMAXK], *gates[AMAXK], *y[AMAXK];
    // reverse pass
    const float *d_memory;       // [B, K, 32]
    float *d_act[AMAXK];         // [B, T_i, 96]
    float *d_x0;                 // [B, T0, D0]
};

bool gru32_all_supported(int H, int D0, int K, int E);
int gru32_fwd_all_launch(const All32Args &a, int D0, bool train, hipStream_t st);
int gru32_bwd_all_launch(const All32Args &a, int D0, hipStream_t st);
size_t gru32_wgrad_all_workspace_bytes(int B, int K, const int *D);
int gru32_wgrad_all_launch(int B, int K, const int *D, const int *T, const float *const *x, const float *const *hs,
                           const float *const *gates, const float *const *d_act, float *const *d_wg, float *const *d_bg,
                           float *const *d_wc, float *const *d_bc, float *workspace, hipStream_t st);

// HPMN_ALL32=0 selects the per-layer kernels
inline bool gru32_all_enabled() {
    static const int on = [] { const char *e = getenv("HPMN_ALL32"); return e ? atoi(e) : 1; }();
    return on != 0;
}

inline void gru32_all_fill(All32Args &a, const HpmnScanDesc &d, const int32_t *len, const void *ids, const float *emb,
                           const float *const *wg, const float *const *bg, const float *const *wc, const float *const *bc,
                           float *memory, float *last) {
    a.B = d.B; a.Tids = d.T; a.F = d.F; a.E = d.E; a.K = d.K; a.front_zero = d.front_zero; a.mask_id0 = d.mask_id0;
    a.V = d.V;
    a.last_t = len[0] + d.last_index;
    for (int i = 0; i < d.K; ++i) {
        a.period[i] = d.periods[i]; a.len[i] = len[i];
        a.wg[i] = wg[i]; a.wc[i] = wc[i];
        if (bg) a.bg[i] = bg[i];
        if (bc) a.bc[i] = bc[i];
    }
    a.ids = ids; a.emb = emb; a.memory = memory; a.last = last;
}

}  // namespace hpmn
