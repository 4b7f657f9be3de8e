// extern "C" surface of the all-layers-in-one-launch scan (gru_pipe_fwd.hip / gru_pipe_bwd.hip).
#include "pipe_common.h"

namespace hpmn {
int embed_gather_sum_launch(const void *ids, const float *emb, float *out, int32_t B, int32_t T, int32_t F, int32_t E,
                            int32_t mask_id0, hipStream_t st);
size_t pipe_sync_bytes(int K, int ntiles);
bool pipe_shape_supported(int H, int D);
int pipe_fwd_launch(const PipeArgs &a, int num_cus, hipStream_t st);
int pipe_bwd_launch(const PipeArgs &a, int num_cus, hipStream_t st);
int embed_gather_seq_launch(const void *ids, const float *emb, float *out, int B, int Tids, int F, int E,
                            int front_zero, int mask_id0, hipStream_t st);

struct Tile128Args {
    int B, T, D, period;
    const float *x, *xp;
    const float *wg, *bg, *wc, *bc;
    float *y;
    float *h_last;
    long h_last_stride;
};
bool tile128_supported(int H, int D);
int tile128_fwd_launch(const Tile128Args &a, hipStream_t st);
bool tile64_supported(int H, int D);
int tile64_fwd_launch(const Tile128Args &a, hipStream_t st);

static constexpr size_t DUMP_BYTES = 16384;   // 256 lanes x 16 B, plus the +2H float offsets of the gate stores

static int device_cus() {
    static int cus = 0;
    if (cus == 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0)
            cus = n;
        else
            cus = 256;
    }
    return cus;
}

static int fill_args(const HpmnPipe &p, bool bwd, PipeArgs &a) {
    if (p.B < 1 || p.K < 1 || p.K > HPMN_MAX_LAYERS || p.H < 1) return HPMN_EINVAL;
    if (!p.sync || !p.x0) return HPMN_EINVAL;
    a.B = p.B; a.K = p.K; a.ntiles = (p.B + TS - 1) / TS; a.train = p.train;
    char *base = reinterpret_cast<char *>(p.sync);
    a.sync = reinterpret_cast<unsigned *>(base + (bwd ? (pipe_sync_bytes(p.K, a.ntiles) + 255) / 256 * 256 : 0));
    a.dump = reinterpret_cast<float *>(base + 2 * ((pipe_sync_bytes(p.K, a.ntiles) + 255) / 256 * 256));
    for (int i = 0; i < p.K; ++i) {
        PipeLayer &L = a.L[i];
        L = PipeLayer{};
        if (p.T[i] < 1 || p.period[i] < 1 || p.D[i] < 1) return HPMN_EINVAL;
        if (!pipe_shape_supported(p.H, p.D[i])) return HPMN_EUNSUPPORTED;
        if (i > 0 && p.D[i] != p.H) return HPMN_EINVAL;
        if (i + 1 < p.K && (p.T[i] % p.period[i] != 0 || p.T[i + 1] != p.T[i] / p.period[i])) return HPMN_EINVAL;
        if (!p.wg[i] || !p.wc[i]) return HPMN_EINVAL;
        L.wg = p.wg[i]; L.bg = p.bg[i]; L.wc = p.wc[i]; L.bc = p.bc[i];
        L.T = p.T[i]; L.D = p.D[i];
        // the top layer of the call has no subsampled output -- unless the caller hands it a y buffer (forward only, r4): the
        // call then is a GROUP of layers of a taller stack and the next call's x0 is this y (ops.tiled_forward_inference)
        const bool top_y = !bwd && i + 1 == p.K && p.y[i] != nullptr;
        if (top_y && p.T[i] % p.period[i] != 0) return HPMN_EINVAL;
        L.period = (i + 1 < p.K || top_y) ? p.period[i] : 1;
        L.x = i == 0 ? p.x0 : p.y[i - 1];
        L.y = (i + 1 < p.K || top_y) ? p.y[i] : nullptr;
        if (i + 1 < p.K && !p.y[i]) return HPMN_EINVAL;
        L.hs = p.hs[i]; L.gates = p.gates[i];
        L.h_last_stride = p.mem_stride > 0 ? (long)p.mem_stride : (long)p.K * p.H;
        if (!bwd) {
            if (!p.bg[i] || !p.bc[i] || !p.memory) return HPMN_EINVAL;
            if (p.train && (!p.hs[i] || !p.gates[i])) return HPMN_EINVAL;
            L.h_last = p.memory + (size_t)i * p.H;
        } else {
            if (!p.hs[i] || !p.gates[i] || !p.d_act[i] || !p.d_memory) return HPMN_EINVAL;
            if (i > 0 && !p.d_x[i]) return HPMN_EINVAL;
            L.d_h_last = p.d_memory + (size_t)i * p.H;
            L.d_act = p.d_act[i];
            L.d_x = i > 0 ? p.d_x[i] : nullptr;
            L.d_y = i + 1 < p.K ? p.d_x[i + 1] : nullptr;
        }
    }
    return HPMN_OK;
}
}  // namespace hpmn

using namespace hpmn;

extern "C" {

int hpmn_pipe_supported(int32_t H, int32_t D0) { return pipe_shape_supported(H, D0) ? 1 : 0; }

size_t hpmn_pipe_sync_bytes(int32_t K, int32_t B) {
    if (K < 1 || K > HPMN_MAX_LAYERS || B < 1) return 0;
    const size_t one = (pipe_sync_bytes(K, (B + TS - 1) / TS) + 255) / 256 * 256;
    return 2 * one + DUMP_BYTES;
}

int hpmn_pipe_fwd(const HpmnPipe *p, void *stream) {
    (void)hipGetLastError();
    if (!p) return HPMN_EINVAL;
    if (p->B == 0) return HPMN_OK;
    PipeArgs a;
    const int rc = fill_args(*p, false, a);
    if (rc != HPMN_OK) return rc;
    return pipe_fwd_launch(a, device_cus(), (hipStream_t)stream);
}

int hpmn_pipe_bwd(const HpmnPipe *p, void *stream) {
    (void)hipGetLastError();
    if (!p) return HPMN_EINVAL;
    if (p->B == 0) return HPMN_OK;
    PipeArgs a;
    const int rc = fill_args(*p, true, a);
    if (rc != HPMN_OK) return rc;
    return pipe_bwd_launch(a, device_cus(), (hipStream_t)stream);
}

int hpmn_tile_supported(int32_t H, int32_t D) { return (tile128_supported(H, D) || tile64_supported(H, D)) ? 1 : 0; }

int hpmn_tile_fwd(const HpmnTileFwd *p, void *stream) {
    (void)hipGetLastError();
    if (!p || p->B < 0 || p->T < 1 || p->period < 1 || p->D < 1 || p->H < 1) return HPMN_EINVAL;
    if (!tile128_supported(p->H, p->D) && !tile64_supported(p->H, p->D)) return HPMN_EUNSUPPORTED;
    if (p->B == 0) return HPMN_OK;
    if ((p->x == nullptr) == (p->xp == nullptr)) return HPMN_EINVAL;          // exactly one of the two
    if (p->H == 64 && p->xp != nullptr) return HPMN_EUNSUPPORTED;              // (H = 64 always projects in the kernel)
    if (!p->wg || !p->wc || !p->h_last || (p->x && (!p->bg || !p->bc))) return HPMN_EINVAL;
    if (p->y && p->T % p->period != 0) return HPMN_EINVAL;
    Tile128Args a;
    a.B = p->B; a.T = p->T; a.D = p->D; a.period = p->period;
    a.x = p->x; a.xp = p->xp; a.wg = p->wg; a.bg = p->bg; a.wc = p->wc; a.bc = p->bc;
    a.y = p->y; a.h_last = p->h_last; a.h_last_stride = (long)p->h_last_stride;
    return p->H == 64 ? tile64_fwd_launch(a, (hipStream_t)stream) : tile128_fwd_launch(a, (hipStream_t)stream);
}

int hpmn_embed_gather_seq(const void *ids, const float *emb, float *out, int32_t B, int32_t Tids, int32_t F,
                          int32_t E, int32_t front_zero, int64_t V, int32_t mask_id0, void *stream) {
    (void)hipGetLastError();
    if (B < 0 || Tids < 1 || F < 1 || E < 4 || front_zero < 0 || V < 1) return HPMN_EINVAL;
    if (E % 4 != 0) return HPMN_EUNSUPPORTED;
    if (B == 0) return HPMN_OK;
    if (!ids || !emb || !out) return HPMN_EINVAL;
    return embed_gather_seq_launch(ids, emb, out, B, Tids, F, E, front_zero, mask_id0, (hipStream_t)stream);
}

/* out[b, f*E:(f+1)*E] += sum_t emb[ids[b,t,f]] (id-0 mask as in hpmn_embed_gather): the gather consumed in place. */
int hpmn_embed_gather_sum(const void *ids, const float *emb, float *out, int32_t B, int32_t T, int32_t F, int32_t E,
                          int64_t V, int32_t mask_id0, void *stream) {
    (void)hipGetLastError();
    if (B < 0 || T < 1 || F < 1 || E < 4 || V < 1) return HPMN_EINVAL;
    if (B == 0) return HPMN_OK;
    if (!ids || !emb || !out) return HPMN_EINVAL;
    return embed_gather_sum_launch(ids, emb, out, B, T, F, E, mask_id0, (hipStream_t)stream);
}

}  // extern "C"
