// TF-form Adam with per-element gradient clip, one launch over a flat parameter buffer.
// HBM-bound: 16 B read (p,g,m,v) + 12 B written (p,m,v) per element, float4-vectorised.
#include <cstdlib>

#include "common.h"

namespace hpmn {

__device__ __forceinline__ void adam_elem(float &p, float g, float &m, float &v, float lr_t, float b1,
                                          float b2, float eps, float clip, float gs) {
    g *= gs;
    g = fminf(fmaxf(g, -clip), clip);
    m = fmaf(b1, m, (1.f - b1) * g);
    v = fmaf(b2, v, (1.f - b2) * g * g);
    p -= lr_t * m / (sqrtf(v) + eps);
}

// CLEAR (hpmn_adam_step_clear, ABI v13): the gradient is CONSUMED -- written back as zeros -- so that the caller's next step
// needs no clearing launch in front of its first kernel (the Amazon step began with a 6.4 us fill of its 16 MB gradient buffer
// on the only stream it has; +4 B per element here, in a launch that streams 28)
template <bool CLEAR>
__global__ __launch_bounds__(256) void adam_kernel_v4(float4 *__restrict__ p, float4 *__restrict__ g,
                                                      float4 *__restrict__ m, float4 *__restrict__ v, long n4,
                                                      float lr_t, float b1, float b2, float eps, float clip,
                                                      float gs) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 pp = p[i], mm = m[i], vv = v[i];
        const float4 gg = g[i];
        if (CLEAR) g[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        adam_elem(pp.x, gg.x, mm.x, vv.x, lr_t, b1, b2, eps, clip, gs);
        adam_elem(pp.y, gg.y, mm.y, vv.y, lr_t, b1, b2, eps, clip, gs);
        adam_elem(pp.z, gg.z, mm.z, vv.z, lr_t, b1, b2, eps, clip, gs);
        adam_elem(pp.w, gg.w, mm.w, vv.w, lr_t, b1, b2, eps, clip, gs);
        p[i] = pp; m[i] = mm; v[i] = vv;
    }
}

template <bool CLEAR>
__global__ __launch_bounds__(256) void adam_kernel_v1(float *__restrict__ p, float *__restrict__ g,
                                                      float *__restrict__ m, float *__restrict__ v, long n,
                                                      float lr_t, float b1, float b2, float eps, float clip,
                                                      float gs) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float pp = p[i], mm = m[i], vv = v[i];
        const float gi = g[i];
        if (CLEAR) g[i] = 0.f;
        adam_elem(pp, gi, mm, vv, lr_t, b1, b2, eps, clip, gs);
        p[i] = pp; m[i] = mm; v[i] = vv;
    }
}

// Row-wise ("lazy") variant for tables too large for the dense update: only the rows listed in row_ids are
// touched; their gradients come from a COMPACT buffer g[u, :] (row u of it belongs to table row row_ids[u]).
// One float4 per thread, E/4 adjacent lanes per row.  A DEVIATION from the reference's dense TF Adam (rows with
// a zero gradient but non-zero moments do not move) -- the semantics of tf.contrib.opt.LazyAdamOptimizer.
__global__ __launch_bounds__(256) void adam_rows_kernel(float4 *__restrict__ p, const float4 *__restrict__ g,
                                                        float4 *__restrict__ m, float4 *__restrict__ v,
                                                        const int64_t *__restrict__ row_ids, long n_rows, int E4,
                                                        float lr_t, float b1, float b2, float eps, float clip,
                                                        float gs) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n_rows * E4; i += stride) {
        const long u = i / E4;
        const long j = row_ids[u] * E4 + (i - u * E4);
        float4 pp = p[j], mm = m[j], vv = v[j];
        const float4 gg = g[i];
        adam_elem(pp.x, gg.x, mm.x, vv.x, lr_t, b1, b2, eps, clip, gs);
        adam_elem(pp.y, gg.y, mm.y, vv.y, lr_t, b1, b2, eps, clip, gs);
        adam_elem(pp.z, gg.z, mm.z, vv.z, lr_t, b1, b2, eps, clip, gs);
        adam_elem(pp.w, gg.w, mm.w, vv.w, lr_t, b1, b2, eps, clip, gs);
        p[j] = pp; m[j] = mm; v[j] = vv;
    }
}

// Dense table Adam in TWO passes with identical arithmetic (include/hpmn_hip.h, hpmn_adam_step_table): the rows a batch
// does not touch have an exactly-zero gradient, so their update (m = b1 m, v = b2 v, p -= lr_t m / (sqrt v + eps)) needs
// nothing of the step and runs early on another stream; the touched rows follow behind the scatter.
// (ids outside [0, V) are skipped: padding entries of a gathered id list are -1, and a raw-ABI caller's bad id must not
//  write outside the flags)
__global__ __launch_bounds__(256) void table_mark_kernel(const void *__restrict__ ids, long n, uint8_t *__restrict__ flags,
                                                         long V, int id_flags) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const long id = load_id(ids, i, id_flags);
        if (id >= 0 && id < V) flags[id] = 1;
    }
}

// PASS 0: rows with flag == 0 (gradient taken as zero, not read).  PASS 1: rows with flag != 0: the gradient row is
// consumed and CLEARED and so is the flag, which leaves the gradient table and the flags all-zero for the next step.
// One float4 per thread, E4 = E/4 adjacent lanes per row (256 % E4 == 0: a row never straddles a wave, so every lane
// of a row has read the flag before one of them clears it).
template <int PASS>
__global__ __launch_bounds__(256) void adam_table_kernel(float4 *__restrict__ p, float4 *__restrict__ g,
                                                         float4 *__restrict__ m, float4 *__restrict__ v,
                                                         uint8_t *__restrict__ flags, long n4, int e4_shift, float lr_t,
                                                         float b1, float b2, float eps, float clip, float gs) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const long row = i >> e4_shift;
        const bool marked = flags[row] != 0;
        if (marked != (PASS == 1)) continue;
        float4 pp = p[i], mm = m[i], vv = v[i];
        float4 gg = make_float4(0.f, 0.f, 0.f, 0.f);
        if (PASS == 1) {
            gg = g[i];
            g[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        adam_elem(pp.x, gg.x, mm.x, vv.x, lr_t, b1, b2, eps, clip, gs);
        adam_elem(pp.y, gg.y, mm.y, vv.y, lr_t, b1, b2, eps, clip, gs);
        adam_elem(pp.z, gg.z, mm.z, vv.z, lr_t, b1, b2, eps, clip, gs);
        adam_elem(pp.w, gg.w, mm.w, vv.w, lr_t, b1, b2, eps, clip, gs);
        p[i] = pp; m[i] = mm; v[i] = vv;
        if (PASS == 1 && (i & ((1L << e4_shift) - 1)) == 0) flags[row] = 0;
    }
}

int table_mark_launch(const void *ids, int64_t n, uint8_t *flags, int64_t V, int32_t id_flags, hipStream_t st) {
    if (n == 0) return HPMN_OK;
    long blocks = (n + 255) / 256;
    if (blocks > 256L * 8) blocks = 256L * 8;
    hipLaunchKernelGGL(table_mark_kernel, dim3((unsigned)blocks), dim3(256), 0, st, ids, (long)n, flags, (long)V, (int)id_flags);
    return check_launch();
}

int adam_table_launch(float *p, float *g, float *m, float *v, uint8_t *flags, int64_t V, int E, int pass, float lr_t,
                      float b1, float b2, float eps, float clip, float gs, hipStream_t st) {
    if (V == 0) return HPMN_OK;
    int shift = 0;
    while ((1 << shift) < E / 4) ++shift;
    const long n4 = V * (long)(E / 4);
    long blocks = (n4 + 255) / 256;
    // Pass 0 has the whole forward and BPTT to finish in and shares the chip with their latency-critical waves: a
    // THIN grid (HPMN_ADAM_EARLY_WGS workgroups, default 256 -- one per CU) trickles through the table instead
    // of flooding every SIMD and the memory system at once (measured: with 4096 workgroups the layer-0 forward beside
    // it took 585 instead of 483 us).  Pass 1 is on the serial tail: full width.
    // (256, not the 192 of round 2: the pass has to be DONE when layer 0's forward launch is -- the two-layer launches
    //  that follow it hold every register of their CUs and stretch by whatever still runs beside them; C3 step
    //  2.823 / 2.771 / 2.786 ms at 192 / 256 / 320.)
    // The thin grid moves ~2.5 TB/s: right for the 1.2 GB of the reference tables (0.45 ms, layer 0's forward takes as long),
    // hopeless for a table sized to HBM (256 M rows = 98 GB of optimiser traffic: 45 ms against a 10 ms chain --
    // measured 43 vs 33 ms/step) -- so it widens in proportion, up to the full width.
    static const long early = [] { const char *e = getenv("HPMN_ADAM_EARLY_WGS"); return e ? atol(e) : 256L; }();
    long thin = early > 0 ? early : 256L * 16;
    const double gb = (double)V * E * 24.0 / 1.2e9;
    if (gb > 1.0) thin = (long)(thin * gb);
    if (thin > 256L * 16) thin = 256L * 16;
    const long cap = pass == 0 ? thin : 256L * 16;
    if (blocks > cap) blocks = cap;
    // (r4, measured and removed: the late pass compacted per wave -- 64 flags in one load, a ballot, the marked rows processed
    //  densely 16 at a time -- 137 us in the step's tail where this strided loop takes 124: a wave's marked rows went through
    //  one dependent row round trip per group, the strided loop has every lane's row loads in flight at once)
    if (pass == 0)
        hipLaunchKernelGGL(adam_table_kernel<0>, dim3((unsigned)blocks), dim3(256), 0, st, (float4 *)p, (float4 *)g,
                           (float4 *)m, (float4 *)v, flags, n4, shift, lr_t, b1, b2, eps, clip, gs);
    else
        hipLaunchKernelGGL(adam_table_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, st, (float4 *)p, (float4 *)g,
                           (float4 *)m, (float4 *)v, flags, n4, shift, lr_t, b1, b2, eps, clip, gs);
    return check_launch();
}

int adam_rows_launch(float *p, const float *g, float *m, float *v, const int64_t *row_ids, int64_t n_rows, int E,
                     float lr_t, float b1, float b2, float eps, float clip, float gs, hipStream_t st) {
    if (n_rows == 0) return HPMN_OK;
    const long n4 = n_rows * (E / 4);
    long blocks = (n4 + 255) / 256;
    if (blocks > 256L * 16) blocks = 256L * 16;
    hipLaunchKernelGGL(adam_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (float4 *)p, (const float4 *)g,
                       (float4 *)m, (float4 *)v, row_ids, (long)n_rows, E / 4, lr_t, b1, b2, eps, clip, gs);
    return check_launch();
}

template <bool CLEAR>
static int adam_launch_t(float *p, float *g, float *m, float *v, int64_t n, float lr_t, float b1, float b2,
                         float eps, float clip, float gs, hipStream_t st) {
    if (n == 0) return HPMN_OK;
    const bool aligned = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) |
                           reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v)) & 15) == 0;
    if (aligned && n % 4 == 0) {
        const long n4 = n / 4;
        long blocks = (n4 + 255) / 256;
        if (blocks > 256L * 16) blocks = 256L * 16;
        hipLaunchKernelGGL(adam_kernel_v4<CLEAR>, dim3((unsigned)blocks), dim3(256), 0, st, (float4 *)p,
                           (float4 *)g, (float4 *)m, (float4 *)v, n4, lr_t, b1, b2, eps, clip, gs);
    } else {
        long blocks = (n + 255) / 256;
        if (blocks > 256L * 16) blocks = 256L * 16;
        hipLaunchKernelGGL(adam_kernel_v1<CLEAR>, dim3((unsigned)blocks), dim3(256), 0, st, p, g, m, v, (long)n, lr_t,
                           b1, b2, eps, clip, gs);
    }
    return check_launch();
}

int adam_launch(float *p, const float *g, float *m, float *v, int64_t n, float lr_t, float b1, float b2,
                float eps, float clip, float gs, hipStream_t st) {
    return adam_launch_t<false>(p, const_cast<float *>(g), m, v, n, lr_t, b1, b2, eps, clip, gs, st);      // (never written)
}
int adam_clear_launch(float *p, float *g, float *m, float *v, int64_t n, float lr_t, float b1, float b2,
                      float eps, float clip, float gs, hipStream_t st) {
    return adam_launch_t<true>(p, g, m, v, n, lr_t, b1, b2, eps, clip, gs, st);
}

}  // namespace hpmn
