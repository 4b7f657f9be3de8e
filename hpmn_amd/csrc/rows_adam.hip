// The table update of a training step from COMPACT gradient rows: no dense [V, E] gradient table (include/hpmn_hip.h,
// hpmn_rows_sum_adam / hpmn_table_mark_ranks).
//
// Reference semantics (code/hpmn.py:204-214): the embedding gradient -- IndexedSlices over the batch's rows -- is densified,
// clipped per element and fed to a dense TF-form Adam over the whole table.  A row nobody touches has an exactly-zero
// gradient and takes the early pass of adam.hip (hpmn_adam_step_table, pass 0); what is left for the step's serial tail is
// the update of the TOUCHED rows, and those exist compactly: the deterministic scatter (scatter_sorted.hip) writes one summed
// gradient row per distinct table row of the batch, ascending.  Under data parallel every rank holds such a
// list; after the all-gather the buffers are
//     ids  [world, ids_stride]      rank r's distinct rows, ascending, the first len[r] entries valid
//     rows [world, rows_stride, E]  their gradient rows
// and ONE launch does what the host framework needed ~60 kernels for: per distinct row of the union, the
// ranks' rows are added in RANK ORDER 0..world-1 (the same addends in the same order on every replica: bit-identical tables),
// clipped, and the TF-form Adam update is applied to param / m / v in place.
//
// Who owns a row of the union, and where it sits in the other ranks' lists: flags[row] -- the byte the two-pass table Adam
// keeps per table row -- carries one BIT PER RANK (hpmn_table_mark_ranks, run from the early-gathered lists underneath the
// forward).  The entry (r, j) whose rank is the LOWEST set bit owns the row; it looks the row up in the lists of the higher
// ranks whose bits are set (a binary search each: the lists are L2-resident, the probes of adjacent entries coincide in their
// first levels), adds, updates, and clears the byte -- which leaves the flags all-zero for the next step, as pass 1 of
// hpmn_adam_step_table does.  Entries of non-owners read one id and one byte and leave.
// world == 1 is the single-GPU tail: scatter -> this, instead of scatter-into-dense-table -> hpmn_adam_step_table pass 1.
#include "common.h"

namespace hpmn {

constexpr int RMAX = HPMN_MAX_RANKS;

struct RowsAdamK {
    int world, E4, id_flags, counts_stride;
    const void *ids;
    long ids_stride;
    const int *counts;            // device list lengths (optional)
    long len[RMAX], first[RMAX], n[RMAX];
    long ent_off[RMAX + 1];       // prefix sums of n[]
    const float4 *rows;
    long rows_stride;             // in rows
    uint8_t *flags;
    float4 *p, *m, *v;
    long V;
    float lr_t, b1, b2, eps, clip, gs;
    const int *bstart;            // optional bucket index over the lists: [world, bstride], see table_mark_ranks_kernel
    long bstride;
    int bshift;
};

__device__ __forceinline__ void adam_elem_r(float &p, float g, float &m, float &v, float lr_t, float b1, float b2, float eps,
                                            float clip, float gs) {
    // (the arithmetic of adam.hip's adam_elem, operation for operation: the two-pass dense update and this one must agree
    //  bit for bit on the same gradient row)
    g *= gs;
    g = fminf(fmaxf(g, -clip), clip);
    m = fmaf(b1, m, (1.f - b1) * g);
    v = fmaf(b2, v, (1.f - b2) * g * g);
    p -= lr_t * m / (sqrtf(v) + eps);
}

// first index in [0, len) of the ascending list with list[i] >= x
__device__ __forceinline__ long lower_bound_id(const void *__restrict__ ids, long base, long len, long x, int id_flags) {
    long lo = 0, hi = len;
    while (lo < hi) {
        const long mid = (lo + hi) >> 1;
        const long y = load_id(ids, base + mid, id_flags);
        if (y < x) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// world == 1: every entry owns its row.  E4 adjacent lanes per entry.
__global__ __launch_bounds__(256) void rows_adam_one_kernel(RowsAdamK a) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long i = t / a.E4;
    const int e4 = (int)(t - i * a.E4);
    if (i >= a.n[0]) return;
    const long j = a.first[0] + i;
    const long len = a.counts ? (long)a.counts[0] : a.len[0];
    if (j >= len) return;
    const long x = load_id(a.ids, j, a.id_flags);
    if (x < 0 || x >= a.V) return;
    const long xi = x * a.E4 + e4;
    float4 pp = a.p[xi], mm = a.m[xi], vv = a.v[xi];
    const float4 acc = a.rows[i * a.E4 + e4];
    adam_elem_r(pp.x, acc.x, mm.x, vv.x, a.lr_t, a.b1, a.b2, a.eps, a.clip, a.gs);
    adam_elem_r(pp.y, acc.y, mm.y, vv.y, a.lr_t, a.b1, a.b2, a.eps, a.clip, a.gs);
    adam_elem_r(pp.z, acc.z, mm.z, vv.z, a.lr_t, a.b1, a.b2, a.eps, a.clip, a.gs);
    adam_elem_r(pp.w, acc.w, mm.w, vv.w, a.lr_t, a.b1, a.b2, a.eps, a.clip, a.gs);
    a.p[xi] = pp; a.m[xi] = mm; a.v[xi] = vv;
    if (e4 == 0) a.flags[x] = 0;
}

// world > 1.  A wave takes 64 consecutive entries (of the concatenated windows) in two phases:
//   1. ONE LANE PER ENTRY: id, flags byte, ownership, and the row's position in the list of every higher rank that holds it --
//      the (up to W - 1) binary searches of a lane advance IN LOCKSTEP, one probe of each per round, so their loads are in
//      flight together (the first version ran them one after the other with E/4 lanes repeating each: 705 us at world = 8 for
//      work whose traffic takes 330);
//   2. E/4 LANES PER ENTRY, 64 / (E/4) entries per round: the entry's facts arrive by shuffle, the ranks' rows are added in
//      rank order, clip + Adam, flag byte cleared.
template <int W>
__global__ __launch_bounds__(256) void rows_sum_adam_kernel(RowsAdamK a) {
    const int lane = threadIdx.x & 63;
    const long wave_e0 = ((long)blockIdx.x * blockDim.x + (threadIdx.x & ~63));
    const long total = a.ent_off[a.world];
    if (wave_e0 >= total) return;
    // ---- phase 1
    const long e = wave_e0 + lane;
    int r = 0;
#pragma unroll
    for (int k = 1; k < W; ++k) r += (k < a.world && e >= a.ent_off[k]) ? 1 : 0;
    long x = -1;
    unsigned mask = 0u;                                   // the higher ranks holding the row; 0: not an owner's entry
    bool own = false;
    if (e < total) {
        const long j = a.first[r] + (e - a.ent_off[r]);
        const long len_r = a.counts ? (long)a.counts[(long)r * a.counts_stride] : a.len[r];
        if (j < len_r) {
            x = load_id(a.ids, (long)r * a.ids_stride + j, a.id_flags);
            if (x >= 0 && x < a.V) {
                const unsigned f = a.flags[x];
                own = f != 0u && (int)__builtin_ctz(f) == r;
                mask = own ? (f & ~((2u << r) - 1u)) : 0u;
            }
        }
    }
    int lo[W], hi[W];
    const long bkt = a.bstart != nullptr && x >= 0 ? (x >> a.bshift) : 0;
#pragma unroll
    for (int r2 = 1; r2 < W; ++r2) {
        lo[r2] = 0;
        hi[r2] = 0;
        if (r2 < a.world && ((mask >> r2) & 1u)) {
            if (a.bstart != nullptr) {
                // the bucket index narrows the search to the list entries of x's bucket (a handful): 3-4 probes instead of 19
                lo[r2] = a.bstart[(long)r2 * a.bstride + bkt];
                hi[r2] = a.bstart[(long)r2 * a.bstride + bkt + 1];
            } else {
                hi[r2] = (int)(a.counts ? (long)a.counts[(long)r2 * a.counts_stride] : a.len[r2]);
            }
        }
    }
    for (;;) {
        bool more = false;
#pragma unroll
        for (int r2 = 1; r2 < W; ++r2) {
            if (lo[r2] < hi[r2]) {
                const int mid = (int)(((unsigned)lo[r2] + (unsigned)hi[r2]) >> 1);
                const long y = load_id(a.ids, (long)r2 * a.ids_stride + mid, a.id_flags);
                if (y < x) lo[r2] = mid + 1; else hi[r2] = mid;
                more = true;
            }
        }
        if (!__any(more)) break;
    }
    // lo[r2] = first entry >= x of list r2: it IS x when the rank's bit was set truthfully; as a window index, -1 if not usable
#pragma unroll
    for (int r2 = 1; r2 < W; ++r2) {
        int w = -1;
        if (r2 < a.world && ((mask >> r2) & 1u)) {
            const long len2 = a.counts ? (long)a.counts[(long)r2 * a.counts_stride] : a.len[r2];
            const long ww = (long)lo[r2] - a.first[r2];
            if (lo[r2] < len2 && ww >= 0 && ww < a.n[r2] && load_id(a.ids, (long)r2 * a.ids_stride + lo[r2], a.id_flags) == x)
                w = (int)ww;
        }
        lo[r2] = w;
    }
    const int x_lo = (int)(x & 0xffffffffL), x_hi = (int)(x >> 32);
    const int own_r = own ? r : -1;
    // ---- phase 2
    const int G = 64 / a.E4;
    const int e4 = lane % a.E4, sub = lane / a.E4;
    for (int it = 0; it < a.E4; ++it) {
        const int src = it * G + sub;                    // the entry (lane of phase 1) this lane group works on
        const int rr = __shfl(own_r, src);
        const int xl = __shfl(x_lo, src), xh = __shfl(x_hi, src);
        int w[W];
#pragma unroll
        for (int r2 = 1; r2 < W; ++r2) w[r2] = __shfl(lo[r2], src);
        if (rr < 0) continue;
        const long xx = (long)(((unsigned long)(unsigned)xh << 32) | (unsigned long)(unsigned)xl);
        const long xi = xx * a.E4 + e4;
        const long i = wave_e0 + src - a.ent_off[rr];
        float4 pp = a.p[xi], mm = a.m[xi], vv = a.v[xi];
        float4 acc = a.rows[((long)rr * a.rows_stride + i) * a.E4 + e4];
        float4 g[W];
#pragma unroll
        for (int r2 = 1; r2 < W; ++r2) {                 // (all of the row's loads requested before the first add)
            g[r2] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (w[r2] >= 0) g[r2] = a.rows[((long)r2 * a.rows_stride + w[r2]) * a.E4 + e4];
        }
#pragma unroll
        for (int r2 = 1; r2 < W; ++r2) {                 // rank order; a rank that does not hold the row adds nothing
            if (w[r2] >= 0) { acc.x += g[r2].x; acc.y += g[r2].y; acc.z += g[r2].z; acc.w += g[r2].w; }
        }
        adam_elem_r(pp.x, acc.x, mm.x, vv.x, a.lr_t, a.b1, a.b2, a.eps, a.clip, a.gs);
        adam_elem_r(pp.y, acc.y, mm.y, vv.y, a.lr_t, a.b1, a.b2, a.eps, a.clip, a.gs);
        adam_elem_r(pp.z, acc.z, mm.z, vv.z, a.lr_t, a.b1, a.b2, a.eps, a.clip, a.gs);
        adam_elem_r(pp.w, acc.w, mm.w, vv.w, a.lr_t, a.b1, a.b2, a.eps, a.clip, a.gs);
        a.p[xi] = pp; a.m[xi] = mm; a.v[xi] = vv;
        if (e4 == 0) a.flags[xx] = 0;                    // (only the owner writes the byte; everybody read it in phase 1 --
    }                                                    //  of THIS wave: entries of other waves that see 0 are no owners either)
}

// flags[row] |= 1 << r for the first len[r] rows of rank r's list.  A byte per row, rows of different ranks may share a
// 32-bit word: the OR is an atomic on the aligned word (off the serial chain: this runs underneath the forward).
// Optionally (bstart != NULL) the same pass builds a BUCKET INDEX over every list for hpmn_rows_sum_adam's searches:
// bstart[r][b] = first entry of list r whose id >> bshift is >= b, for b = 0 .. nb (bstart[r][nb] = len).  The entry that opens a
// bucket fills the slots of the empty buckets in front of it, the last entry the slots behind it (lists are ascending).
__global__ __launch_bounds__(256) void table_mark_ranks_kernel(const void *__restrict__ ids, long ids_stride, int world,
                                                               const int *__restrict__ counts, int counts_stride, long cap,
                                                               uint8_t *__restrict__ flags, long V, int id_flags,
                                                               int *__restrict__ bstart, long bstride, int bshift, long nb) {
    const long stride = (long)gridDim.x * blockDim.x;
    const long total = cap * world;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
        const int r = (int)(t / cap);
        const long j = t - (long)r * cap;
        const long len = counts ? (long)counts[(long)r * counts_stride] : cap;
        if (bstart != nullptr && len == 0 && j == 0)
            for (long b = 0; b <= nb; ++b) bstart[(long)r * bstride + b] = 0;
        if (j >= len) continue;
        const long x = load_id(ids, (long)r * ids_stride + j, id_flags);
        if (bstart != nullptr) {
            // (entries outside [0, V) do not occur in a plan's list; clamped so that the index stays well-formed anyway)
            const long xc = x < 0 ? 0 : (x >= V ? V - 1 : x);
            const long b = xc >> bshift;
            long bp = -1;
            if (j > 0) {
                const long y = load_id(ids, (long)r * ids_stride + j - 1, id_flags);
                bp = (y < 0 ? 0 : (y >= V ? V - 1 : y)) >> bshift;
            }
            for (long bb = bp + 1; bb <= b; ++bb) bstart[(long)r * bstride + bb] = (int)j;
            if (j == len - 1)
                for (long bb = b + 1; bb <= nb; ++bb) bstart[(long)r * bstride + bb] = (int)len;
        }
        if (x < 0 || x >= V) continue;
        unsigned *word = reinterpret_cast<unsigned *>(flags + (x & ~3L));
        const unsigned bit = (1u << r) << (8 * (int)(x & 3L));
        if ((*word & bit) == 0u) atomicOr(word, bit);     // (bits are only ever ADDED while this runs: a set bit seen is set)
    }
}

int table_mark_ranks_launch(const void *ids, int64_t ids_stride, int32_t world, const int32_t *counts, int32_t counts_stride,
                            int64_t cap, uint8_t *flags, int64_t V, int32_t id_flags, int32_t *bstart, int64_t bstride,
                            int32_t bshift, hipStream_t st) {
    if (cap == 0 || world == 0) return HPMN_OK;
    long blocks = (cap * world + 255) / 256;
    if (blocks > 256L * 4) blocks = 256L * 4;            // (a thin grid: it shares the chip with the forward scans)
    const long nb = bstart ? ((V - 1) >> bshift) + 1 : 0;
    hipLaunchKernelGGL(table_mark_ranks_kernel, dim3((unsigned)blocks), dim3(256), 0, st, ids, (long)ids_stride, (int)world,
                       counts, (int)counts_stride, (long)cap, flags, (long)V, (int)id_flags, bstart, (long)bstride, (int)bshift,
                       nb);
    return check_launch();
}

int rows_sum_adam_launch(const HpmnRowsAdam &h, hipStream_t st) {
    RowsAdamK a;
    a.world = h.world; a.E4 = h.E / 4; a.id_flags = h.id_flags; a.counts_stride = h.counts_stride;
    a.ids = h.ids; a.ids_stride = h.ids_stride; a.counts = h.counts;
    a.ent_off[0] = 0;
    for (int r = 0; r < RMAX; ++r) {
        const bool in = r < h.world;
        a.len[r] = in ? h.len[r] : 0; a.first[r] = in ? h.first[r] : 0; a.n[r] = in ? h.n[r] : 0;
        a.ent_off[r + 1] = a.ent_off[r] + a.n[r];
    }
    a.rows = reinterpret_cast<const float4 *>(h.rows); a.rows_stride = h.rows_stride;
    a.flags = h.flags;
    a.p = reinterpret_cast<float4 *>(h.param); a.m = reinterpret_cast<float4 *>(h.m); a.v = reinterpret_cast<float4 *>(h.v);
    a.V = h.V;
    a.lr_t = h.lr_t; a.b1 = h.beta1; a.b2 = h.beta2; a.eps = h.eps; a.clip = h.clip; a.gs = h.grad_scale;
    a.bstart = h.bucket_start; a.bstride = h.bucket_stride; a.bshift = h.bucket_shift;
    const long total = a.ent_off[h.world];
    if (total == 0) return HPMN_OK;
    if (h.world == 1) {
        const long blocks = (total * a.E4 + 255) / 256;
        if (blocks > 0x7fffffffL) return HPMN_EINVAL;
        hipLaunchKernelGGL(rows_adam_one_kernel, dim3((unsigned)blocks), dim3(256), 0, st, a);
        return check_launch();
    }
    const long blocks = (total + 255) / 256;             // one lane per entry in phase 1
    if (blocks > 0x7fffffffL) return HPMN_EINVAL;
    if (h.world <= 2) hipLaunchKernelGGL(rows_sum_adam_kernel<2>, dim3((unsigned)blocks), dim3(256), 0, st, a);
    else if (h.world <= 4) hipLaunchKernelGGL(rows_sum_adam_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(rows_sum_adam_kernel<8>, dim3((unsigned)blocks), dim3(256), 0, st, a);
    return check_launch();
}

}  // namespace hpmn
