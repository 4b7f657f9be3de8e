// Deterministic embedding-gradient scatter: a segmented reduction over the batch's lookups in ROW ORDER (include/hpmn_hip.h,
// hpmn_scatter_plan / hpmn_embed_grad_segsum).  Gradient of Hpmn.embedding (code/hpmn.py:421-422): TF builds IndexedSlices
// and densifies them (:204-205) by summing the slices of equal ids -- here every table row's slices are summed in ONE fixed
// order (ascending lookup index), by exactly one group of lanes, with plain stores: no atomics, so the table gradient is
// bit-reproducible run to run and identical on every data-parallel replica (the atomic kernel of
// embed.hip stays as the plan-less fallback).
//
// The ids are known before the step computes anything, so the ORDER is prepared off the serial chain: a stable sort of the
// flattened ids (host framework: torch.sort on the auxiliary stream) gives perm[j] = lookup index of the j-th entry in row
// order and seg[j] = index of its distinct row; plan_kernel turns that into start[u] (first entry of row u), rows[u] and
// the count.  Behind BPTT:
//   pass 1  one lane group per CHUNK of SCH consecutive entries walks them in order, keeps the running sum of the current
//           row, writes a row that lies inside the chunk straight out and parks the (at most two) runs that cross a chunk
//           border as partials;
//   pass 2  the chunk in which a border-crossing row BEGINS adds that row's partials in chunk order and writes it.
// Output: the compact rows out_rows[u] (what a data-parallel rank sends, what row-wise Adam consumes) and/or
// d_emb[rows[u]] += sum (the dense gradient the two-pass table Adam consumes).
#include <cstdlib>

#include "common.h"

namespace hpmn {

constexpr int SCH = 8;           // entries per chunk.  Small on purpose: the kernel runs beside layer 0's weight gradient, whose
                                 // workgroups hold most of every CU's registers -- at 16 entries (230 registers) its waves waited
                                 // for whole SIMDs to drain (62 us alone, 275 us in the step); at 8 they fit in the gaps

// start[u] = first sorted entry of segment u, rows[u] = its table row, start[U] = n, count[0] = U.
__global__ __launch_bounds__(256) void scatter_plan_kernel(const void *__restrict__ sorted_ids, int id_flags, long n,
                                                           const int *__restrict__ seg, int *__restrict__ start,
                                                           void *__restrict__ rows, int *__restrict__ count) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long j = (long)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += stride) {
        const int u = seg[j];
        if (j == 0 || seg[j - 1] != u) {
            start[u] = (int)j;
            const long id = load_id(sorted_ids, j, id_flags);
            if (id_flags & HPMN_ID_I64) reinterpret_cast<long *>(rows)[u] = id;
            else reinterpret_cast<int *>(rows)[u] = (int)id;
        }
        if (j == n - 1) {
            start[u + 1] = (int)n;
            count[0] = u + 1;
        }
    }
}

struct SegArgs {
    long n;
    const int *perm, *seg, *start;
    const void *rows;
    float *out_rows;      // optional [n, E]
    float *partials;      // [2 * nchunk, E]
    int *heads;           // [nchunk]: chunks in which a border-crossing row BEGINS (appended by pass 1, any order)
    int *nheads;          // [1]
    const float *d_x;     // [B, front_zero + T, F*E]
    float *d_emb;         // optional [V, E]
    const float *d_last;  // optional [B, F*E]
    int T, F, E4, front_zero, id_flags, t_last;
};

// float4 index of lookup q = (b, t, f)'s gradient row (this lane's e4-th piece) in d_x; `at_last`: its step is t_last
__device__ __forceinline__ long lookup_index(const SegArgs &a, int q, int e4, bool &at_last, long &last_index) {
    const int TF = a.T * a.F;
    const int b = q / TF, r = q - b * TF;              // r = t * F + f
    const int t = r / a.F;
    at_last = a.d_last != nullptr && t == a.t_last;
    last_index = ((long)b * a.F + (r - t * a.F)) * a.E4 + e4;
    return ((long)b * (a.front_zero + a.T) + a.front_zero) * a.F * a.E4 + (long)r * a.E4 + e4;
}

__device__ __forceinline__ void write_row(const SegArgs &a, int u, int e4, float4 s) {
    const long row = load_id(a.rows, u, a.id_flags);
    if (id_masked(row, a.id_flags)) s = make_float4(0.f, 0.f, 0.f, 0.f);         // id 0 of the Hpmn class: no gradient
    if (a.out_rows != nullptr) reinterpret_cast<float4 *>(a.out_rows)[(long)u * a.E4 + e4] = s;
    if (a.d_emb != nullptr && !id_masked(row, a.id_flags)) {
        float4 *p = reinterpret_cast<float4 *>(a.d_emb) + row * a.E4 + e4;      // (this launch's only writer of the row)
        float4 o = *p;
        o.x += s.x; o.y += s.y; o.z += s.z; o.w += s.w;
        *p = o;
    }
}

// PASS 1.  thread = (chunk, e4): E4 adjacent lanes own a chunk.  Everything a chunk needs is requested up front in three
// batches of independent loads (order + segment of its entries; their gradient rows and table rows; the table-gradient rows
// it will add to): the first version looked segment extents up inside the serial run loop -- two dependent loads per row,
// 32 rows deep -- and cost the C3 step 0.16 ms.
__global__ __launch_bounds__(256) void segsum_chunks_kernel(SegArgs a, long nchunk) {
    const long g = ((long)blockIdx.x * blockDim.x + threadIdx.x) / a.E4;
    const int e4 = threadIdx.x % a.E4;
    if (g >= nchunk) return;
    const long j0 = g * SCH;
    const int m = (int)((a.n - j0) < SCH ? (a.n - j0) : SCH);
    const long jend = j0 + m;
    int q[SCH], sg[SCH];
#pragma unroll
    for (int i = 0; i < SCH; ++i) {
        const long j = j0 + (i < m ? i : m - 1);
        q[i] = a.perm[j];
        sg[i] = a.seg[j];
    }
    const int sg_prev = j0 > 0 ? a.seg[j0 - 1] : -1;               // does the first run continue one from the chunk before?
    const int sg_next = jend < a.n ? a.seg[jend] : -1;             // does the last run continue into the next chunk?
    float4 v[SCH];
    long row[SCH];
    unsigned last_bits = 0;                                         // entries whose step carries the read path's d_last row
    long last_idx = 0;                                              // (at most one per sequence and id column: rare)
#pragma unroll
    for (int i = 0; i < SCH; ++i) {
        bool at_last;
        long li;
        const long idx = lookup_index(a, q[i], e4, at_last, li);
        v[i] = reinterpret_cast<const float4 *>(a.d_x)[idx];       // (no use of v in this loop: all of the chunk's row loads in flight)
        row[i] = load_id(a.rows, sg[i], a.id_flags);
        if (at_last && i < m) { last_bits |= 1u << i; last_idx = li; }
    }
    if (last_bits != 0) {                                           // joined to the lookup's row BEFORE the sums, as the atomic kernel does
#pragma unroll
        for (int i = 0; i < SCH; ++i) {
            if ((last_bits >> i) & 1u) {
                bool at_last;
                long li;
                (void)lookup_index(a, q[i], e4, at_last, li);
                const float4 w = reinterpret_cast<const float4 *>(a.d_last)[li];
                v[i].x += w.x; v[i].y += w.y; v[i].z += w.z; v[i].w += w.w;
            }
        }
    }
    (void)last_idx;
    // run sums, left to right: the LAST entry of a run ends up holding the run's sum
    unsigned ends = 0, first_run = 0;                               // bit i: entry i closes a run / that run began at entry 0
    {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        bool from0 = true;
#pragma unroll
        for (int i = 0; i < SCH; ++i) {
            if (i < m) {
                acc.x += v[i].x; acc.y += v[i].y; acc.z += v[i].z; acc.w += v[i].w;
                const bool end = (i == m - 1) || (sg[i + 1 < SCH ? i + 1 : i] != sg[i]);
                if (end) {
                    v[i] = acc;
                    acc = make_float4(0.f, 0.f, 0.f, 0.f);
                    ends |= 1u << i;
                    first_run |= from0 ? (1u << i) : 0u;
                    from0 = false;
                }
            }
        }
    }
    // a run is WHOLE unless it touches a chunk border across which its segment continues
    unsigned whole = 0;
#pragma unroll
    for (int i = 0; i < SCH; ++i) {
        if ((ends >> i) & 1u) {
            const bool left = ((first_run >> i) & 1u) && sg_prev == sg[i];
            const bool right = (i == m - 1) && sg_next == sg[i];
            whole |= (!left && !right) ? (1u << i) : 0u;
        }
    }
    float4 *P = reinterpret_cast<float4 *>(a.partials);
    float4 *D = reinterpret_cast<float4 *>(a.d_emb);
    float4 *O = reinterpret_cast<float4 *>(a.out_rows);
    static_assert(SCH % 8 == 0, "RMW batches of 8");
#pragma unroll
    for (int h = 0; h < SCH; h += 8) {                              // the read-modify-write of the table gradient, 8 rows in flight
        float4 old[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) old[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (D != nullptr) {                                         // (uniform; the loads inside are UNCONDITIONAL: a lane that
#pragma unroll                                                      //  will not use the value reads its own valid row all the same)
            for (int i = 0; i < 8; ++i) old[i] = D[row[h + i] * a.E4 + e4];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int k = h + i;
            if (!((ends >> k) & 1u)) continue;
            if ((whole >> k) & 1u) {
                const bool masked = id_masked(row[k], a.id_flags);
                const float4 s = masked ? make_float4(0.f, 0.f, 0.f, 0.f) : v[k];
                if (O != nullptr) O[(long)sg[k] * a.E4 + e4] = s;
                if (D != nullptr && !masked)
                    D[row[k] * a.E4 + e4] = make_float4(old[i].x + s.x, old[i].y + s.y, old[i].z + s.z, old[i].w + s.w);
            } else {
                P[(2 * g + (((first_run >> k) & 1u) ? 0 : 1)) * a.E4 + e4] = v[k];
                // this run crosses the chunk's right border and BEGAN in this chunk: pass 2 starts from here
                const bool right = (k == m - 1) && sg_next == sg[k];
                const bool left = ((first_run >> k) & 1u) && sg_prev == sg[k];
                if (right && !left && e4 == 0) a.heads[atomicAdd(a.nheads, 1)] = (int)g;
            }
        }
    }
}

// PASS 2.  One WAVE per border-crossing row (pass 1 listed the chunks in which such rows begin): the row's partials -- the head
// chunk's last run, then the first run of every following chunk it reaches into -- are cut into NB consecutive blocks, one per
// lane group; a group adds its block left to right, the block sums are added in block order.  A fixed tree: deterministic,
// and short (the uid rows of XLong are 126 partials: 8 per group).  The first version walked them sequentially in one lane
// group per CHUNK of the batch -- 31 k near-empty waves that queued behind the weight gradient: 250 us in the step.
__global__ __launch_bounds__(64) void segsum_borders_kernel(SegArgs a, long nchunk) {
    const int lane = threadIdx.x;
    const int G = 64 / a.E4;                        // lane groups of a wave
    const int NB = G < 16 ? G : 16;                 // blocks a row's partials are cut into
    const int grp = lane / a.E4, e4 = lane % a.E4;
    const int count = *a.nheads;
    const float4 *P = reinterpret_cast<const float4 *>(a.partials);
    for (int h = blockIdx.x; h < count; h += gridDim.x) {
        const long g = a.heads[h];
        const long j0 = g * SCH;
        const long jend = (j0 + SCH) < a.n ? (j0 + SCH) : a.n;
        const int u = a.seg[jend - 1];
        if (id_masked(load_id(a.rows, u, a.id_flags), a.id_flags)) {       // (the padding id of the Hpmn class: thousands of chunks)
            if (grp == 0) write_row(a, u, e4, make_float4(0.f, 0.f, 0.f, 0.f));
            continue;
        }
        const long s1 = a.start[u + 1];
        const bool only_run = a.seg[j0] == u;       // (the chunk's first run too: parked in slot 0)
        const long last_chunk = (s1 - 1) / SCH;
        const int L = (int)(last_chunk - g) + 1;     // partials of this row
        const int per = (L + NB - 1) / NB;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (grp < NB) {
            const int i0 = grp * per, i1 = (i0 + per) < L ? (i0 + per) : L;
            for (int i = i0; i < i1; i += 8) {
                float4 w[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int ii = (i + k) < i1 ? (i + k) : (i1 - 1);
                    const long c = g + ii;
                    w[k] = P[(ii == 0 ? (2 * g + (only_run ? 0 : 1)) : 2 * c) * a.E4 + e4];
                }
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (i + k < i1) { acc.x += w[k].x; acc.y += w[k].y; acc.z += w[k].z; acc.w += w[k].w; }
            }
        }
        // block sums in block order: group 0 collects (shuffles are wave-wide: every lane takes part)
        float4 tot = acc;
        for (int b = 1; b < NB; ++b) {
            const int src = b * a.E4 + e4;
            const float x = __shfl(acc.x, src), y = __shfl(acc.y, src), z = __shfl(acc.z, src), w = __shfl(acc.w, src);
            if ((long)b * per < L) { tot.x += x; tot.y += y; tot.z += z; tot.w += w; }
        }
        if (grp == 0) write_row(a, u, e4, tot);
    }
}

int scatter_plan_launch(const void *sorted_ids, int32_t id_flags, int64_t n, const int32_t *seg, int32_t *start, void *rows,
                        int32_t *count, hipStream_t st) {
    if (n == 0) return HPMN_OK;
    long blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(scatter_plan_kernel, dim3((unsigned)blocks), dim3(256), 0, st, sorted_ids, (int)id_flags, (long)n, seg,
                       start, rows, count);
    return check_launch();
}

// scratch: the partials [2 * nchunk, E], then the head list [nchunk] and its counter (ints)
size_t segsum_partials_floats(int64_t n, int32_t E) {
    const size_t nchunk = (size_t)((n + SCH - 1) / SCH);
    return 2 * nchunk * E + nchunk + 16;
}
int segsum_chunk_entries() { return SCH; }

int embed_grad_segsum_launch(const HpmnScatterPlan &p, const float *d_x, float *d_emb, int32_t B, int32_t T, int32_t F,
                             int32_t E, int32_t front_zero, int32_t id_flags, const float *d_last, int32_t t_last,
                             hipStream_t st) {
    if (p.n == 0) return HPMN_OK;
    SegArgs a;
    a.n = p.n; a.perm = p.perm; a.seg = p.seg; a.start = p.start; a.rows = p.rows; a.out_rows = p.out_rows;
    a.partials = p.partials; a.d_x = d_x; a.d_emb = d_emb; a.d_last = d_last;
    a.T = T; a.F = F; a.E4 = E / 4; a.front_zero = front_zero; a.id_flags = id_flags; a.t_last = t_last;
    const long nchunk = (p.n + SCH - 1) / SCH;
    a.heads = reinterpret_cast<int *>(p.partials + 2 * nchunk * E);
    a.nheads = a.heads + nchunk;
    if (hipMemsetAsync(a.nheads, 0, sizeof(int), st) != hipSuccess) { set_last_hip_error((int)hipGetLastError()); return HPMN_EHIP; }
    const long threads = nchunk * a.E4;
    const unsigned blocks = (unsigned)((threads + 255) / 256);
    hipLaunchKernelGGL(segsum_chunks_kernel, dim3(blocks), dim3(256), 0, st, a, nchunk);
    int rc = check_launch();
    if (rc != HPMN_OK) return rc;
    const unsigned waves = (unsigned)(nchunk < 2048 ? (nchunk < 1 ? 1 : nchunk) : 2048);
    hipLaunchKernelGGL(segsum_borders_kernel, dim3(waves), dim3(64), 0, st, a, nchunk);
    return check_launch();
}

}  // namespace hpmn
