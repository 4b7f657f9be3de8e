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
#include "common.h"

namespace hpmn {

constexpr int SCH = 16;          // entries per chunk

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
    const float *d_x;     // [B, front_zero + T, F*E]
    float *d_emb;         // optional [V, E]
    const float *d_last;  // optional [B, F*E]
    int T, F, E4, front_zero, id_flags, t_last;
};

// gradient row of lookup q = (b, t, f) as this lane's float4 (e4-th of the row), d_last joined at t == t_last
__device__ __forceinline__ float4 lookup_grad(const SegArgs &a, int q, int e4) {
    const int TF = a.T * a.F;
    const int b = q / TF, r = q - b * TF;              // r = t * F + f
    const long base = ((long)b * (a.front_zero + a.T) + a.front_zero) * a.F * a.E4 + (long)r * a.E4 + e4;
    float4 v = reinterpret_cast<const float4 *>(a.d_x)[base];
    if (a.d_last != nullptr) {
        const int t = r / a.F;
        if (t == a.t_last) {
            const float4 w = reinterpret_cast<const float4 *>(a.d_last)[((long)b * a.F + (r - t * a.F)) * a.E4 + e4];
            v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
        }
    }
    return v;
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
#pragma unroll
    for (int i = 0; i < SCH; ++i) {
        v[i] = lookup_grad(a, q[i], e4);
        row[i] = load_id(a.rows, sg[i], a.id_flags);
    }
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
#pragma unroll
    for (int h = 0; h < SCH; h += 8) {                              // the read-modify-write of the table gradient, 8 rows in flight
        float4 old[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int k = h + i;
            old[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (((whole >> k) & 1u) && D != nullptr && !id_masked(row[k], a.id_flags)) old[i] = D[row[k] * a.E4 + e4];
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
            }
        }
    }
}

// PASS 2.  The chunk in which a row that crosses a chunk border BEGINS: its partial (this chunk's last run), then the first
// run of every following chunk the row reaches into, in chunk order.
__global__ __launch_bounds__(256) void segsum_borders_kernel(SegArgs a, long nchunk) {
    const long g = ((long)blockIdx.x * blockDim.x + threadIdx.x) / a.E4;
    const int e4 = threadIdx.x % a.E4;
    if (g >= nchunk) return;
    const long j0 = g * SCH;
    const long jend = (j0 + SCH) < a.n ? (j0 + SCH) : a.n;
    const int u = a.seg[jend - 1];                                     // the chunk's last run
    const long s0 = a.start[u], s1 = a.start[u + 1];
    if (s0 < j0 || s1 <= jend) return;                                  // begins earlier / ends here: not ours
    if (id_masked(load_id(a.rows, u, a.id_flags), a.id_flags)) {       // (the padding id of the Hpmn class: thousands of chunks)
        write_row(a, u, e4, make_float4(0.f, 0.f, 0.f, 0.f));
        return;
    }
    const bool only_run = a.seg[j0] == u;                               // (the chunk's first run too: it was parked in slot 0)
    const float4 *P = reinterpret_cast<const float4 *>(a.partials);
    float4 acc = P[(2 * g + (only_run ? 0 : 1)) * a.E4 + e4];
    const long last_chunk = (s1 - 1) / SCH;
    constexpr int PB = 8;                                               // partials in flight; added in chunk order all the same
    for (long c0 = g + 1; c0 <= last_chunk; c0 += PB) {
        float4 w[PB];
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            const long c = (c0 + i) <= last_chunk ? (c0 + i) : last_chunk;
            w[i] = P[(2 * c) * a.E4 + e4];
        }
#pragma unroll
        for (int i = 0; i < PB; ++i)
            if (c0 + i <= last_chunk) { acc.x += w[i].x; acc.y += w[i].y; acc.z += w[i].z; acc.w += w[i].w; }
    }
    write_row(a, u, e4, acc);
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

size_t segsum_partials_floats(int64_t n, int32_t E) { return (size_t)(2 * ((n + SCH - 1) / SCH)) * E; }

int embed_grad_segsum_launch(const HpmnScatterPlan &p, const float *d_x, float *d_emb, int32_t B, int32_t T, int32_t F,
                             int32_t E, int32_t front_zero, int32_t id_flags, const float *d_last, int32_t t_last,
                             hipStream_t st) {
    if (p.n == 0) return HPMN_OK;
    SegArgs a;
    a.n = p.n; a.perm = p.perm; a.seg = p.seg; a.start = p.start; a.rows = p.rows; a.out_rows = p.out_rows;
    a.partials = p.partials; a.d_x = d_x; a.d_emb = d_emb; a.d_last = d_last;
    a.T = T; a.F = F; a.E4 = E / 4; a.front_zero = front_zero; a.id_flags = id_flags; a.t_last = t_last;
    const long nchunk = (p.n + SCH - 1) / SCH;
    const long threads = nchunk * a.E4;
    const unsigned blocks = (unsigned)((threads + 255) / 256);
    hipLaunchKernelGGL(segsum_chunks_kernel, dim3(blocks), dim3(256), 0, st, a, nchunk);
    int rc = check_launch();
    if (rc != HPMN_OK) return rc;
    hipLaunchKernelGGL(segsum_borders_kernel, dim3(blocks), dim3(256), 0, st, a, nchunk);
    return check_launch();
}

}  // namespace hpmn
