// hpmn_scatter_plan_build: the deterministic scatter's plan (scatter_sorted.hip) from the batch's ids in ~10 launches
// instead of the host framework's ~45 (torch.sort's merge passes, cumsum, casts, searchsorted: 240 us of queue latency on the
// auxiliary stream at the C3 shape, in FRONT of whatever has to wait for the distinct rows -- under data parallel the id
// exchange, the marking and the early table-Adam pass).  The sort itself is a library call (rocPRIM's LSD radix sort of
// (id, lookup index) pairs over the bits the table size needs: stable, so perm is exactly what torch.sort(stable=True)
// gives); the rest -- segment heads, their running count, first entries / distinct rows / count, the distinct rows per chunk
// of the table's row range -- are one scan and two small kernels.
// Gradient of Hpmn.embedding (code/hpmn.py:421-422 -> IndexedSlices, densified at :204-205): this only fixes the ORDER in
// which equal ids' slices are summed; nothing here touches a float.
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "common.h"

namespace hpmn {

int scatter_plan_launch(const void *sorted_ids, int32_t id_flags, int64_t n, const int32_t *seg, int32_t *start, void *rows,
                        int32_t *count, hipStream_t st);

template <typename K>
struct HeadOf {
    const K *keys;
    __device__ int operator()(int j) const { return j > 0 && keys[j] != keys[j - 1] ? 1 : 0; }
};

struct ChunkBounds {
    long b[HPMN_MAX_CHUNKS + 1];
};

// counts[0] = U; counts[1 + c] = distinct rows in [b_c, b_c+1).  One thread per boundary.
__global__ void plan_chunk_counts_kernel(const void *__restrict__ rows, const int *__restrict__ count, int id_flags,
                                         ChunkBounds bounds, int nb, int *__restrict__ counts) {
    __shared__ int pos[HPMN_MAX_CHUNKS + 1];
    const int c = threadIdx.x;
    const int U = *count;
    if (c <= nb) {
        int lo = 0, hi = U;
        const long x = bounds.b[c];
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (load_id(rows, mid, id_flags) < x) lo = mid + 1; else hi = mid;
        }
        pos[c] = c == nb ? U : lo;
    }
    __syncthreads();
    if (c == 0) counts[0] = U;
    if (c < nb) counts[1 + c] = pos[c + 1] - pos[c];
}

static int key_bits(int64_t V) {
    int b = 1;
    while (b < 63 && (1LL << b) < V) ++b;
    return b;
}

template <typename K>
static hipError_t sort_pairs(void *tmp, size_t &bytes, const K *keys, K *sorted, int32_t *perm, int64_t n, int bits,
                             hipStream_t st) {
    // (the library's default takes its merge sort up to 2^20 items -- a block sort and ten merge rounds of two launches each,
    //  150 us of launch latency for the 1 M lookups of an XLong batch; Onesweep is a histogram, a scan and one launch per 8 bits)
    using cfg = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, 8192>;
    return rocprim::radix_sort_pairs<cfg>(tmp, bytes, keys, sorted, rocprim::counting_iterator<int32_t>(0), perm, (size_t)n, 0u,
                                          (unsigned)bits, st);
}

template <typename K>
static hipError_t scan_heads(void *tmp, size_t &bytes, const K *sorted, int32_t *seg, int64_t n, hipStream_t st) {
    auto heads = rocprim::make_transform_iterator(rocprim::counting_iterator<int>(0), HeadOf<K>{sorted});
    return rocprim::inclusive_scan(tmp, bytes, heads, seg, (size_t)n, rocprim::plus<int>(), st);
}

static size_t up256(size_t x) { return (x + 255) / 256 * 256; }

// workspace: [sorted keys | library scratch (the larger of the sort's and the scan's)]
static int plan_sizes(int64_t n, int32_t id_flags, int64_t V, size_t &keys_bytes, size_t &tmp_bytes) {
    const bool wide = id_flags & HPMN_ID_I64;
    keys_bytes = up256((size_t)n * (wide ? 8 : 4));
    size_t a = 0, b = 0;
    hipError_t e = wide ? sort_pairs<long>(nullptr, a, nullptr, nullptr, nullptr, n, key_bits(V), nullptr)
                        : sort_pairs<int>(nullptr, a, nullptr, nullptr, nullptr, n, key_bits(V), nullptr);
    if (e != hipSuccess) { set_last_hip_error((int)e); return HPMN_EHIP; }
    e = wide ? scan_heads<long>(nullptr, b, nullptr, nullptr, n, nullptr) : scan_heads<int>(nullptr, b, nullptr, nullptr, n, nullptr);
    if (e != hipSuccess) { set_last_hip_error((int)e); return HPMN_EHIP; }
    tmp_bytes = up256(a > b ? a : b);
    return HPMN_OK;
}

size_t scatter_plan_build_workspace_bytes(int64_t n, int32_t id_flags, int64_t V) {
    size_t k = 0, t = 0;
    if (plan_sizes(n, id_flags, V, k, t) != HPMN_OK) return 0;
    return k + t + 256;
}

int scatter_plan_build_launch(const void *ids, int32_t id_flags, int64_t n, int64_t V, void *workspace, size_t workspace_bytes,
                              int32_t *perm, int32_t *seg, int32_t *start, void *rows, int32_t *count,
                              const int64_t *row_bounds, int32_t nb, int32_t *counts, hipStream_t st) {
    size_t keys_bytes = 0, tmp_bytes = 0;
    int rc = plan_sizes(n, id_flags, V, keys_bytes, tmp_bytes);
    if (rc != HPMN_OK) return rc;
    char *ws = reinterpret_cast<char *>((reinterpret_cast<size_t>(workspace) + 255) / 256 * 256);
    if ((size_t)(ws - reinterpret_cast<char *>(workspace)) + keys_bytes + tmp_bytes > workspace_bytes) return HPMN_EINVAL;
    void *sorted = ws, *tmp = ws + keys_bytes;
    const bool wide = id_flags & HPMN_ID_I64;
    const int bits = key_bits(V);
    size_t bytes = tmp_bytes;
    hipError_t e = wide ? sort_pairs<long>(tmp, bytes, reinterpret_cast<const long *>(ids), reinterpret_cast<long *>(sorted), perm,
                                           n, bits, st)
                        : sort_pairs<int>(tmp, bytes, reinterpret_cast<const int *>(ids), reinterpret_cast<int *>(sorted), perm, n,
                                          bits, st);
    if (e != hipSuccess) { set_last_hip_error((int)e); return HPMN_EHIP; }
    bytes = tmp_bytes;
    e = wide ? scan_heads<long>(tmp, bytes, reinterpret_cast<const long *>(sorted), seg, n, st)
             : scan_heads<int>(tmp, bytes, reinterpret_cast<const int *>(sorted), seg, n, st);
    if (e != hipSuccess) { set_last_hip_error((int)e); return HPMN_EHIP; }
    rc = scatter_plan_launch(sorted, id_flags & HPMN_ID_I64, n, seg, start, rows, count, st);
    if (rc != HPMN_OK) return rc;
    if (counts != nullptr) {
        ChunkBounds cb;
        for (int c = 0; c <= nb; ++c) cb.b[c] = row_bounds ? row_bounds[c] : (c == 0 ? 0 : V);
        hipLaunchKernelGGL(plan_chunk_counts_kernel, dim3(1), dim3(64), 0, st, (const void *)rows, (const int *)count,
                           (int)(id_flags & HPMN_ID_I64), cb, (int)nb, counts);
        return check_launch();
    }
    return HPMN_OK;
}

}  // namespace hpmn
