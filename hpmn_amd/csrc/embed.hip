// Embedding gather and embedding-gradient scatter for gfx950.
//
// Both are HBM-bound row traffic: a row is E fp32 (64 B at E=16).  The gather assigns
// E/4 adjacent lanes x float4 to one row so a wave moves 64/(E/4) whole rows per
// instruction (16 rows = 1 KiB at E=16) with each row a single aligned 64-byte request;
// the id stream [N,F] is read coalesced.  The scatter pre-reduces runs of equal ids along
// the time axis in registers (the uid column is constant over a sequence and the padding
// is a run of id 0) and issues one fp32 atomic row add per run.
#include "common.h"

namespace hpmn {

// out[n, f*E + e] = emb[ids[n,f], e] * (mask ? ids != 0 : 1);   one float4 per thread.
__global__ __launch_bounds__(256) void embed_gather_kernel(const int32_t *__restrict__ ids,
                                                           const float *__restrict__ emb,
                                                           float *__restrict__ out, long total4,
                                                           int E4, int F, long ids_stride, int mask_id0) {
    // total4 = N*F*E/4 float4 items; item -> (row = n*F+f, e4); ids row n starts at n*ids_stride
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += stride) {
        const long row = i / E4;
        const int e4 = (int)(i - row * E4);
        const long n = row / F;
        const int id = ids[n * ids_stride + (row - n * F)];
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (!(mask_id0 && id == 0))
            v = reinterpret_cast<const float4 *>(emb)[(long)id * E4 + e4];
        reinterpret_cast<float4 *>(out)[i] = v;
    }
}

// One wave per (sequence b, group of 64/E id columns, time segment); lane = (column, e).
// Walks its segment of t in chunks of SCU steps (ids and gradients of a chunk are loaded
// up-front so HBM latency is paid once per chunk), keeps the running sum of the current run
// of equal ids and flushes it with one atomic per element when the id changes.  Runs are
// merely split at segment boundaries.  E must divide 64.
constexpr int SCU = 8;      // steps per load chunk
constexpr int SSEG = 128;   // steps per wave
__global__ __launch_bounds__(64) void embed_grad_scatter_kernel(
    const int32_t *__restrict__ ids, const float *__restrict__ d_x, float *__restrict__ d_emb, int B,
    int T, int F, int E, int front_zero, int mask_id0, int groups, int nseg, int t_lo, int t_hi,
    const float *__restrict__ d_last, int t_last) {
    const int cpw = 64 / E;                          // id columns per wave
    const int seg = blockIdx.x % nseg;
    const int grp = (blockIdx.x / nseg) % groups;
    const long b = blockIdx.x / (nseg * groups);
    const int lane = threadIdx.x;
    const int f = grp * cpw + lane / E;
    const int e = lane % E;
    if (f >= F) return;
    const int Dx = F * E;
    const int t_begin = t_lo + seg * SSEG;
    const int t_end = (t_begin + SSEG) < t_hi ? (t_begin + SSEG) : t_hi;
    const int32_t *idp = ids + (b * T) * F + f;
    const float *gp = d_x + (b * (long)(front_zero + T) + front_zero) * Dx + f * E + e;
    int run_id = -1;
    float acc = 0.f;
    for (int t0 = t_begin; t0 < t_end; t0 += SCU) {
        int idv[SCU];
        float gv[SCU];
#pragma unroll
        for (int i = 0; i < SCU; ++i) {
            const int t = t0 + i;
            idv[i] = -1;
            gv[i] = 0.f;
            if (t < t_end) {
                idv[i] = idp[(long)t * F];
                gv[i] = gp[(long)t * Dx];
                // the read path's gradient wrt uinp[:, last_index, :] joins the scan's at that step (instead of a
                // row-add launch of its own in front of this one)
                if (d_last != nullptr && t == t_last) gv[i] += d_last[b * Dx + f * E + e];
            }
        }
#pragma unroll
        for (int i = 0; i < SCU; ++i) {
            if (idv[i] < 0) continue;
            if (idv[i] != run_id) {
                if (run_id >= 0 && !(mask_id0 && run_id == 0))
                    atomicAdd(d_emb + (long)run_id * E + e, acc);
                run_id = idv[i];
                acc = 0.f;
            }
            acc += gv[i];
        }
    }
    if (run_id >= 0 && !(mask_id0 && run_id == 0)) atomicAdd(d_emb + (long)run_id * E + e, acc);
}

int embed_gather_launch(const int32_t *ids, int64_t ids_stride, const float *emb, float *out, int64_t N,
                        int32_t F, int32_t E, int32_t mask_id0, hipStream_t st) {
    const long total4 = (long)N * F * (E / 4);
    if (total4 == 0) return HPMN_OK;
    long blocks = (total4 + 255) / 256;
    if (blocks > 256L * 16) blocks = 256L * 16;   // 16 workgroups per CU, grid-stride the rest
    hipLaunchKernelGGL(embed_gather_kernel, dim3((unsigned)blocks), dim3(256), 0, st, ids, emb, out, total4,
                       E / 4, F, (long)ids_stride, mask_id0);
    return check_launch();
}

// steps [t_lo, t_hi) of every sequence (ids time; t_hi == 0: T)
// d_last [B, F*E] (optional): added to the gradient rows of ids step t_last
int embed_grad_scatter_launch(const int32_t *ids, const float *d_x, float *d_emb, int32_t B, int32_t T,
                              int32_t F, int32_t E, int32_t front_zero, int32_t mask_id0, int32_t t_lo, int32_t t_hi,
                              hipStream_t st, const float *d_last, int32_t t_last) {
    if (t_hi <= 0 || t_hi > T) t_hi = T;
    if (t_lo < 0) t_lo = 0;
    if (B == 0 || t_hi <= t_lo) return HPMN_OK;
    const int cpw = 64 / E;
    const int groups = (F + cpw - 1) / cpw;
    const int nseg = (t_hi - t_lo + SSEG - 1) / SSEG;
    hipLaunchKernelGGL(embed_grad_scatter_kernel, dim3((unsigned)(B * groups * nseg)), dim3(64), 0, st, ids,
                       d_x, d_emb, B, T, F, E, front_zero, mask_id0, groups, nseg, t_lo, t_hi, d_last, t_last);
    return check_launch();
}

}  // namespace hpmn
