// Embedding gather and embedding-gradient scatter for gfx950.
//
// Both are HBM-bound row traffic: a row is E fp32 (64 B at E=16).  The gather assigns
// E/4 adjacent lanes x float4 to one row so a wave moves 64/(E/4) whole rows per
// instruction (16 rows = 1 KiB at E=16) with each row a single aligned 64-byte request;
// the id stream [N,F] is read coalesced.  The scatter pre-reduces runs of equal ids along
// the time axis in registers (the uid column is constant over a sequence and the padding
// is a run of id 0) and issues one fp32 atomic row add per run.
#include <cstdlib>

#include "common.h"

namespace hpmn {

// out[n, f*E + e] = emb[ids[n,f], e] * (mask ? ids != 0 : 1);   one float4 per thread.
__global__ __launch_bounds__(256) void embed_gather_kernel(const void *__restrict__ ids,
                                                           const float *__restrict__ emb,
                                                           float *__restrict__ out, long total4,
                                                           int E4, int F, long ids_stride, int mask_id0) {
    // total4 = N*F*E/4 float4 items; item -> (row = n*F+f, e4); ids row n starts at n*ids_stride
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += stride) {
        const long row = i / E4;
        const int e4 = (int)(i - row * E4);
        const long n = row / F;
        const long id = load_id(ids, n * ids_stride + (row - n * F), mask_id0);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (!id_masked(id, mask_id0))
            v = reinterpret_cast<const float4 *>(emb)[id * E4 + e4];
        reinterpret_cast<float4 *>(out)[i] = v;
    }
}

// One wave per (sequence b, group of 64/E id columns, time segment); lane = (column, e).
// Walks its segment of t in chunks of SCU steps (ids and gradients of a chunk are loaded
// up-front so HBM latency is paid once per chunk), keeps the running sum of the current run
// of equal ids and flushes it with one atomic per element when the id changes.  Runs are
// merely split at segment boundaries.  E must divide 64.
constexpr int SCU = 8;      // steps per load chunk
constexpr int SSEG = 128;   // steps per wave at most (r5: fewer where the launch would otherwise be a few hundred waves, below)
// HOT (r6, HPMN_ID_HOT): a run's sum goes into a 64-entry LDS table keyed by the id (a slot is claimed with a compare-and-swap
// by the run's first lane; a slot held by another id: the atomic row add as before) and the table is flushed with one atomic
// row add per entry at the end -- a row that many lookups of the segment share (Zipf's head, Taobao's 4-valued btag column:
// 16 k lookups of a C2 batch on 5 rows) costs the wave ONE global row add instead of one per run.  Uniform ids gain nothing
// and pay the table's traffic: the host picks the form from the first batch (hpmn.py: _scatter_hint).
constexpr int SHT = 64;     // table entries per wave
template <bool HOT>
__global__ __launch_bounds__(64) void embed_grad_scatter_kernel(
    const void *__restrict__ ids, const float *__restrict__ d_x, float *__restrict__ d_emb, int B,
    int T, int F, int E, int front_zero, int mask_id0, int groups, int nseg, int t_lo, int t_hi,
    const float *__restrict__ d_last, int t_last, int sseg) {
    __shared__ long long hkey[HOT ? SHT : 1];
    __shared__ float htab[HOT ? 1024 : 1];                   // [entry][e]: nh = min(64, 1024 / E) entries of E floats
    const int nh = (1024 / E) < SHT ? (1024 / E) : SHT;      // (a power of two: E | 64)
    if constexpr (HOT) {
        hkey[threadIdx.x] = -1;                              // (64 threads == SHT entries)
        __builtin_amdgcn_s_waitcnt(0xc07f);                  // lgkmcnt(0): the wave's own LDS writes have landed (one wave: no barrier)
    }
    const int cpw = 64 / E;                          // E-lane slots per wave
    const int seg = blockIdx.x % nseg;
    const int grp = (blockIdx.x / nseg) % groups;
    const long b = blockIdx.x / (nseg * groups);
    const int lane = threadIdx.x;
    // slots beyond the id columns this wave has take further TIME STEPS instead of idling (r4: XLong has two id columns and
    // four slots -- half of every wave returned at once): slot = (step phase, column), the lane walks steps phase, phase + tpw, ...
    const int fpw = (F - grp * cpw) < cpw ? (F - grp * cpw) : cpw;     // id columns of this wave
    const int tpw = cpw / fpw;                                         // steps a wave takes at once
    const int slot = lane / E;
    if (slot >= tpw * fpw) return;
    const int f = grp * cpw + slot % fpw;
    const int tsub = slot / fpw;
    const int e = lane % E;
    const int Dx = F * E;
    const int t_begin = t_lo + seg * sseg;
    const int t_end = (t_begin + sseg) < t_hi ? (t_begin + sseg) : t_hi;
    const long idp = (b * T) * F + f;               // index of ids[b, 0, f]
    const float *gp = d_x + (b * (long)(front_zero + T) + front_zero) * Dx + f * E + e;
    long run_id = -1;
    float acc = 0.f;
    auto flush_run = [&](long id, float sum) {
        if constexpr (HOT) {
            const int h = (int)((id * 0x9E3779B1u) >> 13) & (nh - 1);
            int ok = 0;
            if (e == 0) {
                // (a fresh entry's row is zeroed by its claimer BEFORE the other lanes add: they learn `ok` through the
                //  shuffle below, i.e. after this lane's LDS operations were issued -- LDS runs a wave's operations in order)
                const long long old = atomicCAS(reinterpret_cast<unsigned long long *>(&hkey[h]), (unsigned long long)-1LL,
                                                (unsigned long long)id);
                ok = (old == -1LL || old == (long long)id) ? ((old == -1LL) ? 2 : 1) : 0;
                if (ok == 2)
                    for (int q = 0; q < E; ++q) htab[h * E + q] = 0.f;
            }
            ok = __shfl(ok, 0, E);                           // (the slot's E lanes are in here together: E | 64)
            if (ok) atomicAdd(&htab[h * E + e], sum);
            else atomicAdd(d_emb + id * E + e, sum);
        } else {
            atomicAdd(d_emb + id * E + e, sum);
        }
    };
    for (int t0 = t_begin + tsub; t0 < t_end; t0 += SCU * tpw) {
        long idv[SCU];
        float gv[SCU];
#pragma unroll
        for (int i = 0; i < SCU; ++i) {
            const int t = t0 + i * tpw;
            idv[i] = -1;
            gv[i] = 0.f;
            if (t < t_end) {
                idv[i] = load_id(ids, idp + (long)t * F, mask_id0);
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
                if (run_id >= 0 && !id_masked(run_id, mask_id0)) flush_run(run_id, acc);
                run_id = idv[i];
                acc = 0.f;
            }
            acc += gv[i];
        }
    }
    if (run_id >= 0 && !id_masked(run_id, mask_id0)) flush_run(run_id, acc);
    if constexpr (HOT) {
        // every lane is back here (the early return above is per SLOT: whole 16-lane groups); the table: entry h, element e
        // (the lanes still here are the wave's ACTIVE slots, 0 .. tpw * fpw - 1: they share the table's entries)
        __builtin_amdgcn_s_waitcnt(0xc07f);
        for (int h = slot; h < nh; h += tpw * fpw) {
            const long long k = hkey[h];
            if (k >= 0) atomicAdd(d_emb + k * E + e, htab[h * E + e]);
        }
    }
}

// The gather CONSUMED IN PLACE: out[b, f*E + e] = sum_t emb[ids[b,t,f], e] (mask as above) -- the pooled form of
// Hpmn.embedding (what the reference's mean-pooling baselines do with it, and what the fused scan kernels do with the rows:
// use them, never store them).  The roofline probe for north_star's "gather at >= 40 % of HBM": 4 B of id + 64 B of row per
// lookup is ALL the traffic there is, where the materialising gather above also writes every row back out.
// One workgroup per (sequence, slice of t); lane = (row slot, float4 of the row); rows of 8 steps in flight per lane.
constexpr int GRU_ = 8;
__global__ __launch_bounds__(256) void embed_gather_sum_kernel(const void *__restrict__ ids, const float *__restrict__ emb,
                                                               float *__restrict__ out, int T, int F, int E4, int mask_id0,
                                                               int slices) {
    const long b = blockIdx.x / slices;
    const int sl = blockIdx.x % slices;
    const int rows = T * F;                              // lookups of this sequence, index r = t * F + f
    const int per = (rows + slices - 1) / slices;
    const int r0 = sl * per, r1 = (r0 + per) < rows ? (r0 + per) : rows;
    const int e4 = threadIdx.x % E4, slot = threadIdx.x / E4, nslot = 256 / E4;
    const long idb = b * (long)rows;                     // index of this sequence's first lookup
    // a lane only ever sees lookups r with r % F == f0 when nslot % F == 0: one accumulator per lane then belongs to one id
    // column; otherwise accumulate per column in F partial sums
    float4 acc[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) acc[f] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int r = r0 + slot; r < r1; r += nslot * GRU_) {
        long id[GRU_];
#pragma unroll
        for (int u = 0; u < GRU_; ++u) {
            const int rr = r + u * nslot;
            id[u] = rr < r1 ? load_id(ids, idb + rr, mask_id0) : -1;
        }
        float4 v[GRU_];
#pragma unroll
        for (int u = 0; u < GRU_; ++u) {
            const bool keep = id[u] >= 0 && !id_masked(id[u], mask_id0);
            // (non-temporal: a row is used once; measured 25.9 -> 22.8 us at the C3 shape on a cold 4 GiB table,
            //  tools/micro/gather_line.hip -- the 128-byte line per 64-byte row stays whatever the cache policy bits say)
            typedef float v4f_ __attribute__((ext_vector_type(4)));
            v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (keep) {
                const v4f_ w = __builtin_nontemporal_load(reinterpret_cast<const v4f_ *>(emb) + id[u] * E4 + e4);
                v[u] = make_float4(w[0], w[1], w[2], w[3]);
            }
        }
#pragma unroll
        for (int u = 0; u < GRU_; ++u) {
            const int f = (r + u * nslot) % F;
#pragma unroll
            for (int ff = 0; ff < 4; ++ff)
                if (ff == f) { acc[ff].x += v[u].x; acc[ff].y += v[u].y; acc[ff].z += v[u].z; acc[ff].w += v[u].w; }
        }
    }
    // sum over the row slots: lanes of a wave that hold the same float4 of a row are E4 apart (xor-shuffles), the four waves
    // meet in LDS, and ONE atomic per element and workgroup goes to memory (per-lane atomics: 512 adds per address, serialised)
    __shared__ float4 part[4][4][64];                    // [wave][column][e4]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int f = 0; f < 4; ++f) {
        float4 a = acc[f];
        for (int m = E4; m < 64; m <<= 1) {
            a.x += __shfl_xor(a.x, m); a.y += __shfl_xor(a.y, m); a.z += __shfl_xor(a.z, m); a.w += __shfl_xor(a.w, m);
        }
        if (lane < E4) part[wave][f][lane] = a;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < F * E4; i += 256) {
        const int f = i / E4, e = i - f * E4;
        float4 a = part[0][f][e];
        for (int w = 1; w < 4; ++w) { const float4 p = part[w][f][e]; a.x += p.x; a.y += p.y; a.z += p.z; a.w += p.w; }
        float *o = out + b * (long)F * E4 * 4 + (long)f * E4 * 4 + 4 * e;
        atomicAdd(o, a.x); atomicAdd(o + 1, a.y); atomicAdd(o + 2, a.z); atomicAdd(o + 3, a.w);
    }
}

// out [B, F*E] must be zeroed by the caller
int embed_gather_sum_launch(const void *ids, const float *emb, float *out, int32_t B, int32_t T, int32_t F, int32_t E,
                            int32_t mask_id0, hipStream_t st) {
    // a row's E/4 float4 lanes must sit inside one wave (part[][][64], the xor-shuffles): E <= 256
    if (F > 4 || E % 4 != 0 || E / 4 > 64 || 64 % (E / 4) != 0) return HPMN_EUNSUPPORTED;
    if (B == 0) return HPMN_OK;
    int slices = 1;
    while ((long)B * slices < 2048 && slices < 16) slices *= 2;          // enough workgroups to fill the chip
    if (const char *e = getenv("HPMN_GSUM_SLICES")) slices = atoi(e) > 0 ? atoi(e) : slices;
    hipLaunchKernelGGL(embed_gather_sum_kernel, dim3((unsigned)(B * slices)), dim3(256), 0, st, ids, emb, out, T, F, E / 4,
                       mask_id0, slices);
    return check_launch();
}

int embed_gather_launch(const void *ids, int64_t ids_stride, const float *emb, float *out, int64_t N,
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
int embed_grad_scatter_launch(const void *ids, const float *d_x, float *d_emb, int32_t B, int32_t T,
                              int32_t F, int32_t E, int32_t front_zero, int32_t mask_id0, int32_t t_lo, int32_t t_hi,
                              hipStream_t st, const float *d_last, int32_t t_last) {
    if (t_hi <= 0 || t_hi > T) t_hi = T;
    if (t_lo < 0) t_lo = 0;
    if (B == 0 || t_hi <= t_lo) return HPMN_OK;
    const int cpw = 64 / E;
    const int groups = (F + cpw - 1) / cpw;
    // A wave walks its segment in chunks of 8 steps, one memory round trip per chunk: at the Amazon / Taobao shapes (128
    // sequences of 100 / 300 steps) 128-step segments made the launch 128 / 384 waves of 13 / 16 round trips one behind the other --
    // 36 / 50 us on the step's tail for 38 k / 154 k lookups.  Segments shrink (not below 16 steps) until the launch has ~2 000
    // waves; runs of equal ids are merely split at a few more boundaries.  XLong (500 x 1 001 steps) keeps 128.
    const int want_seg = (int)((2048 + (long)B * groups - 1) / ((long)B * groups));
    int sseg = (t_hi - t_lo + want_seg - 1) / want_seg;
    sseg = (sseg + 15) / 16 * 16;
    sseg = sseg < 16 ? 16 : (sseg > SSEG ? SSEG : sseg);
    const int nseg = (t_hi - t_lo + sseg - 1) / sseg;
    // HPMN_SCATTER_HOT=0 / 1: never / always the table form (default: the caller's HPMN_ID_HOT hint)
    static const int hot_env = [] { const char *e = getenv("HPMN_SCATTER_HOT"); return e ? atoi(e) : -1; }();
    const bool hot = hot_env >= 0 ? hot_env != 0 : (mask_id0 & HPMN_ID_HOT) != 0;
    if (hot)
        hipLaunchKernelGGL(embed_grad_scatter_kernel<true>, dim3((unsigned)(B * groups * nseg)), dim3(64), 0, st, ids,
                           d_x, d_emb, B, T, F, E, front_zero, mask_id0, groups, nseg, t_lo, t_hi, d_last, t_last, sseg);
    else
        hipLaunchKernelGGL(embed_grad_scatter_kernel<false>, dim3((unsigned)(B * groups * nseg)), dim3(64), 0, st, ids,
                           d_x, d_emb, B, T, F, E, front_zero, mask_id0, groups, nseg, t_lo, t_hi, d_last, t_last, sseg);
    return check_launch();
}

}  // namespace hpmn
