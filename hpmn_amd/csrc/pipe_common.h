// Shared pieces of the batch-tiled MFMA scan kernels (gru_pipe_fwd.hip / gru_pipe_bwd.hip), H = 64.
//
// Why this decomposition (DESIGN_HISTORY.md section 3.7): one sequence per wave (gru_scan_*.hip) is latency-bound at
// ~1250-1600 cycles per step because a single wave issues one instruction per ~5 cycles and a step needs ~100
// packed FMAs plus two LDS broadcast round trips.  Here a workgroup owns a TILE of 16 sequences, which makes the
// recurrent product of a step a real [3H x H] x [H x 16] contraction, and the 16x16x32 f16 MFMA does 8192 MACs in
// 16 cycles.  fp32 accuracy is kept by splitting every operand into two halves, x = hi + lo with
// hi = f16(x), lo = f16(x - hi) (both round-to-nearest, so |x - hi - lo| <= 2^-22 |x|, or 2^-25 absolute in the
// f16 subnormal range, which the matrix pipe does not flush), and issuing three MFMAs per tile:
//     W h  ~=  W_hi h_hi + W_hi h_lo + W_lo h_hi          (the dropped W_lo h_lo term is <= 2^-22 |W||h|)
// accumulated in fp32.  That is 3/16 of the f32-MFMA time for fp32-class results (parity tests: same 1e-4 /
// 2e-4 bars as the VALU kernels, measured errors of the same order).
//
// Layout of a step's "B operand" (the [K=64 x N=16] matrix h, r*h, x_t, ...): lane (g = lane/16, n = lane%16)
// of an MFMA holds the 8 k-slots (s, g, e=0..7) of column n for k-step s.  The hardware pairs A's slot (s,g,e)
// with B's slot (s,g,e), so the assignment slot -> hidden unit is free as long as A (the weights, prepared once)
// and B agree.  We choose
//     unit(s, g, e) = 16 (2 s + e/4) + 4 g + e%4
// because the C/D layout of the same instruction gives lane (g, n) of wave w the outputs 16 w + 4 g + j, j=0..3
// of column n: the four values a lane has just produced are exactly slots e = 4 (w%2) + j of (s = w/2, g) -- the
// lane writes 8 bytes (hi) + 8 bytes (lo) to LDS, every reader pulls its 16-byte slot group back with one
// ds_read_b128, and no value ever changes its lane group.
#pragma once
#include "common.h"

namespace hpmn {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;

constexpr int TS = 16;              // sequences per tile (= MFMA N)
constexpr int PH = 64;              // hidden size these kernels are written for
constexpr int ROWB = 160;           // bytes per sequence row of an operand image: 128 data + 32 pad
                                    // (n*160 + g*16 is conflict-free for ds_read_b128's 16-lane groups)
constexpr int IMG = TS * ROWB;      // one image (hi or lo half of a [64 x 16] operand)

// x = hi + lo, four values at once; returned as packed f16 pairs ready for an 8-byte LDS store
__device__ __forceinline__ void split4(const f4 v, uint2 &hi, uint2 &lo) {
    const h4 a = {(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
    const f4 r = {v.x - (float)a.x, v.y - (float)a.y, v.z - (float)a.z, v.w - (float)a.w};
    const h4 b = {(_Float16)r.x, (_Float16)r.y, (_Float16)r.z, (_Float16)r.w};
    hi = __builtin_bit_cast(uint2, a);
    lo = __builtin_bit_cast(uint2, b);
}

__device__ __forceinline__ void split8(const float (&v)[8], h8 &hi, h8 &lo) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        hi[e] = (_Float16)v[e];
        lo[e] = (_Float16)(v[e] - (float)hi[e]);
    }
}

// slot (s, g, e) of an operand image -> hidden unit / feature index
__device__ __forceinline__ int slot_unit(int s, int g, int e) { return 16 * (2 * s + (e >> 2)) + 4 * g + (e & 3); }

// acc += W x with the three-product split (A = weights of one 16-row tile, B = operand column block)
__device__ __forceinline__ f4 mfma3(const h8 a_hi, const h8 a_lo, const h8 b_hi, const h8 b_lo, f4 acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi, b_hi, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi, b_lo, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_lo, b_hi, acc, 0, 0, 0);
    return acc;
}

// writer side of an operand image: lane (w, g, n) owns units 16w + 4g + 0..3 of column n
__device__ __forceinline__ int img_wr_off(int w, int g, int n) { return n * ROWB + (w >> 1) * 64 + g * 16 + (w & 1) * 8; }
// reader side: k-step s of lane (g, n)
__device__ __forceinline__ int img_rd_off(int s, int g, int n) { return n * ROWB + s * 64 + g * 16; }

// LDS barrier that does NOT drain global loads/stores (what __syncthreads() would do)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

// 16 bytes to / from another workgroup's view of memory: agent-scope (sc1) 8-byte accesses -- written through
// to memory by the producer, never served from this CU's L1 or a stale L2 line on the consumer
__device__ __forceinline__ void store4_agent(float *p, const f4 v) {
#ifdef HPMN_DBG_PLAIN_ST
    *reinterpret_cast<f4 *>(p) = v; return;
#endif
    __hip_atomic_store(reinterpret_cast<u64 *>(p), __builtin_bit_cast(u64, f2{v.x, v.y}), RLX_AGENT);
    __hip_atomic_store(reinterpret_cast<u64 *>(p) + 1, __builtin_bit_cast(u64, f2{v.z, v.w}), RLX_AGENT);
}
__device__ __forceinline__ f4 load4_agent(const float *p) {
#ifdef HPMN_DBG_PLAIN_LD
    return *reinterpret_cast<const f4 *>(p);
#endif
    const u64 a = __hip_atomic_load(reinterpret_cast<const u64 *>(p), RLX_AGENT);
    const u64 b = __hip_atomic_load(reinterpret_cast<const u64 *>(p) + 1, RLX_AGENT);
    const f2 x = __builtin_bit_cast(f2, a), y = __builtin_bit_cast(f2, b);
    return f4{x.x, x.y, y.x, y.y};
}

// ---- launch descriptors (host -> kernel, by value) -----------------------------------------------------------
struct PipeLayer {
    const float *wg, *bg, *wc, *bc;
    const float *x;        // [B, T, D] input rows (layer 0: the materialised gather; layer i: y of layer i-1)
    float *y;              // [B, T/period, H] every period-th output, or NULL (last layer)
    float *hs, *gates;     // saved states [B,T+1,H] / [B,T,3H] (training) or NULL
    float *h_last;         // final state goes to h_last[b * h_last_stride + 0..H)
    long h_last_stride;
    // reverse pass
    const float *d_h_last; // [B] rows, same stride as h_last
    float *d_act;          // [B, T, 3H]
    float *d_x;            // [B, T, D]: gradient wrt this layer's input rows (= d_y of the layer below), or NULL
    const float *d_y;      // [B, T/period, H] gradient wrt the subsampled outputs (d_x of the layer above), or NULL
    int T, D, period, pad;
};

struct PipeArgs {
    int B, K, ntiles, train;
    unsigned *sync;        // [0] ticket, [1] error word, [2 ...] progress[K][ntiles][4]
    float *dump;           // >= 4 KB sink for the stores of a partial tile's dead columns
    PipeLayer L[HPMN_MAX_LAYERS];
};

constexpr int WAIT_AHEAD = 6;                   // rows a waiting consumer lets the producer get ahead (see wait_rows)
constexpr unsigned PIPE_SPIN_LIMIT = 1u << 26;   // bounded spins: a lost hand-off ends in an error code, not a hang

}  // namespace hpmn
