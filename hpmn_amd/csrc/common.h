// Shared helpers for the gfx950 kernels of libhpmn_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "hpmn_hip.h"

namespace hpmn {

void set_last_hip_error(int e);

// Record a launch failure and translate it to the ABI's error code.
inline int check_launch() {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_last_hip_error((int)e);
        return HPMN_EHIP;
    }
    return HPMN_OK;
}

// One wave == one workgroup in the scan kernels, so cross-lane hand-offs through LDS need
// no s_barrier: LDS instructions of one wave execute in order.  This only stops the
// compiler from moving LDS accesses across the hand-off point.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// sigmoid / tanh on the transcendental unit (v_exp_f32 + v_rcp_f32, ~1 ulp each): the
// absolute error (~1e-7) is far inside the 1e-4 logit tolerance of the parity tests.
__device__ __forceinline__ float fast_sigmoid(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __expf(-x));
}
__device__ __forceinline__ float fast_tanh(float x) {
    return 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(-2.0f * x)) - 1.0f;
}

// The same with the exponent pre-scaled by the caller (z = -log2(e) * x, resp. -2 log2(e) * x): the scan
// kernels fold the scale into their register-stationary weights, which takes two dependent
// multiplies off the serial chain of every step.
constexpr float NEG_LOG2E = -1.4426950408889634f;
__device__ __forceinline__ float sigmoid_scaled(float z) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z));
}
__device__ __forceinline__ float tanh_scaled(float z2) {
    return fmaf(2.0f, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z2)), -1.0f);
}

typedef float f2 __attribute__((ext_vector_type(2)));

// Table ids arrive as int32 or -- for tables beyond 2^31 - 1 rows (BASELINE configs[4] sized to HBM) -- int64: bit
// HPMN_ID_I64 of the id-flags word every entry point with ids carries (bit HPMN_ID_MASK0 is the id-0 mask of
// code/hpmn.py:417-422).  The flag is a kernel argument, so the branch is wave-uniform (s_cbranch around two loads); row
// arithmetic is 64-bit everywhere.
// BRANCH-FREE on purpose: a load inside a (wave-uniform) branch is waited for at the join, s_waitcnt vmcnt(0) -- seen in the
// ISA of the sorted scatter, where sixteen row loads meant to be in flight together were serialised by the id load between
// them.  Two 4-byte loads that are valid for either width (int32: element i twice; int64: its low and high word) and a select.
__device__ __forceinline__ long load_id(const void *__restrict__ ids, long i, int id_flags) {
    const int w = (id_flags & HPMN_ID_I64) ? 1 : 0;
    const int *p = reinterpret_cast<const int *>(ids) + (i << w);
    const int lo = p[0], hi = p[w];
    const long wide = (long)(((unsigned long)(unsigned)hi << 32) | (unsigned long)(unsigned)lo);
    return w ? wide : (long)lo;
}
__device__ __forceinline__ bool id_masked(long id, int id_flags) { return (id_flags & HPMN_ID_MASK0) && id == 0; }

// Progress counters between the waves of a workgroup live in LDS and MUST be accessed as LDS: a `volatile int *`
// parameter is a GENERIC pointer, for which hipcc emits flat_load/flat_store ... sc0 sc1 followed by
// s_waitcnt vmcnt(0) lgkmcnt(0) -- every poll then drains every outstanding global store of the wave (seen in the
// ISA of the fused forward kernel: a full store drain every ~8 steps of the scan wave).  These two go through an
// address_space(3) pointer: plain ds_read_b32 / ds_write_b32, lgkmcnt only.
typedef __attribute__((address_space(3))) volatile int lds_int;
// (every lane reads the same word: handing the value back through v_readfirstlane makes it -- and every comparison and
//  loop on it -- wave-uniform, i.e. s_cmp + s_cbranch instead of v_cmp + exec-mask bookkeeping around the poll loops)
__device__ __forceinline__ int lds_counter_peek(int *p) { return __builtin_amdgcn_readfirstlane(*(lds_int *)p); }
// Publishing: data first, then the counter.  The LDS unit executes the DS instructions of one wave in the order they
// were issued, so the counter write cannot overtake the data writes in front of it and no s_waitcnt is needed in
// between (with one, every publish stalled its wave for an LDS write latency -- on the serial chain, once per step);
// the compiler barrier keeps the program order.  HPMN_LDS_PUBLISH_WAIT restores the wait (tools, bisecting).
__device__ __forceinline__ void lds_counter_set(int *p, int v) {
#ifdef HPMN_LDS_PUBLISH_WAIT
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#else
    asm volatile("" ::: "memory");
#endif
    *(lds_int *)p = v;
    asm volatile("" ::: "memory");
}

// "Use" a whole group of just-loaded LDS values in one place: the compiler's waitcnt pass then emits ONE
// s_waitcnt (for the newest of them) in front of this statement instead of one in front of each first
// use.  Every s_waitcnt costs the single wave of a scan workgroup a 4-cycle issue slot
// (tools/micro: pk_fma pairs separated by s_waitcnt/s_nop run at 7.5 instead of 5.5 cycles per pk_fma).
typedef float v4f __attribute__((ext_vector_type(4)));
// The two accumulators ride along so that the statement stays behind the previous group's FMAs (the
// scheduler otherwise hoists it and the last group is waited for with nothing left to overlap).
template <int G>
__device__ __forceinline__ void land_group(v4f (&v)[G], f2 &a, f2 &b) {
    static_assert(G == 2 || G == 4, "group sizes in use");
    if constexpr (G == 4) asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(a), "+v"(b));
    else                  asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(a), "+v"(b));
}

// Make a register-stationary value "defined here" for the compiler's s_waitcnt bookkeeping.  Weights
// are loaded once before the time loop; without this the waitcnt pass keeps treating them as results of
// still-pending loads on the loop back-edge and emits s_waitcnt vmcnt(0..2) in front of their uses in
// EVERY step, which drains the step's own prefetch loads and output stores (an HBM round trip on the
// serial chain; seen in the ISA of the scan kernels).
__device__ __forceinline__ void settle(float &x) { asm volatile("" : "+v"(x)); }
__device__ __forceinline__ void settle(f2 &x) { asm volatile("" : "+v"(x)); }

// BPTT through h = u h_prev + (1 - u) c WITHOUT the stored candidate (include/hpmn_hip.h, HPMN_BWD_CANDIDATE_FROM_HS):
// q = (1 - u) c = h - u h_prev;  k1 = (1 - u)(1 - c^2) = (1 - u) - q c;  k2 = (h_prev - c) u (1 - u) = u ((1 - u) h_prev - q).
// Where u rounds to 1 the candidate cannot be recovered and is not needed: k1 = 0, |k2| <= one rounding of h (true value 0).
// c = q / (1 - u) carries an absolute error of eps |h| / (1 - u); it only enters k1 multiplied by q = O(1 - u).
__device__ __forceinline__ void gru_coeff_from_states(float h_new, float h_prev, float u, float omu, float &k1, float &k2) {
    const float q = fmaf(-u, h_prev, h_new);
    const float c = omu > 0.f ? q * __builtin_amdgcn_rcpf(omu) : 0.f;
    k1 = fmaf(-q, c, omu);
    k2 = u * fmaf(h_prev, omu, -q);
}

// acc0/acc1 += sum over NQ float4's of a wave-uniform LDS row (a broadcast) times the lane's packed
// weights w[2*q], w[2*q+1] (pairs over consecutive k).  The LDS reads are software-pipelined in
// groups of G float4 (G reads in flight while the previous group's packed FMAs issue) and fenced
// with a compiler memory barrier so that not all NQ reads are hoisted at once -- with 3H weights resident
// that costs 4*NQ VGPRs and pushes the weights into AGPRs (measured: +98 v_accvgpr_read per step).
template <int NQ, int G = 4>
__device__ __forceinline__ void bcast_matvec(const float4 *row4, const f2 *w, f2 &acc0, f2 &acc1) {
    static_assert(NQ % G == 0, "groups");
    const v4f *row = reinterpret_cast<const v4f *>(row4);
    v4f cur[G], nxt[G];
#pragma unroll
    for (int i = 0; i < G; ++i) cur[i] = row[i];
#pragma unroll
    for (int g = 0; g < NQ / G; ++g) {
        if (g + 1 < NQ / G) {
#pragma unroll
            for (int i = 0; i < G; ++i) nxt[i] = row[(g + 1) * G + i];
        }
        asm volatile("" ::: "memory");   // no later LDS read may be hoisted above this point
        land_group<G>(cur, acc0, acc1);
#pragma unroll
        for (int i = 0; i < G; ++i) {
            const int q = g * G + i;
            acc0 = __builtin_elementwise_fma(f2{cur[i].x, cur[i].y}, w[2 * q], acc0);
            acc1 = __builtin_elementwise_fma(f2{cur[i].z, cur[i].w}, w[2 * q + 1], acc1);
        }
#pragma unroll
        for (int i = 0; i < G; ++i) cur[i] = nxt[i];
    }
}

// bcast_matvec whose accumulators START here (first products are multiplies: no zero-initialising moves, which cost
// the single wave of a scan an issue slot each).  OFF: first float4 of the row (the caller may split a product in two).
template <int NQ, int G = 4>
__device__ __forceinline__ void bcast_matvec_first(const float4 *row4, const f2 *w, f2 &acc0, f2 &acc1) {
    static_assert(NQ % G == 0, "groups");
    const v4f *row = reinterpret_cast<const v4f *>(row4);
    v4f cur[G], nxt[G];
#pragma unroll
    for (int i = 0; i < G; ++i) cur[i] = row[i];
#pragma unroll
    for (int g = 0; g < NQ / G; ++g) {
        if (g + 1 < NQ / G) {
#pragma unroll
            for (int i = 0; i < G; ++i) nxt[i] = row[(g + 1) * G + i];
        }
        asm volatile("" ::: "memory");
        if (g == 0) asm volatile("" : "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2]), "+v"(cur[3]));
        else        land_group<G>(cur, acc0, acc1);
#pragma unroll
        for (int i = 0; i < G; ++i) {
            const int q = g * G + i;
            if (q == 0) {
                acc0 = f2{cur[i].x, cur[i].y} * w[0];
                acc1 = f2{cur[i].z, cur[i].w} * w[1];
            } else {
                acc0 = __builtin_elementwise_fma(f2{cur[i].x, cur[i].y}, w[2 * q], acc0);
                acc1 = __builtin_elementwise_fma(f2{cur[i].z, cur[i].w}, w[2 * q + 1], acc1);
            }
        }
#pragma unroll
        for (int i = 0; i < G; ++i) cur[i] = nxt[i];
    }
}

// K-SPLIT mat-vec, y[unit] = sum_k x[k] W[unit][k] over 64 k with lane == unit on both sides, for the scan waves that
// share a CU's LDS pipe.  A broadcast ds_read_b128 returns 1 KB to the wave (8 cycles of the CU's 128 B/clk return
// path) however few distinct bytes it carries, and bcast_matvec needs 16 of them: with two sequences (four waves) per
// CU the pipe is busy ~60 % of a step and every round trip on the serial chain queues behind the neighbours' reads
// (measured: the same kernel runs 18 % faster with one sequence per CU).  Here KS adjacent lanes share KS units: lane
// (g = lane / KS, s = lane % KS) multiplies the KS rows of units KS g .. KS g + KS-1 over its own 64/KS-wide slice of
// k -- 16/KS reads instead of 16, the same 32 packed FMAs -- and the slices are summed with quad_perm DPP adds
// (full-rate VALU, no LDS), after which the lane keeps the sum of unit KS g + s == lane.
//   w[j][i]: row of unit KS*(lane/KS) + j, k = (64/KS)*(lane%KS) + 2i, 2i+1      (loaded by split_matvec_weights)
template <int KS>
__device__ __forceinline__ void split_matvec_weights(const float *wrows, long row_stride, int lane, f2 (&w)[KS][32 / KS]) {
    const int g = lane / KS, s = lane % KS;
#pragma unroll
    for (int j = 0; j < KS; ++j)
#pragma unroll
        for (int i = 0; i < 32 / KS; ++i)
            w[j][i] = *reinterpret_cast<const f2 *>(wrows + (long)(KS * g + j) * row_stride + (64 / KS) * s + 2 * i);
#pragma unroll
    for (int j = 0; j < KS; ++j)
#pragma unroll
        for (int i = 0; i < 32 / KS; ++i) settle(w[j][i]);
}
// the same from a [k][unit] matrix (the forward's layout: element (unit, k) at wk[k * k_stride + unit]), times scale
template <int KS>
__device__ __forceinline__ void split_matvec_weights_t(const float *wk, long k_stride, float scale, int lane, f2 (&w)[KS][32 / KS]) {
    const int g = lane / KS, s = lane % KS;
#pragma unroll
    for (int j = 0; j < KS; ++j)
#pragma unroll
        for (int i = 0; i < 32 / KS; ++i) {
            const long k = (64 / KS) * s + 2 * i;
            w[j][i] = f2{wk[k * k_stride + KS * g + j], wk[(k + 1) * k_stride + KS * g + j]} * scale;
        }
#pragma unroll
    for (int j = 0; j < KS; ++j)
#pragma unroll
        for (int i = 0; i < 32 / KS; ++i) settle(w[j][i]);
}

template <int CTRL>
__device__ __forceinline__ float dpp_quad(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}

// x: the 64-float operand row in LDS (16-byte aligned); returns y[lane]
template <int KS>
__device__ __forceinline__ float split_matvec(const float *x, const f2 (&w)[KS][32 / KS], int lane) {
    static_assert(KS == 2 || KS == 4, "quad_perm reductions");
    constexpr int NQ = 16 / KS;          // float4s of this lane's k slice
    constexpr int G = 4;
    const v4f *row = reinterpret_cast<const v4f *>(x + (64 / KS) * (lane % KS));
    f2 acc[KS][2];
    v4f cur[G], nxt[G];
#pragma unroll
    for (int i = 0; i < G; ++i) cur[i] = row[i];
#pragma unroll
    for (int g = 0; g < NQ / G; ++g) {
        if (g + 1 < NQ / G) {
#pragma unroll
            for (int i = 0; i < G; ++i) nxt[i] = row[(g + 1) * G + i];
        }
        asm volatile("" ::: "memory");
        if (g == 0) asm volatile("" : "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2]), "+v"(cur[3]));
        else        asm volatile("" : "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2]), "+v"(cur[3]), "+v"(acc[0][0]), "+v"(acc[0][1]));
#pragma unroll
        for (int i = 0; i < G; ++i) {
            const int q = g * G + i;
            const f2 lo = {cur[i].x, cur[i].y}, hi = {cur[i].z, cur[i].w};
#pragma unroll
            for (int j = 0; j < KS; ++j) {
                if (q == 0) {
                    acc[j][0] = lo * w[j][0];
                    acc[j][1] = hi * w[j][1];
                } else {
                    acc[j][0] = __builtin_elementwise_fma(lo, w[j][2 * q], acc[j][0]);
                    acc[j][1] = __builtin_elementwise_fma(hi, w[j][2 * q + 1], acc[j][1]);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < G; ++i) cur[i] = nxt[i];
    }
    float sj[KS];
#pragma unroll
    for (int j = 0; j < KS; ++j) {
        const f2 t = acc[j][0] + acc[j][1];
        sj[j] = t.x + t.y;
    }
    if constexpr (KS == 2) {
        // lane keeps unit (lane & 1): add the partner's slice of that unit
        const float mine = (lane & 1) ? sj[1] : sj[0];
        const float give = (lane & 1) ? sj[0] : sj[1];
        return mine + dpp_quad<0xB1>(give);                      // quad_perm [1,0,3,2]
    } else {
        const bool b0 = lane & 1, b1 = lane & 2;
        // stage 1 (xor 1): of each unit pair keep the one with my low bit, hand the other to the neighbour
        const float k01 = b0 ? sj[1] : sj[0], g01 = b0 ? sj[0] : sj[1];
        const float k23 = b0 ? sj[3] : sj[2], g23 = b0 ? sj[2] : sj[3];
        const float r01 = k01 + dpp_quad<0xB1>(g01);
        const float r23 = k23 + dpp_quad<0xB1>(g23);
        // stage 2 (xor 2): keep the pair with my high bit
        const float kk = b1 ? r23 : r01, gg = b1 ? r01 : r23;
        return kk + dpp_quad<0x4E>(gg);                          // quad_perm [2,3,0,1]
    }
}

// Two weight sets over the same operand row (the forward's reset and update gates), KS = 2.
__device__ __forceinline__ void split_matvec2x(const float *x, const f2 (&wa)[2][16], const f2 (&wb)[2][16], int lane,
                                               float &ya, float &yb) {
    constexpr int NQ = 8, G = 4;
    const v4f *row = reinterpret_cast<const v4f *>(x + 32 * (lane & 1));
    f2 a[2][2], b[2][2];
    v4f cur[G], nxt[G];
#pragma unroll
    for (int i = 0; i < G; ++i) cur[i] = row[i];
#pragma unroll
    for (int g = 0; g < NQ / G; ++g) {
        if (g + 1 < NQ / G) {
#pragma unroll
            for (int i = 0; i < G; ++i) nxt[i] = row[(g + 1) * G + i];
        }
        asm volatile("" ::: "memory");
        if (g == 0) asm volatile("" : "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2]), "+v"(cur[3]));
        else        asm volatile("" : "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2]), "+v"(cur[3]), "+v"(a[0][0]), "+v"(b[0][0]));
#pragma unroll
        for (int i = 0; i < G; ++i) {
            const int q = g * G + i;
            const f2 lo = {cur[i].x, cur[i].y}, hi = {cur[i].z, cur[i].w};
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (q == 0) {
                    a[j][0] = lo * wa[j][0]; a[j][1] = hi * wa[j][1];
                    b[j][0] = lo * wb[j][0]; b[j][1] = hi * wb[j][1];
                } else {
                    a[j][0] = __builtin_elementwise_fma(lo, wa[j][2 * q], a[j][0]);
                    a[j][1] = __builtin_elementwise_fma(hi, wa[j][2 * q + 1], a[j][1]);
                    b[j][0] = __builtin_elementwise_fma(lo, wb[j][2 * q], b[j][0]);
                    b[j][1] = __builtin_elementwise_fma(hi, wb[j][2 * q + 1], b[j][1]);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < G; ++i) cur[i] = nxt[i];
    }
    float sa[2], sb[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const f2 ta = a[j][0] + a[j][1], tb = b[j][0] + b[j][1];
        sa[j] = ta.x + ta.y;
        sb[j] = tb.x + tb.y;
    }
    const bool odd = lane & 1;
    ya = (odd ? sa[1] : sa[0]) + dpp_quad<0xB1>(odd ? sa[0] : sa[1]);
    yb = (odd ? sb[1] : sb[0]) + dpp_quad<0xB1>(odd ? sb[0] : sb[1]);
}

// Same row feeding two weight sets (the r and u columns of the gate kernel).
template <int NQ, int G = 4>
__device__ __forceinline__ void bcast_matvec2(const float4 *row4, const f2 *wa, const f2 *wb, f2 &a0, f2 &b0) {
    static_assert(NQ % G == 0, "groups");
    const v4f *row = reinterpret_cast<const v4f *>(row4);
    v4f cur[G], nxt[G];
#pragma unroll
    for (int i = 0; i < G; ++i) cur[i] = row[i];
#pragma unroll
    for (int g = 0; g < NQ / G; ++g) {
        if (g + 1 < NQ / G) {
#pragma unroll
            for (int i = 0; i < G; ++i) nxt[i] = row[(g + 1) * G + i];
        }
        asm volatile("" ::: "memory");
        land_group<G>(cur, a0, b0);
#pragma unroll
        for (int i = 0; i < G; ++i) {
            const int q = g * G + i;
            const f2 lo = {cur[i].x, cur[i].y}, hi = {cur[i].z, cur[i].w};
            a0 = __builtin_elementwise_fma(lo, wa[2 * q], a0);
            b0 = __builtin_elementwise_fma(lo, wb[2 * q], b0);
            a0 = __builtin_elementwise_fma(hi, wa[2 * q + 1], a0);
            b0 = __builtin_elementwise_fma(hi, wb[2 * q + 1], b0);
        }
#pragma unroll
        for (int i = 0; i < G; ++i) cur[i] = nxt[i];
    }
}

}  // namespace hpmn
