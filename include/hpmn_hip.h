/*
 * hpmn_hip.h -- C ABI of libhpmn_hip.so, the MI355X (gfx950) implementation of the
 * HPMN hot path: embedding gather -> K-layer periodic GRU memory update -> (saved
 * states for BPTT) -> reverse scan -> embedding-gradient scatter -> TF-form Adam.
 *
 * The reference (alimamarankgroup/HPMN, Python 2.7 + TensorFlow 1.4) has no FFI: the
 * path is inline TF graph code.  Each entry point below names the reference lines it
 * replaces (paths relative to the reference checkout); INTEGRATION.md shows the
 * ctypes stub a maintainer would add on the reference side.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer borrowed from the caller (PyTorch tensors in
 *    this repo); the library never allocates device memory, frees or synchronises (the only objects it
 *    creates are the helper stream + two events of an HpmnTrainCtx);
 *  - all work is enqueued on `stream` (a hipStream_t passed as void*); no host sync;
 *  - all tensors are dense row-major, fp32 unless stated;
 *  - TABLE IDS are int32 or int64: every entry point that takes ids (`const void *ids`) also carries an id-flags word
 *    -- the argument / struct field historically named `mask_id0` -- whose bit HPMN_ID_MASK0 asks for the id-0 mask of
 *    code/hpmn.py:417-422 and whose bit HPMN_ID_I64 says the ids are int64.  The reference feeds int32 placeholders
 *    (code/hpmn.py:248-251); a table of more than 2^31 - 1 rows (BASELINE configs[4], "tables sized to 288 GB") needs the
 *    wide form -- a documented deviation (hard part 4).  Row arithmetic is 64-bit throughout;
 *  - return value: 0 (HPMN_OK) or a negative HPMN_E* code; a failing HIP runtime call
 *    is reported as HPMN_EHIP and its hipError_t kept in hpmn_last_hip_error();
 *  - nothing throws across the ABI; thread-safe for distinct streams (the only global
 *    state is the thread-local last-hip-error word).
 */
#ifndef HPMN_HIP_H_
#define HPMN_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HPMN_ABI_VERSION 14
#define HPMN_ID_MASK0 1   /* id-flags bit 0: id 0 gathers a zero row and receives no gradient (the Hpmn class)  */
#define HPMN_ID_I64 2     /* id-flags bit 1: the ids tensor is int64 (default: int32)                          */
#define HPMN_ID_HOT 4     /* id-flags bit 2 (ABI v14; a HINT to the gradient scatter, ignored elsewhere): many lookups share
                           * few rows (a heavy-tailed id law, a 4-valued behaviour-tag column): hpmn_embed_grad_scatter / the
                           * scatter inside hpmn_scan_bwd pre-reduce equal ids of a wave's time segment in an LDS table before
                           * the atomic row adds -- same sums, fewer atomics on the hot rows                           */
#define HPMN_MAX_LAYERS 12
#define HPMN_MAX_CHUNKS 32 /* row-range chunks hpmn_scatter_plan_build counts distinct rows for                        */
#define HPMN_MAX_RANKS 8   /* data-parallel ranks hpmn_rows_sum_adam tells apart (a rank bit per flags byte)        */
/* Saved gates without the candidate (ABI v11): the forward leaves the candidate third of every gates row unwritten
 * (HpmnGruFusedFwd.flags), the reverse scan recovers what it needs of it from the saved states it reads anyway
 * (HpmnGruBwd.flags): h_t = u h_{t-1} + (1 - u) c gives q = (1 - u) c = h_t - u h_{t-1}, and the candidate only ever enters
 * BPTT through q: (1 - u)(1 - c^2) = (1 - u) - q c, (h_{t-1} - c) u (1 - u) = u ((1 - u) h_{t-1} - q); c = q / (1 - u) for
 * the one remaining factor (taken as 0 where u rounds to 1: both coefficients vanish there).  A third of the saved-gates
 * traffic (two 128-byte lines per step at H = 64, written by the forward and read by the reverse scan) disappears; the
 * coefficients differ from the float64 ones by <= ~2.4e-7 ABSOLUTE (they are O(1)) -- 3.2e-7 in the corner u = 1 - 2^-24, where q is
 * smaller than the forward's own rounding of h (tests/test_candidate_elision_cpu.py, tests/test_gpu_parity.py).  hpmn_gru_candidate_elision(H, B) != 0 where both sides support it. */
#define HPMN_FWD_NO_CANDIDATE 1      /* HpmnGruFusedFwd.flags bit 0: gates[..., 2H:3H] is NOT written            */
#define HPMN_BWD_CANDIDATE_FROM_HS 1 /* HpmnGruBwd.flags bit 0: gates[..., 2H:3H] is NOT read (see above)        */
#define HPMN_BWD_SCATTER_INLOOP 2    /* HpmnGruBwd.flags bit 1 (ABI v14): with d_emb AND d_x set (H = 64, D <= 32) the input
                                      * gradient's tiles are added to d_emb INSIDE the scan's loop and d_x is SCRATCH (it does
                                      * not hold the input gradient afterwards).  Without the bit d_emb + d_x means: d_x is
                                      * written, the scatter runs as the launch's epilogue.                              */

enum {
    HPMN_OK = 0,
    HPMN_EINVAL = -1,      /* null pointer / non-positive size / inconsistent desc      */
    HPMN_EUNSUPPORTED = -2,/* shape outside what the gfx950 kernels are instantiated for */
    HPMN_EHIP = -3,        /* HIP runtime error, see hpmn_last_hip_error()               */
    HPMN_ENODEVICE = -4    /* no gfx950 device visible                                   */
};

int hpmn_abi_version(void);
/* 1 when the library was built with -DHPMN_LEGACY_KERNELS: the measured-slower kernel generations (first-generation fused
 * forward, helper-wave reverse scan, the tile kernel's training backward hpmn_pipe_bwd) are compiled in and selectable
 * through their switches; 0 (default build): those switches fall through to the current kernels, hpmn_pipe_bwd returns
 * HPMN_EUNSUPPORTED. */
int hpmn_has_legacy_kernels(void);
const char *hpmn_strerror(int code);
int hpmn_last_hip_error(void);
/* 1 when a (H, D) GRU layer shape has a kernel instantiation, else 0. */
int hpmn_gru_shape_supported(int32_t H, int32_t D);

/* ------------------------------------------------------------------------------------
 * Embedding gather.  Replaces Hpmn.embedding / Hpmn_Industry.embedding
 * (code/hpmn.py:414-423, :266-276):  out[n, f*E:(f+1)*E] = emb[ids[n,f]] * (mask_id0 ?
 * ids[n,f] != 0 : 1).  Also the gather-only roofline micro-benchmark.
 *   ids [N,F] int32 / int64 (mask_id0 = id flags, see Conventions), emb [V,E], out [N, F*E].   E % 4 == 0.
 * ---------------------------------------------------------------------------------- */
int hpmn_embed_gather(const void *ids, const float *emb, float *out,
                      int64_t N, int32_t F, int32_t E, int64_t V, int32_t mask_id0,
                      void *stream);

/* ------------------------------------------------------------------------------------
 * One GRU layer of build_memory, forward, in two launches.  Together they replace one
 * iteration of the loop at code/hpmn.py:116-128: tf.nn.rnn_cell.GRUCell (arithmetic
 * mirrored at code/util.py:81-110 without line 108) driven by tf.nn.dynamic_rnn from a
 * zero state with no sequence_length masking (code/rnn.py:583-588, 754-768), followed by
 * the "take every period-th output" reshape+gather (code/hpmn.py:124-128).
 *
 *   wg [D+H, 2H], bg [2H], wc [D+H, H], bc [H]      TF variable layout (input rows first)
 *
 * (1) hpmn_gru_input_proj -- the time-parallel half (the x rows of _Linear,
 *     code/util.py:88-95, 99-107):
 *        xp[b,t, 0:2H] = s (x[b,t] wg[0:D] + bg),    xp[b,t, 2H:3H] = 2s (x[b,t] wc[0:D] + bc),
 *     s = -log2(e): xp is an opaque hand-over buffer between (1) and (2), kept in the exponent
 *     domain of the scan's exp2-based sigmoid/tanh so no multiply sits on the serial chain.
 *     Input is EITHER x [B,T,D] (layers >= 1: the subsampled outputs of the layer below)
 *     OR, when x == NULL, gathered on the fly from (ids [B,Tids,F], emb [V,E]) with
 *     `front_zero` all-zero steps in front (code/hpmn.py:288-289), T == front_zero+Tids,
 *     D == F*E, and the id-0 mask of code/hpmn.py:417-422 when mask_id0 != 0 -- i.e. it
 *     also replaces Hpmn.embedding (code/hpmn.py:414-423 / :266-276) for the scan.
 *        x_out : optional [B,T,D], the materialised gathered input (training only).
 *
 * (2) hpmn_gru_scan_fwd -- the serial half: for t in 0..T-1
 *        [r,u] = sigmoid(xp[:, t, 0:2H]/s + h wg[D:]);  c = tanh(xp[:, t, 2H:]/2s + (r*h) wc[D:])
 *        h = u*h + (1-u)*c
 *   h_last  : final state, written at h_last[b*h_last_stride + 0..H)   (memory[:, i, :])
 *   y       : optional [B, T/period, H]  = outputs[:, period-1::period, :]   (next layer's input)
 *   hs      : optional [B, T+1, H]  hs[b,0]=0, hs[b,t+1] = state after step t      (training)
 *   gates   : optional [B, T, 3H]   (r, u, c) per step                               (training)
 * ---------------------------------------------------------------------------------- */
typedef struct HpmnInputProj {
    int32_t B, T, D, H;
    const float *x;
    const void *ids;
    const float *emb;
    int32_t Tids, F, E, front_zero, mask_id0;
    int64_t V;
    const float *wg, *bg, *wc, *bc;
    float *xp;     /* [B, T, 3H] */
    float *x_out;  /* optional [B, T, D] (gather mode) */
    int32_t t_begin, t_len;  /* only steps [t_begin, t_begin+t_len) of every sequence; t_len == 0: all T */
} HpmnInputProj;

int hpmn_gru_input_proj(const HpmnInputProj *args, void *stream);

typedef struct HpmnGruFwd {
    int32_t B, T, D, H;
    const float *xp;   /* [B, T, 3H] from hpmn_gru_input_proj */
    const float *wg, *wc;
    float *h_last;
    int64_t h_last_stride;
    float *y;
    int32_t period;
    float *hs;
    float *gates;
    /* time-chunked launches (cross-layer pipelining): run only steps [t_begin, t_end) (both multiples of 2
     * and of period; t_end == 0: to T) starting from h_init[b*h_init_stride + 0..H) (NULL: zero state);
     * h_last always receives the state after the last step run. */
    int32_t t_begin, t_end;
    const float *h_init;
    int64_t h_init_stride;
} HpmnGruFwd;

int hpmn_gru_scan_fwd(const HpmnGruFwd *args, void *stream);

/* ------------------------------------------------------------------------------------
 * (1)+(2) fused: the forward of one GRU layer in ONE launch, without the xp hand-over buffer.
 * A workgroup owns a sequence and runs two specialised waves on two SIMDs of its CU: a
 * projection wave computes the input half x_t W[0:D] + b (gathering x_t from (ids, emb) for
 * layer 0) a few steps AHEAD of the recurrence and hands it over through an LDS ring; the
 * scan wave runs the serial recurrence exactly as hpmn_gru_scan_fwd does.  Same arguments as
 * HpmnInputProj + HpmnGruFwd (whole sequences only: no t_begin/t_end), same results.
 * hpmn_gru_fused_fwd_supported(H, D, gather) tells which shapes have this path (others:
 * HPMN_EUNSUPPORTED -- call (1) then (2)).
 * ---------------------------------------------------------------------------------- */
typedef struct HpmnGruFusedFwd {
    int32_t B, T, D, H;
    const float *x;            /* [B,T,D], or NULL: gather from (ids, emb) as in HpmnInputProj */
    const void *ids;
    const float *emb;
    int32_t Tids, F, E, front_zero, mask_id0;
    int64_t V;
    const float *wg, *bg, *wc, *bc;
    float *x_out;              /* optional [B,T,D]: the gathered input (training, gather mode)   */
    float *h_last;
    int64_t h_last_stride;
    float *y;                  /* optional [B, T/period, H] */
    int32_t period;
    float *hs;                 /* optional [B,T+1,H]  (training) */
    float *gates;              /* optional [B,T,3H]   (training) */
    float *last;               /* optional [B,D]: a copy of the (gathered, masked) input row of step last_t -- the read
                                * path's uinp[:, last_index, :] -- written by the launch itself.  Only where
                                * hpmn_gru_fused_fwd_writes_last() != 0 (HPMN_EUNSUPPORTED otherwise); NULL: none */
    int32_t last_t;
    int32_t flags;             /* HPMN_FWD_NO_CANDIDATE (ABI v11; this word was padding before: 0 = the old behaviour) */
} HpmnGruFusedFwd;

int hpmn_gru_fused_fwd_supported(int32_t H, int32_t D, int32_t gather);
int hpmn_gru_fused_fwd_writes_last(void);
int hpmn_gru_fused_fwd(const HpmnGruFusedFwd *args, void *stream);

/* ------------------------------------------------------------------------------------
 * TWO consecutive layers of build_memory in ONE launch (H = 64): layer i+1 runs while layer i runs.
 * code/hpmn.py:124-128 hands layer i+1 every period-th output of layer i -- a pipeline that one launch
 * per layer serialises.  A workgroup owns two sequences and both layers of them (eight waves, two per
 * SIMD: chain + producer wave of either layer, as in hpmn_gru_fused_fwd); the rows that fire go from the
 * lower layer's producer wave to the upper layer's through an LDS ring behind two LDS counters -- no
 * global flag, no dependence on dispatch order.
 *   lo : the lower layer, exactly as for hpmn_gru_fused_fwd (gather or x rows; lo.y may be NULL when nobody
 *        else needs the subsampled outputs, e.g. in inference)
 *   up : the upper layer: up.T == lo.T / lo.period, up.D == H, up.B == lo.B; up.x is IGNORED (the rows never
 *        leave the CU); everything else as for hpmn_gru_fused_fwd.  (lo.hs == NULL) == (up.hs == NULL).
 *   img_lo, img_up : the layers' projection weights as MFMA operand images (hpmn_gru_proj_images; img_lo is only
 *        read when lo.D == 64), or NULL: the call builds them itself, in `scratch` (>= hpmn_gru_pair_fwd_scratch_bytes()
 *        bytes of device memory, 16-byte aligned, rewritten by every such call on `stream`)
 *   flags : bit 0 swaps which SIMD pair hosts the upper layer's chain / producer waves (measurement switch)
 * Results are bit-identical to two hpmn_gru_fused_fwd calls.  One workgroup per CU is resident (two waves
 * per SIMD at 256 registers): meant for (B + 1) / 2 <= number of CUs; larger batches run but serialise.
 * ---------------------------------------------------------------------------------- */
typedef struct HpmnGruPairFwd {
    HpmnGruFusedFwd lo, up;
    void *scratch;
    const float *img_lo, *img_up;
    int32_t flags, pad_;
} HpmnGruPairFwd;

/* MFMA operand images of the input-projection weights of n layers (H = 64, D[i] in {32, 64}) in one launch: img[i]
 * receives hpmn_gru_proj_image_floats(D[i]) floats (16-byte aligned).  They depend on the weights only: a caller that
 * runs several pair launches per step builds all of them once, in front of the first. */
size_t hpmn_gru_proj_image_floats(int32_t D);
int hpmn_gru_proj_images(int32_t n, const float *const *wg, const float *const *bg, const float *const *wc,
                         const float *const *bc, const int32_t *D, float *const *img, void *stream);

int hpmn_gru_pair_fwd_supported(int32_t H, int32_t D_lo, int32_t gather);
size_t hpmn_gru_pair_fwd_scratch_bytes(void);
int hpmn_gru_pair_fwd(const HpmnGruPairFwd *args, void *stream);

/* ------------------------------------------------------------------------------------
 * One GRU layer, reverse scan (BPTT) -- the serial part of the gradient of
 * hpmn_gru_scan_fwd (TF autodiff through the while_loop of code/hpmn.py:119-120).
 *   d_h_last [B] rows at d_h_last[b*stride + 0..H) : gradient wrt the final state
 *   d_y      : optional [B, T/period, H]  gradient wrt the subsampled outputs
 *   hs, gates: as saved by the forward
 *   d_act    : out [B, T, 3H] = gradients wrt the pre-activations (a_r, a_u, a_c)
 * Weight / input gradients are then plain GEMMs over d_act (host side):
 *   dWg = [x | h_prev]^T d_act[:, :2H],  dWc = [x | r*h_prev]^T d_act[:, 2H:],
 *   dx  = d_act[:, :2H] Wg[:D]^T + d_act[:, 2H:] Wc[:D]^T.
 * ---------------------------------------------------------------------------------- */
typedef struct HpmnGruBwd {
    int32_t B, T, D, H;
    const float *wg, *wc;
    const float *hs, *gates;
    const float *d_h_last;
    int64_t d_h_last_stride;
    const float *d_y;
    int32_t period;
    float *d_act;
    /* time-chunked launches: reverse steps t_end-1 .. t_begin (t_end == 0: T).  The gradient flowing into
     * step t_end-1 is d_h_last when t_end == T, else dh_carry[b*H + 0..H) left by the launch for the later
     * chunk; when t_begin > 0 the gradient wrt the state before step t_begin is written to dh_carry. */
    int32_t t_begin, t_end;
    float *dh_carry;
    /* optional [B, T, D]: the gradient wrt the layer's input rows, d_act [wg[0:D] | wc[0:D]]^T, produced by a third
     * wave of the scan's workgroups underneath the scan (what hpmn_gru_input_grad computes as a launch of its own).
     * Only where hpmn_gru_scan_bwd_fuses_dx(H, B) != 0 and D is 16, 32 or 64; elsewhere it must be NULL
     * (HPMN_EUNSUPPORTED otherwise). */
    float *d_x;
    /* optional, with the fused input gradient of LAYER 0 only (D == F * E, E == 16): instead of (d_x == NULL) or besides
     * writing d_x, the launch adds the input gradient straight into the embedding-table gradient --
     *   d_emb[ids[b, t - front_zero, f]] += d_x[b, t, f*E:(f+1)*E]  for t >= front_zero (skipping id 0 when mask_id0),
     *   with d_last[b] (the read path's gradient wrt uinp[:, last_index, :], may be NULL) added at step last_t --
     * what hpmn_embed_grad_scatter does as a launch of its own behind this one.  Runs of equal ids (the constant uid column,
     * padding) are summed in registers before one atomic row add.  NULL d_emb: off. */
    const void *scatter_ids;   /* [B, Tids, F] */
    float *d_emb;                 /* [V, E] */
    const float *d_last;          /* [B, D] */
    int32_t Tids, F, E, front_zero, mask_id0, last_t;
    int32_t flags, pad_;          /* HPMN_BWD_CANDIDATE_FROM_HS (ABI v11) */
} HpmnGruBwd;

int hpmn_gru_scan_bwd(const HpmnGruBwd *args, void *stream);
int hpmn_gru_scan_bwd_fuses_dx(int32_t H, int32_t B);
/* 1 where hpmn_gru_fused_fwd / hpmn_gru_pair_fwd honour HPMN_FWD_NO_CANDIDATE and hpmn_gru_scan_bwd / hpmn_gru_pair_bwd
 * honour HPMN_BWD_CANDIDATE_FROM_HS for a batch of B sequences (elsewhere the flags give HPMN_EUNSUPPORTED) */
int hpmn_gru_candidate_elision(int32_t H, int32_t B);
/* 1 where HpmnGruBwd.d_emb (the scatter fused into the launch) is supported */
int hpmn_gru_scan_bwd_fuses_scatter(int32_t H, int32_t B, int32_t D, int32_t F, int32_t E);

/* The reverse scans of TWO consecutive layers in ONE launch (H = 64), the mirror of hpmn_gru_pair_fwd: the lower layer's
 * scan runs while the upper layer's does.  A workgroup owns two sequences and both layers (eight waves); the upper
 * layer's input gradient -- the d_y of the lower layer's firing steps -- is formed on the matrix cores underneath the upper
 * layer's own iterations and handed over through an LDS ring; it is never written to memory.
 *   up : the upper layer as for hpmn_gru_scan_bwd (whole sequences: t_begin == t_end == 0); up.d_y from memory or NULL;
 *        up.d_x is IGNORED; up.D == H
 *   lo : the lower layer; lo.d_y is IGNORED (it comes from `up`); lo.T == up.T * lo.period; lo.d_x optional
 *        (D in {16, 32, 64}): the lower layer's input gradient, an epilogue of the launch shared by all four waves of the
 *        sequence
 * Results equal two hpmn_gru_scan_bwd calls (d_act bit-identical).  (B + 1) / 2 <= number of CUs, as for the forward.
 * HPMN_BWD_CANDIDATE_FROM_HS must be set on both layers or on neither (HPMN_EUNSUPPORTED otherwise: the switch is one
 * template argument of the launch). */
typedef struct HpmnGruPairBwd {
    HpmnGruBwd lo, up;
    int32_t flags, pad_;
} HpmnGruPairBwd;

int hpmn_gru_pair_bwd_supported(int32_t H, int32_t D_lo);
int hpmn_gru_pair_bwd(const HpmnGruPairBwd *args, void *stream);

/* ------------------------------------------------------------------------------------
 * One GRU layer, parameter and input gradients -- the time-parallel half of BPTT (TF
 * autodiff of the two _Linear matmuls, code/util.py:88-107, summed over all time steps):
 *   d_wg [D+H,2H] += [x | h_prev]^T d_act[:, 0:2H]     d_bg [2H] += sum d_act[:, 0:2H]
 *   d_wc [D+H, H] += [x | r*h_prev]^T d_act[:, 2H:]    d_bc [H]  += sum d_act[:, 2H:]
 *   d_x  [B,T,D]   = d_act [wg[0:D] | wc[0:D]]^T       (optional; overwritten, not added)
 * x [B,T,D] is the layer input, hs/gates as saved by hpmn_gru_scan_fwd, d_act from
 * hpmn_gru_scan_bwd.  The d_w / d_b outputs are ACCUMULATED (+=; they are views of the
 * optimiser's pre-zeroed flat gradient buffer in this repo) by a deterministic two-stage
 * reduction through `workspace` (>= hpmn_gru_param_grads_workspace_bytes(B,T,D,H) bytes).
 * ---------------------------------------------------------------------------------- */
typedef struct HpmnGruWgrad {
    int32_t B, T, D, H;
    const float *x, *hs, *gates, *d_act;
    const float *wg, *wc;
    float *d_wg, *d_bg, *d_wc, *d_bc;
    float *d_x;
    float *workspace;
    int32_t seq_per_wg;   /* set by the library */
    int32_t t_begin, t_len;  /* steps [t_begin, t_begin+t_len) of every sequence (the reduction of the parameter
                              * gradients and d_x alike); 0,0: all.  Lets the caller start the reduction of the
                              * late steps while the reverse scan is still working on the early ones. */
    int32_t whole_cu;        /* != 0: nothing latency-critical runs beside or behind this launch (the weight gradient of
                              * layer 0 at the end of BPTT): it may fill the CUs.  0: it shares the chip with a reverse
                              * scan and is capped at one workgroup per CU (DESIGN_HISTORY.md 3.9) */
} HpmnGruWgrad;

size_t hpmn_gru_param_grads_workspace_bytes(int32_t B, int32_t T, int32_t D, int32_t H);
int hpmn_gru_param_grads(const HpmnGruWgrad *args, void *stream);
/* Only the input gradient d_x [B,T,D] = d_act [B,T,3H] [wg[0:D] | wc[0:D]]^T (the part of the above that
 * is on BPTT's serial chain; the caller may run hpmn_gru_param_grads with d_x == NULL on another stream). */
int hpmn_gru_input_grad(const float *d_act, const float *wg, const float *wc, float *d_x,
                        int32_t B, int32_t T, int32_t D, int32_t H, int32_t t_begin, int32_t t_len,
                        void *stream);

/* ------------------------------------------------------------------------------------
 * Whole build_memory forward (code/hpmn.py:113-129 without the covariance loss): K
 * layers chained through a caller-provided workspace.  Inference form (no saved
 * states).   memory [B,K,H];  last [B, F*E] = uinp[:, last_index, :] (code/hpmn.py:439
 * last_index=-1, :292 last_index=-2), may be NULL.
 * ---------------------------------------------------------------------------------- */
typedef struct HpmnScanDesc {
    int32_t B, T, F, E, H, K;        /* T = user_maxlen as fed by the loader          */
    int32_t front_zero;              /* 23 for Hpmn_Industry, 0 for Hpmn              */
    int32_t mask_id0;                /* 1 for Hpmn, 0 for Hpmn_Industry               */
    int32_t last_index;              /* -1 or -2                                      */
    int64_t V;
    int32_t periods[HPMN_MAX_LAYERS];/* li_layer[i]                                   */
} HpmnScanDesc;

size_t hpmn_scan_workspace_bytes(const HpmnScanDesc *desc);
int hpmn_scan_fwd(const HpmnScanDesc *desc, const void *ids, const float *emb,
                  const float *const *wg, const float *const *bg,
                  const float *const *wc, const float *const *bc,
                  float *memory, float *last, void *workspace, void *stream);


/* ------------------------------------------------------------------------------------
 * Whole build_memory in TRAINING form and its BPTT, as two calls over one workspace (the
 * hpmn_scan_fwd(..., saved, ...) / hpmn_scan_bwd(...) pair).  Replaces, for the "User" branch, everything TF
 * executes for code/hpmn.py:113-129 in sess.run(train_step) (code/hpmn.py:482, :336) and TF's autodiff of it,
 * including the densified embedding gradient (code/hpmn.py:204-205).
 *
 *   hpmn_scan_fwd_train : ids [B,T,F], emb [V,E], K layers' (wg, bg, wc, bc)  ->  memory [B,K,H],
 *                         last [B,F*E] (= uinp[:, last_index, :], may be NULL); saved states stay in `workspace`
 *   hpmn_scan_bwd       : d_memory [B,K,H], d_last [B,F*E] (may be NULL) + the workspace of the forward call
 *                         ->  d_wg/d_bg/d_wc/d_bc[i] += (K pointers each, the optimiser's pre-zeroed buffers),
 *                             d_emb [V,E] += scatter of the input gradient (pre-zeroed by the caller)
 *
 * The serial chain (reverse scans, input gradients, scatter) runs on `stream`; the weight-gradient reductions run
 * on a helper stream owned by the context, forked and joined with events on `stream`.  With defer_join != 0
 * hpmn_scan_bwd returns with the helper stream still busy: d_emb is complete on `stream`, the d_w* buffers are not
 * until hpmn_train_join(ctx, stream) -- lets the caller update the (large) table underneath them.
 * A context serves one stream at a time; contexts are cheap (one stream, two events) and device-bound.
 * workspace: hpmn_scan_train_workspace_bytes(desc) bytes; hpmn_scan_train_layout gives the byte offsets (from
 * the 256-byte-aligned base) of the saved tensors for hosts that want to inspect them.
 * ---------------------------------------------------------------------------------- */
typedef struct HpmnTrainCtx HpmnTrainCtx;
typedef struct HpmnTrainLayout {
    int32_t K, pad;
    int32_t T[HPMN_MAX_LAYERS];
    uint64_t x0;                           /* [B, T[0], F*E] materialised layer-0 input                     */
    uint64_t xp[HPMN_MAX_LAYERS];          /* [B, T[i], 3H]  (only used by the two-kernel layers)          */
    uint64_t hs[HPMN_MAX_LAYERS];          /* [B, T[i]+1, H]                                               */
    uint64_t gates[HPMN_MAX_LAYERS];       /* [B, T[i], 3H]                                                */
    uint64_t y[HPMN_MAX_LAYERS];           /* [B, T[i]/period[i], H]  (i < K-1)                            */
    uint64_t d_act[HPMN_MAX_LAYERS];       /* [B, T[i], 3H]                                                */
    uint64_t d_x[HPMN_MAX_LAYERS];         /* [B, T[i], D_i]                                               */
    uint64_t wgrad_ws, total_bytes;
    uint64_t pair_ws;                      /* K operand images of the two-layer launches (hpmn_gru_proj_images) */
    uint64_t wgrad_ws_layer[HPMN_MAX_LAYERS]; /* [0]: the slabs of the all-layer weight-gradient launch (H = 32) */
} HpmnTrainLayout;

int hpmn_train_ctx_create(HpmnTrainCtx **ctx);
void hpmn_train_ctx_destroy(HpmnTrainCtx *ctx);
size_t hpmn_scan_train_workspace_bytes(const HpmnScanDesc *desc);
int hpmn_scan_train_layout(const HpmnScanDesc *desc, HpmnTrainLayout *out);
int hpmn_scan_fwd_train(HpmnTrainCtx *ctx, const HpmnScanDesc *desc, const void *ids, const float *emb,
                        const float *const *wg, const float *const *bg, const float *const *wc,
                        const float *const *bc, float *memory, float *last, void *workspace, void *stream);
int hpmn_scan_bwd(HpmnTrainCtx *ctx, const HpmnScanDesc *desc, const void *ids, const float *const *wg,
                  const float *const *wc, const float *d_memory, const float *d_last, float *const *d_wg,
                  float *const *d_bg, float *const *d_wc, float *const *d_bc, float *d_emb, void *workspace,
                  int32_t defer_join, void *stream);
int hpmn_train_join(HpmnTrainCtx *ctx, void *stream);
/* Measurement: with the probe enabled, hpmn_scan_bwd brackets the reverse-scan launch of LAYER 0 (the step's dominant
 * kernel) with timing events on `stream`; hpmn_train_probe_ms waits for the last bracket and returns its duration.  This
 * is the launch INSIDE a real step, weight-gradient kernels live beside it -- what bench.py's roofline entry quotes. */
int hpmn_train_probe(HpmnTrainCtx *ctx, int32_t enable);
int hpmn_train_probe_ms(HpmnTrainCtx *ctx, float *ms);
/* Scheduling hook: layer 0's reverse-scan launch is the longest of the step and, unlike the two-layer launches in front of
 * it, leaves room on its CUs -- work that only has to be done by the end of the step (the caller's early table-Adam pass)
 * belongs beside it.  With the mark enabled hpmn_scan_bwd records an event on `stream` in front of that launch;
 * hpmn_train_wait_layer0_reverse (after hpmn_scan_bwd returned) makes another stream wait for it.  No-op where the step has
 * no separate layer-0 launch (H = 32). */
int hpmn_train_mark_layer0_reverse(HpmnTrainCtx *ctx, int32_t enable);
int hpmn_train_wait_layer0_reverse(HpmnTrainCtx *ctx, void *stream);

/* ------------------------------------------------------------------------------------
 * DETERMINISTIC embedding-gradient scatter (r4): the gradient of Hpmn.embedding (code/hpmn.py:421-422; TF sums the
 * IndexedSlices of equal ids when it densifies them, :204-205) as a segmented reduction over the batch's lookups in ROW
 * ORDER.  Every table row's gradient rows are added in ONE fixed order (ascending lookup index b*T*F + t*F + f) by one
 * group of lanes and stored plainly -- no atomics: bit-reproducible run to run, identical on every data-parallel replica.
 *
 * The order depends on the ids only, so it is prepared off the serial chain:
 *   the caller sorts the n = B*T*F flattened ids STABLY: sorted_ids [n], perm [n] int32 (perm[j] = lookup index of the
 *   j-th entry in row order), and seg [n] int32 (seg[j] = number of distinct ids in front of entry j: an inclusive
 *   prefix count of "differs from its predecessor" minus 1) -- torch.sort / cumsum in this repo;
 *   hpmn_scatter_plan  ->  start [n+1] int32 (start[u] = first entry of row u; start[U] = n), rows [n] (ids' width:
 *                          rows[u] = table row of segment u, ascending), count [1] int32 = U  (only [0, U] / [0, U) written)
 *   hpmn_embed_grad_segsum : out_rows[u, :] = sum_j d_x(lookup perm[j]) over start[u] <= j < start[u+1]   (optional, [n, E])
 *                            d_emb[rows[u], :] += the same sum                                            (optional, [V, E])
 *     with d_last[b] (the read path's gradient wrt uinp[:, last_index, :], may be NULL) joined at step t_last, and id 0
 *     skipped (zero row in out_rows) under HPMN_ID_MASK0 -- exactly what hpmn_embed_grad_scatter adds atomically.
 *     partials: scratch of hpmn_embed_grad_segsum_partials_floats(n, E) floats.   E % 4 == 0, 256 % (E/4) == 0.
 * (rows, out_rows, count) are at the same time the touched-rows list a data-parallel rank exchanges and the compact
 * gradient hpmn_adam_step_rows consumes.
 * hpmn_train_set_scatter_plan(ctx, plan): the NEXT hpmn_scan_bwd on ctx scatters through the plan instead of the
 * atomic kernel (plan == NULL or plan->n == 0: atomic); the plan is copied, its arrays must stay alive until that call's
 * work has run.  One-shot: cleared by the hpmn_scan_bwd that used it.
 * ---------------------------------------------------------------------------------- */
typedef struct HpmnScatterPlan {
    int64_t n;               /* B*T*F lookups of the batch                                        */
    const int32_t *perm;     /* [n]                                                                */
    const int32_t *seg;      /* [n]                                                                */
    const int32_t *start;    /* [n+1]                                                              */
    const void *rows;        /* [n] int32 / int64 (HPMN_ID_I64 of the id flags)                    */
    const int32_t *count;    /* [1]                                                                */
    float *out_rows;         /* optional [n, E]                                                    */
    float *partials;         /* hpmn_embed_grad_segsum_partials_floats(n, E) floats                */
} HpmnScatterPlan;

int hpmn_scatter_plan(const void *sorted_ids, int32_t id_flags, int64_t n, const int32_t *seg, int32_t *start, void *rows,
                      int32_t *count, void *stream);
/* The whole plan from the UNSORTED ids in one call (ABI v12; csrc/plan_build.hip): a stable LSD radix sort of (id, lookup
 * index) pairs over the bits V needs (library call: rocPRIM), the segment scan, hpmn_scatter_plan, and -- counts != NULL --
 * counts[0] = U, counts[1 + c] = distinct rows in [row_bounds[c], row_bounds[c + 1]) for the nb <= HPMN_MAX_CHUNKS chunks of
 * the table's row range a data-parallel exchange sends them in (row_bounds: HOST array [nb + 1], NULL with nb = 1: [0, V)).
 * ids [n] int32 / int64 (HPMN_ID_I64), every id in [0, V); perm / seg [n], start [n + 1], rows [>= n], count [1], counts
 * [1 + nb]: device.  workspace: hpmn_scatter_plan_build_workspace_bytes(n, id_flags, V) bytes (0: the library refused). */
size_t hpmn_scatter_plan_build_workspace_bytes(int64_t n, int32_t id_flags, int64_t V);
int hpmn_scatter_plan_build(const void *ids, int32_t id_flags, int64_t n, int64_t V, void *workspace, size_t workspace_bytes,
                            int32_t *perm, int32_t *seg, int32_t *start, void *rows, int32_t *count,
                            const int64_t *row_bounds, int32_t nb, int32_t *counts, void *stream);
size_t hpmn_embed_grad_segsum_partials_floats(int64_t n, int32_t E);
/* entries per chunk of the reduction (its summation order: a row's entries inside one chunk left to right; a row that spans
 * chunks = its per-chunk sums cut into min(16, 256/E) consecutive blocks, each block left to right, blocks in order) */
int hpmn_embed_grad_segsum_chunk(void);
int hpmn_embed_grad_segsum(const HpmnScatterPlan *plan, const float *d_x, float *d_emb, int32_t B, int32_t T, int32_t F,
                           int32_t E, int32_t front_zero, int32_t id_flags, const float *d_last, int32_t t_last,
                           void *stream);
int hpmn_train_set_scatter_plan(HpmnTrainCtx *ctx, const HpmnScatterPlan *plan);

/* ------------------------------------------------------------------------------------
 * build_memory with ALL K layers in ONE launch (H = 64): forward hpmn_pipe_fwd, BPTT hpmn_pipe_bwd.
 * Replaces the whole loop of code/hpmn.py:116-128 (and TF's autodiff of it) at once: a workgroup owns a
 * tile of 16 sequences of ONE layer and runs that layer's recurrence on the matrix cores (the [3H x H] x
 * [H x 16] product of a step as 16x16x32 f16 MFMAs on operands split x = hi + lo, three products per tile,
 * fp32 accumulate: fp32-class results at 3/16 of the f32-MFMA time); the workgroups of layer i+1 consume
 * every period-th state of layer i while layer i is still running (hand-off through the y rows in memory +
 * per-wave progress words), so all K layers -- "fire every 2^k steps" -- advance concurrently.
 *
 *   x0      [B, T[0], D[0]]   layer-0 input rows (hpmn_embed_gather_seq: gather + zero prefix)
 *   y[i]    [B, T[i]/period[i], H]  every period-th output of layer i, i < K-1 (= the input rows of layer
 *                             i+1: T[i+1] == T[i]/period[i], D[i+1] == H); hpmn_pipe_fwd also accepts y[K-1] (ABI v10):
 *                             the call is then a GROUP of layers of a taller stack and y[K-1] the next group's x0
 *   hs[i]   [B, T[i]+1, H], gates[i] [B, T[i], 3H]   saved states as in hpmn_gru_scan_fwd (train != 0, and bwd)
 *   memory  [B, K, H] out (fwd);  d_memory [B, K, H] in (bwd): gradient wrt memory (row stride mem_stride)
 *   d_act[i] [B, T[i], 3H] out (bwd);  d_x[i] [B, T[i], H] out for i >= 1 (the gradient wrt y[i-1]; layer 0's
 *   input gradient is hpmn_gru_input_grad(d_act[0]) as before).  Weight gradients: hpmn_gru_param_grads.
 *   sync    scratch of hpmn_pipe_sync_bytes(K, B) bytes (progress words, re-zeroed on the stream by every call)
 * D[0] in {16, 32, 48, 64}; other shapes: HPMN_EUNSUPPORTED (use the per-layer entry points above).
 * ---------------------------------------------------------------------------------- */
typedef struct HpmnPipe {
    int32_t B, K, H, train;
    int32_t T[HPMN_MAX_LAYERS], D[HPMN_MAX_LAYERS], period[HPMN_MAX_LAYERS];
    const float *wg[HPMN_MAX_LAYERS], *bg[HPMN_MAX_LAYERS], *wc[HPMN_MAX_LAYERS], *bc[HPMN_MAX_LAYERS];
    const float *x0;
    float *y[HPMN_MAX_LAYERS], *hs[HPMN_MAX_LAYERS], *gates[HPMN_MAX_LAYERS];
    float *memory;
    const float *d_memory;
    float *d_act[HPMN_MAX_LAYERS], *d_x[HPMN_MAX_LAYERS];
    void *sync;
    int64_t mem_stride;   /* floats between consecutive sequences' rows of memory / d_memory; 0: K*H.  Lets the K
                           * layers of this call be layers j..j+K-1 of a taller stack (memory + j*H, stride K_all*H) */
} HpmnPipe;

int hpmn_pipe_supported(int32_t H, int32_t D0);
size_t hpmn_pipe_sync_bytes(int32_t K, int32_t B);
int hpmn_pipe_fwd(const HpmnPipe *args, void *stream);
int hpmn_pipe_bwd(const HpmnPipe *args, void *stream);
/* out[b, t, f*E:(f+1)*E] = t < front_zero ? 0 : emb[ids[b, t-front_zero, f]] * (mask_id0 ? id != 0 : 1):
 * Hpmn.embedding (code/hpmn.py:414-423 / :266-276) with the zero prefix of code/hpmn.py:288-289.
 *   ids [B, Tids, F] int32 / int64 (id flags in mask_id0), emb [V, E], out [B, front_zero+Tids, F*E] */
int hpmn_embed_gather_seq(const void *ids, const float *emb, float *out, int32_t B, int32_t Tids, int32_t F,
                          int32_t E, int32_t front_zero, int64_t V, int32_t mask_id0, void *stream);

/* The gather CONSUMED IN PLACE: out[b, f*E:(f+1)*E] += sum_t emb[ids[b,t,f]] * (mask_id0 ? id != 0 : 1) -- Hpmn.embedding
 * (code/hpmn.py:414-423) followed by a sum over time, without ever storing the gathered rows.  This is how the fused scan
 * kernels use the rows, and the roofline probe for north_star's gather target (4 B of id + 64 B of row per lookup is all the
 * traffic there is).  ids [B,T,F] (F <= 4), emb [V,E], out [B, F*E] pre-zeroed by the caller. */
int hpmn_embed_gather_sum(const void *ids, const float *emb, float *out, int32_t B, int32_t T, int32_t F, int32_t E,
                          int64_t V, int32_t mask_id0, void *stream);

/* ------------------------------------------------------------------------------------
 * Memory read path: covariance regulariser (code/hpmn.py:161-170), multi-hop attention over
 * the K memory slots (query_memory :172-182, attention :133-146), prediction head in
 * inference-mode batch-norm (build_fc_net :190-199) and the loss of :202-207.
 *
 * All read-path variables live in ONE contiguous fp32 range `params` (n_params floats; in this
 * repo a sub-range of the flat parameter buffer) addressed by the offsets below, TF layout
 * (dense kernels [in,out]):
 *   off_wq/off_bq  User/dense {kernel [D0,H], bias [H]}       off_map  User/map [H,H]
 *   off_att[h][0..5]  hop h: dense_{3h+1..3h+3} {kernel,bias}: [4H,80],[80],[80,40],[40],[40,1],[1]
 *   off_gamma/off_beta  output/bn1 [H+D0]                      off_fc[0..5]  fc1,fc2,fc3 {kernel,bias}
 *
 * hpmn_read_fwd      : memory [B,K,H], last [B,D0] -> pred [B], optional logit [B], optional
 *                      att_w0 [B,K] (first-hop weights, code/hpmn.py:182); *mem_loss += sum_b covreg_b.
 * hpmn_read_fwd_bwd  : training.  loss = inv_global_batch * sum_b logloss_b + memory_reg * sum_b covreg_b
 *                      (log-loss is a MEAN over the global batch, the regulariser a SUM).
 *                      mask1 [B,200] / mask2 [B,80]: dropout keep masks (0/1) or NULL; outputs scaled
 *                      by 1/keep_prob.  Writes pred [B], d_memory [B,K,H], d_last [B,D0];
 *                      loss_out[0] += sum_b logloss_b, loss_out[1] += sum_b covreg_b;
 *                      d_params[0..n_params) += gradients.  The training launch does not form them (BPTT
 *                      waits for d_memory / d_last only): it leaves the operand rows of every weight-gradient
 *                      product in `workspace` (>= hpmn_read_workspace_bytes[_n](); zero it ONCE after
 *                      allocation -- only parameter positions of its slab part are ever rewritten), and two
 *                      more launches form gW = X^T dY over the batch rows in 16 row chunks and add the chunks
 *                      in a fixed order (deterministic).
 *                      r5: for K <= 8 slots, H <= 128 the training launch runs its dense layers on the bf16 matrix
 *                      pipe with three-plane split operands (the six products of order <= 2: fp32-equivalent) out
 *                      of weight FRAGMENT images it builds in the same workspace with one small launch in front of
 *                      itself (the size functions include them); HPMN_READ_BF16=0 keeps the fp32 launch.
 * ---------------------------------------------------------------------------------- */
typedef struct HpmnReadDesc {
    int32_t B, K, H, D0, hop;
    int32_t off_wq, off_bq, off_map;
    int32_t off_att[4][6];
    int32_t off_gamma, off_beta;
    int32_t off_fc[6];
    int32_t n_params;
    /* Dropout masks generated IN the kernel (hpmn_read_fwd_bwd with mask1 == mask2 == NULL, keep_prob < 1 and
     * dropout_seed != 0): unit j of sample b of layer l is kept iff a splitmix64 hash of (seed, l, b, j) maps
     * below keep_prob.  Callers that need TF-reproducible or externally supplied masks pass mask1/mask2. */
    uint64_t dropout_seed;
} HpmnReadDesc;

size_t hpmn_read_workspace_bytes(const HpmnReadDesc *desc);
size_t hpmn_read_workspace_bytes_n(int32_t nb, const HpmnReadDesc *const *desc);      /* (two branches: more rows on the tape) */
int hpmn_read_fwd(const HpmnReadDesc *desc, const float *params, const float *memory, const float *last,
                  float *pred, float *logit, float *att_w0, float *mem_loss, void *stream);
int hpmn_read_fwd_bwd(const HpmnReadDesc *desc, const float *params, const float *memory, const float *last,
                      const int32_t *label, const float *mask1, const float *mask2, float keep_prob,
                      float inv_global_batch, float memory_reg, float *pred, float *loss_out,
                      float *d_memory, float *d_last, float *d_params, float *workspace, void *stream);
/* The same for graphs that execute the ITEM branch (code/hpmn.py:444-462, Industry :297-317): nb = 2 branches in the order
 * of the head's concat (user, item) -- repre = [query_u, last_u, query_i, last_i], memory_loss = umloss + imloss -- or nb = 1
 * (either branch alone; identical to the calls above).  desc[b] carries branch b's K, H, D0, hop and the offsets of ITS
 * dense / map / attention variables; the head's offsets (off_gamma, off_beta: sum_b (H_b + D0_b) wide, off_fc), n_params,
 * dropout_seed and B are read from desc[0].  memory / last / att_w0 / d_memory / d_last: nb pointers each.  One launch:
 * both attention stacks, the head, the loss and every gradient of them. */
int hpmn_read_fwd_n(int32_t nb, const HpmnReadDesc *const *desc, const float *params, const float *const *memory,
                    const float *const *last, float *pred, float *logit, float *const *att_w0, float *mem_loss,
                    void *stream);
int hpmn_read_fwd_bwd_n(int32_t nb, const HpmnReadDesc *const *desc, const float *params, const float *const *memory,
                        const float *const *last, const int32_t *label, const float *mask1, const float *mask2,
                        float keep_prob, float inv_global_batch, float memory_reg, float *pred, float *loss_out,
                        float *const *d_memory, float *const *d_last, float *d_params, float *workspace, void *stream);
/* BPTT only waits for d_memory / d_last.  With d_params == NULL hpmn_read_fwd_bwd[_n] stops after the training launch,
 * and this call (any stream ordered behind it, before `workspace` is used again) forms the weight gradients from the
 * rows it left in `workspace` and adds them to d_params -- off the serial chain. */
int hpmn_read_param_grads(const HpmnReadDesc *desc, float *d_params, float *workspace, void *stream);
int hpmn_read_param_grads_n(int32_t nb, const HpmnReadDesc *const *desc, float *d_params, float *workspace, void *stream);
/* The same, and the step's loss scalars with it (they need every workgroup of the training launch, like the gradients do):
 * loss3 = {loss_acc[0], loss_acc[1], inv_global_batch * loss_acc[0] + memory_reg * loss_acc[1]} -- the last is
 * cross_entropy of code/hpmn.py:202-207 without the l2 term -- and loss_acc[0..1] (the loss_out the training launch added
 * into) cleared for the next step.  Replaces four framework launches on a few bytes. */
int hpmn_read_param_grads_loss_n(int32_t nb, const HpmnReadDesc *const *desc, float *d_params, float *workspace,
                                 float *loss_acc, float inv_global_batch, float memory_reg, float *loss3, void *stream);

/* ------------------------------------------------------------------------------------
 * Embedding-gradient scatter-add: gradient of hpmn_embed_gather / the gather inside the
 * layer-0 scan (TF: IndexedSlices densified by the l2 term, code/hpmn.py:204-205).
 *   d_emb[ids[b,t,f]] += d_x[b, front_zero + t, f*E:(f+1)*E]   (skipping id 0 when mask_id0)
 * Runs of equal ids along t (the constant uid column, the id-0 padding) are pre-reduced
 * in registers before one atomic row add.  d_emb must be zeroed by the caller.
 *   ids [B,T,F], d_x [B, front_zero+T, F*E], d_emb [V,E]
 * ---------------------------------------------------------------------------------- */
int hpmn_embed_grad_scatter(const void *ids, const float *d_x, float *d_emb,
                            int32_t B, int32_t T, int32_t F, int32_t E, int32_t front_zero,
                            int64_t V, int32_t mask_id0, void *stream);

/* ------------------------------------------------------------------------------------
 * Optimiser step.  Replaces code/hpmn.py:209-214: per-element clip_by_value(g,-1,1) then
 * tf.train.AdamOptimizer in its TF form
 *   m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;  p -= lr_t m / (sqrt(v) + eps),
 *   lr_t = lr sqrt(1-b2^t)/(1-b1^t)  (computed by the caller).
 * One launch over a flat buffer holding every variable (dense over the embedding table,
 * as TF does).  grad_scale is applied before the clip (1/world for averaged gradients;
 * 1.0 normally).  n % 4 == 0 and 16-byte aligned pointers give the vectorised path.
 * ---------------------------------------------------------------------------------- */
int hpmn_adam_step(float *param, const float *grad, float *m, float *v, int64_t n,
                   float lr_t, float beta1, float beta2, float eps, float clip,
                   float grad_scale, void *stream);
/* (ABI v13) The same update, and the gradient it has consumed is written back as ZEROS: a caller that keeps one flat
 * gradient buffer (code/hpmn.py:204-214 densifies every gradient) then needs no clearing launch in front of its next
 * step -- at the Amazon shape that launch was 6 us of a 0.26 ms step, on the only stream. */
int hpmn_adam_step_clear(float *param, float *grad, float *m, float *v, int64_t n,
                         float lr_t, float beta1, float beta2, float eps, float clip,
                         float grad_scale, void *stream);
/* (ABI v14) ONE library call per training step -- the reference runs a step as ONE sess.run(train_step)
 * (code/hpmn.py:336, :482).  Replaces, for a graph that executes the "User" branch only, everything that call runs:
 *   hpmn_scan_fwd_train -> hpmn_read_fwd_bwd (weight gradients deferred) -> hpmn_scan_bwd (defer_join) ->
 *   hpmn_read_param_grads_loss_n -> hpmn_adam_step_clear over the table range [0, n_emb) -> hpmn_train_join ->
 *   hpmn_adam_step_clear over the dense variables [n_emb, n_total)
 * i.e. forward, loss, BPTT, the densified table gradient, per-element clip and the dense TF-form Adam of code/hpmn.py:202-214
 * in the plain single-process form (one sweep over the table; the two-pass / compact-row / data-parallel forms keep their
 * separate calls).  Every variable lives in ONE flat fp32 buffer (`param`; `grad`, `m`, `v` alike): the table first, then the
 * dense variables; off_gru[i] = element offsets of layer i's {gates kernel, gates bias, candidate kernel, candidate bias},
 * off_read = first read-path variable (HpmnReadDesc's offsets are relative to it).  `grad` must be all-zero on entry
 * (clear_grad_first != 0: the call zeroes it first) and is all-zero again when the call's work has run; `loss_acc` [2]
 * likewise.  loss3 [3] <- {log-loss sum, memory-loss sum, cross_entropy}.  mask1 / mask2: NULL (masks drawn in the kernel
 * from read.dropout_seed when keep_prob < 1) or caller-supplied dropout masks.  All pointers are device pointers owned by the
 * caller; asynchronous on `stream`; nothing is allocated.  What used to cost the Python harness ~200 us of enqueue work per
 * step (four calls + glue) against ~250 us of device time at the Amazon shape is one struct and one call. */
typedef struct HpmnTrainStep {
    HpmnScanDesc scan;
    HpmnReadDesc read;
    const void *ids;                       /* [B, T, F], width per scan.mask_id0's HPMN_ID_I64 bit */
    const int32_t *label;                  /* [B] */
    float *param, *grad, *m, *v;
    int64_t n_emb, n_total;
    int64_t off_gru[HPMN_MAX_LAYERS][4];
    int64_t off_read;
    float *memory, *last, *pred;           /* [B,K,H], [B,D0], [B] */
    float *d_memory, *d_last;              /* scratch of the same shapes */
    void *scan_workspace;                  /* hpmn_scan_train_workspace_bytes(&scan) */
    float *read_workspace;                 /* hpmn_read_workspace_bytes(&read), zero-initialised ONCE by the caller */
    float *loss_acc, *loss3;
    const float *mask1, *mask2;
    float keep_prob, inv_global_batch, memory_reg;
    float lr_t, beta1, beta2, eps, clip;
    int32_t clear_grad_first;
} HpmnTrainStep;
int hpmn_train_step(HpmnTrainCtx *ctx, const HpmnTrainStep *step, void *stream);
/* Row-wise ("lazy") form for embedding tables too large for the dense sweep (BASELINE configs[4]: a table
 * sized to HBM cannot also hold a dense gradient, and 28 B/element of dense Adam traffic over 10^9+ rows is the
 * whole step): the same update applied only to the n_rows table rows row_ids[u] (distinct), whose clipped
 * gradients are the rows of the COMPACT buffer grad_rows [n_rows, E].  A documented DEVIATION from
 * code/hpmn.py:209-214 (TF's dense Adam keeps moving a row whose moments are non-zero even when its gradient is
 * zero) -- the semantics of TF's LazyAdamOptimizer.  param / m / v: [V, E]; E % 4 == 0. */
int hpmn_adam_step_rows(float *param, const float *grad_rows, float *m, float *v, const int64_t *row_ids,
                        int64_t n_rows, int32_t E, float lr_t, float beta1, float beta2, float eps, float clip,
                        float grad_scale, void *stream);
/* The DENSE table update of hpmn_adam_step (same arithmetic, same result) in two passes, so that most of it leaves
 * the serial tail of the step: a row no id of the batch points at has an exactly-zero gradient, and its update
 * (m = b1 m, v = b2 v, p -= lr_t m / (sqrt(v) + eps)) depends on nothing the step computes.
 *   hpmn_table_mark_rows : flags[id] = 1 for the n_ids ids of the batch (flags: V bytes, all zero before)
 *   pass 0               : every row with flag == 0, gradient taken as zero (not read) -- any time after the marking,
 *                          on any stream: the step's gather and scatter only touch marked rows
 *   pass 1               : every row with flag != 0, behind the scatter; consumes AND CLEARS the row's gradient and
 *                          its flag, leaving grad [V,E] and flags all-zero for the next step (the caller must not
 *                          clear the table gradient densely any more, only make sure it starts all-zero)
 * param / grad / m / v: [V, E], 16-byte aligned; E/4 a power of two <= 64 (a row's lanes share one wave).  ids outside
 * [0, V) are ignored by the marking (padding entries of gathered id lists are -1).  id_flags: HPMN_ID_I64 for int64 ids. */
int hpmn_table_mark_rows(const void *ids, int64_t n_ids, uint8_t *flags, int64_t V, int32_t id_flags, void *stream);
int hpmn_adam_step_table(float *param, float *grad, float *m, float *v, uint8_t *flags, int64_t V, int32_t E,
                         int32_t pass, float lr_t, float beta1, float beta2, float eps, float clip, float grad_scale,
                         void *stream);

/* The update of the TOUCHED table rows from COMPACT gradient rows -- the single-GPU tail and the data-parallel exchange
 * in one launch, with no dense [V, E] gradient table anywhere (ABI v12; csrc/rows_adam.hip).  Reference semantics:
 * code/hpmn.py:204-214 (the IndexedSlices of :421-422 densified, clipped per element, dense TF Adam); SURVEY.md 8e (replicated
 * tables: every rank needs the SUM of all ranks' gradient rows, clipped after the sum).
 *   ids   [world, ids_stride]       rank r's distinct table rows, ASCENDING (hpmn_scatter_plan's `rows`), as the all-gather
 *                                   leaves them; the first len[r] entries are valid (counts != NULL: the DEVICE value
 *                                   counts[r * counts_stride] instead -- the single-GPU step never learns its count on the host)
 *   rows  [world, rows_stride, E]   gradient rows (hpmn_embed_grad_segsum's `out_rows`): rows[r][i] belongs to list entry
 *                                   first[r] + i.  The call consumes the WINDOW first[r] <= j < first[r] + n[r] of every
 *                                   rank's list; windows of different calls must cover disjoint TABLE-ROW RANGES that are
 *                                   the same on every rank (a chunk of the exchange = a range of table rows), so that a
 *                                   row's entries all lie in the windows of one call.
 *   flags [V] bytes                 bit r set <=> rank r's list holds the row (hpmn_table_mark_ranks; world == 1: any
 *                                   non-zero byte, e.g. hpmn_table_mark_rows').  The base must be 4-byte aligned and the
 *                                   allocation a multiple of 4 bytes.
 * Per distinct row of the union: the lowest rank holding it adds the ranks' rows in rank order 0..world-1, clips, applies
 * the TF-form Adam update of hpmn_adam_step to param / m / v [V, E] in place and clears the row's flag byte -- with
 * hpmn_adam_step_table(pass 0) over the unflagged rows this IS the dense update of hpmn_adam_step, bit for bit on the same
 * gradient rows.  E/4 a power of two <= 64; 1 <= world <= HPMN_MAX_RANKS. */
typedef struct HpmnRowsAdam {
    int32_t world, E;
    int32_t id_flags;                   /* HPMN_ID_I64: ids are int64                                       */
    int32_t counts_stride;              /* ints between two ranks' entries of `counts`                      */
    const void *ids;
    int64_t ids_stride;
    const int32_t *counts;              /* optional (device)                                                */
    int64_t len[HPMN_MAX_RANKS];        /* list lengths (used when counts == NULL)                          */
    int64_t first[HPMN_MAX_RANKS];
    int64_t n[HPMN_MAX_RANKS];
    const float *rows;
    int64_t rows_stride;                /* rows between two ranks' blocks of `rows`                         */
    uint8_t *flags;
    float *param, *m, *v;
    int64_t V;
    float lr_t, beta1, beta2, eps, clip, grad_scale;
    /* optional bucket index over the lists (hpmn_table_mark_ranks builds it): bucket_start[r * bucket_stride + b] = first entry
     * of list r whose id >> bucket_shift is >= b, b = 0 .. ((V - 1) >> bucket_shift) + 1.  NULL: every search covers a whole list */
    const int32_t *bucket_start;
    int64_t bucket_stride;
    int32_t bucket_shift, pad_;
} HpmnRowsAdam;
int hpmn_rows_sum_adam(const HpmnRowsAdam *args, void *stream);
/* flags[row] |= 1 << r for the first (counts ? counts[r * counts_stride] : cap) entries of ids[r, :], r < world.
 * Entries outside [0, V) are ignored.  Runs underneath the forward (atomic OR on the aligned 32-bit word).
 * bucket_start != NULL: the same pass writes the bucket index HpmnRowsAdam describes (bucket_stride >= ((V-1) >> bucket_shift) + 2
 * ints per rank; every slot is written). */
int hpmn_table_mark_ranks(const void *ids, int64_t ids_stride, int32_t world, const int32_t *counts, int32_t counts_stride,
                          int64_t cap, uint8_t *flags, int64_t V, int32_t id_flags, int32_t *bucket_start, int64_t bucket_stride,
                          int32_t bucket_shift, void *stream);

/* Evaluation on the matrix cores, one layer per launch (ABI v12; csrc/gru_tile64.hip, gru_tile128.hip): ONE layer of build_memory,
 * forward only, 16-sequence tiles, split-f16 operands (three products, fp32 accumulate) -- tf.nn.dynamic_rnn(GRUCell(H)) + the
 * every-period-th output of code/hpmn.py:118-128.  H = 64: x [B, T, D], D in {16, 32, 48, 64}.  H = 128: either x [B, T, D]
 * (D = 32: layer 0's gathered rows; D = 128: the outputs of the layer below; projected in the kernel) or xp [B, T, 384] (rows
 * hpmn_gru_input_proj produced) is given, the other is NULL.  y [B, T / period, H] may be NULL (top layer);
 * h_last[b * h_last_stride + 0..H-1] receives the final state (= memory[:, i, :]). */
typedef struct HpmnTileFwd {
    int32_t B, T, D, H;
    int32_t period, pad_;
    const float *x, *xp;
    const float *wg, *bg, *wc, *bc;
    float *y;
    float *h_last;
    int64_t h_last_stride;
} HpmnTileFwd;
int hpmn_tile_supported(int32_t H, int32_t D);
int hpmn_tile_fwd(const HpmnTileFwd *args, void *stream);

/* ------------------------------------------------------------------------------------
 * Online incremental memory update -- the serving-time form of build_memory: ONE new event per
 * user and call, against a persisted per-user state store.  Reference: the "incremental_update"
 * cascade of code/srnn.py:727-748 (layer i is updated iff index % 2^i == 0, its input being the
 * state layer i-1 has just produced) and its state store / scatter update (:790-796), generalised
 * to the periods of code/hpmn.py:113-129:
 *
 *   n = ++count[u]                                   events of user u so far (1-based)
 *   s_0 = GRU_0(x, s_0)                              every event
 *   for i = 1..K-1 while n % (p_0 p_1 .. p_{i-1}) == 0:   s_i = GRU_i(s_{i-1}, s_i)
 *
 * After T events the store holds exactly the `memory` hpmn_scan_fwd computes over those T steps
 * (every layer's last firing), so a user's lifelong sequence never has to be replayed.
 *   user  [B] int32   row of the store touched by event b (distinct within one call)
 *   x     [B, D]      the event's input row (embedding gather: hpmn_embed_gather)
 *   state [U, K, H]   in/out;   count [U] int32 in/out
 *   wg/bg/wc/bc       K pointers each, TF layout; layer 0 has D input rows, layers >= 1 have H
 * H in {32, 64, 128}; D <= 128.
 * ---------------------------------------------------------------------------------- */
typedef struct HpmnOnlineUpdate {
    int32_t B, D, H, K;
    int32_t periods[HPMN_MAX_LAYERS];
    const int32_t *user;
    const float *x;
    float *state;
    int32_t *count;
    const float *wg[HPMN_MAX_LAYERS], *bg[HPMN_MAX_LAYERS], *wc[HPMN_MAX_LAYERS], *bc[HPMN_MAX_LAYERS];
} HpmnOnlineUpdate;

int hpmn_memory_update(const HpmnOnlineUpdate *args, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* HPMN_HIP_H_ */
