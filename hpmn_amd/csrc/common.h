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

}  // namespace hpmn
