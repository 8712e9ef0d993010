// sp_common.h - shared helpers of libsimple_pose_hip (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>

#include "simple_pose_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define SP_WAVE 64

// ---- error plumbing (thread-local message, never throws across the C boundary) ----------------
void sp_set_error(const char* fmt, ...);
int sp_check_launch(const char* what);

// ---- kernel-name query (sp_conv2d_kernel_name): while a query is active on the calling thread, a launch function records the name
// of the kernel instantiation it WOULD launch (as rocprofv3 reports it, without the namespace / argument decoration) and returns
// SP_OK without launching - the name comes out of the dispatch code itself, so it cannot drift from it
bool sp_name_query_active();
void sp_name_query_begin();
const char* sp_name_query_end();
void sp_name_query_set(const char* fmt, ...);

#define SP_REQUIRE(cond, ...)        \
    do {                             \
        if (!(cond)) {               \
            sp_set_error(__VA_ARGS__); \
            return SP_EINVAL;        \
        }                            \
    } while (0)

static inline int sp_ceil_div(long long a, long long b) { return (int)((a + b - 1) / b); }

// ---- device helpers ------------------------------------------------------------------------------
// torch.max semantics: larger value wins, NaN beats everything, ties -> lower index.
__device__ __forceinline__ bool sp_better(float a, int ia, float b, int ib) {
    const bool an = (a != a), bn = (b != b);
    if (an || bn) return an && (!bn || ia < ib);
    return (a > b) || (a == b && ia < ib);
}

__device__ __forceinline__ void sp_wave_argmax(float& v, int& i) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_xor(v, off, SP_WAVE);
        const int oi = __shfl_xor(i, off, SP_WAVE);
        if (sp_better(ov, oi, v, i)) { v = ov; i = oi; }
    }
}
