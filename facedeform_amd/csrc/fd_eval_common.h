// fd_eval_common.h -- device helpers the evaluation translation units share (fd_eval.hip, fd_eval_shared.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "fd_internal.h"

namespace fd {
namespace {

constexpr unsigned kMaxCUs = 1024;     // bound of the diagnostics' per-CU tables

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

// ---- fp32 epilogue, same operation order as the reference -------------------
__device__ __forceinline__ void normalize3(float &x, float &y, float &z)
{
    const float l2 = x * x + y * y + z * z;
    if (l2 > 0.f) {
        const float inv = 1.f / sqrtf(l2);
        x *= inv; y *= inv; z *= inv;
    }
}

// d2 * log2|d2| with the DX9 multiply (0 * anything = 0): rounding can leave a vertex that sits
// on a centre at exactly zero or at a tiny negative d2; the first gives 0 * -inf = 0 here and the
// second an error of order 1e-7 * 23, the size of the rounding of d2 itself.  No clamp needed.
extern "C" __device__ float fd_fmul_legacy(float, float) __asm("llvm.amdgcn.fmul.legacy");
__device__ __forceinline__ float d2_log_d2(float d)
{
    return fd_fmul_legacy(d, __builtin_amdgcn_logf(__builtin_fabsf(d)));
}

}  // namespace
}  // namespace fd
