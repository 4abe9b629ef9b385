// sweep_common.hpp -- what the two sweep kernels of the general FIR path share (sweep_ops.hip: horizontal pass first;
// sweep_vh_ops.hip: vertical pass first).
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>

namespace {

// tables the kernel only reads, at wave-uniform addresses: through the constant address space, so that hipcc may use scalar
// loads (a plain global pointer next to the kernel's own stores gets vector loads: nothing tells it the two never alias)
typedef const __attribute__((address_space(4))) uint32_t *konst;
__device__ __forceinline__ konst as_konst(const void *p) { return (konst)(uintptr_t)p; }

// x * w with "0 * anything = 0" (measured on gfx950, tools/legacy_mul_test.hip: bit-equal to v_mul_f32 whenever neither
// operand is zero, +0 when either is).  The accumulator slots that do not take a source row get weight 0 from the host:
// acc + 0 is acc whatever the row holds, Inf and NaN included, so the slots need no branch.  A sum never is -0 (it starts
// at +0 and x + (-x) = +0), so the +0 this gives where v_mul_f32 gives -0 adds up to the same bits.
__device__ __forceinline__ float mul_zero_wins(float x, float w) {
    float r;
    asm("v_mul_legacy_f32 %0, %1, %2" : "=v"(r) : "s"(w), "v"(x));
    return r;
}


}  // namespace
