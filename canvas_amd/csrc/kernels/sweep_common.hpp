// sweep_common.hpp -- what the two per-target-line sweeps of the per-line-table FIR path share (sweep_hv_ops.hip: horizontal
// pass first; sweep_vh_ops.hip: vertical pass first).
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>

namespace {

// tables the kernel only reads, at wave-uniform addresses: through the constant address space, so that hipcc may use scalar
// loads (a plain global pointer next to the kernel's own stores gets vector loads: nothing tells it the two never alias)
typedef const __attribute__((address_space(4))) uint32_t *konst;
__device__ __forceinline__ konst as_konst(const void *p) { return (konst)(uintptr_t)p; }

}  // namespace
