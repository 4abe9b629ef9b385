// lut_common.hpp -- pieces shared by the LDS-table kernels (color_ops.hip, chain_ops.hip).
#pragma once
#include "kernels.h"
#include "pixel_math.hpp"
#include <string.h>

namespace cvs {

constexpr int kWG = 1024;                  // 16 waves: the LDS table allows one workgroup per CU
constexpr int kLutHalfs = 65536;

// plain: no colour stage at all (layers go straight to the stack); cross: the two layers are crossfaded with the
// weights wa, wb (video_mix.c:193-205) instead of stacked
struct Mat { float m[9]; int plain; int cross; float wa, wb; };

// The same parameters as the kernels use them: named scalars, built at the top of a kernel straight from the
// kernel-argument struct.  (Handing the argument struct -- or its array -- down by reference made the compiler keep a
// private copy of it in scratch memory, which turns scratch on for the whole kernel.)
struct MatR { float m0, m1, m2, m3, m4, m5, m6, m7, m8; int plain, cross; float wa, wb; };
#define CVS_MAT_REGS(k) { (k).m[0], (k).m[1], (k).m[2], (k).m[3], (k).m[4], (k).m[5], (k).m[6], (k).m[7], (k).m[8], (k).plain, (k).cross, (k).wa, (k).wb }

// Pointers that arrive inside a job record are generic to the compiler, which then emits flat_load
// (counts on BOTH vmcnt and lgkmcnt and so serialises against the LDS gathers).  Tell it they are global.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef const u32x4 __attribute__((address_space(1))) *g_cu4;
typedef u32x4 __attribute__((address_space(1))) *g_u4;
typedef const u32x2 __attribute__((address_space(1))) *g_cu2;
typedef u32x2 __attribute__((address_space(1))) *g_u2;
__device__ __forceinline__ uint4 ld4(const void *p, size_t i) { u32x4 v = ((g_cu4)p)[i]; return make_uint4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ void st4(void *p, size_t i, uint4 v) { u32x4 t = { v.x, v.y, v.z, v.w }; ((g_u4)p)[i] = t; }
__device__ __forceinline__ uint2 ld2(const void *p, size_t i) { u32x2 v = ((g_cu2)p)[i]; return make_uint2(v.x, v.y); }
__device__ __forceinline__ void st2(void *p, size_t i, uint2 v) { u32x2 t = { v.x, v.y }; ((g_u2)p)[i] = t; }

// 128 KiB table -> LDS: 8 sweeps of 1024 lanes x 16 B
__device__ __forceinline__ void stage_lut(uint16_t *lds, const uint16_t *__restrict__ table) {
    const uint4 *src = reinterpret_cast<const uint4 *>(table);
    uint4 *dst = reinterpret_cast<uint4 *>(lds);
#pragma unroll
    for (int i = 0; i < kLutHalfs * 2 / 16 / kWG; i++) dst[i * kWG + threadIdx.x] = src[i * kWG + threadIdx.x];
    __syncthreads();
}

template <bool IN_LDS>
__device__ __forceinline__ uint32_t gather2(const uint16_t *lut, uint32_t pair) {
    // two halfs packed in one dword -> two gathers -> repack
    uint32_t lo = lut[pair & 0xFFFFu], hi = lut[pair >> 16];
    return lo | (hi << 16);
}


inline unsigned persistent_grid(int cus, size_t work_items) {
    size_t need = (work_items + kWG - 1) / kWG;
    size_t g = cus > 0 ? (size_t)cus : 256;
    if (need < g) g = need ? need : 1;
    return (unsigned)g;
}

inline Mat make_mat(const float *m) {
    Mat r;
    r.plain = m ? 0 : 1;
    r.cross = 0; r.wa = r.wb = 0.0f;
    for (int i = 0; i < 9; i++) r.m[i] = m ? m[i] : (i % 4 == 0 ? 1.0f : 0.0f);
    return r;
}

}  // namespace cvs
