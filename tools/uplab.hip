// uplab.hip -- how fast can a 2x enlargement by direct gather run?  (tools/, not part of the library)
// 1920x1080 rgba_f16 -> 3840x2160 rgba_f16; a lane owns two adjacent target columns, a wave a strip of 128 columns and a
// segment of LINES target lines; per line: for each of the lane's (up to) 4 source columns a 2-tap vertical sum from a
// two-row register window (rows loaded as they come into reach, one row ahead), then a 2-tap horizontal sum per pixel,
// truncate, one 16-byte store.  No LDS.  Weights are made up (0.25 / 0.75): only the data movement and instruction mix matter.
// Result (MI355X): 0.0170 ms at 16 lines per wave against k_fir_vh's 0.0223 -- but with ONE source frame, which stays in L2.
// Built into the library with the real tables (round 3, not kept) the same loop took 0.0243 ms on sources that come from HBM
// (rocprofv3: 24.0 us per launch): every window move waits for a row requested one or two moves earlier, and hipcc's counted
// wait for it cannot skip the stores issued in between (gfx9 counts stores in vmcnt and their number per move varies with
// the table), so it waits for the NEWEST request as well.  k_fir_vh, three rows ahead through its written-out ring, stays.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/uplab tools/uplab.hip && tools/bin/uplab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

struct Px { f32x2 rg, ba; };
__device__ __forceinline__ Px widen(uint2 v) {
    Px p;
    p.rg = f32x2{ (float)__builtin_bit_cast(_Float16, (uint16_t)(v.x & 0xffff)), (float)__builtin_bit_cast(_Float16, (uint16_t)(v.x >> 16)) };
    p.ba = f32x2{ (float)__builtin_bit_cast(_Float16, (uint16_t)(v.y & 0xffff)), (float)__builtin_bit_cast(_Float16, (uint16_t)(v.y >> 16)) };
    return p;
}
__device__ __forceinline__ uint32_t pk(float a, float b) { return __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(a, b)); }

template <int LINES>
__global__ __launch_bounds__(64) void k_up(const uint2 *__restrict__ src, u32x4 *__restrict__ dst, int sw, int sh, int tw, int th) {
    const int lane = threadIdx.x;
    const int tcol = ((int)blockIdx.x * 64 + lane) * 2;                  // target columns tcol, tcol + 1
    const int t0 = (int)blockIdx.y * LINES;
    if (tcol >= tw) return;
    // source columns: target x reads source (x - 1) / 2 and that + 1 (clamped): for the pair (2k, 2k + 1): k - 1, k, k, k + 1 -> 3 distinct
    const int k = tcol / 2;
    const int c0 = max(k - 1, 0), c1 = k, c2 = min(k + 1, sw - 1);
    int srow = max((t0 - 1) / 2, 0);                                     // first source row of the window
    Px w0[3], w1[3];
    uint2 nx[3];
    auto ld = [&](int r, int c) { return src[(size_t)min(r, sh - 1) * sw + c]; };
    w0[0] = widen(ld(srow, c0)); w0[1] = widen(ld(srow, c1)); w0[2] = widen(ld(srow, c2));
    w1[0] = widen(ld(srow + 1, c0)); w1[1] = widen(ld(srow + 1, c1)); w1[2] = widen(ld(srow + 1, c2));
    nx[0] = ld(srow + 2, c0); nx[1] = ld(srow + 2, c1); nx[2] = ld(srow + 2, c2);
    for (int t = t0; t < min(t0 + LINES, th); t++) {
        const int need = max((t - 1) / 2, 0);
        if (need > srow) {                                               // uniform
#pragma unroll
            for (int c = 0; c < 3; c++) { w0[c] = w1[c]; w1[c] = widen(nx[c]); }
            srow++;
            nx[0] = ld(srow + 2, c0); nx[1] = ld(srow + 2, c1); nx[2] = ld(srow + 2, c2);
        }
        const float wa = (t & 1) ? 0.75f : 0.25f, wb = 1.0f - wa;
        Px m[3];
#pragma unroll
        for (int c = 0; c < 3; c++) { m[c].rg = w0[c].rg * wa + w1[c].rg * wb; m[c].ba = w0[c].ba * wa + w1[c].ba * wb; }
        const f32x2 org0 = m[0].rg * 0.25f + m[1].rg * 0.75f, oba0 = m[0].ba * 0.25f + m[1].ba * 0.75f;
        const f32x2 org1 = m[1].rg * 0.75f + m[2].rg * 0.25f, oba1 = m[1].ba * 0.75f + m[2].ba * 0.25f;
        dst[((size_t)t * tw + tcol) / 2] = u32x4{ pk(org0.x, org0.y), pk(oba0.x, oba0.y), pk(org1.x, org1.y), pk(oba1.x, oba1.y) };
    }
}

template <int LINES>
void run(const uint2 *src, std::vector<u32x4 *> &dsts, int sw, int sh, int tw, int th) {
    dim3 grid((tw / 2 + 63) / 64, (th + LINES - 1) / LINES);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 4; i++) hipLaunchKernelGGL(k_up<LINES>, grid, dim3(64), 0, 0, src, dsts[i % dsts.size()], sw, sh, tw, th);
    CK(hipDeviceSynchronize());
    float best = 1e9f;
    for (int rep = 0; rep < 5; rep++) {
        CK(hipEventRecord(e0));
        for (int i = 0; i < 16; i++) hipLaunchKernelGGL(k_up<LINES>, grid, dim3(64), 0, 0, src, dsts[i % dsts.size()], sw, sh, tw, th);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms / 16 < best) best = ms / 16;
    }
    const double bytes = (double)sw * sh * 8 + (double)tw * th * 8;
    printf("direct gather, %3d lines per wave (%d waves): %.4f ms  %.2f TB/s (%.3f of 8)\n", LINES, grid.x * grid.y, best, bytes / best / 1e9, bytes / best / 8e9);
}

int main() {
    const int sw = 1920, sh = 1080, tw = 3840, th = 2160;
    uint2 *src; CK(hipMalloc(&src, (size_t)sw * sh * 8));
    std::vector<uint2> h((size_t)sw * sh);
    for (size_t i = 0; i < h.size(); i++) h[i] = make_uint2(0x38003400u + (uint32_t)(i % 251), 0x3c003a00u);
    CK(hipMemcpy(src, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    std::vector<u32x4 *> dsts(8);
    for (auto &d : dsts) CK(hipMalloc(&d, (size_t)tw * th * 8));
    run<8>(src, dsts, sw, sh, tw, th);
    run<16>(src, dsts, sw, sh, tw, th);
    run<32>(src, dsts, sw, sh, tw, th);
    run<64>(src, dsts, sw, sh, tw, th);
    run<135>(src, dsts, sw, sh, tw, th);
    return 0;
}
