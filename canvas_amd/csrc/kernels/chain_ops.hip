// chain_ops.hip -- the fused colour-matrix + alpha-over chain (BASELINE config 2, the north star).
//
// What the reference does per layer (src/cprocess/color.c:104-165) and per stack
// (src/cprocess/workspace.c:530-544, src/cprocess/main.c:43-71,115-139):
//   [LUT over all 4 halfs] -> widen -> 3x3 in f32 -> truncate to half -> [LUT] -> widen ->
//   over(acc, layer, 1.0) for every layer above the first -> truncate to half,
// each arrow a pass over memory with f32 intermediates.  Here a pixel goes through the whole
// chain in registers: 8 B read per layer pixel + 8 B written per output pixel.
//
// Shape of the kernel (MI355X):
//   * persistent grid, ONE workgroup per CU: the 128 KiB transfer table lives in LDS (160 KiB per CU),
//     staged once per launch; a launch carries up to 32 frames so the staging (~3 us) is amortised;
//   * 512 lanes per workgroup (2 waves per SIMD): with one 16-byte load per layer per lane in flight
//     that is 16 KiB outstanding per CU, the amount at which this chip's HBM streams fastest for a
//     2-read + 1-write pattern (tools/membench.hip: 4 MiB in flight chip-wide -> 6.2 TB/s; 16 MiB -> 5.3);
//   * a lane owns pixel PAIRS: one global_load_dwordx4 per layer, one global_store_dwordx4, all
//     non-temporal (nothing is re-read); the pair's arithmetic runs as packed f32;
//   * hand-counted software pipeline (below): the loads of trip t+1 are in flight during the
//     arithmetic of trip t;
//   * job records arrive in the kernel-argument segment (scalar loads, nothing to recycle).
// Bound: HBM.  Algorithmic bytes per output pixel: 8 * (nlayers + 1)  (config 2: 24).
#include "lut_common.hpp"
#include "grade.hpp"
#include "chain_math.hpp"
#include <stdlib.h>

using namespace cvs;

namespace {

// Job records travel as kernel arguments (<= 4 KiB per launch).  The production kernel takes up to four layers, so
// its records are compact (48 B) and a launch carries 64 frames: a launch costs ~24 us beyond its per-frame time
// (table staging, ramp, drain), which is worth amortising.  The first version keeps the full 8-layer record.
constexpr int kJobsPerLaunchV0 = 32, kJobsPerLaunch = 64, kFusedLayers = 4;
struct BatchV0 { cvk_chain_job jobs[kJobsPerLaunchV0]; };
struct JobC { void *out; const void *layer[kFusedLayers]; uint64_t npixels; };
struct Batch { JobC jobs[kJobsPerLaunch]; };
static_assert(sizeof(Batch) + 128 <= 4096, "kernel arguments must fit the 4 KiB segment");

// ---------------------------------------------------------------- v0: first correct version (kept for A/B)

template <int MAXL, bool PRE, bool POST>
__device__ __forceinline__ uint2 chain_pixel(const uint2 (&px)[MAXL], int nl, const MatR &mat, const uint16_t *lut, const uint16_t *post) {
    if (MAXL >= 2 && mat.cross) return narrow(blend_cross(widen(px[0]), widen(px[MAXL >= 2 ? 1 : 0]), mat.wa, mat.wb));
    px32 acc = mat.plain ? widen(px[0]) : grade<PRE, POST>(px[0], mat, lut, post);
#pragma unroll
    for (int k = 1; k < MAXL; k++)      // static indices only: a runtime-indexed array would live in scratch
        if (k < nl) acc = blend_over(acc, mat.plain ? widen(px[k]) : grade<PRE, POST>(px[k], mat, lut, post), 1.0f);
    return narrow(acc);
}

template <int NL, bool PRE, bool POST>
__global__ __launch_bounds__(kWG) void k_chain_v0(BatchV0 batch, int njobs, Mat kmat,
                                                  const uint16_t *__restrict__ pre, const uint16_t *__restrict__ post) {
    const MatR mat = CVS_MAT_REGS(kmat);
    __shared__ uint16_t lut[(PRE || POST) ? kLutHalfs : 1];
    if (PRE) stage_lut(lut, pre);
    else if (POST) stage_lut(lut, post);
    constexpr int MAXL = NL > 0 ? NL : CVK_CHAIN_MAX_LAYERS;
    const size_t stride = (size_t)gridDim.x * kWG;
    for (int j = 0; j < njobs; j++) {
        const cvk_chain_job &job = batch.jobs[j];
        const int nl = NL > 0 ? NL : job.nlayers;
        const size_t npairs = job.npixels / 2;
        for (size_t i = (size_t)blockIdx.x * kWG + threadIdx.x; i < npairs; i += stride) {
            uint4 a[MAXL];
#pragma unroll
            for (int k = 0; k < MAXL; k++) if (k < nl) a[k] = ld4(job.layer[k], i);
            uint2 p0[MAXL], p1[MAXL];
#pragma unroll
            for (int k = 0; k < MAXL; k++) { p0[k] = make_uint2(a[k].x, a[k].y); p1[k] = make_uint2(a[k].z, a[k].w); }
            uint2 r0 = chain_pixel<MAXL, PRE, POST>(p0, nl, mat, lut, post), r1 = chain_pixel<MAXL, PRE, POST>(p1, nl, mat, lut, post);
            st4(job.out, i, make_uint4(r0.x, r0.y, r1.x, r1.y));
        }
        if ((job.npixels & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
            uint2 p[MAXL];
#pragma unroll
            for (int k = 0; k < MAXL; k++) if (k < nl) p[k] = ld2(job.layer[k], job.npixels - 1);
            st2(job.out, job.npixels - 1, chain_pixel<MAXL, PRE, POST>(p, nl, mat, lut, post));
        }
    }
}

// ---------------------------------------------------------------- production kernel
//
// hipcc sinks a prefetch load into the iteration that consumes it (its value is only used after the
// back-edge) and answers loop-carried loads with vmcnt(0), so in compiler-visible form the loads of
// trip t+1 never overlap the arithmetic of trip t.  The pixel loads are therefore issued from inline
// asm (invisible to both passes) and waited for by hand (cdna_hip_programming.md 5.7, form (ii)).
//
// One trip of one lane = one pixel pair:
//     asm loads   nxt <- next trip            NL x global_load_dwordx4 ... nt   (indices clamped, never predicated)
//     arithmetic  cur -> res                  chain_math.hpp
//     s_waitcnt vmcnt(0) naming nxt           BEFORE this trip's store is issued
//     store       res                         global_store_dwordx4 ... nt
// Why the wait sits before the store: measured on gfx950, vmcnt(N) with N younger STORES outstanding
// does not guarantee that older LOADS have landed (stores retire early; outputs were wrong until this
// was changed).  At this point only the next trip's loads (needed now anyway) and the previous trip's
// store (a whole trip old) are outstanding; the new store then drains under the next trip's arithmetic.
//
// The trip loop is unrolled by two with the register sets swapping roles, so no cur = nxt copies.

template <bool NT>
__device__ __forceinline__ void asm_ld4(u32x4 &dst, const void *base, size_t idx) {
    const u32x4 *p = reinterpret_cast<const u32x4 *>(base) + idx;
    if (NT) asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(dst) : "v"(p));
    else    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(p));
}

template <int N, int P, int L>
__device__ __forceinline__ void wait_vm(u32x4 (&r)[P][L]) {
    static_assert(P * L <= 8 && P <= 2, "operand list below covers 2 x 4");
    if constexpr (P * L == 1) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(r[0][0]) : "i"(N) : "memory");
    else if constexpr (P == 1 && L == 2) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(r[0][0]), "+v"(r[0][1]) : "i"(N) : "memory");
    else if constexpr (P == 1 && L == 3) asm volatile("s_waitcnt vmcnt(%3)" : "+v"(r[0][0]), "+v"(r[0][1]), "+v"(r[0][2]) : "i"(N) : "memory");
    else if constexpr (P == 1 && L == 4) asm volatile("s_waitcnt vmcnt(%4)" : "+v"(r[0][0]), "+v"(r[0][1]), "+v"(r[0][2]), "+v"(r[0][3]) : "i"(N) : "memory");
    else if constexpr (P == 2 && L == 1) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(r[0][0]), "+v"(r[1][0]) : "i"(N) : "memory");
    else if constexpr (P == 2 && L == 2) asm volatile("s_waitcnt vmcnt(%4)" : "+v"(r[0][0]), "+v"(r[0][1]), "+v"(r[1][0]), "+v"(r[1][1]) : "i"(N) : "memory");
    else if constexpr (P == 2 && L == 3) asm volatile("s_waitcnt vmcnt(%6)" : "+v"(r[0][0]), "+v"(r[0][1]), "+v"(r[0][2]), "+v"(r[1][0]), "+v"(r[1][1]), "+v"(r[1][2]) : "i"(N) : "memory");
    else asm volatile("s_waitcnt vmcnt(%8)" : "+v"(r[0][0]), "+v"(r[0][1]), "+v"(r[0][2]), "+v"(r[0][3]), "+v"(r[1][0]), "+v"(r[1][1]), "+v"(r[1][2]), "+v"(r[1][3]) : "i"(N) : "memory");
}

__device__ __forceinline__ void stage_lut_any(uint16_t *lds, const uint16_t *__restrict__ table) {
    const uint4 *src = reinterpret_cast<const uint4 *>(table);
    uint4 *dst = reinterpret_cast<uint4 *>(lds);
    constexpr int kWords = kLutHalfs * 2 / 16;           // 8192 16-byte words
    if (kWords % blockDim.x == 0) {
        // every workgroup reads the same 128 KiB at the same moment: start each one at a different slice so that
        // the requests of the 32 workgroups behind one L2 spread over its channels instead of queueing on a few
        const int slices = kWords / (int)blockDim.x, rot = (int)(blockIdx.x >> 3);
        for (int it = 0; it < slices; it++) {
            const int i = ((it + rot) % slices) * (int)blockDim.x + (int)threadIdx.x;
            dst[i] = src[i];
        }
    } else {
        for (int i = threadIdx.x; i < kWords; i += blockDim.x) dst[i] = src[i];
    }
    __syncthreads();
}



enum { DIAG_NONE = 0, DIAG_MEMORY_ONLY = 1, DIAG_COMPUTE_ONLY = 2 };

// NL in 1..4, the same for every job of the batch; every job has npixels >= 2
template <int NL, bool PRE, bool POST, int DIAG>
__global__ __launch_bounds__(kWG) void k_chain(Batch batch, int njobs, Mat kmat,
                                               const uint16_t *__restrict__ pre, const uint16_t *__restrict__ post) {
    const MatR mat = CVS_MAT_REGS(kmat);
    __shared__ uint16_t lut[(PRE || POST) ? kLutHalfs : 1];
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t lane = (size_t)blockIdx.x * blockDim.x + threadIdx.x;

    u32x4 A[1][NL], B[1][NL];
    u32x4 diag_acc = { 0, 0, 0, 0 };
    {
        // the first trip of the first frame goes out before the table is staged
        const JobC &job = batch.jobs[0];
        const size_t last = job.npixels / 2 - 1;
        const size_t idx = lane < last ? lane : last;
#pragma unroll
        for (int k = 0; k < NL; k++) asm_ld4<true>(A[0][k], job.layer[k], idx);
    }
    if (PRE) stage_lut_any(lut, pre);
    else if (POST) stage_lut_any(lut, post);
    wait_vm<0>(A);

    for (int j = 0; j < njobs; j++) {
        const JobC &job = batch.jobs[j];
        const size_t npairs = job.npixels / 2;
        const bool more_jobs = j + 1 < njobs;
        const JobC &njob = batch.jobs[more_jobs ? j + 1 : j];
        const size_t nnpairs = njob.npixels / 2;
        // frame base pointers as wave-uniform values (scalar loads): the per-lane choice in the prefetch
        // is then a register select, not a vector load of the job record
        const void *lp[NL], *nlp[NL];
#pragma unroll
        for (int k = 0; k < NL; k++) { lp[k] = job.layer[k]; nlp[k] = njob.layer[k]; }
        void *const outp = job.out;

        auto trip = [&](u32x4 (&cur)[1][NL], u32x4 (&nxt)[1][NL], size_t base) {
            const size_t nbase = base + stride;
            const bool same = nbase < npairs;          // next trip in this frame, else first trip of the next one
            size_t idx = same ? nbase : lane;
            const size_t last = (same ? npairs : nnpairs) - 1;
            idx = idx < last ? idx : last;
#pragma unroll
            for (int k = 0; k < NL; k++) {
                if (DIAG == DIAG_COMPUTE_ONLY) nxt[0][k] = cur[0][k];
                else asm_ld4<true>(nxt[0][k], same ? lp[k] : nlp[k], idx);
            }
            __builtin_amdgcn_sched_barrier(0);
            u32x4 res;
            if (DIAG == DIAG_MEMORY_ONLY) {
                res = cur[0][0];
#pragma unroll
                for (int k = 1; k < NL; k++) res ^= cur[0][k];
            } else {
                res = chain_pair_lean<NL, PRE, POST>(cur[0], mat, lut, post);
            }
            __builtin_amdgcn_sched_barrier(0);
            wait_vm<0>(nxt);
            __builtin_amdgcn_sched_barrier(0);
            if (DIAG == DIAG_COMPUTE_ONLY) diag_acc ^= res;
            else __builtin_nontemporal_store(res, (g_u4)outp + base);
        };

        size_t base = lane;
        bool in_b = false;                              // which register set holds this lane's next input
        while (base < npairs) {
            trip(A, B, base);
            base += stride;
            if (!(base < npairs)) { in_b = true; break; }
            trip(B, A, base);
            base += stride;
        }
        if (in_b) {
#pragma unroll
            for (int k = 0; k < NL; k++) A[0][k] = B[0][k];
        }
        // a lane that had no trip in this frame starts the next frame cold
        if (more_jobs && !(lane < npairs)) {
            const size_t last = nnpairs - 1;
            const size_t idx = lane < last ? lane : last;
#pragma unroll
            for (int k = 0; k < NL; k++) asm_ld4<true>(A[0][k], nlp[k], idx);
            wait_vm<0>(A);
        }
        if (DIAG == DIAG_COMPUTE_ONLY && lane < npairs) ((g_u4)outp)[lane] = diag_acc;
        // odd pixel count: the last pixel on its own (scalar form of the same arithmetic)
        if ((job.npixels & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
            uint2 px[NL];
#pragma unroll
            for (int k = 0; k < NL; k++) px[k] = ld2(job.layer[k], job.npixels - 1);
            st2(job.out, job.npixels - 1, chain_pixel<NL, PRE, POST>(px, NL, mat, lut, post));
        }
    }
}

template <int NL, int DIAG>
int launch(const Batch &jobs, int njobs, const Mat &mat, const uint16_t *pre, const uint16_t *post, unsigned grid, unsigned block, hipStream_t s) {
    if (pre && post)  hipLaunchKernelGGL((k_chain<NL, true, true, DIAG>), dim3(grid), dim3(block), 0, s, jobs, njobs, mat, pre, post);
    else if (pre)     hipLaunchKernelGGL((k_chain<NL, true, false, DIAG>), dim3(grid), dim3(block), 0, s, jobs, njobs, mat, pre, post);
    else if (post)    hipLaunchKernelGGL((k_chain<NL, false, true, DIAG>), dim3(grid), dim3(block), 0, s, jobs, njobs, mat, pre, post);
    else              hipLaunchKernelGGL((k_chain<NL, false, false, DIAG>), dim3(grid), dim3(block), 0, s, jobs, njobs, mat, pre, post);
    return (int)hipGetLastError();
}

template <int NL>
int launch_v0(const BatchV0 &jobs, int njobs, const Mat &mat, const uint16_t *pre, const uint16_t *post, unsigned grid, hipStream_t s) {
    if (pre && post)  hipLaunchKernelGGL((k_chain_v0<NL, true, true>), dim3(grid), dim3(kWG), 0, s, jobs, njobs, mat, pre, post);
    else if (pre)     hipLaunchKernelGGL((k_chain_v0<NL, true, false>), dim3(grid), dim3(kWG), 0, s, jobs, njobs, mat, pre, post);
    else if (post)    hipLaunchKernelGGL((k_chain_v0<NL, false, true>), dim3(grid), dim3(kWG), 0, s, jobs, njobs, mat, pre, post);
    else              hipLaunchKernelGGL((k_chain_v0<NL, false, false>), dim3(grid), dim3(kWG), 0, s, jobs, njobs, mat, pre, post);
    return (int)hipGetLastError();
}

}  // namespace

// Development knobs, read per call (none of them changes results):
//   CVS_CHAIN_VARIANT  unset/1 = production kernel; 0 = the simple first version (A/B reference);
//                      10 = memory traffic only, 12 = arithmetic only (diagnostic builds, wrong output)
//   CVS_CHAIN_BLOCK    lanes per workgroup (default 512 = 2 waves per SIMD; one workgroup per CU)
static int chain_dispatch(const cvk_chain_job *jobs, int njobs, int uniform_layers, const Mat &mat,
                          const uint16_t *pre, const uint16_t *post, int cus, void *stream);

extern "C" int cvk_chain_color_over(const cvk_chain_job *jobs, int njobs, int uniform_layers, const float *m,
                                    const uint16_t *pre, const uint16_t *post, int cus, void *stream) {
    return chain_dispatch(jobs, njobs, uniform_layers, make_mat(m), pre, post, cus, stream);
}

// two-layer jobs, crossfaded: out = cross(layer[0], layer[1]) with weights wa, wb
extern "C" int cvk_chain_cross(const cvk_chain_job *jobs, int njobs, float wa, float wb, int cus, void *stream) {
    Mat mat = make_mat(NULL);
    mat.cross = 1; mat.wa = wa; mat.wb = wb;
    return chain_dispatch(jobs, njobs, 2, mat, NULL, NULL, cus, stream);
}

static int chain_dispatch(const cvk_chain_job *jobs, int njobs, int uniform_layers, const Mat &mat,
                          const uint16_t *pre, const uint16_t *post, int cus, void *stream) {
    const unsigned grid = (unsigned)(cus > 0 ? cus : 256);
    hipStream_t s = (hipStream_t)stream;
    const char *env = getenv("CVS_CHAIN_VARIANT");
    const int variant = env ? atoi(env) : 1;
    const char *bl = getenv("CVS_CHAIN_BLOCK");
    unsigned block = bl ? (unsigned)atoi(bl) : 512u;
    if (block < 64 || block > (unsigned)kWG || (block & 63u)) block = 512u;
    const bool fused_kernel = variant != 0 && uniform_layers >= 1 && uniform_layers <= kFusedLayers;
    int per_launch = fused_kernel ? kJobsPerLaunch : kJobsPerLaunchV0;
    if (const char *pl = getenv("CVS_CHAIN_JOBS_PER_LAUNCH")) { const int v = atoi(pl); if (v >= 1 && v < per_launch) per_launch = v; }
    for (int first = 0; first < njobs; first += per_launch) {
        const int n = njobs - first < per_launch ? njobs - first : per_launch;
        int rc;
        if (fused_kernel) {
            Batch b;
            memset(&b, 0, sizeof b);
            for (int i = 0; i < n; i++) {
                const cvk_chain_job &src = jobs[first + i];
                b.jobs[i].out = src.out;
                for (int k = 0; k < uniform_layers; k++) b.jobs[i].layer[k] = src.layer[k];
                b.jobs[i].npixels = src.npixels;
            }
#define CVK_DISPATCH(NLV)                                                                              \
            (variant == 10 ? launch<NLV, DIAG_MEMORY_ONLY>(b, n, mat, pre, post, grid, block, s)          \
             : variant == 12 ? launch<NLV, DIAG_COMPUTE_ONLY>(b, n, mat, pre, post, grid, block, s)       \
                             : launch<NLV, DIAG_NONE>(b, n, mat, pre, post, grid, block, s))
            switch (uniform_layers) {
            case 1: rc = CVK_DISPATCH(1); break;
            case 2: rc = CVK_DISPATCH(2); break;
            case 3: rc = CVK_DISPATCH(3); break;
            default: rc = CVK_DISPATCH(4); break;
            }
#undef CVK_DISPATCH
        } else {
            BatchV0 b;
            memset(&b, 0, sizeof b);
            memcpy(b.jobs, jobs + first, sizeof(cvk_chain_job) * (size_t)n);
            switch (uniform_layers) {
            case 1: rc = launch_v0<1>(b, n, mat, pre, post, grid, s); break;
            case 2: rc = launch_v0<2>(b, n, mat, pre, post, grid, s); break;
            case 3: rc = launch_v0<3>(b, n, mat, pre, post, grid, s); break;
            case 4: rc = launch_v0<4>(b, n, mat, pre, post, grid, s); break;
            default: rc = launch_v0<0>(b, n, mat, pre, post, grid, s); break;     // 5..8 layers or mixed counts
            }
        }
        if (rc != 0) return rc;
    }
    return 0;
}
