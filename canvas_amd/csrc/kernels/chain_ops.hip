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
//   * persistent grid, ONE 1024-lane workgroup per CU: the 128 KiB transfer table lives in LDS
//     (160 KiB per CU), staged once per launch; a launch carries up to 32 frames so the staging
//     (~3 us) is amortised;
//   * a lane owns pixel PAIRS: one global_load_dwordx4 per layer, one global_store_dwordx4; the
//     arithmetic of the pair runs as packed f32 (v_pk_mul_f32 / v_pk_add_f32);
//   * software pipeline: the loads of the NEXT trip (same frame, or the first trip of the next
//     frame in the batch) are issued before the ~500 VALU instructions of the current trip, so
//     with only 16 waves per CU there are still P x nlayers x 16 B per lane in flight all the
//     time (P = pairs per trip) -- enough bytes in flight per CU to cover HBM latency;
//   * job records arrive in the kernel-argument segment (scalar loads, nothing to recycle).
// Bound: HBM.  Algorithmic bytes per output pixel: 8 * (nlayers + 1)  (config 2: 24).
#include "lut_common.hpp"
#include "grade.hpp"
#include "chain_math.hpp"
#include <stdlib.h>

using namespace cvs;

namespace {

constexpr int kJobsPerLaunch = 32;
struct Batch { cvk_chain_job jobs[kJobsPerLaunch]; };

// ---------------------------------------------------------------- v0: first correct version (kept for A/B)

template <int MAXL, bool PRE, bool POST>
__device__ __forceinline__ uint2 chain_pixel(const uint2 (&px)[MAXL], int nl, const Mat &mat, const uint16_t *lut, const uint16_t *post) {
    px32 acc = grade<PRE, POST>(px[0], mat, lut, post);
#pragma unroll
    for (int k = 1; k < MAXL; k++)      // static indices only: a runtime-indexed array would live in scratch
        if (k < nl) acc = blend_over(acc, grade<PRE, POST>(px[k], mat, lut, post), 1.0f);
    return narrow(acc);
}

template <int NL, bool PRE, bool POST>
__global__ __launch_bounds__(kWG) void k_chain_v0(Batch batch, int njobs, Mat mat,
                                                  const uint16_t *__restrict__ pre, const uint16_t *__restrict__ post) {
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

// ---------------------------------------------------------------- v1: packed pair math + software pipeline

template <bool PRE, bool POST>
__device__ __forceinline__ px32x2 grade2(uint4 p, const Mat &mat, const uint16_t *lds_lut, const uint16_t *glb_post) {
    if (PRE) {
        p.x = gather2<true>(lds_lut, p.x); p.y = gather2<true>(lds_lut, p.y);
        p.z = gather2<true>(lds_lut, p.z); p.w = gather2<true>(lds_lut, p.w);
    }
    uint4 h = narrow2(mat3x2(widen2(p), mat.m));
    if (POST) {
        if (PRE) { h.x = gather2<false>(glb_post, h.x); h.y = gather2<false>(glb_post, h.y); h.z = gather2<false>(glb_post, h.z); h.w = gather2<false>(glb_post, h.w); }
        else     { h.x = gather2<true>(lds_lut, h.x);   h.y = gather2<true>(lds_lut, h.y);   h.z = gather2<true>(lds_lut, h.z);   h.w = gather2<true>(lds_lut, h.w); }
    }
    return widen2(h);
}

template <int MAXL, bool PRE, bool POST>
__device__ __forceinline__ uint4 chain_pair(const uint4 (&w)[MAXL], int nl, const Mat &mat, const uint16_t *lut, const uint16_t *post) {
    px32x2 acc = grade2<PRE, POST>(w[0], mat, lut, post);
#pragma unroll
    for (int k = 1; k < MAXL; k++)
        if (k < nl) acc = blend_over2_mix1(acc, grade2<PRE, POST>(w[k], mat, lut, post));
    return narrow2(acc);
}

// P pairs per trip; trip t of a job covers pair indices t*P*stride + p*stride + lane  (p < P)
template <int NL, int P, bool PRE, bool POST, int DIAG = 0>
__global__ __launch_bounds__(kWG) void k_chain(Batch batch, int njobs, Mat mat,
                                               const uint16_t *__restrict__ pre, const uint16_t *__restrict__ post) {
    __shared__ uint16_t lut[(PRE || POST) ? kLutHalfs : 1];
    constexpr int MAXL = NL > 0 ? NL : CVK_CHAIN_MAX_LAYERS;
    const size_t stride = (size_t)gridDim.x * kWG;
    const size_t lane = (size_t)blockIdx.x * kWG + threadIdx.x;

    uint4 cur[P][MAXL], nxt[P][MAXL];
    // first trip of the first frame goes out before the table is staged
    {
        const cvk_chain_job &job = batch.jobs[0];
        const int nl = NL > 0 ? NL : job.nlayers;
        const size_t npairs = job.npixels / 2;
#pragma unroll
        for (int p = 0; p < P; p++) {
            const size_t idx = lane + (size_t)p * stride;
#pragma unroll
            for (int k = 0; k < MAXL; k++) if (k < nl && idx < npairs) cur[p][k] = ld4(job.layer[k], idx);
        }
    }
    if (PRE) stage_lut(lut, pre);
    else if (POST) stage_lut(lut, post);

    for (int j = 0; j < njobs; j++) {
        const cvk_chain_job &job = batch.jobs[j];
        const int nl = NL > 0 ? NL : job.nlayers;
        const size_t npairs = job.npixels / 2;
        const bool more_jobs = j + 1 < njobs;
        const cvk_chain_job &njob = batch.jobs[more_jobs ? j + 1 : j];
        const int nnl = NL > 0 ? NL : njob.nlayers;
        const size_t nnpairs = njob.npixels / 2;

        for (size_t base = lane; base < npairs; base += (size_t)P * stride) {
            // prefetch: next trip of this frame, or the first trip of the next frame
            const size_t nbase = base + (size_t)P * stride;
            if (nbase < npairs) {
#pragma unroll
                for (int p = 0; p < P; p++) {
                    const size_t idx = nbase + (size_t)p * stride;
#pragma unroll
                    for (int k = 0; k < MAXL; k++) if (k < nl && idx < npairs) nxt[p][k] = ld4(job.layer[k], idx);
                }
            } else if (more_jobs) {
#pragma unroll
                for (int p = 0; p < P; p++) {
                    const size_t idx = lane + (size_t)p * stride;
#pragma unroll
                    for (int k = 0; k < MAXL; k++) if (k < nnl && idx < nnpairs) nxt[p][k] = ld4(njob.layer[k], idx);
                }
            }
#pragma unroll
            for (int p = 0; p < P; p++) {
                const size_t idx = base + (size_t)p * stride;
                if (idx < npairs) {
                    if (DIAG == 1) {                      // diagnostic build: memory traffic only
                        uint4 x = cur[p][0];
#pragma unroll
                        for (int k = 1; k < MAXL; k++) if (k < nl) { x.x ^= cur[p][k].x; x.y ^= cur[p][k].y; x.z ^= cur[p][k].z; x.w ^= cur[p][k].w; }
                        st4(job.out, idx, x);
                    } else {
                        st4(job.out, idx, chain_pair<MAXL, PRE, POST>(cur[p], nl, mat, lut, post));
                    }
                }
            }
#pragma unroll
            for (int p = 0; p < P; p++)
#pragma unroll
                for (int k = 0; k < MAXL; k++) cur[p][k] = nxt[p][k];
        }
        // lanes that had no trip in this frame still need the first trip of the next one
        if (lane >= npairs && more_jobs) {
#pragma unroll
            for (int p = 0; p < P; p++) {
                const size_t idx = lane + (size_t)p * stride;
#pragma unroll
                for (int k = 0; k < MAXL; k++) if (k < nnl && idx < nnpairs) cur[p][k] = ld4(njob.layer[k], idx);
            }
        }
        // odd pixel count: the last pixel on its own (scalar path)
        if ((job.npixels & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
            uint2 px[MAXL];
#pragma unroll
            for (int k = 0; k < MAXL; k++) if (k < nl) px[k] = ld2(job.layer[k], job.npixels - 1);
            st2(job.out, job.npixels - 1, chain_pixel<MAXL, PRE, POST>(px, nl, mat, lut, post));
        }
    }
}


// ---------------------------------------------------------------- v2: hand-counted software pipeline
//
// hipcc sinks a prefetch load into the iteration that consumes it (the value is only used after the
// back-edge), and its s_waitcnt pass answers loop-carried loads with vmcnt(0); either way the loads of
// trip t+1 never overlap the arithmetic of trip t.  So the pixel loads are issued from inline asm
// (invisible to both passes) and waited for by hand (cdna_hip_programming.md section 5.7, form (ii)):
//
//     asm loads  nxt <- trip t+1          P*NL x global_load_dwordx4 ... nt
//     compute + store trip t              (compiler-visible; ns = stores actually issued, <= P)
//     s_waitcnt vmcnt(ns) naming nxt      in-order counter: everything older than the ns stores is back
//     cur = nxt
//
// Every trip issues exactly P*NL loads (indices are clamped into the frame instead of predicated), so
// the only varying count is ns, which is wave-uniform and picked with a scalar branch.

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
    for (int i = threadIdx.x; i < kLutHalfs * 2 / 16; i += blockDim.x) dst[i] = src[i];
    __syncthreads();
}

__device__ __forceinline__ void st4_nt(void *p, size_t i, uint4 v) {
    u32x4 t = { v.x, v.y, v.z, v.w };
    __builtin_nontemporal_store(t, (g_u4)p + i);
}

// NL in 1..4 (uniform per batch), P in {1, 2}; every job has npixels >= 2
template <int NL, int P, bool PRE, bool POST, bool NT, int DIAG = 0, bool LEAN = false, int PF = 0>
__global__ __launch_bounds__(kWG) void k_chain_pipe(Batch batch, int njobs, Mat mat,
                                                    const uint16_t *__restrict__ pre, const uint16_t *__restrict__ post) {
    __shared__ uint16_t lut[(PRE || POST) ? kLutHalfs : 1];
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t lane = (size_t)blockIdx.x * blockDim.x + threadIdx.x;

    u32x4 cur[P][NL], nxt[P][NL];
    uint4 diag_acc = make_uint4(0, 0, 0, 0);
    {
        const cvk_chain_job &job = batch.jobs[0];
        const size_t last = job.npixels / 2 - 1;
#pragma unroll
        for (int p = 0; p < P; p++) {
            size_t idx = lane + (size_t)p * stride;
            idx = idx < last ? idx : last;
#pragma unroll
            for (int k = 0; k < NL; k++) asm_ld4<NT>(cur[p][k], job.layer[k], idx);
        }
    }
    if (PRE) stage_lut_any(lut, pre);
    else if (POST) stage_lut_any(lut, post);
    wait_vm<0>(cur);

    for (int j = 0; j < njobs; j++) {
        const cvk_chain_job &job = batch.jobs[j];
        const size_t npairs = job.npixels / 2;
        const bool more_jobs = j + 1 < njobs;
        const cvk_chain_job &njob = batch.jobs[more_jobs ? j + 1 : j];
        const size_t nnpairs = njob.npixels / 2;
        bool primed = false;                  // this lane already holds its first trip of the next frame
        // frame base pointers as wave-uniform values (scalar loads); the per-lane choice below is then a
        // register select, not a second, vector, load of the job record
        const void *lp[NL], *nlp[NL];
#pragma unroll
        for (int k = 0; k < NL; k++) { lp[k] = job.layer[k]; nlp[k] = njob.layer[k]; }

        for (size_t base = lane; base < npairs; base += (size_t)P * stride) {
            const size_t nbase = base + (size_t)P * stride;
            const bool same = nbase < npairs;
            const size_t first = same ? nbase : lane;
            const size_t last = (same ? npairs : nnpairs) - 1;
            if (!same) primed = true;
            auto prefetch = [&]() {
#pragma unroll
                for (int p = 0; p < P; p++) {
                    size_t idx = first + (size_t)p * stride;
                    idx = idx < last ? idx : last;
#pragma unroll
                    for (int k = 0; k < NL; k++) { if (DIAG == 2) nxt[p][k] = cur[p][k]; else asm_ld4<NT>(nxt[p][k], same ? lp[k] : nlp[k], idx); }
                }
                __builtin_amdgcn_sched_barrier(0);
            };
            if (PF == 0) prefetch();

            // results first, into registers ...
            uint4 res[P];
            if (LEAN && PF > 0) {
                // same arithmetic as chain_pair_lean, opened up so that the prefetch can be issued part-way
                // through the trip: fewer bytes in flight per CU for the same latency cover
                px32x2 acc[P];
#pragma unroll
                for (int p = 0; p < P; p++) acc[p] = grade_pair<PRE, POST>(cur[p][0], mat, lut, post);
                __builtin_amdgcn_sched_barrier(0);
                if (PF == 1) prefetch();
#pragma unroll
                for (int k = 1; k < NL; k++) {
#pragma unroll
                    for (int p = 0; p < P; p++) acc[p] = over_pair(acc[p], grade_pair<PRE, POST>(cur[p][k], mat, lut, post));
                }
                __builtin_amdgcn_sched_barrier(0);
                if (PF == 2) prefetch();
#pragma unroll
                for (int p = 0; p < P; p++) { const u32x4 r = narrow_pair(acc[p]); res[p] = make_uint4(r.x, r.y, r.z, r.w); }
            } else {
#pragma unroll
            for (int p = 0; p < P; p++) {
                const size_t idx = base + (size_t)p * stride;
                if (idx < npairs) {
                    if (LEAN) {
                        const u32x4 r = chain_pair_lean<NL, PRE, POST>(cur[p], mat, lut, post);
                        res[p] = make_uint4(r.x, r.y, r.z, r.w);
                    } else {
                        uint4 w[NL];
#pragma unroll
                        for (int k = 0; k < NL; k++) w[k] = make_uint4(cur[p][k].x, cur[p][k].y, cur[p][k].z, cur[p][k].w);
                        res[p] = chain_pair<NL, PRE, POST>(w, NL, mat, lut, post);
                    }
                }
            }
            }
            __builtin_amdgcn_sched_barrier(0);
            // ... then ONE wait.  Measured on gfx950: a younger store can retire before an older load, so
            // vmcnt(number of younger stores) does not guarantee the loads are back.  Waiting here, BEFORE
            // this trip's stores are issued, leaves only the next trip's loads (needed now anyway) and the
            // previous trip's stores (issued a whole trip ago) to wait for; this trip's stores then drain
            // under the next trip's arithmetic.
            wait_vm<0>(nxt);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int p = 0; p < P; p++) {
                const size_t idx = base + (size_t)p * stride;
                if (idx < npairs) {
                    if (DIAG == 2) { diag_acc.x ^= res[p].x; diag_acc.y ^= res[p].y; diag_acc.z ^= res[p].z; diag_acc.w ^= res[p].w; }   // compute-only build: no traffic
                    else if (NT) st4_nt(job.out, idx, res[p]); else st4(job.out, idx, res[p]);
                }
            }
#pragma unroll
            for (int p = 0; p < P; p++)
#pragma unroll
                for (int k = 0; k < NL; k++) cur[p][k] = nxt[p][k];
        }
        // a lane with no trip in this frame (or whose last trip was not the frame's last) starts the next frame cold
        if (more_jobs && !primed) {
            const size_t last = nnpairs - 1;
#pragma unroll
            for (int p = 0; p < P; p++) {
                size_t idx = lane + (size_t)p * stride;
                idx = idx < last ? idx : last;
#pragma unroll
                for (int k = 0; k < NL; k++) asm_ld4<NT>(cur[p][k], nlp[k], idx);
            }
            wait_vm<0>(cur);
        }
        if (DIAG == 2 && lane < npairs) st4(job.out, lane, diag_acc);
        if ((job.npixels & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
            uint2 px[NL];
#pragma unroll
            for (int k = 0; k < NL; k++) px[k] = ld2(job.layer[k], job.npixels - 1);
            st2(job.out, job.npixels - 1, chain_pixel<NL, PRE, POST>(px, NL, mat, lut, post));
        }
    }
}

template <int NL, int P>
int launch_pipe_diag(const Batch &jobs, int njobs, const Mat &mat, const uint16_t *pre, const uint16_t *post, unsigned grid, unsigned block, hipStream_t s) {
    if (pre) hipLaunchKernelGGL((k_chain_pipe<NL, P, true, false, true, 2>), dim3(grid), dim3(block), 0, s, jobs, njobs, mat, pre, post);
    else     hipLaunchKernelGGL((k_chain_pipe<NL, P, false, false, true, 2>), dim3(grid), dim3(block), 0, s, jobs, njobs, mat, pre, post);
    return (int)hipGetLastError();
}

template <int NL, int P, int PF>
int launch_lean_pf(const Batch &jobs, int njobs, const Mat &mat, const uint16_t *pre, const uint16_t *post, unsigned grid, unsigned block, hipStream_t s) {
    if (pre) hipLaunchKernelGGL((k_chain_pipe<NL, P, true, false, true, 0, true, PF>), dim3(grid), dim3(block), 0, s, jobs, njobs, mat, pre, post);
    else     hipLaunchKernelGGL((k_chain_pipe<NL, P, false, false, true, 0, true, PF>), dim3(grid), dim3(block), 0, s, jobs, njobs, mat, pre, post);
    return (int)hipGetLastError();
}

template <int NL, int P, int DIAG>
int launch_lean(const Batch &jobs, int njobs, const Mat &mat, const uint16_t *pre, const uint16_t *post, unsigned grid, unsigned block, hipStream_t s) {
    if (pre && post)  hipLaunchKernelGGL((k_chain_pipe<NL, P, true, true, true, DIAG, true>), dim3(grid), dim3(block), 0, s, jobs, njobs, mat, pre, post);
    else if (pre)     hipLaunchKernelGGL((k_chain_pipe<NL, P, true, false, true, DIAG, true>), dim3(grid), dim3(block), 0, s, jobs, njobs, mat, pre, post);
    else if (post)    hipLaunchKernelGGL((k_chain_pipe<NL, P, false, true, true, DIAG, true>), dim3(grid), dim3(block), 0, s, jobs, njobs, mat, pre, post);
    else              hipLaunchKernelGGL((k_chain_pipe<NL, P, false, false, true, DIAG, true>), dim3(grid), dim3(block), 0, s, jobs, njobs, mat, pre, post);
    return (int)hipGetLastError();
}

template <int NL, int P, bool NT>
int launch_pipe(const Batch &jobs, int njobs, const Mat &mat, const uint16_t *pre, const uint16_t *post, unsigned grid, unsigned block, hipStream_t s) {
    if (pre && post)  hipLaunchKernelGGL((k_chain_pipe<NL, P, true, true, NT>), dim3(grid), dim3(block), 0, s, jobs, njobs, mat, pre, post);
    else if (pre)     hipLaunchKernelGGL((k_chain_pipe<NL, P, true, false, NT>), dim3(grid), dim3(block), 0, s, jobs, njobs, mat, pre, post);
    else if (post)    hipLaunchKernelGGL((k_chain_pipe<NL, P, false, true, NT>), dim3(grid), dim3(block), 0, s, jobs, njobs, mat, pre, post);
    else              hipLaunchKernelGGL((k_chain_pipe<NL, P, false, false, NT>), dim3(grid), dim3(block), 0, s, jobs, njobs, mat, pre, post);
    return (int)hipGetLastError();
}

template <int NL, int P>
int launch_v1(const Batch &jobs, int njobs, const Mat &mat, const uint16_t *pre, const uint16_t *post, unsigned grid, hipStream_t s) {
    if (pre && post)  hipLaunchKernelGGL((k_chain<NL, P, true, true>), dim3(grid), dim3(kWG), 0, s, jobs, njobs, mat, pre, post);
    else if (pre)     hipLaunchKernelGGL((k_chain<NL, P, true, false>), dim3(grid), dim3(kWG), 0, s, jobs, njobs, mat, pre, post);
    else if (post)    hipLaunchKernelGGL((k_chain<NL, P, false, true>), dim3(grid), dim3(kWG), 0, s, jobs, njobs, mat, pre, post);
    else              hipLaunchKernelGGL((k_chain<NL, P, false, false>), dim3(grid), dim3(kWG), 0, s, jobs, njobs, mat, pre, post);
    return (int)hipGetLastError();
}

template <int NL, int P>
int launch_diag(const Batch &jobs, int njobs, const Mat &mat, const uint16_t *pre, const uint16_t *post, unsigned grid, hipStream_t s) {
    hipLaunchKernelGGL((k_chain<NL, P, false, false, 1>), dim3(grid), dim3(kWG), 0, s, jobs, njobs, mat, pre, post);
    return (int)hipGetLastError();
}

template <int NL>
int launch_v0(const Batch &jobs, int njobs, const Mat &mat, const uint16_t *pre, const uint16_t *post, unsigned grid, hipStream_t s) {
    if (pre && post)  hipLaunchKernelGGL((k_chain_v0<NL, true, true>), dim3(grid), dim3(kWG), 0, s, jobs, njobs, mat, pre, post);
    else if (pre)     hipLaunchKernelGGL((k_chain_v0<NL, true, false>), dim3(grid), dim3(kWG), 0, s, jobs, njobs, mat, pre, post);
    else if (post)    hipLaunchKernelGGL((k_chain_v0<NL, false, true>), dim3(grid), dim3(kWG), 0, s, jobs, njobs, mat, pre, post);
    else              hipLaunchKernelGGL((k_chain_v0<NL, false, false>), dim3(grid), dim3(kWG), 0, s, jobs, njobs, mat, pre, post);
    return (int)hipGetLastError();
}

}  // namespace

// CVS_CHAIN_VARIANT (development knob, read per call): 0 = v0, 1 = pipelined P=1, 2 = pipelined P=2 (default)
extern "C" int cvk_chain_color_over(const cvk_chain_job *jobs, int njobs, int uniform_layers, const float m[9],
                                    const uint16_t *pre, const uint16_t *post, int cus, void *stream) {
    Mat mat = make_mat(m);
    unsigned grid = (unsigned)(cus > 0 ? cus : 256);
    hipStream_t s = (hipStream_t)stream;
    const char *env = getenv("CVS_CHAIN_VARIANT");
    const int variant = env ? atoi(env) : 2;
    const char *bl = getenv("CVS_CHAIN_BLOCK");
    const unsigned block = bl ? (unsigned)atoi(bl) : (unsigned)kWG;
    const char *gm = getenv("CVS_DIAG_GRIDMUL");
    const unsigned gridmul = gm ? (unsigned)atoi(gm) : 1u;   /* diagnostic kernels use no LDS: more workgroups per CU fit */
    for (int first = 0; first < njobs; first += kJobsPerLaunch) {
        int n = njobs - first < kJobsPerLaunch ? njobs - first : kJobsPerLaunch;
        Batch b;
        memset(&b, 0, sizeof b);
        memcpy(b.jobs, jobs + first, sizeof(cvk_chain_job) * (size_t)n);
        int rc;
#define CVK_DISPATCH(NLV)                                                                   \
        (variant == 0 ? launch_v0<NLV>(b, n, mat, pre, post, grid, s)                      \
         : variant == 6 ? launch_lean<NLV, 2, 0>(b, n, mat, pre, post, grid, block, s)      \
         : variant == 7 ? launch_lean<NLV, 1, 0>(b, n, mat, pre, post, grid, block, s)      \
         : variant == 71 ? launch_lean_pf<NLV, 1, 1>(b, n, mat, pre, post, grid, block, s)  \
         : variant == 72 ? launch_lean_pf<NLV, 1, 2>(b, n, mat, pre, post, grid, block, s)  \
         : variant == 61 ? launch_lean_pf<NLV, 2, 1>(b, n, mat, pre, post, grid, block, s)  \
         : variant == 62 ? launch_lean_pf<NLV, 2, 2>(b, n, mat, pre, post, grid, block, s)  \
         : variant == 16 ? launch_lean<NLV, 2, 2>(b, n, mat, pre, post, grid, block, s)     \
         : variant == 12 ? launch_pipe_diag<NLV, 2>(b, n, mat, pre, post, grid, block, s)   \
         : variant == 13 ? launch_pipe_diag<NLV, 1>(b, n, mat, pre, post, grid, block, s)   \
         : variant == 10 ? launch_diag<NLV, 2>(b, n, mat, pre, post, grid * gridmul, s)     \
         : variant == 11 ? launch_diag<NLV, 1>(b, n, mat, pre, post, grid * gridmul, s)     \
         : variant == 1 ? launch_v1<NLV, 1>(b, n, mat, pre, post, grid, s)                 \
         : variant == 3 ? launch_pipe<NLV, 1, true>(b, n, mat, pre, post, grid, block, s)   \
         : variant == 4 ? launch_pipe<NLV, 2, true>(b, n, mat, pre, post, grid, block, s)   \
         : variant == 5 ? launch_pipe<NLV, 2, false>(b, n, mat, pre, post, grid, block, s)  \
                        : launch_v1<NLV, 2>(b, n, mat, pre, post, grid, s))
        switch (uniform_layers) {
        case 1: rc = CVK_DISPATCH(1); break;
        case 2: rc = CVK_DISPATCH(2); break;
        case 3: rc = CVK_DISPATCH(3); break;
        case 4: rc = CVK_DISPATCH(4); break;
        default: rc = variant == 0 ? launch_v0<0>(b, n, mat, pre, post, grid, s) : launch_v1<0, 1>(b, n, mat, pre, post, grid, s); break;
        }
#undef CVK_DISPATCH
        if (rc != 0) return rc;
    }
    return 0;
}
