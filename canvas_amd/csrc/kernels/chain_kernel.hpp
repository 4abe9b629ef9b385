// chain_kernel.hpp -- the production chain kernel (trip loop, batch-wide chunk walk) as a template, shared by
// chain_ops.hip (1..4 layers, 1 024-lane launch bound: 128 VGPRs) and chain_deep_ops.hip (5..8 layers, 512-lane launch
// bound: the two register sets of eight 16-byte words need more than 128 VGPRs).  See chain_ops.hip for the design.
#pragma once
#include "lut_common.hpp"
#include "grade.hpp"
#include "chain_math.hpp"
#include <string.h>

namespace {

using namespace cvs;

// Job records travel as kernel arguments (<= 4 KiB per launch): compact records of ML layer pointers.
template <int ML> struct JobCT { void *out; const void *layer[ML]; uint64_t npixels; };
template <int ML> struct BatchT {
    static constexpr int kJobs = ML == 4 ? 64 : 48;
    JobCT<ML> jobs[kJobs];
};
// Cut the next launch off a run of jobs: as many frames as the record array and the byte budget allow (at least one).
template <int ML>
inline int fill_batch(BatchT<ML> &b, const cvk_chain_job *jobs, int njobs, int nlayers, uint64_t bytes_per_launch) {
    memset(&b, 0, sizeof b);
    uint64_t bytes = 0;
    int n = 0;
    while (n < njobs && n < BatchT<ML>::kJobs) {
        const cvk_chain_job &src = jobs[n];
        const uint64_t job_bytes = src.npixels * 8 * (uint64_t)(nlayers + 1);
        if (n > 0 && bytes + job_bytes > bytes_per_launch) break;        // the next launch takes it
        b.jobs[n].out = src.out;
        for (int k = 0; k < nlayers; k++) b.jobs[n].layer[k] = src.layer[k];
        b.jobs[n].npixels = src.npixels;
        bytes += job_bytes;
        n++;
    }
    return n;
}
static_assert(sizeof(BatchT<4>) + 160 <= 4096 && sizeof(BatchT<8>) + 160 <= 4096, "kernel arguments must fit the 4 KiB segment");

template <int MAXL, bool PRE, bool POST>
__device__ __forceinline__ uint2 chain_pixel(const uint2 (&px)[MAXL], int nl, const MatR &mat, const uint16_t *lut, const uint16_t *post) {
    if (MAXL >= 2 && mat.cross) return narrow(blend_cross(widen(px[0]), widen(px[MAXL >= 2 ? 1 : 0]), mat.wa, mat.wb));
    px32 acc = mat.plain ? widen(px[0]) : grade<PRE, POST>(px[0], mat, lut, post);
#pragma unroll
    for (int k = 1; k < MAXL; k++)      // static indices only: a runtime-indexed array would live in scratch
        if (k < nl) acc = blend_over(acc, mat.plain ? widen(px[k]) : grade<PRE, POST>(px[k], mat, lut, post), 1.0f);
    return narrow(acc);
}

// ---------------------------------------------------------------- production kernel
//
// hipcc sinks a prefetch load into the iteration that consumes it (its value is only used after the
// back-edge) and answers loop-carried loads with vmcnt(0), so in compiler-visible form the loads of
// trip t+1 never overlap the arithmetic of trip t.  The pixel loads are therefore issued from inline
// asm (invisible to both passes) and waited for by hand (cdna_hip_programming.md 5.7, form (ii)).
//
// One trip of one lane = one pixel pair:
//     asm loads   nxt <- next chunk           NL x global_load_dwordx4 v, v_off, s[base] nt   (lane offsets clamped, never predicated)
//     arithmetic  cur -> res                  chain_math.hpp
//     s_waitcnt vmcnt(0) naming nxt           BEFORE this trip's store is issued
//     store       res                         global_store_dwordx4 ... nt
// Why the wait sits before the store: measured on gfx950, vmcnt(N) with N younger STORES outstanding
// does not guarantee that older LOADS have landed (outputs were wrong until this was changed).  At this point only
// the next trip's loads (needed now anyway) and the previous trip's store (a whole trip old) are outstanding; the
// new store then drains under the next trip's arithmetic.
// The trip loop is unrolled by two with the register sets swapping roles, so no cur = nxt copies.

// address = 64-bit scalar base + 32-bit unsigned lane offset (the global saddr form): no vector address arithmetic
__device__ __forceinline__ void asm_ld4s(u32x4 &dst, const void *sbase, uint32_t voff) {
    asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "=v"(dst) : "v"(voff), "s"(sbase));
}

template <int L>
__device__ __forceinline__ void wait_vm0(u32x4 (&r)[L]) {
    static_assert(L >= 1 && L <= 8, "operand lists below cover 1..8 layers");
    if constexpr (L == 1) asm volatile("s_waitcnt vmcnt(0)" : "+v"(r[0]) : : "memory");
    else if constexpr (L == 2) asm volatile("s_waitcnt vmcnt(0)" : "+v"(r[0]), "+v"(r[1]) : : "memory");
    else if constexpr (L == 3) asm volatile("s_waitcnt vmcnt(0)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]) : : "memory");
    else if constexpr (L == 4) asm volatile("s_waitcnt vmcnt(0)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]) : : "memory");
    else if constexpr (L == 5) asm volatile("s_waitcnt vmcnt(0)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]) : : "memory");
    else if constexpr (L == 6) asm volatile("s_waitcnt vmcnt(0)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]) : : "memory");
    else if constexpr (L == 7) asm volatile("s_waitcnt vmcnt(0)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]) : : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : : "memory");
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

enum { DIAG_NONE = 0, DIAG_MEMORY_ONLY = 1, DIAG_COMPUTE_ONLY = 2 };     // != DIAG_NONE only exists in -DCVS_DIAG builds (tools/)

// Position of a workgroup in the batch's run of chunks; every member is wave-uniform (SGPRs).  The frame's pointers
// ride along so that the argument segment is read once per frame crossed, not once per trip.
template <int NL>
struct Walk {
    unsigned job;           // index into the batch, == njobs when past the end
    unsigned chunk;         // chunk of that job
    unsigned npairs;        // pixel pairs of that job
    unsigned nchunks;       // ceil(npairs / L)
    const char *layer[NL];
    char *out;
};

// NL in 1..4, the same for every job of the batch; every job has 2 <= npixels and npixels * 8 < 4 GiB;
// blockDim.x == 1 << lshift
// Whether an instance fetches through the asm pipeline above.  The property it needs -- no instruction touches a register
// between the load into it and the wait -- is checked on the machine code of every instance at build time
// (tools/check_asm_loads.py, run by the Makefile): an instance that hipcc compiles with a copy of an in-flight register in
// front of the wait is listed here and takes plain loads, waits left to the compiler (slower, never wrong).
//   <6, GRADE, PRE, !POST>: ROCm 7.2's hipcc turns the two register sets of six layers into one set + sixteen v_mov_b64
//   on a loop edge, placed before the wait.
constexpr bool chain_hand_pipelined(int nl, int mode, bool pre, bool post) {
    return !(nl == 6 && mode == CHAIN_GRADE && pre && !post);
}

template <int NL, int MODE, bool PRE, bool POST, int DIAG>
__global__ __launch_bounds__(NL <= 4 ? kWG : 512) void k_chain(BatchT<(NL <= 4 ? 4 : 8)> batch, int njobs_, Mat kmat, int lshift,
                                               const uint16_t *__restrict__ pre, const uint16_t *__restrict__ post) {
    constexpr bool HAND = chain_hand_pipelined(NL, MODE, PRE, POST);
    const MatR mat = CVS_MAT_REGS(kmat);
    __shared__ uint16_t lut[(PRE || POST) ? kLutHalfs : 1];
    const unsigned L = 1u << lshift, G = gridDim.x, tid = threadIdx.x, njobs = (unsigned)njobs_;

    auto load_job = [&](Walk<NL> &w) {
        const auto &job = batch.jobs[w.job];
        w.npairs = (unsigned)(job.npixels >> 1);
        w.nchunks = (w.npairs + L - 1) >> lshift;
#pragma unroll
        for (int k = 0; k < NL; k++) w.layer[k] = reinterpret_cast<const char *>(job.layer[k]);
        w.out = reinterpret_cast<char *>(job.out);
    };
    // move `w` forward by `by` chunks, crossing into later frames as needed
    auto advance = [&](Walk<NL> &w, unsigned by) {
        w.chunk += by;
        while (w.chunk >= w.nchunks) {
            w.chunk -= w.nchunks;
            if (++w.job >= njobs) { w.job = njobs; w.chunk = 0; return; }
            load_job(w);
        }
    };
    // lanes past the end of a frame's last chunk read its last pair again (clamped, never predicated) and do not store
    auto valid_of = [&](const Walk<NL> &w) -> unsigned { const unsigned left = w.npairs - (w.chunk << lshift); return left < L ? left : L; };
    auto issue = [&](u32x4 (&dst)[NL], const Walk<NL> &w) {
        const unsigned valid = valid_of(w);
        const uint32_t voff = (tid < valid ? tid : valid - 1) << 4;
#pragma unroll
        for (int k = 0; k < NL; k++) {
            const char *base = w.layer[k] + ((size_t)w.chunk << (lshift + 4));
            if constexpr (HAND) asm_ld4s(dst[k], base, voff);
            else dst[k] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(base + voff));
        }
    };

    Walk<NL> cur;
    cur.job = 0; cur.chunk = 0;
    load_job(cur);
    advance(cur, blockIdx.x);
    if (cur.job >= njobs) return;               // fewer chunks in the whole batch than workgroups (uniform: no barrier is left behind)
    Walk<NL> next = cur;                        // the chunk after `cur` (job == njobs: there is none)
    advance(next, G);

    u32x4 A[NL], B[NL];
    u32x4 diag_acc = { 0, 0, 0, 0 };
    issue(A, cur);                              // the first trip goes out before the table is staged
    if (PRE) stage_lut_any(lut, pre);
    else if (POST) stage_lut_any(lut, post);
    if constexpr (HAND) wait_vm0(A);
    if (next.job < njobs) issue(B, next);

    // One trip.  On entry `now` holds chunk `cur` and the loads of chunk `next` are in flight into `nxt`.
    // The order inside matters: everything that does not need the loaded data -- the arithmetic on `now` and the walk
    // to the chunk after next -- runs BEFORE the wait, so that between "the loads have landed" and "the following
    // loads are issued" there is nothing but this trip's store.  (The first form of this kernel walked after the
    // wait: some forty scalar instructions on the load -> issue -> load critical loop.)
    auto trip = [&](u32x4 (&now)[NL], u32x4 (&nxt)[NL]) -> bool {
        u32x4 res;
        if (DIAG == DIAG_MEMORY_ONLY) {
            res = now[0];
#pragma unroll
            for (int k = 1; k < NL; k++) res ^= now[k];
        } else {
            res = chain_pair_lean<NL, PRE, POST, MODE>(now, mat, lut, post);
        }
        Walk<NL> after = next;
        if (after.job < njobs) advance(after, G);
        const bool store_it = tid < valid_of(cur);
        g_u4 out = (g_u4)(cur.out + ((size_t)cur.chunk << (lshift + 4)));
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (HAND) wait_vm0(nxt);
        __builtin_amdgcn_sched_barrier(0);
        if (DIAG == DIAG_COMPUTE_ONLY) diag_acc ^= res;
        else if (store_it) __builtin_nontemporal_store(res, out + tid);
        if (after.job < njobs) {
            if (DIAG == DIAG_COMPUTE_ONLY) {
#pragma unroll
                for (int k = 0; k < NL; k++) now[k] = nxt[k];
            } else {
                issue(now, after);              // into the registers this trip has just consumed
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        cur = next;
        next = after;
        return cur.job < njobs;
    };
    while (trip(A, B) && trip(B, A)) { }
    if (DIAG == DIAG_COMPUTE_ONLY) ((g_u4)batch.jobs[0].out)[(size_t)blockIdx.x * L + tid] = diag_acc;

}

// Odd pixel counts: the last pixel of such a frame has no partner.  One lane per frame of the batch, the scalar form of
// the same arithmetic, tables read from global memory -- a launch of its own (only when a batch has such a frame) so
// that the trip loop's kernel carries none of this code.
template <int NL, int MODE, bool PRE, bool POST>
__global__ __launch_bounds__(64) void k_chain_tail(BatchT<(NL <= 4 ? 4 : 8)> batch, int njobs, Mat kmat, const uint16_t *__restrict__ pre, const uint16_t *__restrict__ post) {
    MatR mat = CVS_MAT_REGS(kmat);
    mat.plain = MODE == CHAIN_PLAIN; mat.cross = MODE == CHAIN_CROSS;
    const int j = (int)threadIdx.x;
    if (j >= njobs) return;
    const auto &job = batch.jobs[j];
    if (!(job.npixels & 1)) return;
    uint2 px[NL];
#pragma unroll
    for (int k = 0; k < NL; k++) px[k] = ld2(job.layer[k], job.npixels - 1);
    st2(job.out, job.npixels - 1, chain_pixel<NL, PRE, POST>(px, NL, mat, PRE ? pre : post, post));
}

template <int NL, int MODE, int DIAG>
int launch(const BatchT<(NL <= 4 ? 4 : 8)> &jobs, int njobs, const Mat &mat, const uint16_t *pre, const uint16_t *post, unsigned grid, unsigned block, int lshift, hipStream_t s) {
    bool odd = false;
    for (int i = 0; i < njobs; i++) odd = odd || (jobs.jobs[i].npixels & 1);
    if constexpr (MODE != CHAIN_GRADE) {
        hipLaunchKernelGGL((k_chain<NL, MODE, false, false, DIAG>), dim3(grid), dim3(block), 0, s, jobs, njobs, mat, lshift, pre, post);
        if (odd) hipLaunchKernelGGL((k_chain_tail<NL, MODE, false, false>), dim3(1), dim3(64), 0, s, jobs, njobs, mat, pre, post);
    } else {
        if (pre && post)  hipLaunchKernelGGL((k_chain<NL, MODE, true, true, DIAG>), dim3(grid), dim3(block), 0, s, jobs, njobs, mat, lshift, pre, post);
        else if (pre)     hipLaunchKernelGGL((k_chain<NL, MODE, true, false, DIAG>), dim3(grid), dim3(block), 0, s, jobs, njobs, mat, lshift, pre, post);
        else if (post)    hipLaunchKernelGGL((k_chain<NL, MODE, false, true, DIAG>), dim3(grid), dim3(block), 0, s, jobs, njobs, mat, lshift, pre, post);
        else              hipLaunchKernelGGL((k_chain<NL, MODE, false, false, DIAG>), dim3(grid), dim3(block), 0, s, jobs, njobs, mat, lshift, pre, post);
        if (odd) {
            if (pre && post)  hipLaunchKernelGGL((k_chain_tail<NL, MODE, true, true>), dim3(1), dim3(64), 0, s, jobs, njobs, mat, pre, post);
            else if (pre)     hipLaunchKernelGGL((k_chain_tail<NL, MODE, true, false>), dim3(1), dim3(64), 0, s, jobs, njobs, mat, pre, post);
            else if (post)    hipLaunchKernelGGL((k_chain_tail<NL, MODE, false, true>), dim3(1), dim3(64), 0, s, jobs, njobs, mat, pre, post);
            else              hipLaunchKernelGGL((k_chain_tail<NL, MODE, false, false>), dim3(1), dim3(64), 0, s, jobs, njobs, mat, pre, post);
        }
    }
    return (int)hipGetLastError();
}


}  // namespace
