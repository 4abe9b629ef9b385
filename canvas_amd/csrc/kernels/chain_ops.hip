// chain_ops.hip -- the fused colour-matrix + alpha-over chain (BASELINE config 2, the north star).
//
// What the reference does per layer (src/cprocess/color.c:104-165) and per stack
// (src/cprocess/workspace.c:530-544, src/cprocess/main.c:43-71,115-139):
//   [LUT over all 4 halfs] -> widen -> 3x3 in f32 -> truncate to half -> [LUT] -> widen ->
//   over(acc, layer, 1.0) for every layer above the first -> truncate to half,
// each arrow a pass over memory with f32 intermediates.  Here a pixel goes through the whole
// chain in registers: 8 B read per layer pixel + 8 B written per output pixel.
//
// Shape of the kernel (MI355X), second generation (what changed and why: DESIGN.md 4.1):
//   * 256 workgroups of 512 lanes, one per CU: the 128 KiB transfer table lives in LDS (160 KiB per CU);
//   * a lane owns pixel PAIRS: one global_load_dwordx4 per layer, one global_store_dwordx4, all non-temporal;
//   * the batch of frames is ONE run of 8 KiB chunks (512 pairs): workgroup b takes chunks b, b + 256, b + 512, ...
//     of the whole batch, walking from frame to frame without a per-frame tail.  The walk is wave-uniform, so it
//     lives in SGPRs: a load is `global_load_dwordx4 v, v_lane_offset, s[chunk base]` and the trip loop carries no
//     vector address arithmetic at all (the first generation spent 18 VALU instructions per trip on 64-bit
//     per-lane indices and frame-boundary selects);
//   * software pipeline by hand: the loads of trip t+1 are in flight during the arithmetic of trip t (inline asm,
//     hipcc would sink them; cdna_hip_programming.md 5.7);
//   * SHORT LAUNCHES: the host cuts a batch into launches of about eight 4K frames (kBytesPerLaunch).  Measured: in a
//     long launch the persistent workgroups drift apart (some CUs stream a little faster for the whole launch), the
//     2 MiB window of addresses in flight chip-wide smears out and HBM efficiency falls -- a plain 2-read + 1-write
//     stream drops from 0.80 of 8 TB/s at 4 frames per launch to 0.715 at 64.  A kernel boundary is the cheapest
//     rendezvous there is (~3.5 us including the table staging).
// Bound: HBM.  Algorithmic bytes per output pixel: 8 * (nlayers + 1)  (config 2: 24).
#include "lut_common.hpp"
#include "grade.hpp"
#include "chain_math.hpp"
#include <stdlib.h>

using namespace cvs;

namespace {

// Job records travel as kernel arguments (<= 4 KiB per launch).  The production kernel takes up to four layers, so
// its records are compact (48 B) and a launch can carry 64 frames; the first version keeps the full 8-layer record.
constexpr int kJobsPerLaunchV0 = 32, kJobsPerLaunch = 64, kFusedLayers = 4;
// bytes (read + written) one launch of the production kernel moves before the host starts the next one:
// eight 4K two-layer frames.  See "SHORT LAUNCHES" above; tools/chainlab.hip and profiles/r02/launch_split.txt.
constexpr uint64_t kBytesPerLaunch = 8ull * 3840 * 2160 * 24;
constexpr int kChainBlock = 512, kChainBlockLog2 = 9;
struct BatchV0 { cvk_chain_job jobs[kJobsPerLaunchV0]; };
struct JobC { void *out; const void *layer[kFusedLayers]; uint64_t npixels; };
struct Batch { JobC jobs[kJobsPerLaunch]; };
static_assert(sizeof(Batch) + 128 <= 4096, "kernel arguments must fit the 4 KiB segment");

// ---------------------------------------------------------------- v0: first correct version
// Serves stacks of 5..8 layers, batches with mixed layer counts and frames too large for 32-bit chunk offsets.

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
    static_assert(L >= 1 && L <= 4, "operand lists below cover 1..4 layers");
    if constexpr (L == 1) asm volatile("s_waitcnt vmcnt(0)" : "+v"(r[0]) : : "memory");
    else if constexpr (L == 2) asm volatile("s_waitcnt vmcnt(0)" : "+v"(r[0]), "+v"(r[1]) : : "memory");
    else if constexpr (L == 3) asm volatile("s_waitcnt vmcnt(0)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]) : : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]) : : "memory");
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
template <int NL, int MODE, bool PRE, bool POST, int DIAG>
__global__ __launch_bounds__(kWG) void k_chain(Batch batch, int njobs_, Mat kmat, int lshift,
                                               const uint16_t *__restrict__ pre, const uint16_t *__restrict__ post) {
    const MatR mat = CVS_MAT_REGS(kmat);
    __shared__ uint16_t lut[(PRE || POST) ? kLutHalfs : 1];
    const unsigned L = 1u << lshift, G = gridDim.x, tid = threadIdx.x, njobs = (unsigned)njobs_;

    auto load_job = [&](Walk<NL> &w) {
        const JobC &job = batch.jobs[w.job];
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
        for (int k = 0; k < NL; k++) asm_ld4s(dst[k], w.layer[k] + ((size_t)w.chunk << (lshift + 4)), voff);
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
    wait_vm0(A);
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
        wait_vm0(nxt);
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
__global__ __launch_bounds__(64) void k_chain_tail(Batch batch, int njobs, Mat kmat, const uint16_t *__restrict__ pre, const uint16_t *__restrict__ post) {
    MatR mat = CVS_MAT_REGS(kmat);
    mat.plain = MODE == CHAIN_PLAIN; mat.cross = MODE == CHAIN_CROSS;
    const int j = (int)threadIdx.x;
    if (j >= njobs) return;
    const JobC &job = batch.jobs[j];
    if (!(job.npixels & 1)) return;
    uint2 px[NL];
#pragma unroll
    for (int k = 0; k < NL; k++) px[k] = ld2(job.layer[k], job.npixels - 1);
    st2(job.out, job.npixels - 1, chain_pixel<NL, PRE, POST>(px, NL, mat, PRE ? pre : post, post));
}

template <int NL, int MODE, int DIAG>
int launch(const Batch &jobs, int njobs, const Mat &mat, const uint16_t *pre, const uint16_t *post, unsigned grid, unsigned block, int lshift, hipStream_t s) {
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

template <int DIAG>
int launch_nl(int nl, const Batch &b, int n, const Mat &mat, const uint16_t *pre, const uint16_t *post, unsigned grid, unsigned block, int lshift, hipStream_t s) {
    if (mat.cross) return launch<2, CHAIN_CROSS, DIAG>(b, n, mat, pre, post, grid, block, lshift, s);
    if (mat.plain) {
        switch (nl) {
        case 1: return launch<1, CHAIN_PLAIN, DIAG>(b, n, mat, pre, post, grid, block, lshift, s);
        case 2: return launch<2, CHAIN_PLAIN, DIAG>(b, n, mat, pre, post, grid, block, lshift, s);
        case 3: return launch<3, CHAIN_PLAIN, DIAG>(b, n, mat, pre, post, grid, block, lshift, s);
        default: return launch<4, CHAIN_PLAIN, DIAG>(b, n, mat, pre, post, grid, block, lshift, s);
        }
    }
    switch (nl) {
    case 1: return launch<1, CHAIN_GRADE, DIAG>(b, n, mat, pre, post, grid, block, lshift, s);
    case 2: return launch<2, CHAIN_GRADE, DIAG>(b, n, mat, pre, post, grid, block, lshift, s);
    case 3: return launch<3, CHAIN_GRADE, DIAG>(b, n, mat, pre, post, grid, block, lshift, s);
    default: return launch<4, CHAIN_GRADE, DIAG>(b, n, mat, pre, post, grid, block, lshift, s);
    }
}

template <int NL>
int launch_v0(const BatchV0 &jobs, int njobs, const Mat &mat, const uint16_t *pre, const uint16_t *post, unsigned grid, hipStream_t s) {
    if (pre && post)  hipLaunchKernelGGL((k_chain_v0<NL, true, true>), dim3(grid), dim3(kWG), 0, s, jobs, njobs, mat, pre, post);
    else if (pre)     hipLaunchKernelGGL((k_chain_v0<NL, true, false>), dim3(grid), dim3(kWG), 0, s, jobs, njobs, mat, pre, post);
    else if (post)    hipLaunchKernelGGL((k_chain_v0<NL, false, true>), dim3(grid), dim3(kWG), 0, s, jobs, njobs, mat, pre, post);
    else              hipLaunchKernelGGL((k_chain_v0<NL, false, false>), dim3(grid), dim3(kWG), 0, s, jobs, njobs, mat, pre, post);
    return (int)hipGetLastError();
}

// What tools/ may turn (only a -DCVS_DIAG build reads them, once per call; the shipped library has no knobs on this path):
//   CVS_CHAIN_VARIANT  1 = production kernel (default); 0 = the first version; 10 = memory traffic only, 12 = arithmetic
//                      only (both produce WRONG pixels: timing only)
//   CVS_CHAIN_BLOCK    lanes per workgroup: 256, 512 (default) or 1024
//   CVS_CHAIN_LAUNCH_MB  bytes moved per launch before the batch is cut, in MiB (default kBytesPerLaunch)
struct Tuning { int variant; unsigned block; int lshift; uint64_t bytes_per_launch; };

Tuning tuning() {
    Tuning t = { 1, (unsigned)kChainBlock, kChainBlockLog2, kBytesPerLaunch };
#ifdef CVS_DIAG
    if (const char *e = getenv("CVS_CHAIN_VARIANT")) t.variant = atoi(e);
    if (const char *e = getenv("CVS_CHAIN_BLOCK")) {
        const int b = atoi(e);
        if (b == 256) { t.block = 256; t.lshift = 8; } else if (b == 1024) { t.block = 1024; t.lshift = 10; }
    }
    if (const char *e = getenv("CVS_CHAIN_LAUNCH_MB")) { const long long mb = atoll(e); if (mb > 0) t.bytes_per_launch = (uint64_t)mb << 20; }
#endif
    return t;
}

thread_local int t_launches = 0;      // launches of the trip-loop kernel the calling thread's last chain call made

int chain_dispatch(const cvk_chain_job *jobs, int njobs, int uniform_layers, const Mat &mat,
                   const uint16_t *pre, const uint16_t *post, int cus, void *stream) {
    const unsigned grid = (unsigned)(cus > 0 ? cus : 256);
    hipStream_t s = (hipStream_t)stream;
    const Tuning t = tuning();
    bool fused_kernel = t.variant != 0 && uniform_layers >= 1 && uniform_layers <= kFusedLayers;
    for (int i = 0; fused_kernel && i < njobs; i++)             // chunk offsets are 32-bit
        if (jobs[i].npixels < 2 || jobs[i].npixels * 8 >= (1ull << 32)) fused_kernel = false;
    int first = 0;
    while (first < njobs) {
        int n = 0, rc;
        if (fused_kernel) {
            Batch b;
            memset(&b, 0, sizeof b);
            uint64_t bytes = 0;
            while (first + n < njobs && n < kJobsPerLaunch) {
                const cvk_chain_job &src = jobs[first + n];
                const uint64_t job_bytes = src.npixels * 8 * (uint64_t)(uniform_layers + 1);
                if (n > 0 && bytes + job_bytes > t.bytes_per_launch) break;        // the next launch takes it
                b.jobs[n].out = src.out;
                for (int k = 0; k < uniform_layers; k++) b.jobs[n].layer[k] = src.layer[k];
                b.jobs[n].npixels = src.npixels;
                bytes += job_bytes;
                n++;
            }
#ifdef CVS_DIAG
            if (t.variant == 10) rc = launch_nl<DIAG_MEMORY_ONLY>(uniform_layers, b, n, mat, pre, post, grid, t.block, t.lshift, s);
            else if (t.variant == 12) rc = launch_nl<DIAG_COMPUTE_ONLY>(uniform_layers, b, n, mat, pre, post, grid, t.block, t.lshift, s);
            else
#endif
            rc = launch_nl<DIAG_NONE>(uniform_layers, b, n, mat, pre, post, grid, t.block, t.lshift, s);
        } else {
            BatchV0 b;
            memset(&b, 0, sizeof b);
            n = njobs - first < kJobsPerLaunchV0 ? njobs - first : kJobsPerLaunchV0;
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
        first += n;
        t_launches++;
    }
    return 0;
}

}  // namespace

extern "C" void cvk_chain_count_reset(void) { t_launches = 0; }
extern "C" int cvk_chain_count(void) { return t_launches; }

extern "C" int cvk_chain_color_over(const cvk_chain_job *jobs, int njobs, int uniform_layers, const float *m,
                                    const uint16_t *pre, const uint16_t *post, int cus, void *stream) {
    return chain_dispatch(jobs, njobs, uniform_layers, make_mat(m), pre, post, cus, stream);
}

// two-layer jobs, crossfaded: out = cross(layer[0], layer[1]) with weights wa, wb
extern "C" int cvk_chain_cross(const cvk_chain_job *jobs, int njobs, float wa, float wb, int cus, void *stream) {
    Mat mat = make_mat(NULL);
    mat.plain = 0; mat.cross = 1; mat.wa = wa; mat.wb = wb;
    return chain_dispatch(jobs, njobs, 2, mat, NULL, NULL, cus, stream);
}
