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
//   * (chain_kernel.hpp; instantiated here for 1..4 layers and in chain_deep_ops.hip for 5..8)
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
#include "chain_kernel.hpp"
#include <stdlib.h>

// chain_deep_ops.hip: the same kernel for 5..8 layers (one launch of up to 48 frames; *taken = frames consumed).  Plain build
// only: in the contracted build (CVS_CONTRACT) stacks of more than four layers take the first-version kernel below.
extern "C" int cvk_chain_deep(const cvk_chain_job *jobs, int njobs, int nlayers, const cvs::Mat *mat, const uint16_t *pre, const uint16_t *post,
                              unsigned grid, unsigned block, int lshift, int diag, uint64_t bytes_per_launch, void *stream, int *taken);

namespace {

// Job records travel as kernel arguments (<= 4 KiB per launch).  The production kernel takes up to four layers, so
// its records are compact (48 B) and a launch can carry 64 frames; the first version keeps the full 8-layer record.
#ifdef CVS_CONTRACT
constexpr int kJobsPerLaunchV0 = 32, kFusedLayers = 4;
#else
constexpr int kJobsPerLaunchV0 = 32, kFusedLayers = 8;
#endif
typedef BatchT<4> Batch;
// bytes (read + written) one launch of the production kernel moves before the host starts the next one:
// eight 4K two-layer frames.  See "SHORT LAUNCHES" above; tools/chainlab.hip and profiles/r02/launch_split.txt.
constexpr uint64_t kBytesPerLaunch = 8ull * 3840 * 2160 * 24;
constexpr int kChainBlock = 512, kChainBlockLog2 = 9;
struct BatchV0 { cvk_chain_job jobs[kJobsPerLaunchV0]; };

// ---------------------------------------------------------------- v0: first correct version
// Serves batches with mixed layer counts and frames too large for 32-bit chunk offsets.

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
#ifndef CVS_CONTRACT
        if (fused_kernel && uniform_layers > 4) {
            const int diag = t.variant == 10 ? DIAG_MEMORY_ONLY : t.variant == 12 ? DIAG_COMPUTE_ONLY : DIAG_NONE;
            rc = cvk_chain_deep(jobs + first, njobs - first, uniform_layers, &mat, pre, post, grid, t.block, t.lshift, diag, t.bytes_per_launch, stream, &n);
        } else
#endif
        if (fused_kernel) {
            Batch b;
            n = fill_batch(b, jobs + first, njobs - first, uniform_layers, t.bytes_per_launch);
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
