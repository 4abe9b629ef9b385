// membench2.hip -- which launch shape streams the chain's access pattern (NL read streams + 1 write stream of
// 66 MB frames, a batch of frames per launch) fastest WHEN THE KERNEL ALSO CARRIES THE CHAIN'S ARITHMETIC?
// Diagnostic only; not part of the library.  Extends tools/membench.hip by the things the production kernel has and
// that first sweep did not: a software pipeline (next trip's loads in flight during this trip's work and store), a
// workgroup-uniform chunk mapping over the whole batch (no per-frame tail), VALU ballast and LDS gathers of the
// production kernel's size, and a 128 KiB / 64 KiB LDS footprint that caps residency the way the transfer table does.
//
//   hipcc --offload-arch=gfx950 -O3 tools/membench2.hip -o /tmp/membench2 && /tmp/membench2 > gpurun_out/membench2.txt
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kMaxJobs = 16;
struct Job { u32x4 *out; const u32x4 *in[3]; };
struct Jobs { Job j[kMaxJobs]; };

__device__ __forceinline__ void asm_ld(u32x4 &dst, const u32x4 *p) {
    asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(dst) : "v"(p));
}
template <int N>
__device__ __forceinline__ void touch(u32x4 (&r)[N]) {       // names the registers a wait covers
#pragma unroll
    for (int i = 0; i < N; i++) asm volatile("" : "+v"(r[i]));
}
__device__ __forceinline__ void wait0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// the chain's per-trip work in miniature: `gathers` 2-byte LDS reads at data-dependent addresses and `fmas`
// dependent-by-four VALU operations per row
template <int NL>
__device__ __forceinline__ u32x4 work(const u32x4 *v, const uint16_t *lds, unsigned lds_mask, int gathers, int fmas) {
    u32x4 x = v[0];
#pragma unroll
    for (int k = 1; k < NL; k++) x ^= v[k];
    if (gathers > 0) {
        uint32_t acc = 0;
        for (int g = 0; g < gathers; g += 4) {
            acc += lds[(x.x >> (g & 15)) & lds_mask];
            acc += lds[(x.y >> (g & 15)) & lds_mask];
            acc += lds[(x.z >> (g & 15)) & lds_mask];
            acc += lds[(x.w >> (g & 15)) & lds_mask];
        }
        x.x ^= acc;
    }
    if (fmas > 0) {
        f32x4 f = __builtin_bit_cast(f32x4, x);
        const f32x4 a = { 1.0001f, 0.9999f, 1.0002f, 0.9998f }, b = { 0.5f, 0.25f, 0.125f, 0.0625f };
        for (int i = 0; i < fmas; i += 4) f = __builtin_elementwise_fma(f, a, b);
        x = __builtin_bit_cast(u32x4, f);
    }
    return x;
}

// Row r (0 <= r < nrows) is `L` consecutive 16-byte words of the batch's virtual space (frame after frame); every
// frame holds a whole number of rows.  Trip t of workgroup b covers U rows:
//   MAP 0 (interleaved):  rows (t*U + u) * G + b      -- what a grid-stride loop does, U trips at once
//   MAP 1 (contiguous):   rows (t*G + b) * U + u      -- each workgroup reads U*L*16 contiguous bytes per stream
template <int NL, int U, int PIPE, int MAP>
__global__ void k_stream2(Jobs jobs, int njobs, unsigned rows_per_frame, unsigned inv_rpf, unsigned lds_mask, int gathers, int fmas) {
    extern __shared__ uint16_t lds[];
    const unsigned L = blockDim.x, G = gridDim.x, b = blockIdx.x, tid = threadIdx.x;
    for (unsigned i = tid; i <= lds_mask; i += L) lds[i] = (uint16_t)(i * 2654435761u >> 16);
    __syncthreads();
    const unsigned nrows = rows_per_frame * (unsigned)njobs;
    const unsigned ntrips = (nrows + G * U - 1) / (G * U);

    auto row_of = [&](unsigned t, int u) -> unsigned { return MAP == 0 ? (t * U + u) * G + b : (t * G + b) * U + u; };
    auto issue = [&](u32x4 (&dst)[U * NL], unsigned t) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            unsigned r = row_of(t, u);
            r = r < nrows ? r : nrows - 1;                       // clamped, never predicated
            const unsigned job = __umulhi(r, inv_rpf), in_frame = r - job * rows_per_frame;   // wave-uniform; exact for r < 2^20
            const Job &jb = jobs.j[job];
#pragma unroll
            for (int k = 0; k < NL; k++) asm_ld(dst[u * NL + k], jb.in[k] + (size_t)in_frame * L + tid);
        }
    };
    auto finish = [&](u32x4 (&cur)[U * NL], unsigned t) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            const unsigned r = row_of(t, u);
            const u32x4 x = work<NL>(&cur[u * NL], lds, lds_mask, gathers, fmas);
            if (r < nrows) {
                const unsigned job = __umulhi(r, inv_rpf), in_frame = r - job * rows_per_frame;
                __builtin_nontemporal_store(x, jobs.j[job].out + (size_t)in_frame * L + tid);
            }
        }
    };

    u32x4 A[U * NL], B[U * NL];
    if (PIPE) {
        issue(A, 0);
        wait0(); touch(A);
        for (unsigned t = 0; t < ntrips; t += 2) {
            issue(B, t + 1 < ntrips ? t + 1 : t);
            __builtin_amdgcn_sched_barrier(0);
            // arithmetic first, then the wait for the prefetch, then the store (the production kernel's order)
            u32x4 res[U];
#pragma unroll
            for (int u = 0; u < U; u++) res[u] = work<NL>(&A[u * NL], lds, lds_mask, gathers, fmas);
            __builtin_amdgcn_sched_barrier(0);
            wait0(); touch(B);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < U; u++) {
                const unsigned r = row_of(t, u);
                if (r < nrows) { const unsigned job = __umulhi(r, inv_rpf), in_frame = r - job * rows_per_frame;
                    __builtin_nontemporal_store(res[u], jobs.j[job].out + (size_t)in_frame * L + tid); }
            }
            if (t + 1 >= ntrips) break;
            issue(A, t + 2 < ntrips ? t + 2 : t + 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < U; u++) res[u] = work<NL>(&B[u * NL], lds, lds_mask, gathers, fmas);
            __builtin_amdgcn_sched_barrier(0);
            wait0(); touch(A);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < U; u++) {
                const unsigned r = row_of(t + 1, u);
                if (r < nrows) { const unsigned job = __umulhi(r, inv_rpf), in_frame = r - job * rows_per_frame;
                    __builtin_nontemporal_store(res[u], jobs.j[job].out + (size_t)in_frame * L + tid); }
            }
        }
    } else {
        for (unsigned t = 0; t < ntrips; t++) {
            issue(A, t);
            wait0(); touch(A);
            __builtin_amdgcn_sched_barrier(0);
            finish(A, t);
        }
    }
}

struct Shape { int nl, lanes, wg_per_cu, u, pipe, map, lds_kib, gathers, fmas; };

template <int NL, int U, int PIPE, int MAP>
static void launch(const Shape &s, Jobs &jobs, int njobs, unsigned rows_per_frame, hipStream_t st) {
    const unsigned lds_bytes = (unsigned)s.lds_kib * 1024u;
    const unsigned mask = lds_bytes ? lds_bytes / 2 - 1 : 0;
    auto kern = k_stream2<NL, U, PIPE, MAP>;
    CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL(kern, dim3(256 * s.wg_per_cu), dim3(s.lanes), lds_bytes ? lds_bytes : 2, st, jobs, njobs, rows_per_frame, (unsigned)((0x100000000ull + rows_per_frame - 1) / rows_per_frame), mask,
                       lds_bytes ? s.gathers : 0, s.fmas);
}

template <int NL, int U>
static void launch_pm(const Shape &s, Jobs &jobs, int njobs, unsigned rpf, hipStream_t st) {
    if (s.pipe && s.map) launch<NL, U, 1, 1>(s, jobs, njobs, rpf, st);
    else if (s.pipe) launch<NL, U, 1, 0>(s, jobs, njobs, rpf, st);
    else if (s.map) launch<NL, U, 0, 1>(s, jobs, njobs, rpf, st);
    else launch<NL, U, 0, 0>(s, jobs, njobs, rpf, st);
}

static void launch_any(const Shape &s, Jobs &jobs, int njobs, unsigned rpf, hipStream_t st) {
    if (s.nl == 2) { if (s.u == 1) launch_pm<2, 1>(s, jobs, njobs, rpf, st); else if (s.u == 2) launch_pm<2, 2>(s, jobs, njobs, rpf, st); else launch_pm<2, 4>(s, jobs, njobs, rpf, st); }
    else           { if (s.u == 1) launch_pm<3, 1>(s, jobs, njobs, rpf, st); else if (s.u == 2) launch_pm<3, 2>(s, jobs, njobs, rpf, st); else launch_pm<3, 4>(s, jobs, njobs, rpf, st); }
}

int main(int argc, char **argv) {
    const size_t W = 3840, H = 2160, bytes = W * H * 8, words = bytes / 16, slot = 64u << 20;
    const int njobs = 8, reps = argc > 1 ? atoi(argv[1]) : 5;
    char *arena;
    CK(hipMalloc((void **)&arena, slot * 4 * njobs));
    CK(hipMemset(arena, 0x3b, slot * 4 * njobs));
    Jobs jobs;
    for (int j = 0; j < njobs; j++) {
        for (int k = 0; k < 3; k++) jobs.j[j].in[k] = (const u32x4 *)(arena + slot * (4 * j + k));
        jobs.j[j].out = (u32x4 *)(arena + slot * (4 * j + 3));
    }
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));

    std::vector<Shape> shapes;
    // work levels: none; the production kernel's (16 gathers + 160 VALU per row of a 2-layer chain)
    const int works[2][2] = { { 0, 0 }, { 16, 160 } };
    for (int wk = 0; wk < 2; wk++)
        for (int lanes : { 256, 512, 1024 })
            for (int wg : { 1, 2 })
                for (int u : { 1, 2, 4 })
                    for (int pipe : { 0, 1 })
                        for (int map : { 0, 1 }) {
                            if (lanes * wg > 1024) continue;
                            if (u == 1 && map == 1) continue;              // identical to map 0
                            if (lanes * u > 2048) continue;
                            // LDS footprint: 128 KiB caps residency at one workgroup per CU, 64 KiB at two
                            const int lds = wg == 1 ? 128 : 64;
                            shapes.push_back({ 2, lanes, wg, u, pipe, map, lds, works[wk][0], works[wk][1] });
                        }
    // the same without an LDS footprint, no work: the first sweep's regime (residency not capped)
    for (int lanes : { 256, 512 })
        for (int wg : { 1, 2, 4 })
            for (int u : { 1, 2 })
                shapes.push_back({ 2, lanes, wg, u, 0, 0, 0, 0, 0 });
    // three layers (config 4) on the best candidates
    for (int lanes : { 256, 512 })
        for (int u : { 1, 2 })
            for (int pipe : { 0, 1 })
                shapes.push_back({ 3, lanes, 1, u, pipe, 0, 128, 0, 0 });

    printf("%-3s %-5s %-3s %-2s %-4s %-3s %-4s %-4s %-4s %9s %8s %6s\n", "NL", "lanes", "wg", "U", "pipe", "map", "lds", "gath", "fma", "ms", "GB/s", "of8T");
    // interleave: every shape once per round, so that drift between rounds hits all shapes alike
    std::vector<std::vector<float>> t(shapes.size());
    for (int r = 0; r < reps + 1; r++)
        for (size_t i = 0; i < shapes.size(); i++) {
            const Shape &s = shapes[i];
            const unsigned rpf = (unsigned)(words / s.lanes);
            CK(hipEventRecord(e0, st));
            launch_any(s, jobs, njobs, rpf, st);
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            CK(hipGetLastError());
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (r > 0) t[i].push_back(ms);
        }
    for (size_t i = 0; i < shapes.size(); i++) {
        const Shape &s = shapes[i];
        std::sort(t[i].begin(), t[i].end());
        const float ms = t[i][t[i].size() / 2];
        const double gb = (double)bytes * (s.nl + 1) * njobs / (ms * 1e-3) / 1e9;
        printf("%-3d %-5d %-3d %-2d %-4d %-3d %-4d %-4d %-4d %9.4f %8.0f %6.3f\n", s.nl, s.lanes, s.wg_per_cu, s.u, s.pipe, s.map, s.lds_kib, s.gathers, s.fmas, ms, gb, gb / 8000.0);
    }
    return 0;
}
