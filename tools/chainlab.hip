// chainlab.hip -- which FEATURE of the production chain kernel costs streaming rate (and brings the per-process
// level)?  Starts from the plain grid-stride 2-read + 1-write loop that streams at ~0.78 of 8 TB/s in every process and
// adds the production kernel's features one at a time, all variants interleaved in one process, 64 frames per launch
// over a ring of 8 frame sets in one arena (bench.py's layout).  Diagnostic only; not part of the library.
//   hipcc --offload-arch=gfx950 -O3 tools/chainlab.hip -o tools/bin/chainlab
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef const u32x4 __attribute__((address_space(1))) *g_cu4;
typedef u32x4 __attribute__((address_space(1))) *g_u4;

constexpr int kJobs = 64;
struct Job { void *out; const void *in[2]; uint64_t npairs; };
struct Batch { Job j[kJobs]; };

enum { F_ASM = 1, F_PIPE = 2, F_LUT = 4, F_IDX64 = 8, F_CLAMP = 16, F_CHUNK = 32, F_GATHER = 64, F_RESYNC = 128, F_TICKET = 256, F_TICKET8 = 512, F_ROT = 1024 };

// timing-only rendezvous of the (co-resident) workgroups: no data is handed over, so no fences; bounded spin
__device__ __forceinline__ void resync(unsigned *ctr, unsigned target) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int spin = 0; spin < (1 << 20) && __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target; spin++) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
}


__device__ __forceinline__ void asm_ld(u32x4 &dst, const void *base, size_t idx) {
    const u32x4 *p = reinterpret_cast<const u32x4 *>(base) + idx;
    asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(dst) : "v"(p));
}
__device__ __forceinline__ void wait2(u32x4 &a, u32x4 &b) { asm volatile("s_waitcnt vmcnt(0)" : "+v"(a), "+v"(b) : : "memory"); }

__device__ __forceinline__ void stage(uint16_t *lds, const uint16_t *table) {
    const uint4 *src = reinterpret_cast<const uint4 *>(table);
    uint4 *dst = reinterpret_cast<uint4 *>(lds);
    const int slices = 8192 / (int)blockDim.x, rot = (int)(blockIdx.x >> 3);
    for (int it = 0; it < slices; it++) {
        const int i = ((it + rot) % slices) * (int)blockDim.x + (int)threadIdx.x;
        dst[i] = src[i];
    }
    __syncthreads();
}

template <int F>
__device__ __forceinline__ u32x4 combine(u32x4 x, u32x4 y, const uint16_t *lds) {
    u32x4 r = x ^ y;
    if (F & F_GATHER) {            // 16 two-byte gathers at data-dependent addresses, like the transfer table's
        uint32_t acc = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) { acc += lds[x[k] & 0xFFFFu]; acc += lds[x[k] >> 16]; acc += lds[y[k] & 0xFFFFu]; acc += lds[y[k] >> 16]; }
        r.x ^= acc;
    }
    return r;
}

template <int F>
__global__ __launch_bounds__(1024) void k_lab(Batch batch, int njobs, const uint16_t *table, unsigned *ctr, int every) {
    __shared__ uint16_t lds[(F & (F_LUT | F_GATHER)) ? 65536 : 1];
    if (F & (F_LUT | F_GATHER)) stage(lds, table);
    if (F & (F_TICKET | F_TICKET8)) {
        // dynamic chunk assignment: chunk ids come from a device-wide counter (F_TICKET) or from one of eight counters
        // chosen by blockIdx % 8, the chunk being ticket * 8 + blockIdx % 8 (F_TICKET8); wave 0 draws the ticket two
        // trips ahead, hands it to the other waves through LDS, one barrier per trip.  All frames equal here.
        __shared__ unsigned slot[4];
        const unsigned L = blockDim.x, per_frame = (unsigned)(batch.j[0].npairs / L), total = per_frame * (unsigned)njobs;
        unsigned *my = (F & F_TICKET8) ? ctr + 32 * (blockIdx.x & 7) : ctr;
        auto draw = [&]() -> unsigned {
            unsigned t = 0;
            if (threadIdx.x == 0) t = __hip_atomic_fetch_add(my, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return t;
        };
        auto chunk_of = [&](unsigned t) -> unsigned { return (F & F_TICKET8) ? t * 8 + (blockIdx.x & 7) : t; };
        unsigned t0 = draw(), t1 = draw();
        if (threadIdx.x == 0) { slot[0] = chunk_of(t0); slot[1] = chunk_of(t1); }
        __syncthreads();
        unsigned c = slot[0], n = 0;
        u32x4 a0, a1;
        if (c < total) { const unsigned j = c / per_frame, i = c - j * per_frame; asm_ld(a0, batch.j[j].in[0], (size_t)i * L + threadIdx.x); asm_ld(a1, batch.j[j].in[1], (size_t)i * L + threadIdx.x); }
        while (c < total) {
            const unsigned nc = slot[(n + 1) & 3];
            const unsigned t2 = draw();                                  // for trip n + 2
            const unsigned pc = nc < total ? nc : c;
            const unsigned pj = pc / per_frame, pi = pc - pj * per_frame;
            u32x4 b0, b1;
            asm_ld(b0, batch.j[pj].in[0], (size_t)pi * L + threadIdx.x);
            asm_ld(b1, batch.j[pj].in[1], (size_t)pi * L + threadIdx.x);
            __builtin_amdgcn_sched_barrier(0);
            const u32x4 r = combine<F>(a0, a1, lds);
            __builtin_amdgcn_sched_barrier(0);
            wait2(b0, b1);
            __builtin_amdgcn_sched_barrier(0);
            const unsigned j = c / per_frame, i = c - j * per_frame;
            __builtin_nontemporal_store(r, (g_u4)batch.j[j].out + (size_t)i * L + threadIdx.x);
            if (threadIdx.x == 0) slot[(n + 2) & 3] = chunk_of(t2);
            __syncthreads();
            a0 = b0; a1 = b1; c = nc; n++;
        }
        return;
    }
    if (F & F_CHUNK) {
        // workgroup-uniform chunk walk over the whole batch: chunk c = blockDim.x pairs; frames hold whole chunks
        const unsigned L = blockDim.x, G = gridDim.x, nj_all = (unsigned)njobs;
        unsigned j = 0, c = blockIdx.x, nch = (unsigned)(batch.j[0].npairs / L);
        while (j < nj_all && c >= nch) { c -= nch; j++; nch = j < nj_all ? (unsigned)(batch.j[j].npairs / L) : 1; }
        if (j >= nj_all) return;
        unsigned off = blockIdx.x;                 // F_ROT: position inside the stripe of G chunks, rotated by `every` per trip
        // frame pointers live in registers (wave-uniform) and are re-read from the argument segment only when the
        // walk crosses into another frame
        const void *ci0 = batch.j[j].in[0], *ci1 = batch.j[j].in[1];
        void *co = batch.j[j].out;
        u32x4 a0, a1;
        asm_ld(a0, ci0, (size_t)c * L + threadIdx.x); asm_ld(a1, ci1, (size_t)c * L + threadIdx.x);
        for (;;) {
            unsigned noff = off;
            if (F & F_ROT) { noff = off + (unsigned)every; if (noff >= G) noff -= G; }
            unsigned nj = j, nc = c + G + noff - off, nnch = nch;
            off = noff;
            const void *ni0 = ci0, *ni1 = ci1;
            void *no = co;
            if (nc >= nnch) {
                while (nj < nj_all && nc >= nnch) { nc -= nnch; nj++; nnch = nj < nj_all ? (unsigned)(batch.j[nj].npairs / L) : 1; }
                if (nj < nj_all) { ni0 = batch.j[nj].in[0]; ni1 = batch.j[nj].in[1]; no = batch.j[nj].out; }
            }
            const bool more = nj < nj_all;
            u32x4 b0, b1;
            asm_ld(b0, more ? ni0 : ci0, (size_t)(more ? nc : c) * L + threadIdx.x);
            asm_ld(b1, more ? ni1 : ci1, (size_t)(more ? nc : c) * L + threadIdx.x);
            __builtin_amdgcn_sched_barrier(0);
            const u32x4 r = combine<F>(a0, a1, lds);
            __builtin_amdgcn_sched_barrier(0);
            wait2(b0, b1);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_nontemporal_store(r, (g_u4)co + (size_t)c * L + threadIdx.x);
            if (!more) break;
            a0 = b0; a1 = b1; j = nj; c = nc; nch = nnch; ci0 = ni0; ci1 = ni1; co = no;
        }
        return;
    }
    for (int jn = 0; jn < njobs; jn++) {
        const Job &job = batch.j[jn];
        if ((F & F_RESYNC) && jn > 0 && jn % every == 0) resync(ctr, (unsigned)(jn / every) * gridDim.x);
        if (F & F_IDX64) {
            const size_t stride = (size_t)gridDim.x * blockDim.x, npairs = job.npairs;
            const size_t lane = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
            if (F & F_PIPE) {
                u32x4 a0, a1, b0, b1;
                size_t idx = lane < npairs - 1 ? lane : npairs - 1;
                asm_ld(a0, job.in[0], idx); asm_ld(a1, job.in[1], idx);
                wait2(a0, a1);
                for (size_t base = lane; base < npairs; base += stride) {
                    size_t nidx = base + stride;
                    nidx = nidx < npairs - 1 ? nidx : npairs - 1;
                    asm_ld(b0, job.in[0], nidx); asm_ld(b1, job.in[1], nidx);
                    __builtin_amdgcn_sched_barrier(0);
                    const u32x4 r = combine<F>(a0, a1, lds);
                    __builtin_amdgcn_sched_barrier(0);
                    wait2(b0, b1);
                    __builtin_amdgcn_sched_barrier(0);
                    __builtin_nontemporal_store(r, (g_u4)job.out + base);
                    a0 = b0; a1 = b1;
                }
            } else {
                for (size_t i = lane; i < npairs; i += stride) {
                    u32x4 x, y;
                    if (F & F_ASM) { asm_ld(x, job.in[0], i); asm_ld(y, job.in[1], i); wait2(x, y); }
                    else { x = __builtin_nontemporal_load((g_cu4)job.in[0] + i); y = __builtin_nontemporal_load((g_cu4)job.in[1] + i); }
                    __builtin_nontemporal_store(combine<F>(x, y, lds), (g_u4)job.out + i);
                }
            }
        } else {
            const unsigned stride = gridDim.x * blockDim.x, npairs = (unsigned)job.npairs;
            for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < npairs; i += stride) {
                u32x4 x, y;
                if (F & F_ASM) { asm_ld(x, job.in[0], i); asm_ld(y, job.in[1], i); wait2(x, y); }
                else { x = __builtin_nontemporal_load((g_cu4)job.in[0] + i); y = __builtin_nontemporal_load((g_cu4)job.in[1] + i); }
                __builtin_nontemporal_store(combine<F>(x, y, lds), (g_u4)job.out + i);
            }
        }
    }
}

struct Variant { const char *name; int flags; int lanes; int split; int streams = 1; };   // split: issue the batch as launches of this many frames (0: one launch)

static unsigned *g_ctr; static int g_every = 8;
template <int F>
static void go(const Batch &b, int njobs, const uint16_t *table, int lanes, hipStream_t s) {
    if (F & (F_RESYNC | F_TICKET | F_TICKET8)) CK(hipMemsetAsync(g_ctr, 0, 1024, s));
    hipLaunchKernelGGL(k_lab<F>, dim3(256), dim3(lanes), 0, s, b, njobs, table, g_ctr, g_every);
}

static void launch(int f, const Batch &b, int njobs, const uint16_t *table, int lanes, hipStream_t s) {
    switch (f) {
#define C(X) case X: go<X>(b, njobs, table, lanes, s); break;
    C(F_CHUNK | F_ROT) C(F_TICKET) C(F_TICKET8) C(F_TICKET | F_LUT | F_GATHER) C(F_TICKET8 | F_LUT | F_GATHER) C(F_RESYNC) C(F_RESYNC | F_LUT) C(0) C(F_ASM) C(F_IDX64) C(F_IDX64 | F_ASM) C(F_IDX64 | F_ASM | F_PIPE) C(F_LUT) C(F_LUT | F_IDX64 | F_ASM | F_PIPE)
    C(F_CHUNK) C(F_CHUNK | F_LUT) C(F_CHUNK | F_LUT | F_GATHER) C(F_LUT | F_GATHER | F_IDX64 | F_ASM | F_PIPE) C(F_LUT | F_GATHER)
#undef C
    default: printf("no instance for flags %d\n", f); exit(1);
    }
}

int main(int argc, char **argv) {
    const size_t bytes = 3840ull * 2160 * 8, pairs = bytes / 16, slot = 64u << 20;
    const int ring = 8, njobs = argc > 1 ? atoi(argv[1]) : 64, rounds = argc > 2 ? atoi(argv[2]) : 5;
    char *arena;
    CK(hipMalloc((void **)&arena, slot * 3 * ring));
    CK(hipMemset(arena, 0x3b, slot * 3 * ring));
    uint16_t *table;
    CK(hipMalloc((void **)&table, 131072));
    CK(hipMemset(table, 0x11, 131072));
    Batch b;
    for (int j = 0; j < kJobs; j++) {
        const int g = j % ring;
        b.j[j].in[0] = arena + slot * (3 * g);
        b.j[j].in[1] = arena + slot * (3 * g + 1);
        b.j[j].out = arena + slot * (3 * g + 2);
        b.j[j].npairs = pairs;
    }
    CK(hipMalloc((void **)&g_ctr, 1024));
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t e0, e1, ej;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&ej));
    hipStream_t s2;
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    const Variant vs[] = {
        { "plain, one launch                            512", 0, 512, 0 },
        { "plain, launches of 8 frames back to back     512", 0, 512, 8 },
        { "plain, launches of 16 frames back to back    512", 0, 512, 16 },
        { "plain, launches of 4 frames back to back     512", 0, 512, 4 },
        { "plain, launches of 4, alternating 2 streams  512", 0, 512, 4, 2 },
        { "plain, launches of 2, alternating 2 streams  512", 0, 512, 2, 2 },
        { "plain, launches of 8, alternating 2 streams  512", 0, 512, 8, 2 },
        { "chunk walk + table + gathers, launches of 4, 2 streams", F_CHUNK | F_LUT | F_GATHER, 512, 4, 2 },
        { "chunk walk + table + gathers, launches of 4, 1 stream ", F_CHUNK | F_LUT | F_GATHER, 512, 4 },
        { "plain, one launch, rendezvous every 8 frames 512", F_RESYNC, 512, -8 },
        { "plain, one launch, rendezvous every 2 frames 512", F_RESYNC, 512, -2 },
        { "plain, one launch, rendezvous every frame    512", F_RESYNC, 512, -1 },
        { "plain + table, launches of 8 frames          512", F_LUT, 512, 8 },
        { "plain + table, rendezvous every 8 frames     512", F_RESYNC | F_LUT, 512, -8 },
        { "plain + table, rendezvous every 2 frames     512", F_RESYNC | F_LUT, 512, -2 },
        { "pipelined + table (prod. mem-only), 1 launch 512", F_LUT | F_IDX64 | F_ASM | F_PIPE, 512, 0 },
        { "pipelined + table, launches of 8 frames      512", F_LUT | F_IDX64 | F_ASM | F_PIPE, 512, 8 },
        { "chunk walk, one launch                       512", F_CHUNK, 512, 0 },
        { "chunk walk, launches of 8 frames             512", F_CHUNK, 512, 8 },
        { "chunk walk rotated by 1 per trip, one launch 512", F_CHUNK | F_ROT, 512, -1 },
        { "chunk walk rotated by 8 per trip, one launch 512", F_CHUNK | F_ROT, 512, -8 },
        { "chunk walk rotated by 37 per trip, 1 launch  512", F_CHUNK | F_ROT, 512, -37 },
        { "chunk walk rotated by 101 per trip, 1 launch 512", F_CHUNK | F_ROT, 512, -101 },
        { "chunk walk rotated by 128 per trip, 1 launch 512", F_CHUNK | F_ROT, 512, -128 },
        { "tickets (8 counters), one launch             512", F_TICKET8, 512, 0 },
        { "chunk walk + table + gathers, launches of 8  512", F_CHUNK | F_LUT | F_GATHER, 512, 8 },
    };
    const int nv = (int)(sizeof vs / sizeof vs[0]);
    std::vector<std::vector<float>> t(nv);
    for (int r = 0; r < rounds + 1; r++)
        for (int v = 0; v < nv; v++) {
            CK(hipEventRecord(e0, s));
            if (vs[v].split > 0 && vs[v].split < njobs) {
                int turn = 0;
                if (vs[v].streams == 2) { CK(hipEventRecord(ej, s)); CK(hipStreamWaitEvent(s2, ej, 0)); }
                for (int first = 0; first < njobs; first += vs[v].split, turn++) {
                    Batch part = b;
                    for (int q = 0; q < vs[v].split && first + q < njobs; q++) part.j[q] = b.j[first + q];
                    launch(vs[v].flags, part, std::min(vs[v].split, njobs - first), table, vs[v].lanes, (vs[v].streams == 2 && (turn & 1)) ? s2 : s);
                }
                if (vs[v].streams == 2) { CK(hipEventRecord(ej, s2)); CK(hipStreamWaitEvent(s, ej, 0)); }
            } else if (vs[v].split < 0) { g_every = -vs[v].split; launch(vs[v].flags, b, njobs, table, vs[v].lanes, s); }
            else launch(vs[v].flags, b, njobs, table, vs[v].lanes, s);
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            CK(hipGetLastError());
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (r) t[v].push_back(ms);
        }
    printf("arena %p, %d frames per launch\n", (void *)arena, njobs);
    for (int v = 0; v < nv; v++) {
        std::sort(t[v].begin(), t[v].end());
        const float ms = t[v][t[v].size() / 2];
        printf("%-50s %8.4f ms  %.4f of 8 TB/s  (min %.4f max %.4f)\n", vs[v].name, ms, (double)bytes * 3 * njobs / (ms * 1e-3) / 8e12,
               (double)bytes * 3 * njobs / (t[v].back() * 1e-3) / 8e12, (double)bytes * 3 * njobs / (t[v].front() * 1e-3) / 8e12);
    }
    return 0;
}
