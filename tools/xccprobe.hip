// xccprobe.hip -- is the chain kernel's per-process throughput level tied to the hardware queue's XCC rotation
// (which XCD workgroup 0 of a dispatch lands on)?  For each of 8 streams of ONE process: which XCD block 0..7 ran on,
// and the rate of a 2-read + 1-write streaming kernel over 8 x (2 + 1) 66 MB frames, for every in-kernel rotation of
// the chunk -> workgroup assignment.  Run the binary several times back to back to see both levels.
// Diagnostic only; not part of the library.
//   hipcc --offload-arch=gfx950 -O3 tools/xccprobe.hip -o tools/bin/xccprobe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

struct Job { u32x4 *out; const u32x4 *in[2]; };
struct Jobs { Job j[8]; };

__global__ void k_probe(int *xcc_of_block) {
    if (threadIdx.x == 0) {
        uint32_t id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(id));
        xcc_of_block[blockIdx.x] = (int)id;
    }
}

// persistent 256 x 512 lanes, grid-stride over each frame; rot re-deals the 8 KiB chunks among the workgroups of an
// aligned group of eight (so among the eight XCDs) without changing which addresses are in flight together
__global__ void k_fill(uint32_t *p, size_t n, uint32_t seed) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t x = (uint32_t)i * 2654435761u ^ seed;
        x ^= x << 13; x ^= x >> 17; x ^= x << 5;
        // half-float-like payload: four U[0,1) halfs per 8 bytes would be codes 0..0x3BFF; keep it simple: random 14-bit codes
        p[i] = (x & 0x3FFF3FFFu);
    }
}

__global__ __launch_bounds__(512) void k_stream(Jobs jobs, int njobs, unsigned words, int rot, int lds_halfs) {
    extern __shared__ uint16_t lds[];
    for (int i = threadIdx.x; i < lds_halfs; i += blockDim.x) lds[i] = (uint16_t)i;
    __syncthreads();
    const unsigned b = (blockIdx.x & ~7u) | ((blockIdx.x + (unsigned)rot) & 7u);
    const unsigned stride = gridDim.x * blockDim.x;
    for (int j = 0; j < njobs; j++) {
        const Job &job = jobs.j[j];
        for (unsigned i = b * blockDim.x + threadIdx.x; i < words; i += stride) {
            u32x4 x = __builtin_nontemporal_load(&job.in[0][i]);
            u32x4 y = __builtin_nontemporal_load(&job.in[1][i]);
            x ^= y;
            if (lds_halfs) x.x ^= lds[x.y & (unsigned)(lds_halfs - 1)];
            __builtin_nontemporal_store(x, &job.out[i]);
        }
    }
}

int main(int argc, char **argv) {
    const size_t bytes = 3840ull * 2160 * 8, words = bytes / 16, slot = 64u << 20;
    const int njobs = 8, nstreams = argc > 1 ? atoi(argv[1]) : 8;
    char *arena;
    CK(hipMalloc((void **)&arena, slot * 3 * njobs));
    const int fill = argc > 2 ? atoi(argv[2]) : 0, lds_kib = argc > 3 ? atoi(argv[3]) : 0;
    if (fill) { hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (uint32_t *)arena, slot * 3 * njobs / 4, 12345u); CK(hipDeviceSynchronize()); }
    else CK(hipMemset(arena, 0x3b, slot * 3 * njobs));
    CK(hipFuncSetAttribute((const void *)k_stream, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    Jobs jobs;
    for (int j = 0; j < njobs; j++) {
        jobs.j[j].in[0] = (const u32x4 *)(arena + slot * (3 * j));
        jobs.j[j].in[1] = (const u32x4 *)(arena + slot * (3 * j + 1));
        jobs.j[j].out = (u32x4 *)(arena + slot * (3 * j + 2));
    }
    int *d_xcc, h_xcc[256];
    CK(hipMalloc((void **)&d_xcc, sizeof h_xcc));
    std::vector<hipStream_t> streams(nstreams);
    for (auto &s : streams) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("arena %p fill %s lds %d KiB\n", (void *)arena, fill ? "random" : "constant", lds_kib);
    for (int si = -1; si < nstreams; si++) {
        hipStream_t s = si < 0 ? (hipStream_t)0 : streams[si];
        hipLaunchKernelGGL(k_probe, dim3(256), dim3(64), 0, s, d_xcc);
        CK(hipStreamSynchronize(s));
        CK(hipMemcpy(h_xcc, d_xcc, sizeof h_xcc, hipMemcpyDeviceToHost));
        printf("stream %2d: blocks 0..7 on XCC %d %d %d %d %d %d %d %d |", si, h_xcc[0], h_xcc[1], h_xcc[2], h_xcc[3], h_xcc[4], h_xcc[5], h_xcc[6], h_xcc[7]);
        for (int rot = 0; rot < 8; rot += 2) {
            std::vector<float> t;
            for (int r = 0; r < 6; r++) {
                CK(hipEventRecord(e0, s));
                hipLaunchKernelGGL(k_stream, dim3(256), dim3(512), lds_kib * 1024, s, jobs, njobs, (unsigned)words, rot, lds_kib * 512);
                CK(hipEventRecord(e1, s));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (r) t.push_back(ms);
            }
            std::sort(t.begin(), t.end());
            printf(" %.3f", (double)bytes * 3 * njobs / (t[t.size() / 2] * 1e-3) / 8e12);
        }
        printf("\n");
    }
    return 0;
}
